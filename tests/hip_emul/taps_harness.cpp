// CPU harness for gan_amd/csrc/taps.hip (few-channel convolutions as a dense product plus a sum over taps): the
// kernels' own source compiled for the host (tests/hip_emul/hip/hip_runtime.h) and run under AddressSanitizer on
// buffers of exactly the documented sizes, against direct convolution loops in double precision.
// Test infrastructure only: built and run by tests/test_bn_tile_emulated.py.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <random>

thread_local EmulIdx threadIdx, blockIdx;
EmulIdx blockDim, gridDim;
pthread_barrier_t* emul_barrier;

#include "../../gan_amd/csrc/taps.hip"

namespace gan {
static char g_err[512];
char* error_buffer() { return g_err; }
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
}  // namespace gan

static int g_failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++g_failures; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

struct Buf {
    float* p;
    size_t n;
    explicit Buf(size_t n_) : n(n_) { if (posix_memalign(reinterpret_cast<void**>(&p), 16, (n ? n : 4) * sizeof(float))) abort(); for (size_t i = 0; i < n; ++i) p[i] = NAN; }
    ~Buf() { free(p); }
    void fill(std::mt19937& g) { std::normal_distribution<float> d(0.f, 1.f); for (size_t i = 0; i < n; ++i) p[i] = d(g); }
};

static int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

static double max_err(const Buf& got, const std::vector<double>& want) {
    double worst = 0, scale = 1e-30;
    for (size_t i = 0; i < want.size(); ++i) { worst = std::max(worst, std::fabs(got.p[i] - want[i])); scale = std::max(scale, std::fabs(want[i])); }
    return worst / scale;
}

// one geometry: n x h x w image, cb channels on the wide side, cs on the thin side, k x k filter
static void run_case(int n, int h, int w, int cb, int cs, int k, int reflect, unsigned seed) {
    std::mt19937 gen(seed);
    const int taps = k * k, pad = (k - 1) / 2, ld = (taps * cs + 3) & ~3;
    const long pix = (long)n * h * w;
    auto src = [&](int p, int t, int len) { int i = p + t - pad; if (reflect) i = refl(i, len); return (i < 0 || i >= len) ? -1 : i; };
    // ---------------- head: x [pix][cb] -> y [pix][cs], filter [t][c][o]
    {
        Buf x(pix * cb), wt((size_t)taps * cb * cs), bias(cs), wp((size_t)cb * ld), z(pix * ld), y(pix * cs), dy(pix * cs), e(pix * ld);
        x.fill(gen); wt.fill(gen); bias.fill(gen); dy.fill(gen);
        EXPECT(gan_taps_pack_filter(wt.p, wp.p, taps, cb, cs, ld, 0, nullptr) == 0, "%s", error_buffer());
        for (long p = 0; p < pix; ++p)
            for (int j = 0; j < ld; ++j) { double s = 0; for (int c = 0; c < cb; ++c) s += (double)x.p[p * cb + c] * wp.p[c * ld + j]; z.p[p * ld + j] = (float)s; }
        EXPECT(gan_taps_sum_fwd(z.p, bias.p, y.p, n, h, w, cs, k, k, ld, reflect, nullptr) == 0, "%s", error_buffer());
        std::vector<double> yr(pix * cs), dwr((size_t)taps * cb * cs, 0.0), dxr(pix * cb, 0.0), dbr(cs, 0.0);
        for (int im = 0; im < n; ++im) for (int py = 0; py < h; ++py) for (int px = 0; px < w; ++px) {
            const long p = ((long)im * h + py) * w + px;
            for (int o = 0; o < cs; ++o) { yr[p * cs + o] = bias.p[o]; dbr[o] += dy.p[p * cs + o]; }
            for (int ty = 0; ty < k; ++ty) for (int tx = 0; tx < k; ++tx) {
                const int sy = src(py, ty, h), sx = src(px, tx, w);
                if (sy < 0 || sx < 0) continue;
                const long q = ((long)im * h + sy) * w + sx;
                const int t = ty * k + tx;
                for (int c = 0; c < cb; ++c) for (int o = 0; o < cs; ++o) {
                    const double wv = wt.p[(t * cb + c) * cs + o];
                    yr[p * cs + o] += x.p[q * cb + c] * wv;
                    dwr[(t * cb + c) * cs + o] += (double)x.p[q * cb + c] * dy.p[p * cs + o];
                    dxr[q * cb + c] += (double)dy.p[p * cs + o] * wv;
                }
            }
        }
        EXPECT(max_err(y, yr) < 1e-5, "head forward %dx%dx%d k%d reflect %d: %g", n, h, w, k, reflect, max_err(y, yr));
        // gradients through E
        EXPECT(gan_taps_spread_bwd(dy.p, e.p, n, h, w, cs, k, k, ld, reflect, nullptr) == 0, "%s", error_buffer());
        Buf dwp((size_t)cb * ld), dbp(ld), dw((size_t)taps * cb * cs), db(cs), dx(pix * cb);
        for (int c = 0; c < cb; ++c) for (int j = 0; j < ld; ++j) { double s = 0; for (long q = 0; q < pix; ++q) s += (double)x.p[q * cb + c] * e.p[q * ld + j]; dwp.p[c * ld + j] = (float)s; }
        for (int j = 0; j < ld; ++j) { double s = 0; for (long q = 0; q < pix; ++q) s += e.p[q * ld + j]; dbp.p[j] = (float)s; }
        for (size_t i = 0; i < dw.n; ++i) dw.p[i] = 1.f;
        for (int o = 0; o < cs; ++o) db.p[o] = 1.f;
        EXPECT(gan_taps_unpack_filter_grad(dwp.p, dbp.p, dw.p, db.p, taps, cb, cs, ld, 0, 1, nullptr) == 0, "%s", error_buffer());
        for (auto& v : dwr) v += 1.0;
        for (auto& v : dbr) v += 1.0;
        EXPECT(max_err(dw, dwr) < 1e-5, "head filter gradient: %g", max_err(dw, dwr));
        EXPECT(max_err(db, dbr) < 1e-5, "head bias gradient: %g", max_err(db, dbr));
        for (long q = 0; q < pix; ++q) for (int c = 0; c < cb; ++c) { double s = 0; for (int j = 0; j < ld; ++j) s += (double)e.p[q * ld + j] * wp.p[c * ld + j]; dx.p[q * cb + c] = (float)s; }
        EXPECT(max_err(dx, dxr) < 1e-5, "head input gradient: %g", max_err(dx, dxr));
    }
    // ---------------- stem: input gradient of x [pix][cs] -> y [pix][cb], filter [t][o][c]
    {
        Buf wt((size_t)taps * cs * cb), dy(pix * cb), wp((size_t)cb * ld), z(pix * ld), dx(pix * cs);
        wt.fill(gen); dy.fill(gen);
        EXPECT(gan_taps_pack_filter(wt.p, wp.p, taps, cb, cs, ld, 1, nullptr) == 0, "%s", error_buffer());
        for (long p = 0; p < pix; ++p)
            for (int j = 0; j < ld; ++j) { double s = 0; for (int c = 0; c < cb; ++c) s += (double)dy.p[p * cb + c] * wp.p[c * ld + j]; z.p[p * ld + j] = (float)s; }
        EXPECT(gan_taps_sum_bwd(z.p, dx.p, n, h, w, cs, k, k, ld, reflect, nullptr) == 0, "%s", error_buffer());
        std::vector<double> dxr(pix * cs, 0.0);
        for (int im = 0; im < n; ++im) for (int py = 0; py < h; ++py) for (int px = 0; px < w; ++px) {
            const long p = ((long)im * h + py) * w + px;
            for (int ty = 0; ty < k; ++ty) for (int tx = 0; tx < k; ++tx) {
                const int sy = src(py, ty, h), sx = src(px, tx, w);
                if (sy < 0 || sx < 0) continue;
                const long q = ((long)im * h + sy) * w + sx;
                const int t = ty * k + tx;
                for (int o = 0; o < cs; ++o) for (int c = 0; c < cb; ++c) dxr[q * cs + o] += (double)dy.p[p * cb + c] * wt.p[(t * cs + o) * cb + c];
            }
        }
        EXPECT(max_err(dx, dxr) < 1e-5, "stem input gradient %dx%dx%d k%d reflect %d: %g", n, h, w, k, reflect, max_err(dx, dxr));
        // and the transposed unpacking (gradient of a filter stored [t][o][c])
        Buf dwp((size_t)cb * ld), dw((size_t)taps * cs * cb);
        dwp.fill(gen);
        EXPECT(gan_taps_unpack_filter_grad(dwp.p, nullptr, dw.p, nullptr, taps, cb, cs, ld, 1, 0, nullptr) == 0, "%s", error_buffer());
        bool same = true;
        for (int t = 0; t < taps; ++t) for (int o = 0; o < cs; ++o) for (int c = 0; c < cb; ++c) same = same && dw.p[(t * cs + o) * cb + c] == dwp.p[c * ld + t * cs + o];
        EXPECT(same, "transposed unpack");
    }
}

int main() {
    run_case(2, 9, 7, 5, 3, 7, 1, 1);      // 7 x 7 mirrored, image barely larger than the padding (both borders within reach)
    run_case(1, 12, 10, 6, 3, 7, 1, 2);
    run_case(2, 8, 9, 4, 3, 3, 1, 3);
    run_case(1, 9, 9, 3, 1, 5, 0, 4);      // zero padding, one channel
    run_case(1, 5, 6, 7, 4, 3, 0, 5);
    run_case(1, 4, 4, 2, 2, 7, 1, 6);      // image of 4 pixels, padding 3: every pixel is near both borders
    printf("%s: %d failure(s)\n", g_failures ? "FAILED" : "ok", g_failures);
    return g_failures ? 1 : 0;
}
