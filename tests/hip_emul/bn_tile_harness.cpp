// CPU harness for the two-launch batch-norm kernels of gan_amd/csrc/norm.hip and their variants that take their rows
// from a contraction's split-K slabs.  The kernels' own source is
// compiled for the host through tests/hip_emul/hip/hip_runtime.h and run under AddressSanitizer on buffers of exactly
// the documented sizes: an access outside them aborts the run.  Results are compared with straightforward loops.
// Test infrastructure only: built and run by tests/test_bn_tile_emulated.py.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <random>

thread_local EmulIdx threadIdx, blockIdx;
EmulIdx blockDim, gridDim;
pthread_barrier_t* emul_barrier;

#include "../../gan_amd/csrc/norm.hip"

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
Knobs& knobs() {
    static Knobs k = [] {
        Knobs r;
        memset(&r, 0, sizeof(r));
        r.force_cfg = -1;
        r.two_launch_bn = 1;
        r.rgb_gather = 1;
        return r;
    }();
    return k;
}
}  // namespace gan

using namespace gan;

// 16-byte aligned buffer of exactly n floats (ASan guards both ends)
struct Buf {
    float* p;
    size_t n;
    explicit Buf(size_t n_) : n(n_) {
        if (posix_memalign(reinterpret_cast<void**>(&p), 16, (n ? n : 4) * sizeof(float))) abort();
    }
    ~Buf() { free(p); }
    void fill(std::mt19937& g, float scale = 1.f, float shift = 0.f) {
        std::normal_distribution<float> d(0.f, 1.f);
        for (size_t i = 0; i < n; ++i) p[i] = d(g) * scale + shift;
    }
    void nan() { for (size_t i = 0; i < n; ++i) p[i] = NAN; }
};

static int g_failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++g_failures; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static bool same_bits(const Buf& a, const Buf& b) { return a.n == b.n && memcmp(a.p, b.p, a.n * sizeof(float)) == 0; }

// out[i] = sum_k slab[k][i] + bias[i % c], in the order the library's slab reduction takes it
static void sum_slabs(const Buf& slabs, int splits, long elems, const float* bias, int c, Buf& out) {
    for (long i = 0; i < elems; ++i) {
        float s = slabs.p[i];
        for (int k = 1; k < splits; ++k) s += slabs.p[(long)k * elems + i];
        if (bias) s += bias[i % c];
        out.p[i] = s;
    }
}

static RowSource tensor_source(float* t) { return RowSource{nullptr, t, nullptr, 0, 0}; }

// Forward: statistics riding on the slab sum against statistics of the finished tensor, then the apply launch.
static void forward_case(int groups, int rows, int c, int splits, unsigned seed) {
    std::mt19937 gen(seed);
    TileBn ts;
    EXPECT(bn_ride_plan(groups, rows, c, false, &ts), "no forward plan for %d x %d x %d", groups, rows, c);
    const long elems = (long)groups * rows * c;
    Buf slabs((size_t)splits * elems), bias(c), y0(elems), y1(elems);
    slabs.fill(gen); bias.fill(gen);
    Buf p0((size_t)2 * groups * ts.nb * c), p1((size_t)2 * groups * ts.nb * c);
    p0.nan(); p1.nan(); y1.nan();
    sum_slabs(slabs, splits, elems, bias.p, c, y0);
    BnRide r = BnRide{1, reinterpret_cast<float2*>(p0.p), ts, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    EXPECT(launch_bn_ride(r, tensor_source(y0.p), nullptr, "fwd alone") == 0, "%s", error_buffer());
    r.partial = reinterpret_cast<float2*>(p1.p);
    EXPECT(launch_bn_ride(r, RowSource{slabs.p, y1.p, bias.p, elems, splits}, nullptr, "fwd ride") == 0,
           "%s", error_buffer());
    EXPECT(same_bits(y0, y1), "forward %d x %d x %d splits %d: summed rows differ", groups, rows, c, splits);
    EXPECT(same_bits(p0, p1), "forward %d x %d x %d splits %d: statistics differ", groups, rows, c, splits);
    // the apply launch from those statistics against plain two-pass moments
    Buf gamma(c), beta(c), z(elems), mean(groups * c), rstd(groups * c), mm(c), mv(c);
    gamma.fill(gen, 0.1f, 1.f); beta.fill(gen);
    for (int i = 0; i < c; ++i) { mm.p[i] = 0.f; mv.p[i] = 1.f; }
    EXPECT(gan_bn_fwd_train_from_stats(y1.p, gamma.p, beta.p, mm.p, mv.p, nullptr, z.p, mean.p, rstd.p, groups, rows, c, 1e-5f, 0.9f, 0,
                                       GAN_ACT_NONE, 0.f, p1.p, nullptr) == 0, "%s", error_buffer());
    double worst = 0;
    for (int g = 0; g < groups; ++g)
        for (int ch = 0; ch < c; ++ch) {
            double m = 0, v = 0;
            for (int rr = 0; rr < rows; ++rr) m += y0.p[((long)g * rows + rr) * c + ch];
            m /= rows;
            for (int rr = 0; rr < rows; ++rr) { const double d = y0.p[((long)g * rows + rr) * c + ch] - m; v += d * d; }
            v /= rows;
            worst = std::max(worst, std::fabs(mean.p[g * c + ch] - m));
            worst = std::max(worst, std::fabs(rstd.p[g * c + ch] - 1.0 / std::sqrt(v + 1e-5)) * std::sqrt(v + 1e-5));
        }
    EXPECT(worst < 1e-4, "forward %d x %d x %d: saved statistics off by %g", groups, rows, c, worst);
}

// Backward: sums riding on the slab sum (plain or folded) against sums over the finished gradient, then the apply launch
// with column sums.
static void backward_case(int groups, int rows, int c, int splits, int act, unsigned seed) {
    std::mt19937 gen(seed);
    TileBn ts;
    EXPECT(bn_ride_plan(groups, rows, c, true, &ts), "no backward plan for %d x %d x %d", groups, rows, c);
    const long elems = (long)groups * rows * c;
    Buf slabs((size_t)splits * elems), dy0(elems), dy1(elems), x(elems), gamma(c), beta(c), mean(groups * c), rstd(groups * c);
    slabs.fill(gen); x.fill(gen); gamma.fill(gen, 0.1f, 1.f); beta.fill(gen, 0.1f); mean.fill(gen, 0.1f); rstd.fill(gen, 0.05f, 1.f);
    dy1.nan();
    sum_slabs(slabs, splits, elems, nullptr, c, dy0);
    Buf p0((size_t)2 * groups * ts.nb * c), p1((size_t)2 * groups * ts.nb * c);
    p0.nan(); p1.nan();
    BnRide r = BnRide{2, reinterpret_cast<float2*>(p0.p), ts, x.p, gamma.p, beta.p, mean.p, rstd.p, act, 0.2f};
    EXPECT(launch_bn_ride(r, tensor_source(dy0.p), nullptr, "bwd alone") == 0, "%s", error_buffer());
    r.partial = reinterpret_cast<float2*>(p1.p);
    const RowSource src = RowSource{slabs.p, dy1.p, nullptr, elems, splits};
    EXPECT(launch_bn_ride(r, src, nullptr, "bwd ride") == 0, "%s", error_buffer());
    EXPECT(same_bits(dy0, dy1), "backward %d x %d x %d splits %d: gradient rows differ", groups, rows, c, splits);
    EXPECT(same_bits(p0, p1), "backward %d x %d x %d splits %d: sums differ", groups, rows, c, splits);
    Buf dx(elems), dg(c), db(c), colsum((size_t)groups * ts.nb * c);
    dx.nan(); colsum.nan();
    EXPECT(gan_bn_bwd_tiled(x.p, gamma.p, beta.p, mean.p, rstd.p, dy1.p, dx.p, dg.p, db.p, 0, groups, rows, c, act, 0.2f, p1.p, 1, colsum.p,
                            nullptr) == 0, "%s", error_buffer());
    // the partial column sums add up to the column sums of dx
    double worst = 0, scale = 0;
    for (int ch = 0; ch < c; ++ch) {
        double want = 0, got = 0, mag = 0;
        for (long rr = 0; rr < (long)groups * rows; ++rr) { want += dx.p[rr * c + ch]; mag += std::fabs(dx.p[rr * c + ch]); }
        for (int b = 0; b < groups * ts.nb; ++b) got += colsum.p[(long)b * c + ch];
        worst = std::max(worst, std::fabs(want - got));
        scale = std::max(scale, mag);
    }
    EXPECT(worst <= 1e-5 * scale, "backward %d x %d x %d: column sums off by %g of %g", groups, rows, c, worst, scale);
    // and gan_bn_bwd_tiled computing the sums itself gives the same bits
    Buf p2((size_t)2 * groups * ts.nb * c), dx2(elems), dg2(c), db2(c);
    p2.nan();
    EXPECT(gan_bn_bwd_tiled(x.p, gamma.p, beta.p, mean.p, rstd.p, dy0.p, dx2.p, dg2.p, db2.p, 0, groups, rows, c, act, 0.2f, p2.p, 0, nullptr,
                            nullptr) == 0, "%s", error_buffer());
    EXPECT(same_bits(dx, dx2) && same_bits(dg, dg2) && same_bits(db, db2), "backward %d x %d x %d: apply differs", groups, rows, c);
}

// The gradient of a tensor with two consumers: the two addends presented as two "slabs" whose distance is the distance
// of the two tensors (either sign), summed, stored and reduced in one launch.
static void fan_out_case(int groups, int rows, int c, bool b_first, unsigned seed) {
    std::mt19937 gen(seed);
    TileBn ts;
    EXPECT(bn_ride_plan(groups, rows, c, true, &ts), "no backward plan for %d x %d x %d", groups, rows, c);
    const long elems = (long)groups * rows * c;
    Buf a(elems), b(elems), sum0(elems), sum1(elems), x(elems), gamma(c), beta(c), mean(groups * c), rstd(groups * c);
    a.fill(gen); b.fill(gen); x.fill(gen); gamma.fill(gen, 0.1f, 1.f); beta.fill(gen, 0.1f); mean.fill(gen, 0.1f); rstd.fill(gen, 0.05f, 1.f);
    sum1.nan();
    for (long i = 0; i < elems; ++i) sum0.p[i] = b_first ? b.p[i] + a.p[i] : a.p[i] + b.p[i];
    Buf p0((size_t)2 * groups * ts.nb * c), p1((size_t)2 * groups * ts.nb * c);
    p0.nan(); p1.nan();
    BnRide r = BnRide{2, reinterpret_cast<float2*>(p0.p), ts, x.p, gamma.p, beta.p, mean.p, rstd.p, GAN_ACT_RELU, 0.2f};
    EXPECT(launch_bn_ride(r, tensor_source(sum0.p), nullptr, "sums alone") == 0, "%s", error_buffer());
    r.partial = reinterpret_cast<float2*>(p1.p);
    const float* first = b_first ? b.p : a.p;
    const float* second = b_first ? a.p : b.p;
    EXPECT(launch_bn_ride(r, RowSource{first, sum1.p, nullptr, (long)(second - first), 2}, nullptr, "fan-out") == 0, "%s", error_buffer());
    EXPECT(same_bits(sum0, sum1), "fan-out %d x %d x %d: sums of the addends differ", groups, rows, c);
    EXPECT(same_bits(p0, p1), "fan-out %d x %d x %d: block sums differ", groups, rows, c);
}

int main() {
    fan_out_case(1, 4096, 64, false, 21);
    fan_out_case(2, 1296, 128, true, 22);
    fan_out_case(3, 1300, 64, false, 23);
    forward_case(1, 2048, 64, 1, 1);
    forward_case(1, 1100, 128, 3, 2);
    forward_case(3, 1296, 64, 5, 3);           // row blocks of 62 rows: the row lanes take 3 or 4 rows each
    backward_case(1, 2048, 64, 1, GAN_ACT_NONE, 4);
    backward_case(4, 256, 128, 3, GAN_ACT_LRELU, 5);
    backward_case(3, 1300, 64, 2, GAN_ACT_RELU, 6);
    backward_case(2, 1296, 128, 1, GAN_ACT_NONE, 7);
    backward_case(2, 1296, 128, 4, GAN_ACT_TANH, 8);
    // long row blocks (128 and 600 rows)
    forward_case(1, 16384, 128, 2, 9);
    backward_case(2, 19200, 256, 2, GAN_ACT_RELU, 13);
    printf("%s: %d failure(s)\n", g_failures ? "FAILED" : "ok", g_failures);
    return g_failures ? 1 : 0;
}
