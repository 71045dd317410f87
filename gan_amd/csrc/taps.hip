// Stride-1 convolutions with FEW channels on one side (the 7 x 7 image head 64 -> 3 and the input gradient of the 7 x 7
// stem 3 -> 64 of the CycleGAN generators, cyclegan_module.py:124-125,143-144) as a dense product plus a sum over taps.
//
// With cs = 3 output channels the implicit-GEMM tile is 3 wide: 9 % of a 32-wide MFMA tile does work, and every input
// value is gathered and split 49 times.  Instead, with T = kh * kw taps:
//     Z[p, (t, o)] = sum_c x[p, c] * w[t, c, o]          a plain (pixels x cb) . (cb x T*cs) product: full tiles, x read once
//     y[p, o]      = bias[o] + sum_t Z[R(p + t), (t, o)]  a sum over the taps; R mirrors (REFLECT) or drops (zero padding)
// and for the gradients, with Pre(q, t) = { p : R(p + t) = q } (one pixel inside the image, up to two per axis at a
// mirrored border):
//     E[q, (t, o)] = sum_{p in Pre(q, t)} dy[p, o]        then  dw[t, c, o] = sum_q x[q, c] * E[q, (t, o)]   (dense)
//                                                          and   dx[q, c]   = sum_(t,o) E[q, (t, o)] * w[t, c, o] (dense)
//     dx[q, o]     = sum_t sum_{p in Pre(q, t)} Z[p, (t, o)]   (input gradient of the few-channel INPUT side)
// The dense products run on the contraction kernels (gan_linear_*); the kernels here are the tap sums and the filter
// repacking.  Rows of Z / E are `ld` floats long (T * cs rounded up to a multiple of 4; the pad columns hold zeros).
#include "common.h"

namespace gan {

struct TapGeom {
    int n, h, w;      // image
    int cs;           // channels of the thin side
    int kh, kw;       // filter; the padding is (kh - 1) / 2, (kw - 1) / 2 on each side
    int ld;           // floats per row of Z / E
    int reflect;      // 1: mirrored border, 0: zeros
};

__device__ __forceinline__ int tap_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// The pixels whose tap `t` (offset t - pad) lands on q: candidates[0..count).
__device__ __forceinline__ int tap_preimages(int q, int t, int pad, int n, int reflect, int (&cand)[3]) {
    int count = 0;
    const int d = pad - t;
    int p = q + d;                                   // the image pixel itself
    if (p >= 0 && p < n) cand[count++] = p;
    if (reflect) {
        if (q >= 1 && q <= pad) {                    // mirrored across the low border: index -q
            p = -q + d;
            if (p >= 0 && p < n) cand[count++] = p;
        }
        if (q <= n - 2 && q >= n - 1 - pad) {        // mirrored across the high border: index 2 (n - 1) - q
            p = 2 * (n - 1) - q + d;
            if (p >= 0 && p < n) cand[count++] = p;
        }
    }
    return count;
}

// wp[c][t * cs + o] = w[t * st + c * sc + o * so]; columns T * cs .. ld - 1 are zero.  (Forward head: w is
// [t][c][o]: st = cb * cs, sc = cs, so = 1.  Stem input gradient: w is [t][o][c]: st = cs * cb, sc = 1, so = cb.)
__global__ void taps_pack_filter_kernel(const float* __restrict__ w, float* __restrict__ wp, int taps, int cb, int cs, int ld,
                                        int st, int sc, int so) {
    const int total = cb * ld;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i / ld, j = i - c * ld;
        const int t = j / cs, o = j - t * cs;
        wp[i] = t < taps ? w[t * st + c * sc + o * so] : 0.f;
    }
}

// dw[t * st + c * sc + o * so] (+)= dwp[c][t * cs + o]; db[o] (+)= dbp[centre tap * cs + o] (db may be nullptr)
__global__ void taps_unpack_grad_kernel(const float* __restrict__ dwp, const float* __restrict__ dbp, float* __restrict__ dw,
                                        float* __restrict__ db, int taps, int cb, int cs, int ld, int st, int sc, int so,
                                        int centre, int accumulate) {
    const int total = cb * taps * cs;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i / (taps * cs), j = i - c * (taps * cs);
        const int t = j / cs, o = j - t * cs;
        const int k = t * st + c * sc + o * so;
        dw[k] = (accumulate ? dw[k] : 0.f) + dwp[c * ld + j];
    }
    if (db != nullptr && blockIdx.x == 0 && (int)threadIdx.x < cs)
        db[threadIdx.x] = (accumulate ? db[threadIdx.x] : 0.f) + dbp[centre * cs + threadIdx.x];
}

// The two tap sums read T * CS values per pixel from T different rows of Z: one work-item per pixel is a chain of T
// dependent-latency loads on 1/8 of the chip's wave slots.  A workgroup therefore takes 32 pixels x 8 filter-row
// slots (work-item = (slot, pixel); slot s adds the filter rows s, s + 8, ...), and the slot sums meet in LDS in
// slot order: deterministic, 8 times the loads in flight.
constexpr int kTapPixels = 32, kTapSlots = 8;

template <int CS>
__device__ __forceinline__ void tap_slots_finish(float (&acc)[CS], const float* __restrict__ bias, float* __restrict__ out, long p,
                                                 bool pvalid) {
    __shared__ float red[kTapSlots][kTapPixels][CS];
    const int pl = threadIdx.x % kTapPixels, slot = threadIdx.x / kTapPixels;
#pragma unroll
    for (int o = 0; o < CS; ++o) red[slot][pl][o] = acc[o];
    __syncthreads();
    if (slot == 0 && pvalid) {
#pragma unroll
        for (int o = 0; o < CS; ++o) {
            float v = bias != nullptr ? bias[o] : 0.f;
            for (int sl = 0; sl < kTapSlots; ++sl) v += red[sl][pl][o];
            out[p * CS + o] = v;
        }
    }
}

// y[p, o] = bias[o] + sum_t Z[R(p + t), (t, o)]
template <int CS>
__global__ __launch_bounds__(kTapPixels * kTapSlots) void taps_sum_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                                            float* __restrict__ y, const TapGeom g) {
    const long pixels = (long)g.n * g.h * g.w;
    const int pl = threadIdx.x % kTapPixels, slot = threadIdx.x / kTapPixels;
    const long p = (long)blockIdx.x * kTapPixels + pl;
    const bool pvalid = p < pixels;
    const long pc = pvalid ? p : 0;
    const int px = (int)(pc % g.w);
    const long r = pc / g.w;
    const int py = (int)(r % g.h);
    const long img = r / g.h;
    const int pady = (g.kh - 1) / 2, padx = (g.kw - 1) / 2;
    float acc[CS];
#pragma unroll
    for (int o = 0; o < CS; ++o) acc[o] = 0.f;
    for (int ty = slot; ty < g.kh && pvalid; ty += kTapSlots) {
        int sy = py + ty - pady;
        if (g.reflect) sy = tap_reflect(sy, g.h);
        if (sy < 0 || sy >= g.h) continue;
        for (int tx = 0; tx < g.kw; ++tx) {
            int sx = px + tx - padx;
            if (g.reflect) sx = tap_reflect(sx, g.w);
            if (sx < 0 || sx >= g.w) continue;
            const float* src = z + ((img * g.h + sy) * g.w + sx) * g.ld + (ty * g.kw + tx) * CS;
#pragma unroll
            for (int o = 0; o < CS; ++o) acc[o] += src[o];
        }
    }
    tap_slots_finish<CS>(acc, bias, y, p, pvalid);
}

// dx[q, o] = sum_t sum_{p in Pre(q, t)} Z[p, (t, o)]
template <int CS>
__global__ __launch_bounds__(kTapPixels * kTapSlots) void taps_sum_bwd_kernel(const float* __restrict__ z, float* __restrict__ dx,
                                                                            const TapGeom g) {
    const long pixels = (long)g.n * g.h * g.w;
    const int pl = threadIdx.x % kTapPixels, slot = threadIdx.x / kTapPixels;
    const long q = (long)blockIdx.x * kTapPixels + pl;
    const bool qvalid = q < pixels;
    const long qc = qvalid ? q : 0;
    const int qx = (int)(qc % g.w);
    const long r = qc / g.w;
    const int qy = (int)(r % g.h);
    const long img = r / g.h;
    const int pady = (g.kh - 1) / 2, padx = (g.kw - 1) / 2;
    float acc[CS];
#pragma unroll
    for (int o = 0; o < CS; ++o) acc[o] = 0.f;
    for (int ty = slot; ty < g.kh && qvalid; ty += kTapSlots) {
        int ys[3];
        const int ny = tap_preimages(qy, ty, pady, g.h, g.reflect, ys);
        for (int tx = 0; tx < g.kw; ++tx) {
            int xs[3];
            const int nx = tap_preimages(qx, tx, padx, g.w, g.reflect, xs);
            for (int a = 0; a < ny; ++a)
                for (int b = 0; b < nx; ++b) {
                    const float* src = z + ((img * g.h + ys[a]) * g.w + xs[b]) * g.ld + (ty * g.kw + tx) * CS;
#pragma unroll
                    for (int o = 0; o < CS; ++o) acc[o] += src[o];
                }
        }
    }
    tap_slots_finish<CS>(acc, nullptr, dx, q, qvalid);
}

// E[q, (t, o)] = sum over Pre(q, t) of dy[p, o]: one work-item per (pixel, tap slot); the slot behind the last tap
// writes the zero pad columns of the row.
__global__ __launch_bounds__(256) void taps_spread_bwd_kernel(const float* __restrict__ dy, float* __restrict__ e, const TapGeom g) {
    const int taps = g.kh * g.kw;
    const int slots = taps + (g.ld > taps * g.cs ? 1 : 0);
    const long total = (long)g.n * g.h * g.w * slots;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = (int)(i % slots);
    const long q = i / slots;
    float* row = e + q * g.ld;
    if (t == taps) {
        for (int j = taps * g.cs; j < g.ld; ++j) row[j] = 0.f;
        return;
    }
    const int qx = (int)(q % g.w);
    const long r = q / g.w;
    const int qy = (int)(r % g.h);
    const long img = r / g.h;
    const int ty = t / g.kw, tx = t - ty * g.kw;
    int ys[3], xs[3];
    const int ny = tap_preimages(qy, ty, (g.kh - 1) / 2, g.h, g.reflect, ys);
    const int nx = tap_preimages(qx, tx, (g.kw - 1) / 2, g.w, g.reflect, xs);
    for (int o = 0; o < g.cs; ++o) {
        float s = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) s += dy[((img * g.h + ys[a]) * g.w + xs[b]) * g.cs + o];
        row[t * g.cs + o] = s;
    }
}

static int tap_geom_check(const char* what, const TapGeom& g) {
    GAN_CHECK_ARG(g.n > 0 && g.h > 0 && g.w > 0 && g.cs > 0 && g.kh > 0 && g.kw > 0, "%s: sizes must be positive", what);
    GAN_CHECK_ARG((g.kh & 1) && (g.kw & 1), "%s: odd filter sizes only (symmetric padding)", what);
    GAN_CHECK_ARG(g.cs <= 4, "%s: at most 4 channels on the thin side", what);
    GAN_CHECK_ARG(g.ld >= g.kh * g.kw * g.cs && (g.ld & 3) == 0, "%s: rows of %d floats cannot hold %d x %d x %d columns in multiples of 4",
                  what, g.ld, g.kh, g.kw, g.cs);
    GAN_CHECK_ARG(!g.reflect || ((g.kh - 1) / 2 < g.h && (g.kw - 1) / 2 < g.w), "%s: REFLECT padding must be smaller than the image", what);
    GAN_CHECK_ARG((long)g.n * g.h * g.w * g.ld < (1L << 40) && ((long)g.n * g.h * g.w * (g.kh * g.kw + 1) + 255) / 256 < (1L << 31),
                  "%s: tensor too large", what);
    return 0;
}

static unsigned blocks_for(long items) { return (unsigned)((items + 255) / 256); }

}  // namespace gan

using namespace gan;

extern "C" int gan_taps_pack_filter(const float* w, float* wp, int taps, int cb, int cs, int ld, int transposed,
                                    gan_stream_t stream) {
    GAN_CHECK_ARG(w && wp && taps > 0 && cb > 0 && cs > 0 && ld >= taps * cs, "gan_taps_pack_filter: bad arguments");
    const int st = cb * cs, sc = transposed ? 1 : cs, so = transposed ? cb : 1;
    hipLaunchKernelGGL(taps_pack_filter_kernel, dim3(blocks_for((long)cb * ld)), dim3(256), 0, as_stream(stream), w, wp, taps, cb, cs, ld,
                       st, sc, so);
    return check_launch("gan_taps_pack_filter");
}

extern "C" int gan_taps_unpack_filter_grad(const float* dwp, const float* dbp, float* dw, float* db, int taps, int cb, int cs,
                                           int ld, int transposed, int accumulate, gan_stream_t stream) {
    GAN_CHECK_ARG(dwp && dw && taps > 0 && cb > 0 && cs > 0 && cs <= 256 && ld >= taps * cs, "gan_taps_unpack_filter_grad: bad arguments");
    GAN_CHECK_ARG(db == nullptr || dbp != nullptr, "gan_taps_unpack_filter_grad: db needs the column sums");
    GAN_CHECK_ARG(taps & 1, "gan_taps_unpack_filter_grad: odd tap count (a centre tap) expected");
    const int st = cb * cs, sc = transposed ? 1 : cs, so = transposed ? cb : 1;
    hipLaunchKernelGGL(taps_unpack_grad_kernel, dim3(blocks_for((long)cb * taps * cs)), dim3(256), 0, as_stream(stream), dwp, dbp, dw, db,
                       taps, cb, cs, ld, st, sc, so, taps / 2, accumulate);
    return check_launch("gan_taps_unpack_filter_grad");
}

#define GAN_TAPS_CS_DISPATCH(KERNEL, cs, ...)                                                               \
    switch (cs) {                                                                                            \
        case 1: hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__); break;                                         \
        case 2: hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__); break;                                         \
        case 3: hipLaunchKernelGGL((KERNEL<3>), __VA_ARGS__); break;                                         \
        default: hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__); break;                                        \
    }

extern "C" int gan_taps_sum_fwd(const float* z, const float* bias, float* y, int n, int h, int w, int cs, int kh, int kw,
                                int ld, int reflect, gan_stream_t stream) {
    const TapGeom g = TapGeom{n, h, w, cs, kh, kw, ld, reflect ? 1 : 0};
    if (int rc = tap_geom_check("gan_taps_sum_fwd", g)) return rc;
    GAN_CHECK_ARG(z && y, "gan_taps_sum_fwd: NULL tensor");
    const dim3 grid((unsigned)(((long)n * h * w + kTapPixels - 1) / kTapPixels));
    GAN_TAPS_CS_DISPATCH(taps_sum_fwd_kernel, cs, grid, dim3(kTapPixels * kTapSlots), 0, as_stream(stream), z, bias, y, g);
    return check_launch("gan_taps_sum_fwd");
}

extern "C" int gan_taps_sum_bwd(const float* z, float* dx, int n, int h, int w, int cs, int kh, int kw, int ld, int reflect,
                                gan_stream_t stream) {
    const TapGeom g = TapGeom{n, h, w, cs, kh, kw, ld, reflect ? 1 : 0};
    if (int rc = tap_geom_check("gan_taps_sum_bwd", g)) return rc;
    GAN_CHECK_ARG(z && dx, "gan_taps_sum_bwd: NULL tensor");
    const dim3 grid((unsigned)(((long)n * h * w + kTapPixels - 1) / kTapPixels));
    GAN_TAPS_CS_DISPATCH(taps_sum_bwd_kernel, cs, grid, dim3(kTapPixels * kTapSlots), 0, as_stream(stream), z, dx, g);
    return check_launch("gan_taps_sum_bwd");
}

extern "C" int gan_taps_spread_bwd(const float* dy, float* e, int n, int h, int w, int cs, int kh, int kw, int ld, int reflect,
                                   gan_stream_t stream) {
    const TapGeom g = TapGeom{n, h, w, cs, kh, kw, ld, reflect ? 1 : 0};
    if (int rc = tap_geom_check("gan_taps_spread_bwd", g)) return rc;
    GAN_CHECK_ARG(dy && e, "gan_taps_spread_bwd: NULL tensor");
    const int slots = kh * kw + (ld > kh * kw * cs ? 1 : 0);
    const dim3 grid(blocks_for((long)n * h * w * slots));
    hipLaunchKernelGGL(taps_spread_bwd_kernel, grid, dim3(256), 0, as_stream(stream), dy, e, g);
    return check_launch("gan_taps_spread_bwd");
}
