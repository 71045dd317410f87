// Shared by the contraction translation units (igemm.hip: fp32 MFMA kernel + host side; igemm_split_*.hip: the bf16
// matrix-pipe kernels of igemm_split.h, one translation unit per contraction mode so that they compile side by side).
#pragma once
#include <cstdint>

#include "common.h"

namespace gan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_FPROP = 0, MODE_TPROP = 1, MODE_WGRAD = 2 };
constexpr int BK = 16;
constexpr int kMaxClasses = 4;

// n / d for n*d < 2^32 by multiply-high with m = floor(2^32/d) + 1 (d >= 2); d == 1 is a copy.
struct FastDiv {
    unsigned d, m;
};

inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d ? d : 1;
    f.m = f.d > 1 ? (unsigned)((1ull << 32) / f.d) + 1u : 0u;
    return f;
}

// d == 1 has m == 0: the product term vanishes and the (d == 1) term passes n through, branch free
__device__ __forceinline__ unsigned fdiv(unsigned n, FastDiv f) { return __umulhi(n, f.m) + n * (unsigned)(f.d == 1u); }

struct ClassInfo {
    int ph, pw;    // (q + pad) mod stride of this class's output pixels
    int khc, kwc;  // taps of this class per axis
    int hq0, wq0;  // first quotient index whose pixel is inside the image
    int hq, wq;    // quotient extents
    int M, K;
    FastDiv d_kwc;
};

struct IgemmArgs {
    gan_conv_geom g;
    const float* a;
    const float* b;
    float* out;
    const float* bias;
    int M, N, K;  // FPROP / WGRAD problem; TPROP uses cls[].M / cls[].K and N = g.cb
    int splits, kt_per_split;
    long slab_elems;
    int accumulate;
    int vec_epilogue;  // rows of the output are 16-byte aligned and N % 4 == 0: float4 stores
    unsigned a_bytes, b_bytes;  // sizes of the two operand tensors (buffer range checks of the split kernel)
    int ncls;
    FastDiv d_cb, d_cs, d_kw, d_hsws, d_ws;
    ClassInfo cls[kMaxClasses];
    Epilogue epi;  // applied where the final value is stored: here when splits == 1, else by the slab reduction
    int reflect;   // FPROP / WGRAD: taps that fall outside the big image read its mirror image (tf.pad REFLECT
                   // by pt / pl in front of a VALID convolution) instead of zeros
    int row3_spr;  // > 0: filter-row gather of a big image whose channel count is no multiple of 4 (igemm_split.h, ROW3),
                   // slices per filter row
    int row3_mul;  // ceil(65536 / cb): (e * row3_mul) >> 16 = e / cb for the run elements e < 144
    int grid_mt, grid_nt, grid_z;  // row tiles, column tiles, (parity classes x k splits) of the one-dimensional grid
    int xcd_order;                 // tile_of_workgroup: 0 / 3 launch order (k range / row tile fastest), 1 / 2 XCD-aware
    // Workgroups beyond own_blocks carry the last stage of a bias-gradient column sum (its partial sums were left by
    // a launch ahead of this one).
    int own_blocks;
    ColsumFinal tail;
#ifdef GAN_PHASE_STAMPS
    long long* stamps;  // probe builds: [workgroup][8] clock stamps (igemm_split.h)
#endif
};


// Which tile a workgroup computes.  The grid is one-dimensional.  The hardware hands workgroups to the 8 XCDs
// round robin in launch order (position L runs on XCD L % 8) and every XCD has an L2 of its own, so tiles that
// read the same operand bytes should sit on ONE XCD at the same time.  Launch positions are therefore mapped to
// work indices such that each XCD receives one contiguous range of them, and the work index is decoded with the
// operand-sharing axis fastest:
//   order 1 (by row tile)  all parity classes, k splits and column tiles of one row tile are neighbours: they
//                  gather the same neighbourhood of the image, tap by tap, so the image crosses the fabric once
//                  and the filter once per XCD (FPROP / TPROP whose image outweighs the filter);
//   order 2 (by k range)   all row and column tiles of one (class, k split) are neighbours.  WGRAD: a k split is a
//                  range of images, which every filter tap reads, shifted.  FPROP / TPROP: a k split is a range of
//                  filter taps, read by every row tile, so a large filter crosses the fabric once.
// Orders 0 and 3 are plain launch orders (no XCD grouping) with the k range resp. the row tile fastest.
struct TileId { int mt, nt, z; };
template <int MODE>
__device__ __forceinline__ TileId tile_of_workgroup(const IgemmArgs& p) {
    const unsigned MT = (unsigned)p.grid_mt, NT = (unsigned)p.grid_nt, Z = (unsigned)p.grid_z;
    const unsigned L = blockIdx.x;
    unsigned w = L;
    if (p.xcd_order == 1 || p.xcd_order == 2) {
        const unsigned T = MT * NT * Z;
        const unsigned per = (T + 7u) >> 3;                  // work items of a "tall" XCD
        const unsigned tall = (T & 7u) ? (T & 7u) : 8u;      // XCDs that receive `per` items (the rest per - 1)
        const unsigned xcd = L & 7u, loc = L >> 3;
        w = xcd < tall ? xcd * per + loc : tall * per + (xcd - tall) * (per - 1u) + loc;
    }
    TileId t;
    if (MODE == MODE_WGRAD || p.xcd_order >= 2) {     // 3: row tile fastest in plain launch order
        t.mt = (int)(w % MT);
        const unsigned r = w / MT;
        t.nt = (int)(r % NT);
        t.z = (int)(r / NT);
    } else {
        t.z = (int)(w % Z);
        const unsigned r = w / Z;
        t.nt = (int)(r % NT);
        t.mt = (int)(r / NT);
    }
    return t;
}

// index i of an image row / column of n pixels extended by reflection without repeating the edge (|i| < n beyond it)
__device__ __forceinline__ int reflect_at(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// Launchers of igemm_split_kernel, one per (contraction mode m = 0 FPROP / 1 TPROP / 2 WGRAD, tile family p) and
// translation unit (igemm_split_tu.hip).  cfg: p0 {4: 128x128, 7: the same with two k groups per workgroup},
// p1 {5: 128x64, 8}, p2 {6: 64x64, 9: 128x32 (not TPROP)}.  slow_gather: general gather arithmetic even where the
// cheap one applies (test knob).
// rgb.hip: forward pass over a 3-channel image without a slice pipeline
bool rgb_fprop_ok(const IgemmArgs& a);
int run_rgb_fprop(const IgemmArgs& a, hipStream_t s, const char* what);

#define GAN_DECLARE_SPLIT(m, p) void launch_split_m##m##_p##p(const IgemmArgs& a, int cfg, bool slow_gather, dim3 grid, hipStream_t s)
GAN_DECLARE_SPLIT(0, 0); GAN_DECLARE_SPLIT(0, 1); GAN_DECLARE_SPLIT(0, 2);
GAN_DECLARE_SPLIT(1, 0); GAN_DECLARE_SPLIT(1, 1); GAN_DECLARE_SPLIT(1, 2);
GAN_DECLARE_SPLIT(2, 0); GAN_DECLARE_SPLIT(2, 1); GAN_DECLARE_SPLIT(2, 2);
#undef GAN_DECLARE_SPLIT

}  // namespace gan
