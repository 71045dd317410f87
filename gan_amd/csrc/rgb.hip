// Forward convolution with a 5-row filter over a 3-channel image (the first layer of every discriminator: dcgan.py:328,
// text2image.py:205, cyclegan_module.py:20) on the bf16 matrix pipe, without a slice pipeline.
//
// Such a layer is 0.6 GFLOP on 3 MB of input and 17 MB of output (DCGAN-64, batch 64): nothing to pipeline, and in the
// general kernel (igemm_split.h) its five 16-deep slices are a chain of dependent latencies.  Here a workgroup of four
// waves computes 128 output pixels x 64 channels in ONE pass:
//   * the gather axis is indexed per FILTER ROW as in the general kernel's filter-row gather: the kw * 3 floats a
//     filter row touches are one contiguous run in memory, padded to `spr` slices of 16 (the pad multiplies zeros), so
//     the MFMA k index is (filter row, slice, element) and an A fragment -- 8 consecutive k of one pixel -- is 8
//     consecutive floats, loaded as range-checked dwords straight from global memory into registers;
//   * every load of a pixel's fragments is issued before anything waits (5 x 8 dwords per lane for a 5 x 5 filter);
//   * the filter (kh * spr * 16 virtual rows x 64 columns) is split into its three bf16 parts ONCE per workgroup, by
//     plain vector loads issued right behind the image loads, and kept in LDS in fragment layout;
//   * fp32 values are split exactly into three bf16 parts and six of the nine part products accumulate in fp32, in the
//     order the general kernel uses (igemm_split.h).
// No split-K, no workspace.  Bias and the fused activation are applied on store.
// Measured inside the DCGAN-64 iteration: 15.0 us per batch-64 pass (10.5 us fixed + 4.5 us per 16.8 MB of output)
// against 21.5 on the general kernel.  A 14-k-step variant for the 7 x 7 stem of the CycleGAN generators (216 registers,
// 97 KB of LDS: one workgroup per CU) was slower than the general kernel there and is not kept.
#include "igemm_common.h"

namespace gan {

typedef __bf16 rgb_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned rgb_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kRgbOutOfRange = 0xfffffff0u;

struct RgbArgs {
    gan_conv_geom g;
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int M;                 // n * hs * ws output pixels
    int spr;               // 16-deep slices per filter row
    int run;               // kw * 3 floats of a filter row
    unsigned x_bytes, w_bytes;
    Epilogue epi;
};

// x -> its three bf16 parts (top 16 bits of x, of the remainder, of the second remainder: all exact)
__device__ __forceinline__ void rgb_parts(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u = __float_as_uint(x);
    const float r = x - __uint_as_float(u & 0xffff0000u);
    const unsigned v = __float_as_uint(r);
    const float s = r - __uint_as_float(v & 0xffff0000u);
    h = u >> 16; m = v >> 16; l = __float_as_uint(s) >> 16;
}

__device__ __forceinline__ void rgb_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    unsigned h0, m0, l0, h1, m1, l1;
    rgb_parts(x0, h0, m0, l0);
    rgb_parts(x1, h1, m1, l1);
    h = h0 | (h1 << 16); m = m0 | (m1 << 16); l = l0 | (l1 << 16);
}

// NKS = kh k-steps of 16 (one per filter row of up to 5 columns).
// LDS: three parts x (2 * NKS groups of 8 k) x (64 + 8 columns) x 16 bytes.
template <int NKS>
__global__ __launch_bounds__(256) void rgb_fprop_kernel(const RgbArgs p) {
    constexpr int SPR = 1;
    constexpr int GB = (64 + 8) * 16;          // bytes of one 8-deep group of one part (padded: conflict-free fragment reads)
    constexpr int PB = 2 * NKS * GB;           // one part
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * PB];
    const gan_conv_geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, lrow = lane & 31;
    const int m0 = blockIdx.x * 128 + wave * 32;
    const int n0 = blockIdx.y * 64;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

    // ---- the lane's pixel and its image fragments: 8 consecutive run floats per k-step
    const int m = m0 + lrow;
    const bool mok = m < p.M;
    const unsigned um = mok ? (unsigned)m : 0u;
    const int hw = g.hs * g.ws;
    const int img = um / (unsigned)hw;
    const int rem = um - img * hw;
    const int oy = rem / g.ws, ox = rem - oy * g.ws;
    const int iy0 = oy * g.sh - g.pt, ix0 = ox * g.sw - g.pl;
    float a_raw[NKS][8];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int ta = ks / SPR, sl = ks - ta * SPR;
        const int iy = iy0 + ta;
        const bool rowok = mok & ((unsigned)iy < (unsigned)g.hb);
        const int rowbase = (img * g.hb + iy) * g.wb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = sl * 16 + half * 8 + j;
            const int tc = (e * 11) >> 5;                    // e / 3 for e < 32
            const int col = ix0 + tc;
            const bool ok = rowok & (e < p.run) & ((unsigned)col < (unsigned)g.wb);
            const unsigned off = (unsigned)((rowbase + col) * 3 + (e - 3 * tc)) * 4u;
            a_raw[ks][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_x, (int)(ok ? off : kRgbOutOfRange), 0, 0));
        }
    }

    // ---- the filter: virtual row k = (filter row, run element), 64 columns; split once, fragment layout in LDS
#pragma unroll
    for (int i = 0; i < NKS; ++i) {
        const int u = tid + i * 256;                       // (k, column quad): NKS * 16 rows x 16 quads
        const int k = u >> 4, cq = u & 15;
        const int ta = k / (SPR * 16), e = k - ta * (SPR * 16);
        const bool ok = e < p.run;
        const unsigned off = (unsigned)((ta * p.run + e) * g.cs + n0 + cq * 4) * 4u;
        const rgb_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)(ok ? off : kRgbOutOfRange), 0, 0);
        unsigned char* q = smem + (k >> 3) * GB + (cq * 4) * 16 + (k & 7) * 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned h, mm, l;
            rgb_parts(__uint_as_float(v[c]), h, mm, l);
            *reinterpret_cast<unsigned short*>(q + c * 16) = (unsigned short)h;
            *reinterpret_cast<unsigned short*>(q + c * 16 + PB) = (unsigned short)mm;
            *reinterpret_cast<unsigned short*>(q + c * 16 + 2 * PB) = (unsigned short)l;
        }
    }
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        unsigned ah[4], am[4], al[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rgb_split_pair(a_raw[ks][2 * q], a_raw[ks][2 * q + 1], ah[q], am[q], al[q]);
        const rgb_bf16x8 a0 = __builtin_bit_cast(rgb_bf16x8, rgb_u32x4{ah[0], ah[1], ah[2], ah[3]});
        const rgb_bf16x8 a1 = __builtin_bit_cast(rgb_bf16x8, rgb_u32x4{am[0], am[1], am[2], am[3]});
        const rgb_bf16x8 a2 = __builtin_bit_cast(rgb_bf16x8, rgb_u32x4{al[0], al[1], al[2], al[3]});
        const unsigned char* bs = smem + (2 * ks + half) * GB + lrow * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const rgb_bf16x8 b0 = *reinterpret_cast<const rgb_bf16x8*>(bs + j * 32 * 16);
            const rgb_bf16x8 b1 = *reinterpret_cast<const rgb_bf16x8*>(bs + j * 32 * 16 + PB);
            const rgb_bf16x8 b2 = *reinterpret_cast<const rgb_bf16x8*>(bs + j * 32 * 16 + 2 * PB);
            f32x16 c = acc[j];
            // six of the nine part products, smallest first: l*h, h*l, m*m, m*h, h*m, h*h
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
            acc[j] = c;
        }
    }

    // ---- store: accumulator register r of a lane is row (r & 3) + 8 * (r >> 2) + 4 * half, column lane & 31.
    // The activation is the same for every element: one dispatch, outside the element loops.
    const bool full = m0 + 32 <= p.M;
    auto store_all = [&](auto fn) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + j * 32 + lrow;
            const float bv = p.bias != nullptr ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (full || row < p.M) {      // (`full` is uniform over the wave: no divergence in whole tiles)
                    const long idx = (long)row * g.cs + col;
                    p.y[idx] = fn(acc[j][r] + bv, idx);
                }
            }
        }
    };
    const Epilogue& e = p.epi;
    if (!epi_active(e)) store_all([](float v, long) { return v; });
    else if (e.act == GAN_ACT_LRELU) store_all([&](float v, long) { return fmaxf(v, e.param * v) * e.scale + e.shift; });
    else if (e.act == GAN_ACT_RELU) store_all([&](float v, long) { return fmaxf(v, 0.f) * e.scale + e.shift; });
    else store_all([&](float v, long idx) { return epi_apply(e, v, idx); });
}

// Whether the direct kernel serves this forward pass (3-channel image, output channels a multiple of 64, a filter of
// 5 or 14 k-steps: 5 x 5 and 7 x 7), and the launch.
bool rgb_fprop_ok(const IgemmArgs& a) {
    const gan_conv_geom& g = a.g;
    if (g.cb != 3 || g.cs % 64 != 0 || g.kw * 3 > 32 || a.accumulate || a.a_bytes == 0 || a.b_bytes == 0) return false;
    if (a.epi.gate != nullptr || a.reflect) return false;
    if ((reinterpret_cast<size_t>(a.b) & 15) != 0) return false;
    const int spr = (g.kw * 3 + 15) / 16;
    if (g.kh != 5 || spr != 1) return false;
    if ((long)g.n * g.hs * g.ws * g.cs >= (1L << 31) || (long)g.n * g.hb * g.wb * 3 >= (1L << 29)) return false;
    return true;
}

int run_rgb_fprop(const IgemmArgs& a, hipStream_t s, const char* what) {
    RgbArgs p;
    p.g = a.g;
    p.x = a.a; p.w = a.b; p.bias = a.bias; p.y = a.out;
    p.M = a.g.n * a.g.hs * a.g.ws;
    p.spr = (a.g.kw * 3 + 15) / 16;
    p.run = a.g.kw * 3;
    p.x_bytes = a.a_bytes; p.w_bytes = a.b_bytes;
    p.epi = a.epi;
    const dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)(a.g.cs / 64));
    hipLaunchKernelGGL((rgb_fprop_kernel<5>), grid, dim3(256), 0, s, p);
    return check_launch(what);
}

}  // namespace gan
