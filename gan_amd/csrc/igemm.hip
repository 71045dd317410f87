// Implicit-GEMM convolution family on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// All six TensorFlow conv passes the reference's ops.py needs, plus the three
// matmul passes of ops.linear, reduce to three contraction shapes over the
// geometry {big image, small image, filter [kh][kw][cb][cs]}:
//
//   FPROP  out_small[p, n]   = sum_{tap,c} big[p*s + tap - pad, c] * w[tap][c][n]
//          (conv2d forward, conv2d_transpose dgrad, linear forward)
//   TPROP  out_big[q, n]     = sum_{tap,c} small[(q + pad - tap)/s, c] * w[tap][n][c]
//          (conv2d dgrad, conv2d_transpose forward, linear dgrad); for stride 2 the
//          output pixels are split into s*s parity classes, each a dense stride-1 problem
//   WGRAD  dw[tap][cb][cs]   = sum_p big[p*s + tap - pad, cb] * small[p, cs]
//          (conv2d wgrad, conv2d_transpose wgrad, linear wgrad)
//
// One kernel template serves the three: a 256-thread block (4 waves) owns a BM x BN
// output tile, stages BK=16-deep operand slices through LDS as [k][m] / [k][n] images
// (double buffered, global loads for slice t+1 in flight while slice t is multiplied),
// and each wave accumulates TM x TN 32x32 MFMA tiles.  K can be split over blockIdx.z;
// partial tiles then go to output-shaped slabs that a second kernel sums in fixed order
// (deterministic, no float atomics).
//
// The gather index arithmetic is division free inside the k loop: k -> (tap, channel) and
// pixel -> (image, row, column) use multiply-high by host-computed reciprocals, and every
// global load is unconditional (masked lanes read element 0 and are zeroed by a select), so
// the loop body is one basic block the compiler can interleave with the MFMAs.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>

#include "common.h"

#include "igemm_common.h"
#include "bn_tile.h"

namespace gan {

// Raw gather load: masked-off lanes read element 0 of the tensor (always mapped); the zeroing
// select happens when the registers are written to LDS one slice later, so nothing touches the
// loaded registers -- and no vmcnt wait is needed -- while the loads are in flight.
template <int V>
__device__ __forceinline__ void load_vec(float (&dst)[V], const float* base, long off, bool ok) {
    const float* p = base + (ok ? off : 0L);
    if constexpr (V == 4) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    } else {
        dst[0] = *p;
    }
}

template <int MODE, int BM, int BN, int WM, int WN, int VA, int VB>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs p) {
    static_assert(WM * WN == 4, "four waves per block");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 MFMA tile");
    constexpr bool A_TRANS = (MODE != MODE_WGRAD);  // A source contiguous along k
    constexpr bool B_TRANS = (MODE == MODE_TPROP);  // B source contiguous along k
    // [k][m] images.  A transposing fill writes one dword per lane down a column: a leading
    // dimension = 2 mod 8 spreads the 4 k-chunks of a row over distinct banks.  A direct fill
    // writes 16-byte vectors along the row and only needs 16-byte row alignment.
    constexpr int LDA = BM + (A_TRANS ? 2 : 4);
    constexpr int LDB = BN + (B_TRANS ? 2 : 4);
    constexpr int STAGE = BK * (LDA + LDB);
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const TileId tile = tile_of_workgroup<MODE>(p);
    const int split = tile.z % p.splits;
    const int ci = tile.z / p.splits;
    const gan_conv_geom& g = p.g;

    int M, N, K;
    ClassInfo cl = p.cls[0];
    if constexpr (MODE == MODE_TPROP) {
        // static selects: a runtime index into the by-value argument would spill it to scratch
        if (ci == 1) cl = p.cls[1];
        else if (ci == 2) cl = p.cls[2];
        else if (ci == 3) cl = p.cls[3];
        M = cl.M; K = cl.K; N = g.cb;
    } else {
        M = p.M; N = p.N; K = p.K;
    }
    const int m0 = tile.mt * BM;
    if (m0 >= M) return;  // block-uniform: smaller parity classes have fewer row tiles
    const int n0 = tile.nt * BN;
    const int nkt = (K + BK - 1) / BK;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(nkt, kt0 + p.kt_per_split);
    const int kend = min(K, kt1 * BK);  // this block's k range is [kt0*BK, kend)

    // ---------------------------------------------------------------- A loader set-up
    constexpr int A_CPR = A_TRANS ? BK / VA : BM / VA;  // chunks per tile row (k-row for direct)
    constexpr int A_STEP = 256 / A_CPR;                 // tile rows covered per pass
    constexpr int A_EXT = A_TRANS ? BM : BK;            // tile rows to cover
    constexpr int A_PASS = (A_EXT + A_STEP - 1) / A_STEP;
    const int a_chunk = tid % A_CPR, a_r = tid / A_CPR;
    // gather source (FPROP: big image; TPROP: small image; WGRAD: big image)
    const int SH = (MODE == MODE_TPROP) ? g.hs : g.hb;
    const int SW = (MODE == MODE_TPROP) ? g.ws : g.wb;
    const int SC = (MODE == MODE_TPROP) ? g.cs : g.cb;
    const FastDiv dSC = (MODE == MODE_TPROP) ? p.d_cs : p.d_cb;
    const FastDiv dTW = (MODE == MODE_TPROP) ? cl.d_kwc : p.d_kw;  // taps per filter row in the k decode

    int a_base[A_PASS], a_y0[A_PASS], a_x0[A_PASS];  // A_TRANS: per output-pixel row
    int a_kh = 0, a_kw = 0, a_c = 0;                 // WGRAD: this thread's (tap, channel)
    bool a_mvalid = true;
    if constexpr (A_TRANS) {
        const int row_w = (MODE == MODE_TPROP) ? cl.wq : g.ws;
        const int row_hw = (MODE == MODE_TPROP) ? cl.hq * cl.wq : g.hs * g.ws;
#pragma unroll
        for (int ps = 0; ps < A_PASS; ++ps) {
            const int m = m0 + a_r + ps * A_STEP;
            const bool ok = (a_r + ps * A_STEP < BM) && (m < M);
            const unsigned um = ok ? (unsigned)m : 0u;
            const int img = um / (unsigned)row_hw;
            const int rem = um - img * row_hw;
            const int y = rem / row_w, x = rem - y * row_w;
            a_base[ps] = ok ? img * SH * SW : -1;
            if constexpr (MODE == MODE_TPROP) {
                a_y0[ps] = cl.hq0 + y;
                a_x0[ps] = cl.wq0 + x;
            } else {
                a_y0[ps] = y * g.sh - g.pt;
                a_x0[ps] = x * g.sw - g.pl;
            }
        }
    } else {
        const int m = m0 + a_chunk * VA;
        a_mvalid = m < M;
        const unsigned um = a_mvalid ? (unsigned)m : 0u;
        const int tap = um / (unsigned)g.cb;
        a_c = um - tap * g.cb;
        a_kh = tap / g.kw;
        a_kw = tap - a_kh * g.kw;
    }

    // ---------------------------------------------------------------- B loader set-up
    constexpr int B_CPR = B_TRANS ? BK / VB : BN / VB;
    constexpr int B_STEP = 256 / B_CPR;
    constexpr int B_EXT = B_TRANS ? BN : BK;
    constexpr int B_PASS = (B_EXT + B_STEP - 1) / B_STEP;
    const int b_chunk = tid % B_CPR, b_r = tid / B_CPR;

    float a_reg[A_PASS][VA], b_reg[B_PASS][VB];
    bool a_ok[A_PASS], b_ok[B_PASS];  // validity of the slice held in a_reg / b_reg

    auto load_a = [&](int kt) {
        if constexpr (A_TRANS) {
            const int k = kt * BK + a_chunk * VA;
            const bool kv = k < kend;
            const unsigned uk = kv ? (unsigned)k : 0u;
            const int tapc = fdiv(uk, dSC);
            const int cc = uk - tapc * SC;
            const int ta = fdiv(tapc, dTW);
            const int tb = tapc - ta * (int)dTW.d;
#pragma unroll
            for (int ps = 0; ps < A_PASS; ++ps) {
                int iy = (MODE == MODE_TPROP) ? a_y0[ps] - ta : a_y0[ps] + ta;
                int ix = (MODE == MODE_TPROP) ? a_x0[ps] - tb : a_x0[ps] + tb;
                if (MODE == MODE_FPROP && p.reflect) { iy = reflect_at(iy, SH); ix = reflect_at(ix, SW); }
                const bool ok = kv && a_base[ps] >= 0 && (unsigned)iy < (unsigned)SH && (unsigned)ix < (unsigned)SW;
                const long off = ((long)(a_base[ps] + iy * SW + ix)) * SC + cc;
                load_vec<VA>(a_reg[ps], p.a, off, ok);
                a_ok[ps] = ok;
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < A_PASS; ++ps) {
                const int kr = a_r + ps * A_STEP;
                const int k = kt * BK + kr;
                const bool kv = (kr < BK) && (k < kend) && a_mvalid;
                const unsigned uk = kv ? (unsigned)k : 0u;
                const int img = fdiv(uk, p.d_hsws);
                const int rem = uk - img * (int)p.d_hsws.d;
                const int oy = fdiv(rem, p.d_ws);
                const int ox = rem - oy * g.ws;
                int iy = oy * g.sh + a_kh - g.pt, ix = ox * g.sw + a_kw - g.pl;
                if (p.reflect) { iy = reflect_at(iy, g.hb); ix = reflect_at(ix, g.wb); }
                const bool ok = kv && (unsigned)iy < (unsigned)g.hb && (unsigned)ix < (unsigned)g.wb;
                const long off = ((long)((img * g.hb + iy) * g.wb + ix)) * g.cb + a_c;
                load_vec<VA>(a_reg[ps], p.a, off, ok);
                a_ok[ps] = ok;
            }
        }
    };

    auto load_b = [&](int kt) {
        if constexpr (B_TRANS) {
            // w[tap][n][c]: contiguous along k = (class tap, c)
            const int k = kt * BK + b_chunk * VB;
            const bool kv = k < kend;
            const unsigned uk = kv ? (unsigned)k : 0u;
            const int tapc = fdiv(uk, p.d_cs);
            const int cc = uk - tapc * g.cs;
            const int ta = fdiv(tapc, cl.d_kwc);
            const int tb = tapc - ta * cl.kwc;
            const int wtap = (cl.ph + g.sh * ta) * g.kw + (cl.pw + g.sw * tb);
#pragma unroll
            for (int ps = 0; ps < B_PASS; ++ps) {
                const int nr = b_r + ps * B_STEP;
                const int n = n0 + nr;
                const bool ok = kv && (nr < BN) && (n < N);
                const long off = ((long)(wtap * g.cb + n)) * g.cs + cc;
                load_vec<VB>(b_reg[ps], p.b, off, ok);
                b_ok[ps] = ok;
            }
        } else {
            // row-major [K][N] (filter as [taps*cb][cs], or the small image as [pixels][cs])
            const int n = n0 + b_chunk * VB;
#pragma unroll
            for (int ps = 0; ps < B_PASS; ++ps) {
                const int kr = b_r + ps * B_STEP;
                const int k = kt * BK + kr;
                const bool ok = (kr < BK) && (k < kend) && (n < N);
                const long off = (long)k * N + n;
                load_vec<VB>(b_reg[ps], p.b, off, ok);
                b_ok[ps] = ok;
            }
        }
    };

    auto store_a = [&](int buf) {
        float* As = smem + buf * STAGE;
        if constexpr (A_TRANS) {
#pragma unroll
            for (int ps = 0; ps < A_PASS; ++ps) {
                const int r = a_r + ps * A_STEP;
                if (A_EXT % A_STEP == 0 || r < BM) {
#pragma unroll
                    for (int v = 0; v < VA; ++v) As[(a_chunk * VA + v) * LDA + r] = a_ok[ps] ? a_reg[ps][v] : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < A_PASS; ++ps) {
                const int kr = a_r + ps * A_STEP;
                if (A_EXT % A_STEP == 0 || kr < BK) {
                    if constexpr (VA == 4) {
                        *reinterpret_cast<float4*>(&As[kr * LDA + a_chunk * 4]) =
                            a_ok[ps] ? make_float4(a_reg[ps][0], a_reg[ps][1], a_reg[ps][2], a_reg[ps][3])
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        As[kr * LDA + a_chunk] = a_ok[ps] ? a_reg[ps][0] : 0.f;
                    }
                }
            }
        }
    };
    auto store_b = [&](int buf) {
        float* Bs = smem + buf * STAGE + BK * LDA;
        if constexpr (B_TRANS) {
#pragma unroll
            for (int ps = 0; ps < B_PASS; ++ps) {
                const int r = b_r + ps * B_STEP;
                if (B_EXT % B_STEP == 0 || r < BN) {
#pragma unroll
                    for (int v = 0; v < VB; ++v) Bs[(b_chunk * VB + v) * LDB + r] = b_ok[ps] ? b_reg[ps][v] : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < B_PASS; ++ps) {
                const int kr = b_r + ps * B_STEP;
                if (B_EXT % B_STEP == 0 || kr < BK) {
                    if constexpr (VB == 4) {
                        *reinterpret_cast<float4*>(&Bs[kr * LDB + b_chunk * 4]) =
                            b_ok[ps] ? make_float4(b_reg[ps][0], b_reg[ps][1], b_reg[ps][2], b_reg[ps][3])
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        Bs[kr * LDB + b_chunk] = b_ok[ps] ? b_reg[ps][0] : 0.f;
                    }
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_a = wm * (BM / WM) + (lane & 31);
    const int frag_b = wn * (BN / WN) + (lane & 31);
    const int kh2 = lane >> 5;

    // One k-pair step: fragments of rows kk, kk+1 of the LDS images (lanes 0-31 take row kk,
    // lanes 32-63 row kk+1: the A / B operand layout of the 32x32x2 MFMA).
    auto frag = [&](const float* As, const float* Bs, int kk, float (&af)[TM], float (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = As[(kk + kh2) * LDA + frag_a + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = Bs[(kk + kh2) * LDB + frag_b + j * 32];
    };
    auto mma = [&](const float (&af)[TM], const float (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    };

    if (kt0 < kt1) {
        // Software pipeline, one barrier per slice: while slice kt is multiplied out of LDS buffer
        // `buf`, slice kt+1 (already in registers) is written to the other buffer and the global
        // loads of slice kt+2 are issued, so they have a whole slice of MFMAs to land.  Inside a
        // slice the fragments of step s+1 are read before the MFMAs of step s are issued, and the
        // staging work is spread between the steps so that no gap between two MFMAs grows long.
        load_a(kt0);
        load_b(kt0);
        store_a(0);
        store_b(0);
        load_a(kt0 + 1);
        load_b(kt0 + 1);
        __syncthreads();
        int buf = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            const float* As = smem + buf * STAGE;
            const float* Bs = As + BK * LDA;
            float af0[TM], bf0[TN], af1[TM], bf1[TN];
            // The order below is pinned with sched_barrier: left alone, the scheduler undoes the
            // fragment prefetch and parks the staging work where its latency is exposed.
            frag(As, Bs, 0, af0, bf0);
            frag(As, Bs, 2, af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mma(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            frag(As, Bs, 4, af0, bf0);
            store_a(buf ^ 1);   // slice kt+1, loaded a whole slice ago (past the end: zeros, never read)
            __builtin_amdgcn_sched_barrier(0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            frag(As, Bs, 6, af1, bf1);
            store_b(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            frag(As, Bs, 8, af0, bf0);
            load_a(kt + 2);     // past the end: masked off
            __builtin_amdgcn_sched_barrier(0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            frag(As, Bs, 10, af1, bf1);
            load_b(kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            frag(As, Bs, 12, af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            frag(As, Bs, 14, af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mma(af0, bf0);
            mma(af1, bf1);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---------------------------------------------------------------- epilogue
    // Row -> output offset table (element offset of channel 0, or -1), built once per block.
    __shared__ int rowoff[BM];
    if (tid < BM) {
        const int m = m0 + tid;
        int off = -1;
        if (m < M) {
            if constexpr (MODE == MODE_TPROP) {
                const int hw = cl.hq * cl.wq;
                const int img = m / hw;
                const int rem = m - img * hw;
                const int yq = rem / cl.wq, xq = rem - yq * cl.wq;
                const int y = g.sh * (cl.hq0 + yq) + cl.ph - g.pt;
                const int x = g.sw * (cl.wq0 + xq) + cl.pw - g.pl;
                off = ((img * g.hb + y) * g.wb + x) * g.cb;
            } else {
                off = m * N;
            }
        }
        rowoff[tid] = off;
    }
    __syncthreads();

    float* dst = p.out + (p.splits > 1 ? (long)split * p.slab_elems : 0L);
    const bool direct = p.splits == 1;
    const bool accum = direct && p.accumulate;
    const bool epi_on = direct && epi_active(p.epi);

    if (p.vec_epilogue) {
        // The accumulators sit row-per-register; dword-per-lane stores of them are issue bound.
        // Transpose through LDS instead (the operand buffers are free now) and let every lane store
        // 16 bytes of one output row: R rows of the tile per pass.
        constexpr int R = ((2 * STAGE) / BN / 32) * 32 < BM ? ((2 * STAGE) / BN / 32) * 32 : BM;
        static_assert(R >= 32, "LDS staging must hold a 32-row group of the tile");
        constexpr int F4_ROW = BN / 4;
        const bool has_bias = direct && p.bias != nullptr;
#pragma unroll
        for (int r0 = 0; r0 < BM; r0 += R) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int gbase = wm * (BM / WM) + i * 32;     // first tile row of this 32-row group
                if (gbase >= r0 && gbase < r0 + R) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = gbase - r0 + (r & 3) + 8 * (r >> 2) + 4 * kh2;
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            smem[row * BN + wn * (BN / WN) + j * 32 + (lane & 31)] = acc[i][j][r];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int f = tid; f < R * F4_ROW; f += 256) {
                const int row = f / F4_ROW, c4 = f - row * F4_ROW;
                const int off = (r0 + row < BM) ? rowoff[r0 + row] : -1;
                const int col = n0 + c4 * 4;
                if (off >= 0 && col < N) {
                    float4 v = *reinterpret_cast<const float4*>(&smem[row * BN + c4 * 4]);
                    if (has_bias) {
                        const float4 bb = *reinterpret_cast<const float4*>(p.bias + col);
                        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                    }
                    float4* o = reinterpret_cast<float4*>(dst + off + col);
                    if (accum) {
                        const float4 old = *o;
                        v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                    }
                    if (epi_on) v = epi_apply4(p.epi, v, (long)off + col);
                    *o = v;
                }
            }
            if (r0 + R < BM) __syncthreads();
        }
        return;
    }

    int ncol[TN];
    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ncol[j] = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
        bias_v[j] = (direct && p.bias != nullptr && ncol[j] < N) ? p.bias[ncol[j]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh2;
            const int off = rowoff[row];
            if (off < 0) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (ncol[j] < N) {
                    float v = acc[i][j][r] + bias_v[j];
                    if (accum) v += dst[off + ncol[j]];
                    if (epi_on) v = epi_apply(p.epi, v, (long)off + ncol[j]);
                    dst[off + ncol[j]] = v;
                }
            }
        }
    }
}

// out[i] = (accumulate ? out[i] : 0) + bias[i % c] + sum_s slab[s][i].  The first `own_blocks` workgroups do
// that; any further ones carry the last stage of the bias-gradient column sum (`tail`).
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                    const float* __restrict__ bias, long elems, int c, int splits,
                                    int accumulate, int own_blocks, const ColsumFinal tail, const Epilogue epi) {
    const bool epi_on = epi_active(epi);
    if ((int)blockIdx.x >= own_blocks) {
        colsum_final_block(tail, blockIdx.x - own_blocks);
        return;
    }
    const long stride = (long)own_blocks * blockDim.x;
    if ((elems & 3) == 0 && (c & 3) == 0) {
        const long n4 = elems >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 s = reinterpret_cast<const float4*>(slabs)[i];
            // (unrolled: the loads of several slabs are in flight together; the sum keeps its order)
#pragma unroll 4
            for (int k = 1; k < splits; ++k) {
                const float4 t = reinterpret_cast<const float4*>(slabs + (long)k * elems)[i];
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            if (bias) {
                const float4 bb = *reinterpret_cast<const float4*>(bias + (int)((i * 4) % c));
                s.x += bb.x; s.y += bb.y; s.z += bb.z; s.w += bb.w;
            }
            if (accumulate) {
                const float4 o = reinterpret_cast<const float4*>(out)[i];
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            if (epi_on) s = epi_apply4(epi, s, i * 4);
            reinterpret_cast<float4*>(out)[i] = s;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += stride) {
            float s = slabs[i];
#pragma unroll 4
            for (int k = 1; k < splits; ++k) s += slabs[(long)k * elems + i];
            if (bias) s += bias[(int)(i % c)];
            if (accumulate) s += out[i];
            if (epi_on) s = epi_apply(epi, s, i);
            out[i] = s;
        }
    }
}

// Same sum for many slabs over a small output (thin wgrad: hundreds of K splits): a block covers
// 64 consecutive elements as 16 float4 columns x 16 split lanes; lane j adds slabs j, j+16, ...
// and the 16 partial sums are combined through LDS in a fixed order.  Needs elems % 4 == 0.
__global__ __launch_bounds__(256) void reduce_slabs_wide_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                const float* __restrict__ bias, long elems, int c,
                                                                int splits, int accumulate, int own_blocks,
                                                                const ColsumFinal tail, const Epilogue epi) {
    __shared__ float4 red[16][16];
    if ((int)blockIdx.x >= own_blocks) {          // (uniform per workgroup: no barrier is skipped by part of one)
        colsum_final_block(tail, blockIdx.x - own_blocks);
        return;
    }
    const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i4 = (long)blockIdx.x * 16 + col;
    const long n4 = elems >> 2;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n4) {
#pragma unroll 4
        for (int k = sl; k < splits; k += 16) {
            const float4 t = reinterpret_cast<const float4*>(slabs + (long)k * elems)[i4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
    }
    red[sl][col] = s;
    __syncthreads();
    if (sl == 0 && i4 < n4) {
        for (int j = 1; j < 16; ++j) {
            const float4 t = red[j][col];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (bias) {
            const long e = i4 * 4;
            s.x += bias[(int)(e % c)]; s.y += bias[(int)((e + 1) % c)];
            s.z += bias[(int)((e + 2) % c)]; s.w += bias[(int)((e + 3) % c)];
        }
        if (accumulate) {
            const float4 o = reinterpret_cast<const float4*>(out)[i4];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        if (epi_active(epi)) s = epi_apply4(epi, s, i4 * 4);
        reinterpret_cast<float4*>(out)[i4] = s;
    }
}

// The same for outputs whose element count is not a multiple of 4 (a slab then starts at any dword: no 16-byte
// loads): a block covers 16 consecutive elements x 16 split lanes.
__global__ __launch_bounds__(256) void reduce_slabs_wide_scalar_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                       const float* __restrict__ bias, long elems, int c,
                                                                       int splits, int accumulate, int own_blocks,
                                                                       const ColsumFinal tail, const Epilogue epi) {
    __shared__ float red[16][17];
    if ((int)blockIdx.x >= own_blocks) {
        colsum_final_block(tail, blockIdx.x - own_blocks);
        return;
    }
    const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + col;
    float s = 0.f;
    if (i < elems) {
#pragma unroll 4
        for (int k = sl; k < splits; k += 16) s += slabs[(long)k * elems + i];
    }
    red[sl][col] = s;
    __syncthreads();
    if (sl == 0 && i < elems) {
        for (int j = 1; j < 16; ++j) s += red[j][col];
        if (bias) s += bias[(int)(i % c)];
        if (accumulate) s += out[i];
        if (epi_active(epi)) s = epi_apply(epi, s, i);
        out[i] = s;
    }
}

// Slab sum and the adjoint of tf.pad(REFLECT) in one pass: the slabs hold the gradient w.r.t. the reflect-padded image
// ((h + 2p) x (w + 2p)); every pixel of the unpadded image sums the (at most 3 x 3) padded positions that mirror onto it
// over all slabs, in a fixed order (rows, columns, slabs).
struct FoldReflect {
    int n, h, w, c, p;
};

__global__ void reduce_slabs_fold_reflect_kernel(const float* __restrict__ slabs, float* __restrict__ dx, const FoldReflect f,
                                                 int splits, long slab_elems) {
    const int hp = f.h + 2 * f.p, wp = f.w + 2 * f.p;
    const int cv = (f.c & 3) == 0 ? f.c >> 2 : f.c;   // channels, or channel quads moved 16 bytes at a time
    const long total = (long)f.n * f.h * f.w * cv;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)(i % cv);
        long r = i / cv;
        const int sx = (int)(r % f.w); r /= f.w;
        const int sy = (int)(r % f.h);
        const int img = (int)(r / f.h);
        // padded rows that map to sy: sy+p always; p-sy if 1 <= sy <= p; 2(h-1)-sy+p if h-1-p <= sy <= h-2
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = sy + f.p;
        if (sy >= 1 && sy <= f.p) ys[ny++] = f.p - sy;
        if (sy <= f.h - 2 && sy >= f.h - 1 - f.p) ys[ny++] = 2 * (f.h - 1) - sy + f.p;
        xs[nx++] = sx + f.p;
        if (sx >= 1 && sx <= f.p) xs[nx++] = f.p - sx;
        if (sx <= f.w - 2 && sx >= f.w - 1 - f.p) xs[nx++] = 2 * (f.w - 1) - sx + f.p;
        if ((f.c & 3) == 0) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int a = 0; a < ny; ++a)
                for (int b = 0; b < nx; ++b) {
                    const float4* src = reinterpret_cast<const float4*>(slabs + (((long)img * hp + ys[a]) * wp + xs[b]) * f.c) + ch;
#pragma unroll 4
                    for (int k = 0; k < splits; ++k) {
                        const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + (long)k * slab_elems);
                        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                    }
                }
            reinterpret_cast<float4*>(dx)[i] = s;
        } else {
            float s = 0.f;
            for (int a = 0; a < ny; ++a)
                for (int b = 0; b < nx; ++b) {
                    const float* src = slabs + (((long)img * hp + ys[a]) * wp + xs[b]) * f.c + ch;
                    for (int k = 0; k < splits; ++k) s += src[(long)k * slab_elems];
                }
            dx[i] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// TPROP with a thin output (cb <= 4: the image-producing deconv of a generator, the input
// gradient of a discriminator's first conv).  A 32-wide MFMA tile would be >= 87 % padding in N,
// so this path runs on the vector ALUs instead: one thread owns one output pixel and its cb
// accumulators, a block owns an 8 x 32 patch of one parity class of one image.  The patch's
// input halo is staged through LDS in 16-channel chunks (coalesced 64-byte segments per pixel);
// filter values are the same for every lane of the block (tap and channel are loop counters), so
// they arrive through the scalar data path and feed the FMAs as scalar operands.
// ------------------------------------------------------------------------------------------
constexpr int kThinQH = 8, kThinQW = 32, kThinCC = 16, kThinPix = kThinCC + 4;  // padded pixel stride in LDS
constexpr int kThinMaxTaps = 7;                                                  // per axis and class

struct ThinArgs {
    gan_conv_geom g;
    const float* in;    // small image [n][hs][ws][cs]
    const float* w;     // [kh][kw][cb][cs]
    float* out;         // big image [n][hb][wb][cb]
    const float* bias;  // [cb] or null
    int ncls, tiles_h, tiles_w;
    ClassInfo cls[kMaxClasses];
    Epilogue epi;
};

// MAXT bounds the taps per axis of any class and sizes the LDS halo (3: 5x5 stride 2 -> 27 KB, five
// blocks per CU; 7: up to 7x7 stride 1 -> 42.5 KB).
// TAIL: any channel count and alignment of the small image (the y-conditioned MNIST generator's last layer reads
// 138 channels): the halo is staged dword by dword with zeros beyond the last channel, and the filter values of a
// chunk are fetched one by one with the channel index range-checked, so nothing is read beyond a filter row.
template <int CB, int MAXT, bool TAIL>
__global__ __launch_bounds__(256) void tprop_thin_kernel(const ThinArgs p) {
    __shared__ __attribute__((aligned(16))) float tile[(kThinQH + MAXT - 1) * (kThinQW + MAXT - 1) * kThinPix];
    const gan_conv_geom& g = p.g;
    const int ci = blockIdx.y;
    // This block's class and, at the very end, the epilogue come straight from the kernel-argument segment (the one
    // by-value argument starts at its offset 0) by scalar loads where they are needed, instead of being held in
    // scalar registers from the kernel's entry on: the tap loop runs at the scalar-register limit.
    ClassInfo cl;
    Epilogue epi;
#if defined(__HIP_DEVICE_COMPILE__)
    const ThinArgs* ka = (const ThinArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    __builtin_memcpy(&cl, &ka->cls[ci], sizeof(cl));
#else
    cl = p.cls[0];
    epi = p.epi;
#endif
    int b = blockIdx.x;
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h;
    const int img = b / p.tiles_h;
    const int qh_t = th * kThinQH, qw_t = tw * kThinQW;  // patch origin in the class grid
    if (qh_t >= cl.hq || qw_t >= cl.wq) return;          // block-uniform: before any barrier
    const int khc = cl.khc, kwc = cl.kwc;
    const int RW = kThinQW + kwc - 1, RH = kThinQH + khc - 1;
    // input rows / columns of the halo: i = (hq0 + q) - ta, ta in [0, khc)
    const int i0 = cl.hq0 + qh_t - (khc - 1), j0 = cl.wq0 + qw_t - (kwc - 1);
    const int lqh = threadIdx.x >> 5, lqw = threadIdx.x & 31;
    const float* in_img = p.in + (long)img * g.hs * g.ws * g.cs;
    // accumulators in pairs (even k, odd k): the inner product runs on the packed fp32 FMA, two channels of the
    // chunk per instruction, with the filter pair as a scalar-register operand
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc2[c] = f32x2{0.f, 0.f};

    // thread -> (halo pixel, 4-channel group), 64 pixels further on every pass.  The pixel offsets are the same for
    // every channel chunk, so they are worked out once; the halo of chunk c0 + 16 is fetched into registers before
    // chunk c0 is multiplied, so its memory latency is hidden behind the FMAs instead of standing between two chunks.
    constexpr int NP = ((kThinQH + MAXT - 1) * (kThinQW + MAXT - 1) + 63) / 64;
    const int c4 = threadIdx.x & 3;
    int off[NP];   // element offset of the pixel in its image; -1: outside the image (zeros); -2: beyond the halo
    {
        int r = 0, cc = threadIdx.x >> 2;
#pragma unroll
        for (int t = 0; t < NP; ++t, cc += 64) {
            while (cc >= RW) { cc -= RW; ++r; }   // RW >= 32: at most two wraps per pass
            const int i = i0 + r, j = j0 + cc;
            const bool inside = i >= 0 && i < g.hs && j >= 0 && j < g.ws;
            off[t] = t * 64 + (int)(threadIdx.x >> 2) >= RH * RW ? -2 : (inside ? (i * g.ws + j) * g.cs + c4 * 4 : -1);
        }
    }
    float4 halo[NP];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (off[t] >= 0) {
                const float* src = in_img + off[t] + c0;
                if constexpr (!TAIL) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    const int left = g.cs - (c0 + c4 * 4);   // channels from this thread's first one to the last
                    if (left > 0) v.x = src[0];
                    if (left > 1) v.y = src[1];
                    if (left > 2) v.z = src[2];
                    if (left > 3) v.w = src[3];
                }
            }
            halo[t] = v;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < g.cs; c0 += kThinCC) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NP; ++t)
            if (off[t] != -2) *reinterpret_cast<float4*>(&tile[(t * 64 + (threadIdx.x >> 2)) * kThinPix + c4 * 4]) = halo[t];
        if (c0 + kThinCC < g.cs) fetch(c0 + kThinCC);
        __syncthreads();
        for (int ta = 0; ta < khc; ++ta)
            for (int tb = 0; tb < kwc; ++tb) {
                const float* xp = &tile[((lqh + khc - 1 - ta) * RW + (lqw + kwc - 1 - tb)) * kThinPix];
                f32x2 x[kThinCC / 2];
#pragma unroll
                for (int k = 0; k < kThinCC; k += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(xp + k);
                    x[k / 2] = f32x2{v.x, v.y};
                    x[k / 2 + 1] = f32x2{v.z, v.w};
                }
                // uniform address: tap (cl.ph + sh*ta, cl.pw + sw*tb), channels c0..c0+15
                const float* wp = p.w + ((long)((cl.ph + g.sh * ta) * g.kw + (cl.pw + g.sw * tb)) * g.cb) * g.cs + c0;
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    if constexpr (!TAIL) {
                        const f32x2* wq = reinterpret_cast<const f32x2*>(wp + (long)c * g.cs);   // cs and c0 are multiples of 16
#pragma unroll
                        for (int k = 0; k < kThinCC / 2; ++k) acc2[c] = __builtin_elementwise_fma(x[k], wq[k], acc2[c]);
                    } else {
                        const float* wr = wp + (long)c * g.cs;
                        const int left = g.cs - c0;   // uniform: channels of this chunk that exist
                        if (left >= kThinCC) {        // a whole chunk: 16 consecutive dwords, at any dword address
#pragma unroll
                            for (int k = 0; k < kThinCC / 2; ++k)
                                acc2[c] = __builtin_elementwise_fma(x[k], f32x2{wr[2 * k], wr[2 * k + 1]}, acc2[c]);
                        } else {
#pragma unroll
                            for (int k = 0; k < kThinCC / 2; ++k) {
                                const float w0 = 2 * k < left ? wr[2 * k] : 0.f;
                                const float w1 = 2 * k + 1 < left ? wr[2 * k + 1] : 0.f;
                                acc2[c] = __builtin_elementwise_fma(x[k], f32x2{w0, w1}, acc2[c]);
                            }
                        }
                    }
                }
            }
    }
    float acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = acc2[c].x + acc2[c].y;
    const int qh = qh_t + lqh, qw = qw_t + lqw;
    if (qh < cl.hq && qw < cl.wq) {
        const int h = (cl.hq0 + qh) * g.sh + cl.ph - g.pt, w = (cl.wq0 + qw) * g.sw + cl.pw - g.pl;
        float* o = p.out + (((long)img * g.hb + h) * g.wb + w) * g.cb;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_memcpy(&epi, &ka->epi, sizeof(epi));
#endif
        const bool epi_on = epi_active(epi);
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            float v = acc[c] + (p.bias ? p.bias[c] : 0.f);
            if (epi_on) v = epi_apply(epi, v, (o - p.out) + c);
            o[c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct TileCfg {
    int bm, bn;
};

struct Plan {
    int cfg;  // fp32 MFMA 0: 128x128, 1: 128x64, 2: 64x64, 3: 128x32; split-bf16 MFMA 4: 128x128, 5: 128x64, 6: 64x64,
              // 7: 128x128 and 8: 128x64 with split-K inside the workgroup (two groups of four waves per tile),
              // 9: 128x32 (outputs of a few channels in the forward and weight-gradient modes)
    int va, vb;
    int splits, kt_per_split;
    int mt, nt;
    long out_elems;
    int row3;   // slices per filter row of the 3-channel gather the launch uses (0: none)
    int order;  // pinned tile order (1 launch order, 2 by row tile, 3 by k range) or 0: chosen by tile_order()
};

static const TileCfg kCfg[10] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}, {128, 128}, {128, 64}, {64, 64},
                                 {128, 128}, {128, 64}, {128, 32}};
constexpr int kFirstSplitCfg = 4, kFirstKg2Cfg = 7, kThinSplitCfg = 9, kNumCfg = 10;
static bool is_kg2(int cfg) { return cfg == kFirstKg2Cfg || cfg == kFirstKg2Cfg + 1; }
constexpr int kOrderStride = 16;  // gan_conv_plan_set's cfg argument = tile configuration + 16 * tile order

// Process-wide switch: 1 keeps every contraction on the fp32 MFMA (bitwise an fmaf chain); 0 lets
// layers whose channel counts are multiples of 4 run on the bf16 matrix pipe with fp32 operands
// split into three bf16 parts (igemm_split.h).  GAN_HIP_EXACT_F32 in the environment sets the default.
// It selects plans and therefore workspace sizes: set it before the first query, not while another
// thread is inside a contraction call.
static std::atomic<int> g_exact_f32{-1};
static bool exact_f32() {
    int v = g_exact_f32.load(std::memory_order_relaxed);
    if (v < 0) {
        v = knobs().exact_f32 ? 1 : 0;
        g_exact_f32.store(v, std::memory_order_relaxed);
    }
    return v == 1;
}
void set_exact_f32(int on) { g_exact_f32.store(on ? 1 : 0, std::memory_order_relaxed); }
int get_exact_f32() { return exact_f32() ? 1 : 0; }

static int geom_check(const gan_conv_geom& g) {
    GAN_CHECK_ARG(g.n > 0 && g.hb > 0 && g.wb > 0 && g.cb > 0 && g.hs > 0 && g.ws > 0 && g.cs > 0,
                  "conv geometry: non-positive dimension");
    GAN_CHECK_ARG(g.kh > 0 && g.kw > 0 && g.kh <= 16 && g.kw <= 16, "conv geometry: filter taps must be in 1..16");
    GAN_CHECK_ARG((g.sh == 1 || g.sh == 2) && (g.sw == 1 || g.sw == 2), "conv geometry: strides must be 1 or 2");
    GAN_CHECK_ARG(g.pt >= 0 && g.pl >= 0 && g.pt < g.kh && g.pl < g.kw, "conv geometry: pad_before out of range");
    // the last filter row/column must not start beyond the padded image
    GAN_CHECK_ARG((g.hs - 1) * g.sh - g.pt < g.hb && (g.ws - 1) * g.sw - g.pl < g.wb,
                  "conv geometry: small side too large for big side");
    const long big = (long)g.n * g.hb * g.wb * g.cb, small = (long)g.n * g.hs * g.ws * g.cs;
    const long wel = (long)g.kh * g.kw * g.cb * g.cs;
    GAN_CHECK_ARG(big < (1L << 31) && small < (1L << 31) && wel < (1L << 31),
                  "conv geometry: tensors of 2^31 elements or more are not supported");
    // exactness range of the multiply-high index decodes (n * d < 2^32)
    const uint64_t lim = 1ull << 32;
    const uint64_t taps = (uint64_t)g.kh * g.kw;
    GAN_CHECK_ARG(taps * g.cb * g.cb < lim && taps * g.cs * g.cs < lim,
                  "conv geometry: taps * channels^2 must stay below 2^32");
    const uint64_t hw = (uint64_t)g.hs * g.ws;
    GAN_CHECK_ARG((uint64_t)g.n * hw * hw < lim || (g.n == 1 && (hw - 1) * hw < lim),
                  "conv geometry: batch * (small height * width)^2 must stay below 2^32");
    return 0;
}

static void build_classes(const gan_conv_geom& g, IgemmArgs& a) {
    a.ncls = 0;
    for (int ph = 0; ph < g.sh; ++ph)
        for (int pw = 0; pw < g.sw; ++pw) {
            ClassInfo c;
            c.ph = ph; c.pw = pw;
            c.khc = ph < g.kh ? (g.kh - ph + g.sh - 1) / g.sh : 0;
            c.kwc = pw < g.kw ? (g.kw - pw + g.sw - 1) / g.sw : 0;
            c.hq0 = g.pt > ph ? (g.pt - ph + g.sh - 1) / g.sh : 0;
            c.wq0 = g.pl > pw ? (g.pl - pw + g.sw - 1) / g.sw : 0;
            const int hmax = g.hb - 1 + g.pt - ph, wmax = g.wb - 1 + g.pl - pw;
            c.hq = hmax >= 0 ? hmax / g.sh - c.hq0 + 1 : 0;
            c.wq = wmax >= 0 ? wmax / g.sw - c.wq0 + 1 : 0;
            if (c.hq < 0) c.hq = 0;
            if (c.wq < 0) c.wq = 0;
            c.M = g.n * c.hq * c.wq;
            c.K = c.khc * c.kwc * g.cs;
            if (c.kwc == 0) c.kwc = 1;  // keeps the k decode defined; K is 0 anyway
            c.d_kwc = make_fastdiv((unsigned)c.kwc);
            a.cls[a.ncls++] = c;
        }
}

// Pinned plans: (mode, geometry) -> (tile configuration, split-K factor), filled by the host-side
// tuner.  Consulted by make_plan after the heuristic and before the environment overrides.
struct PlanKey {
    int v[14];
    bool operator<(const PlanKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};

static PlanKey plan_key(int mode, const gan_conv_geom& g) {
    PlanKey k;
    const int f[14] = {mode, g.n, g.hb, g.wb, g.cb, g.hs, g.ws, g.cs, g.kh, g.kw, g.sh, g.sw, g.pt, g.pl};
    memcpy(k.v, f, sizeof(f));
    return k;
}

static std::map<PlanKey, std::pair<int, int>>& plan_table() {
    static std::map<PlanKey, std::pair<int, int>> t;
    return t;
}
static std::mutex g_plan_mu;

static bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// The split kernel's loaders move 16 / 8 bytes at a time: both gathered channel counts must be multiples
// of 4 and both operands below 4 GB (a_bytes / b_bytes are 0 beyond that).
// RGB layers (FPROP / WGRAD over a 3-channel big image) take the split kernel's filter-row gather: slices per
// filter row, or 0 when the geometry does not qualify (a run of kw * 3 floats must fit two 16-deep slices).
// The same gather serves every big image whose channel count is no multiple of 4 (the 1 / 11 / 21-channel layers of
// the y-conditioned MNIST model): a filter row's kw * cb values are one contiguous run whatever cb is (knob
// row_gather_any = 0 keeps those layers on the fp32 kernel).
static int row3_slices(int mode, const gan_conv_geom& g) {
    if (mode == MODE_TPROP || g.cb % 4 == 0) return 0;
    const bool rgb = g.cb == 3 && g.cs % 4 == 0 && g.kw * 3 <= 32;
    if (!rgb && (!knobs().row_gather_any || g.kw * g.cb > 128)) return 0;
    if (knobs().rgb_gather == 0 || (knobs().rgb_gather == 2 && mode == MODE_FPROP)) return 0;
    return (g.kw * g.cb + 15) / 16;
}

static bool split_eligible(int mode, const IgemmArgs& a) {
    const gan_conv_geom& g = a.g;
    // FPROP / WGRAD gather the small side and the filter's cs axis as PAIRS of adjacent channels (8-byte loads at
    // dword-aligned addresses; the last channel of an odd count on its own), so any cs goes there (138 = 128 + 10
    // label channels of the y-conditioned MNIST generator, the 3 and 1 channels of CycleGAN's image and patch heads);
    // results whose rows are not 16-byte multiples leave through the dword epilogue.  The big side is read 16 bytes
    // (FPROP) or 8 bytes (WGRAD) at a time.  TPROP gathers the small side and the filter along cs, 16 bytes at a time,
    // or in pairs when cs is even but no multiple of 4.
    const bool ch = (mode == MODE_TPROP) ? (g.cs % 2 == 0) : (g.cb % 4 == 0 || row3_slices(mode, g) > 0);
    return ch && a.a_bytes != 0 && a.b_bytes != 0;
}

// `a_al` / `b_al`: the operand pointers are 16-byte aligned (vector loads, split kernel eligible).
static Plan make_plan(int mode, const IgemmArgs& a, bool a_al, bool b_al) {
    const gan_conv_geom& g = a.g;
    Plan pl;
    int M, N, K;
    if (mode == MODE_TPROP) {
        M = 0; K = 0;
        for (int i = 0; i < a.ncls; ++i) {
            if (a.cls[i].M > M) M = a.cls[i].M;
            if (a.cls[i].K > K) K = a.cls[i].K;
        }
        N = g.cb;
        pl.out_elems = (long)g.n * g.hb * g.wb * g.cb;
        pl.va = pl.vb = (g.cs % 4 == 0) ? 4 : 1;
    } else {
        M = a.M; N = a.N; K = a.K;
        pl.out_elems = (long)M * N;
        pl.va = (g.cb % 4 == 0) ? 4 : 1;
        pl.vb = (g.cs % 4 == 0) ? 4 : 1;
    }
    const int real_M = M, real_K = K;
    if (const int spr = row3_slices(mode, g)) {   // plan on the virtual sizes; undone below if the fp32 kernel runs
        if (mode == MODE_FPROP) K = g.kh * spr * BK;
        else M = g.kh * spr * BK;
    }
    if (!a_al) pl.va = 1;
    if (!b_al) pl.vb = 1;
    pl.order = 0;
    if (N <= 32) pl.cfg = 3;
    else if (N <= 64) pl.cfg = (M >= 8192) ? 1 : 2;
    else pl.cfg = (M >= 2048 && N >= 128) ? 0 : 2;
    // the split kernel's loaders move 16 / 8 bytes at a time: channel counts must be multiples of 4; mirrored taps
    // are built for its cheap gather arithmetic and for the filter-row gather of 3-channel images only
    const bool reflect_ok = !a.reflect || row3_slices(mode, g) > 0 || (mode == MODE_FPROP ? g.cb % 16 == 0 : g.ws % 4 == 0);
    const bool split_ok = a_al && b_al && split_eligible(mode, a) && !exact_f32() && reflect_ok;
    if (split_ok && pl.cfg == 3 && mode == MODE_FPROP) {
        // the same tile on the bf16 pipe (CycleGAN's 64 -> 3 channel 7x7 image head: 163 -> 133 us); the weight
        // gradient of such a layer is faster on the fp32 tile (190 against 245 us) and keeps it
        pl.cfg = kThinSplitCfg;
    } else if (split_ok && pl.cfg < 3) {
        // the split kernel converts every staged value, so it wants the largest tile that still leaves
        // enough blocks: 128x128 from 1024 rows on (the tuned DCGAN / CycleGAN plans agree)
        if (N >= 128 && M >= 1024) pl.cfg = 0;
        pl.cfg += kFirstSplitCfg;
        // fewer 128x128 tiles than CUs: the k range is split anyway, so split it inside the workgroup first
        // (two wave groups per tile, half the slabs); the tuner's choices on the DCGAN / CycleGAN shapes agree
        const long ncls = mode == MODE_TPROP ? a.ncls : 1;
        if (pl.cfg == kFirstSplitCfg) {
            const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128) * ncls;
            if (tiles <= 256 && (K + BK - 1) / BK >= 24) pl.cfg = kFirstKg2Cfg;
        } else if (pl.cfg == kFirstSplitCfg + 1) {
            // 128x64 tiles: two workgroups of two k groups fit a CU
            const long tiles = (long)((M + 127) / 128) * ((N + 63) / 64) * ncls;
            if (tiles <= 512 && (K + BK - 1) / BK >= 24) pl.cfg = kFirstKg2Cfg + 1;
        }
    }
    int pinned_splits = 0;
    {
        std::lock_guard<std::mutex> lock(g_plan_mu);
        auto it = plan_table().find(plan_key(mode, g));
        if (it != plan_table().end()) {
            pl.cfg = it->second.first % kOrderStride;
            pl.order = it->second.first / kOrderStride;
            pinned_splits = it->second.second;
        }
    }
    if (knobs().force_cfg >= 0) pl.cfg = knobs().force_cfg % kNumCfg;
    // a pinned or forced split configuration on operands that cannot take it (unaligned views, exact mode)
    if (pl.cfg == kThinSplitCfg && (!split_ok || mode == MODE_TPROP)) {
        pl.cfg = 3;          // (the transposed mode has no 32-wide bf16 tile: thin outputs there run on the vector ALUs)
    } else if (is_kg2(pl.cfg) && !split_ok) {
        pl.cfg -= kFirstKg2Cfg;
        pinned_splits *= 2;  // a plan tuned for two k groups per workgroup: the one-group kernel wants twice the slabs
    } else if (pl.cfg >= kFirstSplitCfg && !split_ok) {
        pl.cfg -= kFirstSplitCfg;
    }
    // the filter-row gather tiles the VIRTUAL gather axis (kh x slices-per-row x 16)
    pl.row3 = pl.cfg >= kFirstSplitCfg ? row3_slices(mode, g) : 0;
    M = real_M; K = real_K;
    if (pl.row3 && mode == MODE_FPROP) K = g.kh * pl.row3 * BK;
    if (pl.row3 && mode == MODE_WGRAD) M = g.kh * pl.row3 * BK;
    const int bm = kCfg[pl.cfg].bm, bn = kCfg[pl.cfg].bn;
    pl.mt = (M + bm - 1) / bm;
    pl.nt = (N + bn - 1) / bn;
    const int nkt = (K + BK - 1) / BK;
    const long blocks = (long)pl.mt * pl.nt * (mode == MODE_TPROP ? a.ncls : 1);
    int splits = 1;
    // one co-resident wave of blocks: the fp32 kernel fits 3 blocks per CU, the split kernel 2
    // (LDS and registers allow 2 / 3 / 5 blocks per CU for its 128x128 / 128x64 / 64x64 tiles)
    // (the two-group tiles hold 8 waves each: one 128x128 or two 128x64 workgroups per CU)
    const long target = pl.cfg == 4 ? 512 : (pl.cfg == 5 ? 768 : (pl.cfg == 6 ? 1280 : (pl.cfg == 7 ? 256 : (pl.cfg == 8 ? 512 : (pl.cfg == 9 ? 1024 : 768)))));
    if (blocks < target) {
        splits = (int)(target / blocks);
        // keep >= 12 k-slices per wave group: shorter loops are all prologue
        const int kdiv = is_kg2(pl.cfg) ? 24 : 12;
        const int max_by_k = nkt / kdiv > 0 ? nkt / kdiv : 1;
        if (splits > max_by_k) splits = max_by_k;
        // the slabs are written once and read once: keep them to a few tens of MB
        const long max_by_bytes = (48L << 20) / (pl.out_elems * (long)sizeof(float) + 1);
        if (splits > max_by_bytes) splits = (int)(max_by_bytes > 0 ? max_by_bytes : 1);
        if (splits > 512) splits = 512;
        if (splits < 1) splits = 1;
    }
    if (pinned_splits > 0) splits = pinned_splits;
    if (knobs().force_splits > 0) splits = knobs().force_splits;
    pl.kt_per_split = nkt > 0 ? (nkt + splits - 1) / splits : 1;
    pl.splits = nkt > 0 ? (nkt + pl.kt_per_split - 1) / pl.kt_per_split : 1;
    if (pl.splits < 1) pl.splits = 1;
    return pl;
}

// The vector-ALU path for thin TPROP outputs; a forced tile configuration (tests, tuning) keeps
// the MFMA path.
static bool thin_tprop_ok(const IgemmArgs& a) {
    const gan_conv_geom& g = a.g;
    if (g.cb > 4 || a.accumulate) return false;
    if ((long)g.hs * g.ws * g.cs >= (1L << 31)) return false;   // the kernel keeps pixel offsets within an image as ints
    if (knobs().force_cfg >= 0 || knobs().no_thin) return false;
    for (int i = 0; i < a.ncls; ++i)
        if (a.cls[i].khc > kThinMaxTaps || a.cls[i].kwc > kThinMaxTaps) return false;
    return true;
}

static int run_thin_tprop(const IgemmArgs& a, hipStream_t s, const char* what) {
    ThinArgs t;
    t.g = a.g;
    t.in = a.a; t.w = a.b; t.out = a.out; t.bias = a.bias;
    t.epi = a.epi;
    t.ncls = a.ncls;
    int hq = 0, wq = 0;
    for (int i = 0; i < kMaxClasses; ++i) {
        t.cls[i] = a.cls[i < a.ncls ? i : 0];
        if (i < a.ncls && a.cls[i].hq > hq) hq = a.cls[i].hq;
        if (i < a.ncls && a.cls[i].wq > wq) wq = a.cls[i].wq;
    }
    t.tiles_h = (hq + kThinQH - 1) / kThinQH;
    t.tiles_w = (wq + kThinQW - 1) / kThinQW;
    if (t.tiles_h < 1 || t.tiles_w < 1) return 0;
    dim3 grid((unsigned)(a.g.n * t.tiles_h * t.tiles_w), (unsigned)a.ncls);
    int maxt = 0;
    for (int i = 0; i < a.ncls; ++i) {
        if (a.cls[i].khc > maxt) maxt = a.cls[i].khc;
        if (a.cls[i].kwc > maxt) maxt = a.cls[i].kwc;
    }
    // channel counts that are not multiples of the chunk, or an unaligned image, take the dword-by-dword variant
    const bool tail = (a.g.cs % kThinCC != 0) || !aligned16(a.a) || !aligned16(a.b);
#define GAN_THIN_LAUNCH(CBV)                                                                                      \
    if (tail) {                                                                                                   \
        if (maxt <= 3) hipLaunchKernelGGL((tprop_thin_kernel<CBV, 3, true>), grid, dim3(256), 0, s, t);           \
        else hipLaunchKernelGGL((tprop_thin_kernel<CBV, kThinMaxTaps, true>), grid, dim3(256), 0, s, t);          \
    } else if (maxt <= 3) hipLaunchKernelGGL((tprop_thin_kernel<CBV, 3, false>), grid, dim3(256), 0, s, t);       \
    else hipLaunchKernelGGL((tprop_thin_kernel<CBV, kThinMaxTaps, false>), grid, dim3(256), 0, s, t)
    switch (a.g.cb) {
        case 1: GAN_THIN_LAUNCH(1); break;
        case 2: GAN_THIN_LAUNCH(2); break;
        case 3: GAN_THIN_LAUNCH(3); break;
        default: GAN_THIN_LAUNCH(4); break;
    }
#undef GAN_THIN_LAUNCH
    return check_launch(what);
}

template <int MODE, int BM, int BN, int WM, int WN>
static void launch_vec(const IgemmArgs& a, const Plan& pl, dim3 grid, hipStream_t s) {
    if (pl.va == 4 && pl.vb == 4)
        hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, 4, 4>), grid, dim3(256), 0, s, a);
    else if (pl.va == 1 && pl.vb == 4)
        hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, 1, 4>), grid, dim3(256), 0, s, a);
    else if (pl.va == 4 && pl.vb == 1)
        hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, 4, 1>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, 1, 1>), grid, dim3(256), 0, s, a);
}

template <int MODE>
static void launch_cfg(const IgemmArgs& a, const Plan& pl, dim3 grid, hipStream_t s) {
    switch (pl.cfg) {
        case 0: launch_vec<MODE, 128, 128, 2, 2>(a, pl, grid, s); break;
        case 1: launch_vec<MODE, 128, 64, 2, 2>(a, pl, grid, s); break;
        case 2: launch_vec<MODE, 64, 64, 2, 2>(a, pl, grid, s); break;
        case 3: launch_vec<MODE, 128, 32, 4, 1>(a, pl, grid, s); break;
        default:
            if (pl.cfg == kThinSplitCfg && MODE == MODE_TPROP) { launch_vec<MODE, 128, 32, 4, 1>(a, pl, grid, s); break; }   // (make_plan maps TPROP to 3)
            {
                typedef void (*Launcher)(const IgemmArgs&, int, bool, dim3, hipStream_t);
                static const Launcher table[3][3] = {{launch_split_m0_p0, launch_split_m0_p1, launch_split_m0_p2},
                                                     {launch_split_m1_p0, launch_split_m1_p1, launch_split_m1_p2},
                                                     {launch_split_m2_p0, launch_split_m2_p1, launch_split_m2_p2}};
                const int part = (pl.cfg == 4 || pl.cfg == 7) ? 0 : ((pl.cfg == 5 || pl.cfg == 8) ? 1 : 2);
                table[MODE][part](a, pl.cfg, knobs().slow_gather != 0, grid, s);
            }
            break;
    }
}

// Which XCD-aware tile order a launch uses (tile_of_workgroup).  FPROP / TPROP: by row tile, the image operand
// crosses the fabric about once and the filter once per XCD (8 times); by k range it is the other way round.
static int tile_order(int mode, const IgemmArgs& a, const Plan& pl) {
    int k = knobs().tile_order > 0 ? knobs().tile_order : pl.order;
    if (k >= 1000) {   // test knob: one decimal digit per mode, 1FTW (FPROP, TPROP, WGRAD), 0 = chosen here
        k -= 1000;
        k = mode == MODE_FPROP ? k / 100 : (mode == MODE_TPROP ? (k / 10) % 10 : k % 10);
        if (k == 0) k = pl.order;
    }
    if (k >= 1 && k <= 4) return k - 1;
    // Chosen by what a whole training iteration measures, not by a layer pass timed alone (there, launch after
    // launch with warm L2s, the XCD grouping looks 8-15 % better for the transposed contraction than it is):
    //   WGRAD  by k range: same time as any other order, a fifth of the fabric bytes;
    //   TPROP  row tile fastest in launch order: grouping its parity classes on one XCD costs 3 % of a DCGAN step;
    //   FPROP  by row tile or by k range, whichever moves fewer bytes (same time as launch order).
    if (mode == MODE_WGRAD) return 2;
    if (mode == MODE_TPROP) return 3;
    const gan_conv_geom& g = a.g;
    const double image = 4.0 * g.n * g.hb * g.wb * g.cb;
    const double filter = 4.0 * g.kh * g.kw * g.cb * g.cs;
    return image + 8.0 * filter <= 8.0 * image + filter ? 1 : 2;
}

static size_t plan_ws_bytes(const Plan& pl) {
    return pl.splits > 1 ? (size_t)pl.splits * pl.out_elems * sizeof(float) : 0;
}

// Runs one contraction.  `out_c` is the channel count of the output's last axis (bias period).
// `tail` (optional): a pending column-sum last stage; if the contraction has a slab reduction launch, that launch
// carries it and *tail_done is set.
static size_t plan_ws_bytes_fold(const Plan& pl) { return (size_t)(pl.splits > 1 ? pl.splits : 1) * pl.out_elems * sizeof(float); }

static void launch_fold(const float* slabs, float* dx, const FoldReflect& f, int splits, long slab_elems, hipStream_t s) {
    const long items = (long)f.n * f.h * f.w * ((f.c & 3) == 0 ? f.c >> 2 : f.c);
    long blocks = (items + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(reduce_slabs_fold_reflect_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slabs, dx, f, splits, slab_elems);
}

// `fold` (TPROP only): a.g is the geometry of the reflect-PADDED image; the contraction leaves its result in the
// workspace (one slab when the plan has no split-K) and the fold kernel writes the gradient of the unpadded image.
int run_igemm(int mode, IgemmArgs a, int out_c, void* ws, size_t ws_bytes, hipStream_t s, const char* what,
              const ColsumFinal* tail = nullptr, int* tail_done = nullptr, const FoldReflect* fold = nullptr) {
    ColsumFinal job = ColsumFinal{nullptr, nullptr, 0, 0, 0, 0};
    if (tail != nullptr && tail_done != nullptr && tail->blocks > 0) job = *tail;
    // a batch-norm reduction over the output, asked for by the entry point that called (taken: it never outlives the call)
    const BnRide ride = pending_bn_ride();
    pending_bn_ride().kind = 0;
    if (ride.kind != 0) {
        const long out_elems = mode == MODE_TPROP ? (fold ? (long)fold->n * fold->h * fold->w * fold->c : (long)a.g.n * a.g.hb * a.g.wb * a.g.cb)
                                                  : (long)a.g.n * a.g.hs * a.g.ws * a.g.cs;
        GAN_CHECK_ARG(mode != MODE_WGRAD && job.blocks == 0 && !a.accumulate && !epi_active(a.epi) && ride.s.c == out_c &&
                      (long)ride.s.groups * ride.s.rows * ride.s.c == out_elems,
                      "%s: the batch-norm reduction does not fit this contraction's output", what);
    }
    // the reduction over the finished output, as a launch of its own (no slab reduction launch to ride on)
    auto ride_alone = [&](float* out) -> int {
        return launch_bn_ride(ride, RowSource{nullptr, out, nullptr, 0, 0}, s, what);
    };
    if (mode == MODE_TPROP) build_classes(a.g, a);
    else a.ncls = 1;
    if (mode == MODE_FPROP && knobs().rgb_direct && knobs().force_cfg < 0 && !exact_f32() && rgb_fprop_ok(a)) {
        // first layer over a 3-channel image: one pass, no slice pipeline, no slabs (rgb.hip)
        if (int rc = run_rgb_fprop(a, s, what)) return rc;
        return ride.kind ? ride_alone(a.out) : 0;
    }
    if (mode == MODE_TPROP && thin_tprop_ok(a)) {
        if (fold == nullptr) {
            if (int rc = run_thin_tprop(a, s, what)) return rc;
            return ride.kind ? ride_alone(a.out) : 0;
        }
        const long elems = (long)a.g.n * a.g.hb * a.g.wb * a.g.cb;
        GAN_CHECK_ARG(ws != nullptr && ws_bytes >= (size_t)elems * sizeof(float), "%s: workspace of %zu bytes needed, %zu given",
                      what, (size_t)elems * sizeof(float), ws_bytes);
        float* dx = a.out;
        a.out = static_cast<float*>(ws);
        if (int rc = run_thin_tprop(a, s, what)) return rc;
        launch_fold(static_cast<const float*>(ws), dx, *fold, 1, elems, s);
        if (int rc = check_launch(what)) return rc;
        return ride.kind ? ride_alone(dx) : 0;
    }
    Plan pl = make_plan(mode, a, aligned16(a.a), aligned16(a.b));
    const size_t need = fold ? plan_ws_bytes_fold(pl) : plan_ws_bytes(pl);
    GAN_CHECK_ARG(need == 0 || (ws != nullptr && ws_bytes >= need), "%s: workspace of %zu bytes needed, %zu given",
                  what, need, ws_bytes);
    float* final_out = a.out;
    const float* bias = a.bias;
    const int accumulate = a.accumulate;
    a.splits = pl.splits;
    a.kt_per_split = pl.kt_per_split;
    a.slab_elems = pl.out_elems;
    a.row3_spr = pl.row3;
    if (pl.row3 && mode == MODE_FPROP) a.K = a.g.kh * pl.row3 * BK;
    if (pl.row3 && mode == MODE_WGRAD) a.M = a.g.kh * pl.row3 * BK;
    if (pl.splits > 1 || fold != nullptr) {
        a.out = static_cast<float*>(ws);
        a.bias = nullptr;
        a.accumulate = 0;
    }
    {
        const int n_out = (mode == MODE_TPROP) ? a.g.cb : a.N;
        a.vec_epilogue = ((n_out & 3) == 0 && (pl.out_elems & 3) == 0 && aligned16(a.out) && aligned16(final_out) &&
                          (bias == nullptr || aligned16(bias)) && !knobs().scalar_epilogue) ? 1 : 0;
    }
    a.grid_mt = pl.mt;
    a.grid_nt = pl.nt;
    a.grid_z = pl.splits * a.ncls;
    a.xcd_order = tile_order(mode, a, pl);
    const long own_blocks = (long)pl.mt * pl.nt * pl.splits * a.ncls;
    const bool split_kernel = pl.cfg >= kFirstSplitCfg;
    // the last stage of a pending bias-gradient column sum rides on this launch when no reduction launch follows
    const bool carry = split_kernel && job.blocks > 0 && pl.splits == 1 && !knobs().no_carry;
    a.own_blocks = (int)own_blocks;
    if (carry) a.tail = job;
    dim3 grid((unsigned)(own_blocks + (carry ? job.blocks : 0)));
    switch (mode) {
        case MODE_FPROP: launch_cfg<MODE_FPROP>(a, pl, grid, s); break;
        case MODE_TPROP: launch_cfg<MODE_TPROP>(a, pl, grid, s); break;
        default: launch_cfg<MODE_WGRAD>(a, pl, grid, s); break;
    }
    if (int rc = check_launch(what)) return rc;
    if (carry) *tail_done = 1;
    if (fold != nullptr) {
        launch_fold(static_cast<const float*>(ws), final_out, *fold, pl.splits, pl.out_elems, s);
        if (int rc = check_launch(what)) return rc;
        return ride.kind ? ride_alone(final_out) : 0;
    }
    if (ride.kind && pl.splits > 1 && pl.splits < 16 && (pl.out_elems & 3) == 0 && (out_c & 3) == 0 && knobs().no_bn_ride != 2)
        // ONE launch sums the slabs, adds the bias, stores the output and takes the batch-norm reduction over it
        return launch_bn_ride(ride, RowSource{static_cast<const float*>(ws), final_out, bias, pl.out_elems, pl.splits}, s, what);
    if (pl.splits >= 16 && (pl.out_elems & 3) == 0) {
        const long n4 = pl.out_elems >> 2;
        const int own = (int)((n4 + 15) / 16);
        hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)(own + job.blocks)), dim3(256), 0, s,
                           static_cast<const float*>(ws), final_out, bias, pl.out_elems, out_c, pl.splits, accumulate, own, job, a.epi);
        if (int rc = check_launch(what)) return rc;
        if (job.blocks > 0) *tail_done = 1;
    } else if (pl.splits >= 16) {
        const int own = (int)((pl.out_elems + 15) / 16);
        hipLaunchKernelGGL(reduce_slabs_wide_scalar_kernel, dim3((unsigned)(own + job.blocks)), dim3(256), 0, s,
                           static_cast<const float*>(ws), final_out, bias, pl.out_elems, out_c, pl.splits, accumulate, own, job,
                           a.epi);
        if (int rc = check_launch(what)) return rc;
        if (job.blocks > 0) *tail_done = 1;
    } else if (pl.splits > 1) {
        long work = (pl.out_elems + 3) / 4;
        int blocks = (int)((work + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks + job.blocks), dim3(256), 0, s, static_cast<const float*>(ws),
                           final_out, bias, pl.out_elems, out_c, pl.splits, accumulate, blocks, job, a.epi);
        if (int rc = check_launch(what)) return rc;
        if (job.blocks > 0) *tail_done = 1;
    }
    return ride.kind ? ride_alone(final_out) : 0;
}

size_t igemm_ws_bytes(int mode, IgemmArgs a) {
    if (mode == MODE_TPROP) build_classes(a.g, a);
    else a.ncls = 1;
    // The pointers are not known yet, and their alignment selects the kernel family (16-byte aligned
    // operands take the split kernel, others the fp32 kernel with a different block target and hence
    // split-K factor): report the larger of the two plans' needs.
    const size_t al = plan_ws_bytes(make_plan(mode, a, true, true));
    const size_t un = plan_ws_bytes(make_plan(mode, a, false, false));
    return al > un ? al : un;
}

#ifdef GAN_PHASE_STAMPS
static long long* g_debug_stamps = nullptr;
extern "C" void gan_debug_set_stamps(long long* p) { g_debug_stamps = p; }
#endif

IgemmArgs make_args(int mode, const gan_conv_geom& g, const float* a, const float* b, float* out,
                    const float* bias, int accumulate) {
    IgemmArgs r;
    r.g = g;
    r.a = a; r.b = b; r.out = out; r.bias = bias;
    r.accumulate = accumulate;
    r.splits = 1; r.kt_per_split = 1; r.slab_elems = 0; r.ncls = 1; r.vec_epilogue = 0;
    r.epi = no_epilogue();
    r.reflect = 0;
    r.row3_spr = 0;
    r.row3_mul = (65536 + g.cb - 1) / g.cb;
    r.own_blocks = 0x7fffffff;
#ifdef GAN_PHASE_STAMPS
    r.stamps = g_debug_stamps;
#endif
    r.tail = ColsumFinal{nullptr, nullptr, 0, 0, 0, 0};
    r.d_cb = make_fastdiv((unsigned)g.cb);
    r.d_cs = make_fastdiv((unsigned)g.cs);
    r.d_kw = make_fastdiv((unsigned)g.kw);
    r.d_hsws = make_fastdiv((unsigned)(g.hs * g.ws));
    r.d_ws = make_fastdiv((unsigned)g.ws);
    for (int i = 0; i < kMaxClasses; ++i) r.cls[i] = ClassInfo{0, 0, 1, 1, 0, 0, 0, 0, 0, 0, make_fastdiv(1)};
    const uint64_t big_b = 4ull * g.n * g.hb * g.wb * g.cb, small_b = 4ull * g.n * g.hs * g.ws * g.cs;
    const uint64_t filt_b = 4ull * g.kh * g.kw * g.cb * g.cs;
    const uint64_t cap = 0xffffffe0ull;  // operands beyond 4 GB keep the fp32 kernel (make_plan)
    uint64_t ab, bb;
    if (mode == MODE_FPROP) {
        r.M = g.n * g.hs * g.ws; r.N = g.cs; r.K = g.kh * g.kw * g.cb;
        ab = big_b; bb = filt_b;
    } else if (mode == MODE_WGRAD) {
        r.M = g.kh * g.kw * g.cb; r.N = g.cs; r.K = g.n * g.hs * g.ws;
        ab = big_b; bb = small_b;
    } else {
        r.M = 0; r.N = g.cb; r.K = 0;
        ab = small_b; bb = filt_b;
    }
    r.a_bytes = (unsigned)(ab < cap ? ab : 0);
    r.b_bytes = (unsigned)(bb < cap ? bb : 0);
    return r;
}

int plan_set(int mode, const gan_conv_geom& g, int cfg, int splits) {
    GAN_CHECK_ARG(cfg < 0 || cfg % kOrderStride != kThinSplitCfg || mode != MODE_TPROP,
                  "gan_conv_plan_set: tile configuration %d serves modes 0 and 2 (thin outputs of mode 1 run on the vector "
                  "ALUs); use 3", kThinSplitCfg);
    if (cfg >= 0 && cfg % kOrderStride >= kFirstSplitCfg) {
        const IgemmArgs a = make_args(mode, g, nullptr, nullptr, nullptr, nullptr, 0);
        GAN_CHECK_ARG(split_eligible(mode, a),
                      "gan_conv_plan_set: tile configuration %d (bf16 matrix pipe) needs a gathered-image channel count "
                      "that is a multiple of 4 (forward / weight gradient: or filter rows of at most 128 floats; transposed: "
                      "or an even count) and operands below 4 GB; this geometry runs configurations 0..3", cfg % kOrderStride);
    }
    std::lock_guard<std::mutex> lock(g_plan_mu);
    if (cfg < 0) plan_table().erase(plan_key(mode, g));
    else plan_table()[plan_key(mode, g)] = std::make_pair(cfg, splits);
    return 0;
}

void plan_clear() {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    plan_table().clear();
}

// The plan a call with 16-byte aligned operands would run with right now.
void plan_get(int mode, const gan_conv_geom& g, int* cfg, int* splits, int* thin) {
    static const float dummy[4] __attribute__((aligned(16))) = {0, 0, 0, 0};
    IgemmArgs a = make_args(mode, g, dummy, dummy, nullptr, nullptr, 0);
    if (mode == MODE_TPROP) build_classes(a.g, a);
    else a.ncls = 1;
    const Plan pl = make_plan(mode, a, true, true);
    *cfg = pl.cfg + kOrderStride * pl.order;
    *splits = pl.splits;
    *thin = (mode == MODE_TPROP && thin_tprop_ok(a)) ? 1 : 0;
}

int conv_geom_check(const gan_conv_geom* g) {
    GAN_CHECK_ARG(g != nullptr, "conv geometry: NULL");
    return geom_check(*g);
}

size_t igemm_ws_bytes_geom(int mode, const gan_conv_geom& g) {
    return igemm_ws_bytes(mode, make_args(mode, g, nullptr, nullptr, nullptr, nullptr, 0));
}

int run_igemm_geom(int mode, const gan_conv_geom& g, const float* a, const float* b, float* out, const float* bias,
                   int accumulate, int out_c, void* ws, size_t ws_bytes, hipStream_t s, const char* what,
                   const ColsumFinal* tail, int* tail_done) {
    return run_igemm(mode, make_args(mode, g, a, b, out, bias, accumulate), out_c, ws, ws_bytes, s, what, tail, tail_done);
}

// Reflecting taps: the geometry's pad_before (pt, pl) says how far the image is extended by reflection on each side.
// Forward and weight-gradient contractions only (they gather from the big image).
int run_igemm_geom_reflect(int mode, const gan_conv_geom& g, const float* a, const float* b, float* out, const float* bias,
                           int accumulate, int out_c, void* ws, size_t ws_bytes, hipStream_t s, const char* what,
                           const ColsumFinal* tail, int* tail_done) {
    GAN_CHECK_ARG(mode == MODE_FPROP || mode == MODE_WGRAD, "%s: reflection applies to forward and weight gradient", what);
    GAN_CHECK_ARG(g.pt < g.hb && g.pl < g.wb && (g.hs - 1) * g.sh + g.kh - 1 - g.pt < 2 * g.hb - 1 &&
                  (g.ws - 1) * g.sw + g.kw - 1 - g.pl < 2 * g.wb - 1,
                  "%s: REFLECT padding must be smaller than the image", what);
    IgemmArgs args = make_args(mode, g, a, b, out, bias, accumulate);
    args.reflect = 1;
    return run_igemm(mode, args, out_c, ws, ws_bytes, s, what, tail, tail_done);
}

// Input gradient of a convolution over a reflect-padded image (`g`: the unpadded image, pad_before = the padding on
// each side): the transposed contraction on the padded geometry, then the fold back onto the image.
static gan_conv_geom padded_geom(const gan_conv_geom& g) {
    gan_conv_geom gp = g;
    gp.hb = g.hb + 2 * g.pt;
    gp.wb = g.wb + 2 * g.pl;
    gp.pt = gp.pl = 0;
    return gp;
}

size_t igemm_ws_bytes_dgrad_reflect(const gan_conv_geom& g) {
    const gan_conv_geom gp = padded_geom(g);
    IgemmArgs a = make_args(MODE_TPROP, gp, nullptr, nullptr, nullptr, nullptr, 0);
    build_classes(a.g, a);
    const size_t al = plan_ws_bytes_fold(make_plan(MODE_TPROP, a, true, true));
    const size_t un = plan_ws_bytes_fold(make_plan(MODE_TPROP, a, false, false));
    return al > un ? al : un;
}

int run_dgrad_reflect(const gan_conv_geom& g, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes,
                      hipStream_t s, const char* what) {
    GAN_CHECK_ARG(g.pt == g.pl && g.pt < g.hb && g.pl < g.wb, "%s: REFLECT padding must be the same on both axes and smaller "
                  "than the image", what);
    // The padded geometry below extends the image by pt / pl on BOTH sides and the fold mirrors about those borders:
    // that is the adjoint of the forward pass only when the filter covers the padded image exactly (tf.pad by p on
    // each side + VALID).  A geometry with a different implicit pad behind the image (even filters, stride 2 with a
    // ragged end) would make this silently something else than the forward pass's adjoint: refuse it.
    GAN_CHECK_ARG((g.hs - 1) * g.sh + g.kh == g.hb + 2 * g.pt && (g.ws - 1) * g.sw + g.kw == g.wb + 2 * g.pl,
                  "%s: the filter must cover the reflect-padded image exactly ((hs-1)*sh+kh == hb+2*pt and the same along "
                  "the width): symmetric REFLECT padding in front of a VALID convolution", what);
    const gan_conv_geom gp = padded_geom(g);
    if (int rc = geom_check(gp)) return rc;
    const FoldReflect f = FoldReflect{g.n, g.hb, g.wb, g.cb, g.pt};
    return run_igemm(MODE_TPROP, make_args(MODE_TPROP, gp, dy, w, dx, nullptr, 0), g.cb, ws, ws_bytes, s, what, nullptr, nullptr, &f);
}

// The same with a fused epilogue (activation behind the output, or the activation derivative of a gate tensor).
int run_igemm_geom_epi(int mode, const gan_conv_geom& g, const float* a, const float* b, float* out, const float* bias,
                       const Epilogue& epi, int out_c, void* ws, size_t ws_bytes, hipStream_t s, const char* what) {
    IgemmArgs args = make_args(mode, g, a, b, out, bias, 0);
    args.epi = epi;
    return run_igemm(mode, args, out_c, ws, ws_bytes, s, what, nullptr, nullptr);
}

}  // namespace gan
