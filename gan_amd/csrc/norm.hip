// Batch / instance normalisation fused with the activation that follows it (gfx950).
//
// Every kernel here is bandwidth bound.  x is viewed as [groups][rows][c] with c contiguous.
// Reductions over rows run in two launches: a partial kernel (many blocks, 16-byte loads, four
// rows in flight per thread) and a finalize kernel in which ONE WAVE owns one (group, channel):
// its 64 lanes stride over the block partials and merge them with a fixed shuffle tree, so the
// result is deterministic and the serial chain is at most rblocks/64 long.
//
// Training forward: (1) per-block shifted sums -> (mean, sum of squared deviations),
// (2) Chan merge of the partials, moving-average update, (3) normalise + affine + activation
// (+ residual).  Backward: (1) per-block sums of dz and dz*xhat, (2) merge, dgamma/dbeta, (3) dx.
#include "common.h"
#include "bn_tile.h"

#include <cstdint>
#include <initializer_list>

namespace gan {

constexpr int kChanTile = 64;  // generic path: channels per block (one wave-width)
constexpr int kRowLanes = 4;   // generic path: waves per block, each on its own row

struct BnShape {
    int groups, rows, c;
    int rblocks, rows_per_block;
    int vec4;  // 1: float4 path (c % 4 == 0 and c/4 divides 256)
};

static bool vec4_ok(int c) { return (c % 4 == 0) && (c / 4 <= 256) && (256 % (c / 4) == 0); }

static BnShape bn_shape(int groups, int rows, int c) {
    BnShape s;
    s.groups = groups; s.rows = rows; s.c = c;
    s.vec4 = vec4_ok(c) ? 1 : 0;
    long rb;
    if (s.vec4) {
        const int row_lanes = 256 / (c / 4);
        rb = (rows + row_lanes * 4 - 1) / (row_lanes * 4);  // >= 4 rows per thread
        long cap = 1024 / groups;
        if (cap < 1) cap = 1;
        if (rb > cap) rb = cap;
    } else {
        const int ctiles = (c + kChanTile - 1) / kChanTile;
        long cap = 1024 / ((long)ctiles * groups);
        if (cap < 1) cap = 1;
        rb = (rows + 31) / 32;
        if (rb > cap) rb = cap;
    }
    if (rb < 1) rb = 1;
    s.rows_per_block = (int)((rows + rb - 1) / rb);
    s.rblocks = (rows + s.rows_per_block - 1) / s.rows_per_block;
    return s;
}

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// ---------------------------------------------------------------------------- forward statistics
// partial[(g*rblocks + rb)*c + ch] = {mean, m2} over the block's rows.
// float4 path: thread = (row lane, 4-channel group); sums are taken about the block's first row
// (shifted data), which keeps the one-pass variance free of cancellation.
__global__ __launch_bounds__(256) void bn_partial_stats_v4_kernel(const float* __restrict__ x, float2* __restrict__ partial,
                                                                  BnShape s) {
    __shared__ float4 red1[256];
    __shared__ float4 red2[256];
    const int c4 = s.c >> 2;
    const int cg = threadIdx.x % c4, rl = threadIdx.x / c4, RL = 256 / c4;
    const int rb = blockIdx.x, grp = blockIdx.y;
    const int r0 = rb * s.rows_per_block;
    const int r1 = min(s.rows, r0 + s.rows_per_block);
    const float4* xp = reinterpret_cast<const float4*>(x + ((long)grp * s.rows) * s.c) + cg;
    const float4 shift = xp[(long)r0 * c4];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += RL) {
        const float4 d = f4_sub(xp[(long)r * c4], shift);
        s1 = f4_add(s1, d);
        s2 = f4_fma(d, d, s2);
    }
    red1[threadIdx.x] = s1;
    red2[threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < RL; ++j) {
            s1 = f4_add(s1, red1[cg + j * c4]);
            s2 = f4_add(s2, red2[cg + j * c4]);
        }
        const float inv = 1.0f / (float)(r1 - r0);
        float2* out = partial + ((long)grp * s.rblocks + rb) * s.c + cg * 4;
        out[0] = make_float2(shift.x + s1.x * inv, s2.x - s1.x * s1.x * inv);
        out[1] = make_float2(shift.y + s1.y * inv, s2.y - s1.y * s1.y * inv);
        out[2] = make_float2(shift.z + s1.z * inv, s2.z - s1.z * s1.z * inv);
        out[3] = make_float2(shift.w + s1.w * inv, s2.w - s1.w * s1.w * inv);
    }
}

// generic path (any c): 64-channel tiles, two passes over a row range that stays cache resident
__global__ __launch_bounds__(256) void bn_partial_stats_kernel(const float* __restrict__ x, float2* __restrict__ partial,
                                                               BnShape s) {
    __shared__ float red[kRowLanes][kChanTile];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ch = blockIdx.z * kChanTile + tx;
    const int rb = blockIdx.x, grp = blockIdx.y;
    const int r0 = rb * s.rows_per_block;
    const int r1 = min(s.rows, r0 + s.rows_per_block);
    const bool cv = ch < s.c;
    const float* xp = x + ((long)grp * s.rows) * s.c + ch;
    float sum = 0.f;
    if (cv)
        for (int r = r0 + ty; r < r1; r += kRowLanes) sum += xp[(long)r * s.c];
    red[ty][tx] = sum;
    __syncthreads();
    const float cnt = (float)(r1 - r0);
    const float mean = (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]) / cnt;
    __syncthreads();
    float m2 = 0.f;
    if (cv)
        for (int r = r0 + ty; r < r1; r += kRowLanes) {
            const float d = xp[(long)r * s.c] - mean;
            m2 += d * d;
        }
    red[ty][tx] = m2;
    __syncthreads();
    if (ty == 0 && cv)
        partial[((long)grp * s.rblocks + rb) * s.c + ch] = make_float2(mean, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    const float nn = n + nb;
    if (nn > 0.f) {
        const float f = nb / nn;
        const float delta = mb - mean;
        mean += delta * f;
        m2 += m2b + delta * delta * n * f;
        n = nn;
    }
}

// One wave per (group, channel); with `chain_groups` one wave per channel takes the groups one after
// the other, so that the moving averages move once per group, in group order (several passes through
// one layer stacked along the batch axis).
__global__ __launch_bounds__(256) void bn_finalize_stats_kernel(const float2* __restrict__ partial, float* __restrict__ save_mean,
                                                                float* __restrict__ save_rstd, float* __restrict__ moving_mean,
                                                                float* __restrict__ moving_var, BnShape s, float eps,
                                                                float momentum, int unbiased, int chain_groups) {
    const int widx = (blockIdx.x * 256 + threadIdx.x) >> 6;  // wave-uniform
    const int lane = threadIdx.x & 63;
    if (widx >= (chain_groups ? s.c : s.groups * s.c)) return;
    const int g0 = chain_groups ? 0 : widx / s.c, g1 = chain_groups ? s.groups : g0 + 1;
    const int ch = chain_groups ? widx : widx - g0 * s.c;
    float mm = 0.f, mv = 0.f;
    if (moving_mean != nullptr && lane == 0) { mm = moving_mean[ch]; mv = moving_var[ch]; }
    for (int grp = g0; grp < g1; ++grp) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int rb = lane; rb < s.rblocks; rb += 64) {
            const float2 pr = partial[((long)grp * s.rblocks + rb) * s.c + ch];
            const int r0 = rb * s.rows_per_block;
            chan_merge(n, mean, m2, (float)(min(s.rows, r0 + s.rows_per_block) - r0), pr.x, pr.y);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float nb = __shfl_down(n, off, 64), mb = __shfl_down(mean, off, 64), m2b = __shfl_down(m2, off, 64);
            chan_merge(n, mean, m2, nb, mb, m2b);
        }
        if (lane == 0) {
            const float var = m2 / n;
            save_mean[grp * s.c + ch] = mean;
            save_rstd[grp * s.c + ch] = 1.0f / sqrtf(var + eps);
            const float v = unbiased ? var * (n / fmaxf(n - 1.f, 1.f)) : var;
            mm = mm * momentum + mean * (1.f - momentum);
            mv = mv * momentum + v * (1.f - momentum);
        }
    }
    if (moving_mean != nullptr && lane == 0) { moving_mean[ch] = mm; moving_var[ch] = mv; }
}

// y = act((x-mean)*rstd*gamma+beta) + residual.  mean/rstd are per (group, channel); for the
// inference path the caller passes groups = 1 with mean = moving_mean and var in place of rstd.
template <bool EVAL>
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ mean,
                                const float* __restrict__ rstd_or_var, const float* __restrict__ residual,
                                float* __restrict__ y, int groups, int rows, int c, float eps, int act, float act_param) {
    const unsigned total = (unsigned)groups * rows * c;  // < 2^31 checked on the host
    const unsigned stride = gridDim.x * blockDim.x;
    const unsigned per_group = (unsigned)rows * c;
    if ((c & 3) == 0) {
        for (unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) * 4u; i < total; i += stride * 4u) {
            const unsigned ch = i % (unsigned)c;
            const unsigned gi = (i / per_group) * c + ch;
            const float4 xv = *reinterpret_cast<const float4*>(x + i);
            const float4 mu = *reinterpret_cast<const float4*>(mean + gi);
            float4 rs = *reinterpret_cast<const float4*>(rstd_or_var + gi);
            if (EVAL) {
                rs.x = 1.0f / sqrtf(rs.x + eps); rs.y = 1.0f / sqrtf(rs.y + eps);
                rs.z = 1.0f / sqrtf(rs.z + eps); rs.w = 1.0f / sqrtf(rs.w + eps);
            }
            const float4 ga = *reinterpret_cast<const float4*>(gamma + ch);
            const float4 be = *reinterpret_cast<const float4*>(beta + ch);
            float4 o;
            o.x = apply_act((xv.x - mu.x) * rs.x * ga.x + be.x, act, act_param);
            o.y = apply_act((xv.y - mu.y) * rs.y * ga.y + be.y, act, act_param);
            o.z = apply_act((xv.z - mu.z) * rs.z * ga.z + be.z, act, act_param);
            o.w = apply_act((xv.w - mu.w) * rs.w * ga.w + be.w, act, act_param);
            if (residual) {
                const float4 rv = *reinterpret_cast<const float4*>(residual + i);
                o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
            }
            *reinterpret_cast<float4*>(y + i) = o;
        }
    } else {
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const unsigned ch = i % (unsigned)c;
            const unsigned gi = (i / per_group) * c + ch;
            float rs = rstd_or_var[gi];
            if (EVAL) rs = 1.0f / sqrtf(rs + eps);
            float o = apply_act((x[i] - mean[gi]) * rs * gamma[ch] + beta[ch], act, act_param);
            if (residual) o += residual[i];
            y[i] = o;
        }
    }
}

// ---------------------------------------------------------------------------- backward
__device__ __forceinline__ float dz_of(float xv, float dyv, float mu, float rs, float ga, float be, int act,
                                       float act_param, float& xhat) {
    // (no contraction left to the compiler: kernels that take these sums in different launch shapes must agree bit
    // for bit, and a fused multiply-add formed at one call site and not at another differs in the last bit)
#pragma clang fp contract(off)
    xhat = (xv - mu) * rs;
    if (act == GAN_ACT_NONE) return dyv;
    const float z = fmaf(xhat, ga, be);
    return dyv * act_grad(z, apply_act(z, act, act_param), act, act_param);
}

// partial[(g*rblocks+rb)*c + ch] = {sum dz, sum dz*xhat},  dz = dy * act'(z)
__global__ __launch_bounds__(256) void bn_bwd_partial_v4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                float2* __restrict__ partial, BnShape s, int act, float act_param) {
    __shared__ float4 red1[256];
    __shared__ float4 red2[256];
    const int c4 = s.c >> 2;
    const int cg = threadIdx.x % c4, rl = threadIdx.x / c4, RL = 256 / c4;
    const int rb = blockIdx.x, grp = blockIdx.y;
    const int r0 = rb * s.rows_per_block;
    const int r1 = min(s.rows, r0 + s.rows_per_block);
    const long base = ((long)grp * s.rows) * s.c;
    const float4* xp = reinterpret_cast<const float4*>(x + base) + cg;
    const float4* dp = reinterpret_cast<const float4*>(dy + base) + cg;
    const float4 mu = reinterpret_cast<const float4*>(mean + grp * s.c)[cg];
    const float4 rs = reinterpret_cast<const float4*>(rstd + grp * s.c)[cg];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[cg];
    const float4 be = reinterpret_cast<const float4*>(beta)[cg];
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += RL) {
        const float4 xv = xp[(long)r * c4], dv = dp[(long)r * c4];
        float4 xh, dz;
        dz.x = dz_of(xv.x, dv.x, mu.x, rs.x, ga.x, be.x, act, act_param, xh.x);
        dz.y = dz_of(xv.y, dv.y, mu.y, rs.y, ga.y, be.y, act, act_param, xh.y);
        dz.z = dz_of(xv.z, dv.z, mu.z, rs.z, ga.z, be.z, act, act_param, xh.z);
        dz.w = dz_of(xv.w, dv.w, mu.w, rs.w, ga.w, be.w, act, act_param, xh.w);
        s0 = f4_add(s0, dz);
        s1 = f4_fma(dz, xh, s1);
    }
    red1[threadIdx.x] = s0;
    red2[threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < RL; ++j) {
            s0 = f4_add(s0, red1[cg + j * c4]);
            s1 = f4_add(s1, red2[cg + j * c4]);
        }
        float2* out = partial + ((long)grp * s.rblocks + rb) * s.c + cg * 4;
        out[0] = make_float2(s0.x, s1.x);
        out[1] = make_float2(s0.y, s1.y);
        out[2] = make_float2(s0.z, s1.z);
        out[3] = make_float2(s0.w, s1.w);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float2* __restrict__ partial, BnShape s, int act, float act_param) {
    __shared__ float red0[kRowLanes][kChanTile];
    __shared__ float red1[kRowLanes][kChanTile];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ch = blockIdx.z * kChanTile + tx;
    const int rb = blockIdx.x, grp = blockIdx.y;
    const int r0 = rb * s.rows_per_block;
    const int r1 = min(s.rows, r0 + s.rows_per_block);
    const bool cv = ch < s.c;
    float s0 = 0.f, s1 = 0.f;
    if (cv) {
        const long base = ((long)grp * s.rows) * s.c + ch;
        const float mu = mean[grp * s.c + ch], rs = rstd[grp * s.c + ch];
        const float ga = gamma[ch], be = beta[ch];
        for (int r = r0 + ty; r < r1; r += kRowLanes) {
            const long i = base + (long)r * s.c;
            float xhat;
            const float dz = dz_of(x[i], dy[i], mu, rs, ga, be, act, act_param, xhat);
            s0 += dz;
            s1 += dz * xhat;
        }
    }
    red0[ty][tx] = s0;
    red1[ty][tx] = s1;
    __syncthreads();
    if (ty == 0 && cv)
        partial[((long)grp * s.rblocks + rb) * s.c + ch] =
            make_float2(red0[0][tx] + red0[1][tx] + red0[2][tx] + red0[3][tx],
                        red1[0][tx] + red1[1][tx] + red1[2][tx] + red1[3][tx]);
}

// One wave per channel: sums[(g*c+ch)] = {sum dz, sum dz*xhat} per group; dgamma/dbeta = totals.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float2* __restrict__ partial, float2* __restrict__ sums,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, BnShape s,
                                                              int accumulate) {
    const int ch = (blockIdx.x * 256 + threadIdx.x) >> 6;  // wave-uniform
    const int lane = threadIdx.x & 63;
    if (ch >= s.c) return;
    float tg = 0.f, tb = 0.f;
    for (int grp = 0; grp < s.groups; ++grp) {
        float a0 = 0.f, a1 = 0.f;
        for (int rb = lane; rb < s.rblocks; rb += 64) {
            const float2 pr = partial[((long)grp * s.rblocks + rb) * s.c + ch];
            a0 += pr.x;
            a1 += pr.y;
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {
            sums[grp * s.c + ch] = make_float2(a0, a1);
            tb += a0;
            tg += a1;
        }
    }
    if (lane == 0) {
        if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + tg;
        if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + tb;
    }
}

// dx = gamma*rstd*(dz - sum_dz/rows - xhat*sum_dz_xhat/rows)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float2* __restrict__ sums, float* __restrict__ dx, int groups, int rows,
                                    int c, int act, float act_param) {
    const unsigned total = (unsigned)groups * rows * c;
    const unsigned stride = gridDim.x * blockDim.x;
    const unsigned per_group = (unsigned)rows * c;
    const float inv_rows = 1.0f / (float)rows;
    if ((c & 3) == 0) {
        for (unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) * 4u; i < total; i += stride * 4u) {
            const unsigned ch = i % (unsigned)c;
            const unsigned gi = (i / per_group) * c + ch;
            const float4 xv = *reinterpret_cast<const float4*>(x + i);
            const float4 dv = *reinterpret_cast<const float4*>(dy + i);
            const float4 mu = *reinterpret_cast<const float4*>(mean + gi);
            const float4 rs = *reinterpret_cast<const float4*>(rstd + gi);
            const float4 ga = *reinterpret_cast<const float4*>(gamma + ch);
            const float4 be = *reinterpret_cast<const float4*>(beta + ch);
            const float4 sa = *reinterpret_cast<const float4*>(sums + gi);       // {s0,s1} of ch, ch+1
            const float4 sb = *reinterpret_cast<const float4*>(sums + gi + 2);   // {s0,s1} of ch+2, ch+3
            float xh;
            float4 o;
            float dz = dz_of(xv.x, dv.x, mu.x, rs.x, ga.x, be.x, act, act_param, xh);
            o.x = ga.x * rs.x * (dz - sa.x * inv_rows - xh * (sa.y * inv_rows));
            dz = dz_of(xv.y, dv.y, mu.y, rs.y, ga.y, be.y, act, act_param, xh);
            o.y = ga.y * rs.y * (dz - sa.z * inv_rows - xh * (sa.w * inv_rows));
            dz = dz_of(xv.z, dv.z, mu.z, rs.z, ga.z, be.z, act, act_param, xh);
            o.z = ga.z * rs.z * (dz - sb.x * inv_rows - xh * (sb.y * inv_rows));
            dz = dz_of(xv.w, dv.w, mu.w, rs.w, ga.w, be.w, act, act_param, xh);
            o.w = ga.w * rs.w * (dz - sb.z * inv_rows - xh * (sb.w * inv_rows));
            *reinterpret_cast<float4*>(dx + i) = o;
        }
    } else {
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const unsigned ch = i % (unsigned)c;
            const unsigned gi = (i / per_group) * c + ch;
            float xhat;
            const float ga = gamma[ch], rs = rstd[gi];
            const float dz = dz_of(x[i], dy[i], mean[gi], rs, ga, beta[ch], act, act_param, xhat);
            const float2 sm = sums[gi];
            dx[i] = ga * rs * (dz - sm.x * inv_rows - xhat * (sm.y * inv_rows));
        }
    }
}

// ---------------------------------------------------------------------------- one-launch path
// Small activations (rows <= 1024 per group, c % 4 == 0: the 4x4 layers of the 64x64 networks, the 2-D batch
// norms after linear layers).  The three-launch path above costs 8-11 us however small the tensor is, because
// each launch is a chain of dependent memory round trips.  Here one workgroup owns FOUR channels of one
// group outright: thread t holds rows t, t+256, ... of those channels in registers (one float4 per row), the
// statistics are exact two-pass sums reduced inside the workgroup (shuffle tree + 4 wave partials in LDS, fixed
// order), and the normalised rows are written from the registers: one read, one write, one launch
// (rows 64 / 1024: 3.4 / 5.2 us forward, 4.1 / 8.3 us backward against 7.5 / 8.5 and 9.3 / 10.8).
// A workgroup uses 16 bytes of every 128-byte line it pulls into its CU, which is why this stops paying
// above ~1024 rows (4096 rows: 25 us backward against 12), and the backward pass, whose one workgroup per quad
// must see every group for dgamma / dbeta, keeps the three launches when groups are stacked.
// A 128-byte line holds 8 neighbouring channel quads, so the quads are dealt to workgroups such that the 8
// workgroups sharing a line have ids 8 apart: with workgroups assigned to the 8 XCDs round robin they sit on
// the same XCD and the line crosses the fabric once.
constexpr int kFusedMaxRows = 1024;

__device__ __forceinline__ int fused_quad(int block, int nquads) {
    return (nquads % 8 == 0) ? (block % 8) * (nquads / 8) + block / 8 : block;
}

// sum over the 256 threads of the workgroup, result in every thread; `red` is 4 float4 of LDS
__device__ __forceinline__ float4 block_sum4(float4 v, float4* red) {
    v.x = wave_sum(v.x); v.y = wave_sum(v.y); v.z = wave_sum(v.z); v.w = wave_sum(v.w);
    __syncthreads();                        // the previous use of `red` is over
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return f4_add(f4_add(red[0], red[1]), f4_add(red[2], red[3]));
}

template <int RPT>
__global__ __launch_bounds__(256) void bn_fwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ moving_mean,
                                                           float* __restrict__ moving_var, const float* __restrict__ residual,
                                                           float* __restrict__ y, float* __restrict__ save_mean,
                                                           float* __restrict__ save_rstd, int groups, int rows, int c, float eps,
                                                           float momentum, int unbiased, int act, float act_param,
                                                           int all_groups) {
    __shared__ float4 red[4];
    const int c4 = c >> 2;
    const int q = fused_quad(blockIdx.x, c4);
    const int g0 = all_groups ? 0 : blockIdx.y, g1 = all_groups ? groups : g0 + 1;
    const float4 ga = reinterpret_cast<const float4*>(gamma)[q], be = reinterpret_cast<const float4*>(beta)[q];
    float4 mm = make_float4(0.f, 0.f, 0.f, 0.f), mv = mm;
    if (moving_mean != nullptr) {
        mm = reinterpret_cast<const float4*>(moving_mean)[q];
        mv = reinterpret_cast<const float4*>(moving_var)[q];
    }
    const float n = (float)rows, inv_n = 1.0f / n;
    for (int grp = g0; grp < g1; ++grp) {
        const long base = (long)grp * rows * c4 + q;
        const float4* xp = reinterpret_cast<const float4*>(x) + base;
        float4 v[RPT];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = threadIdx.x + 256 * i;
            v[i] = (r < rows) ? xp[(long)r * c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            s = f4_add(s, v[i]);
        }
        s = block_sum4(s, red);
        const float4 mean = make_float4(s.x * inv_n, s.y * inv_n, s.z * inv_n, s.w * inv_n);
        float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            if (threadIdx.x + 256 * i < rows) {
                const float4 d = f4_sub(v[i], mean);
                m2 = f4_fma(d, d, m2);
            }
        }
        m2 = block_sum4(m2, red);
        const float4 var = make_float4(m2.x * inv_n, m2.y * inv_n, m2.z * inv_n, m2.w * inv_n);
        const float4 rstd = make_float4(1.0f / sqrtf(var.x + eps), 1.0f / sqrtf(var.y + eps), 1.0f / sqrtf(var.z + eps),
                                        1.0f / sqrtf(var.w + eps));
        if (threadIdx.x == 0) {
            reinterpret_cast<float4*>(save_mean + (long)grp * c)[q] = mean;
            reinterpret_cast<float4*>(save_rstd + (long)grp * c)[q] = rstd;
        }
        const float corr = unbiased ? n / fmaxf(n - 1.f, 1.f) : 1.f;
        const float keep = momentum, take = 1.f - momentum;
        mm = make_float4(mm.x * keep + mean.x * take, mm.y * keep + mean.y * take, mm.z * keep + mean.z * take,
                         mm.w * keep + mean.w * take);
        mv = make_float4(mv.x * keep + var.x * corr * take, mv.y * keep + var.y * corr * take,
                         mv.z * keep + var.z * corr * take, mv.w * keep + var.w * corr * take);
        float4* yp = reinterpret_cast<float4*>(y) + base;
        const float4* rp = residual ? reinterpret_cast<const float4*>(residual) + base : nullptr;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = threadIdx.x + 256 * i;
            if (r < rows) {
                float4 o;
                o.x = apply_act((v[i].x - mean.x) * rstd.x * ga.x + be.x, act, act_param);
                o.y = apply_act((v[i].y - mean.y) * rstd.y * ga.y + be.y, act, act_param);
                o.z = apply_act((v[i].z - mean.z) * rstd.z * ga.z + be.z, act, act_param);
                o.w = apply_act((v[i].w - mean.w) * rstd.w * ga.w + be.w, act, act_param);
                if (rp) o = f4_add(o, rp[(long)r * c4]);
                yp[(long)r * c4] = o;
            }
        }
    }
    if (moving_mean != nullptr && threadIdx.x == 0) {
        reinterpret_cast<float4*>(moving_mean)[q] = mm;
        reinterpret_cast<float4*>(moving_var)[q] = mv;
    }
}

// One workgroup per channel quad takes every group in turn: dgamma / dbeta are sums over the groups.
template <int RPT>
__global__ __launch_bounds__(256) void bn_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ dy,
                                                           float* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate, int groups, int rows, int c,
                                                           int act, float act_param) {
    __shared__ float4 red[4];
    const int c4 = c >> 2;
    const int q = fused_quad(blockIdx.x, c4);
    const float4 ga = reinterpret_cast<const float4*>(gamma)[q], be = reinterpret_cast<const float4*>(beta)[q];
    const float inv_rows = 1.0f / (float)rows;
    float4 tg = make_float4(0.f, 0.f, 0.f, 0.f), tb = tg;
    for (int grp = 0; grp < groups; ++grp) {
        const long base = (long)grp * rows * c4 + q;
        const float4* xp = reinterpret_cast<const float4*>(x) + base;
        const float4* dp = reinterpret_cast<const float4*>(dy) + base;
        const float4 mu = reinterpret_cast<const float4*>(mean + (long)grp * c)[q];
        const float4 rs = reinterpret_cast<const float4*>(rstd + (long)grp * c)[q];
        float4 xv[RPT], dv[RPT];
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = threadIdx.x + 256 * i;
            if (r < rows) {
                xv[i] = xp[(long)r * c4];
                dv[i] = dp[(long)r * c4];
                float4 xh, dz;
                dz.x = dz_of(xv[i].x, dv[i].x, mu.x, rs.x, ga.x, be.x, act, act_param, xh.x);
                dz.y = dz_of(xv[i].y, dv[i].y, mu.y, rs.y, ga.y, be.y, act, act_param, xh.y);
                dz.z = dz_of(xv[i].z, dv[i].z, mu.z, rs.z, ga.z, be.z, act, act_param, xh.z);
                dz.w = dz_of(xv[i].w, dv[i].w, mu.w, rs.w, ga.w, be.w, act, act_param, xh.w);
                s0 = f4_add(s0, dz);
                s1 = f4_fma(dz, xh, s1);
            } else {
                xv[i] = dv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        s0 = block_sum4(s0, red);
        s1 = block_sum4(s1, red);
        tb = f4_add(tb, s0);
        tg = f4_add(tg, s1);
        if (dx != nullptr) {
            float4* op = reinterpret_cast<float4*>(dx) + base;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int r = threadIdx.x + 256 * i;
                if (r < rows) {
                    float xh;
                    float4 o;
                    float dz = dz_of(xv[i].x, dv[i].x, mu.x, rs.x, ga.x, be.x, act, act_param, xh);
                    o.x = ga.x * rs.x * (dz - s0.x * inv_rows - xh * (s1.x * inv_rows));
                    dz = dz_of(xv[i].y, dv[i].y, mu.y, rs.y, ga.y, be.y, act, act_param, xh);
                    o.y = ga.y * rs.y * (dz - s0.y * inv_rows - xh * (s1.y * inv_rows));
                    dz = dz_of(xv[i].z, dv[i].z, mu.z, rs.z, ga.z, be.z, act, act_param, xh);
                    o.z = ga.z * rs.z * (dz - s0.z * inv_rows - xh * (s1.z * inv_rows));
                    dz = dz_of(xv[i].w, dv[i].w, mu.w, rs.w, ga.w, be.w, act, act_param, xh);
                    o.w = ga.w * rs.w * (dz - s0.w * inv_rows - xh * (s1.w * inv_rows));
                    op[(long)r * c4] = o;
                }
            }
        }
    }
    if (threadIdx.x == 0) {
        if (dgamma) {
            float4 o = accumulate ? reinterpret_cast<const float4*>(dgamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            reinterpret_cast<float4*>(dgamma)[q] = f4_add(o, tg);
        }
        if (dbeta) {
            float4 o = accumulate ? reinterpret_cast<const float4*>(dbeta)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            reinterpret_cast<float4*>(dbeta)[q] = f4_add(o, tb);
        }
    }
}

static bool fused_bn_ok(int rows, int c, const void* p0, const void* p1, const void* p2, const void* p3) {
    auto a16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return rows <= kFusedMaxRows && (c & 3) == 0 && a16(p0) && a16(p1) && a16(p2) && a16(p3) &&
           !knobs().three_launch_bn;
}

#define GAN_FUSED_BN_DISPATCH(KERNEL, rows, GRID, st, ...)                                            \
    do {                                                                                              \
        if ((rows) <= 256) hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(256), 0, st, __VA_ARGS__);        \
        else if ((rows) <= 512) hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(256), 0, st, __VA_ARGS__);   \
        else hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), 0, st, __VA_ARGS__);                      \
    } while (0)

// ---------------------------------------------------------------------------- two launches, larger tensors
// Inside a training iteration every launch of the three-launch path starts cold, and the finalize launch in the middle
// -- a handful of waves that load the block partials, merge them and store two numbers per channel -- costs as much
// as its neighbours: skipping it (timing experiment) shortens the DCGAN-64 step by 5.5 %.  Here the statistics kernel
// and the apply kernel share ONE decomposition, (row block) x (tile of 64 channels), and every apply workgroup merges
// the partial results of its channel tile itself (all workgroups of a tile the same sums in the same order: identical
// bits), so nothing stands between the two launches.  A workgroup of the second launch reads the rows the workgroup
// of the same grid position read in the first (same XCD: they come out of its L2).  The partials are merged as
// division-free sums about the first block's mean; the row block count is kept small (<= 256 workgroups per launch)
// so that the merge is a few loads per thread.
//
// The rows of the first launch may come from a contraction's split-K slabs instead of a finished tensor (RowSource):
// the launch then sums the slabs, adds the bias, stores the finished row AND takes its statistics, i.e. it replaces
// the contraction's slab reduction launch and the statistics launch.  The sums are taken in the order the separate
// launches take them: same bits.
//
// Channel quad of one row.  KIND 0: from the finished tensor `t`; 1: summed over the slabs (+ bias) and, if `store`,
// stored to src.out.  i4 = float4 index of the quad in the flat [groups * rows][c] tensor, q = its quad within the row.
template <int KIND>
__device__ __forceinline__ float4 row_quad(const RowSource& src, const float4* __restrict__ t, long i4, int q, bool store) {
    if (KIND == 0) return t[i4];
    const long stride4 = src.slab_elems >> 2;
    const float4* sp = reinterpret_cast<const float4*>(src.slabs) + i4;
    float4 v = sp[0];
#pragma unroll 4
    for (int k = 1; k < src.splits; ++k) v = f4_add(v, sp[(long)k * stride4]);
    if (src.bias) v = f4_add(v, reinterpret_cast<const float4*>(src.bias)[q]);
    if (store) reinterpret_cast<float4*>(src.out)[i4] = v;
    return v;
}

// {sum pa', sum pb'} over the nb partials of group `grp` for the thread's channel quad; every thread of the workgroup
// must call it.  MOMENTS: partials are {mean, m2} of n rows and the sums are {n d, m2 + n d^2}, d = mean - shift.
template <bool MOMENTS>
__device__ __forceinline__ void tile_bn_sums(const float2* __restrict__ partial, const TileBn& s, int grp, int ch0, int quad,
                                             int rl, float4 shift, float4 (*red)[256], float4& out_a, float4& out_b) {
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
#pragma unroll 4
    for (int b = rl; b < s.nb; b += 16) {
        const float4* pp = reinterpret_cast<const float4*>(partial + ((long)grp * s.nb + b) * s.c + ch0) + 2 * quad;
        const float4 p0 = pp[0], p1 = pp[1];   // {a, b} of channels 0, 1 and of channels 2, 3 of the quad
        const float4 pa = make_float4(p0.x, p0.z, p1.x, p1.z), pb = make_float4(p0.y, p0.w, p1.y, p1.w);
        if (MOMENTS) {
            const int r0 = b * s.rpb;
            const float n = (float)(min(s.rows, r0 + s.rpb) - r0);
            const float4 d = f4_sub(pa, shift);
            const float4 nd = make_float4(n * d.x, n * d.y, n * d.z, n * d.w);
            sa = f4_add(sa, nd);
            sb = f4_add(sb, f4_fma(nd, d, pb));
        } else {
            sa = f4_add(sa, pa);
            sb = f4_add(sb, pb);
        }
    }
    __syncthreads();
    red[0][threadIdx.x] = sa;
    red[1][threadIdx.x] = sb;
    __syncthreads();
    out_a = red[0][quad];
    out_b = red[1][quad];
    for (int j = 1; j < 16; ++j) {
        out_a = f4_add(out_a, red[0][quad + 16 * j]);
        out_b = f4_add(out_b, red[1][quad + 16 * j]);
    }
}

// {mean, biased variance} of group `grp` for the thread's quad
__device__ __forceinline__ void tile_bn_moments(const float2* __restrict__ partial, const TileBn& s, int grp, int ch0, int quad,
                                                int rl, float4 (*red)[256], float4& mean, float4& var) {
    const float4* p0 = reinterpret_cast<const float4*>(partial + (long)grp * s.nb * s.c + ch0) + 2 * quad;
    const float4 f0 = p0[0], f1 = p0[1];
    const float4 shift = make_float4(f0.x, f0.z, f1.x, f1.z);
    float4 a, b;
    tile_bn_sums<true>(partial, s, grp, ch0, quad, rl, shift, red, a, b);
    const float inv = 1.0f / (float)s.rows;
    mean = make_float4(shift.x + a.x * inv, shift.y + a.y * inv, shift.z + a.z * inv, shift.w + a.w * inv);
    var = make_float4(fmaxf(b.x - a.x * a.x * inv, 0.f) * inv, fmaxf(b.y - a.y * a.y * inv, 0.f) * inv,
                      fmaxf(b.z - a.z * a.z * inv, 0.f) * inv, fmaxf(b.w - a.w * a.w * inv, 0.f) * inv);
}

// {mean, m2} of a row block from its shifted sums.  Contraction is switched off: the kernels that share this function
// must produce the same bits, and a fused multiply-add the compiler forms at one call site and not at another differs
// in the last bit whenever 1 / rows is inexact.
__device__ __forceinline__ void tile_stats_finish(float4 shift, float4 s1, float4 s2, int rows, float4& mean, float4& m2) {
#pragma clang fp contract(off)
    const float inv = 1.0f / (float)rows;
    mean = make_float4(shift.x + s1.x * inv, shift.y + s1.y * inv, shift.z + s1.z * inv, shift.w + s1.w * inv);
    m2 = make_float4(s2.x - s1.x * s1.x * inv, s2.y - s1.y * s1.y * inv, s2.z - s1.z * s1.z * inv, s2.w - s1.w * s1.w * inv);
}

__device__ __forceinline__ void store_partial_quad(float2* partial, long idx, float4 a, float4 b) {
    float4* pp = reinterpret_cast<float4*>(partial + idx);
    pp[0] = make_float4(a.x, b.x, a.y, b.y);
    pp[1] = make_float4(a.z, b.z, a.w, b.w);
}

// grid (nb, c / 64, groups); thread = (row lane 0..15, channel quad 0..15)
template <int KIND>
__global__ __launch_bounds__(256) void bn_tile_stats_kernel(const float* __restrict__ x, float2* __restrict__ partial, TileBn s,
                                                            const RowSource src) {
    __shared__ float4 red[2][256];
    const int quad = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int rb = blockIdx.x, ch0 = blockIdx.y * 64, grp = blockIdx.z;
    const int c4 = s.c >> 2, q = (ch0 >> 2) + quad;
    const int r0 = rb * s.rpb, r1 = min(s.rows, r0 + s.rpb);
    const long rowg0 = (long)grp * s.rows;
    const float4* t = reinterpret_cast<const float4*>(x);
    // (every row lane forms the block's first row itself; only the loop below stores rows)
    const float4 shift = row_quad<KIND>(src, t, (rowg0 + r0) * c4 + q, q, false);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 16) {
        const float4 d = f4_sub(row_quad<KIND>(src, t, (rowg0 + r) * c4 + q, q, true), shift);
        s1 = f4_add(s1, d);
        s2 = f4_fma(d, d, s2);
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < 16; ++j) {
            s1 = f4_add(s1, red[0][quad + 16 * j]);
            s2 = f4_add(s2, red[1][quad + 16 * j]);
        }
        float4 mean, m2;
        tile_stats_finish(shift, s1, s2, r1 - r0, mean, m2);
        store_partial_quad(partial, ((long)grp * s.nb + rb) * s.c + ch0 + quad * 4, mean, m2);
    }
}

// The apply kernels of both directions begin with a chain of dependent latencies (block partials -> LDS merge ->
// statistics) before they can use a single row.  The first kTilePre rows of every row lane (all of them for a row block
// of 64 rows) are therefore requested BEFORE that chain and wait in registers: the two latencies overlap.
constexpr int kTilePre = 4;

__device__ __forceinline__ void tile_bn_prefetch(const float4* p, int r0, int r1, int rl, int c4, float4 (&pre)[kTilePre]) {
#pragma unroll
    for (int j = 0; j < kTilePre; ++j) {
        const int r = r0 + rl + 16 * j;
        pre[j] = r < r1 ? p[(long)r * c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int ACT>
__device__ __forceinline__ void tile_bn_apply_row(float4 v, const float4* rp, float4* yp, long at, float4 mean, float4 rstd,
                                                  float4 ga, float4 be, float act_param) {
    float4 o;
    o.x = apply_act((v.x - mean.x) * rstd.x * ga.x + be.x, ACT, act_param);
    o.y = apply_act((v.y - mean.y) * rstd.y * ga.y + be.y, ACT, act_param);
    o.z = apply_act((v.z - mean.z) * rstd.z * ga.z + be.z, ACT, act_param);
    o.w = apply_act((v.w - mean.w) * rstd.w * ga.w + be.w, ACT, act_param);
    if (rp) o = f4_add(o, rp[at]);
    yp[at] = o;
}

template <int ACT>
__device__ __forceinline__ void tile_bn_apply_rows(const float4* xp, const float4* rp, float4* yp, int r0, int r1, int rl, int c4,
                                                   float4 mean, float4 rstd, float4 ga, float4 be, float act_param,
                                                   const float4 (&pre)[kTilePre]) {
#pragma unroll
    for (int j = 0; j < kTilePre; ++j) {
        const int r = r0 + rl + 16 * j;
        if (r < r1) tile_bn_apply_row<ACT>(pre[j], rp, yp, (long)r * c4, mean, rstd, ga, be, act_param);
    }
#pragma unroll 4
    for (int r = r0 + rl + 16 * kTilePre; r < r1; r += 16)
        tile_bn_apply_row<ACT>(xp[(long)r * c4], rp, yp, (long)r * c4, mean, rstd, ga, be, act_param);
}

// The activation is the same for every element of a launch: the row loops are instantiated per activation and the
// dispatch happens once, outside them.
#define GAN_ACT_DISPATCH(act, CALL)                                                 \
    switch (act) {                                                                  \
        case GAN_ACT_RELU: { constexpr int ACT = GAN_ACT_RELU; CALL; } break;       \
        case GAN_ACT_LRELU: { constexpr int ACT = GAN_ACT_LRELU; CALL; } break;     \
        case GAN_ACT_TANH: { constexpr int ACT = GAN_ACT_TANH; CALL; } break;       \
        case GAN_ACT_SIGMOID: { constexpr int ACT = GAN_ACT_SIGMOID; CALL; } break; \
        default: { constexpr int ACT = GAN_ACT_NONE; CALL; } break;                 \
    }

__global__ __launch_bounds__(256) void bn_tile_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ moving_mean,
                                                            float* __restrict__ moving_var, const float* __restrict__ residual,
                                                            float* __restrict__ y, float* __restrict__ save_mean,
                                                            float* __restrict__ save_rstd, const float2* __restrict__ partial,
                                                            TileBn s, float eps, float momentum, int unbiased, int act,
                                                            float act_param) {
    __shared__ float4 red[2][256];
    const int quad = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int rb = blockIdx.x, ch0 = blockIdx.y * 64, grp = blockIdx.z;
    const int c4 = s.c >> 2, q = (ch0 >> 2) + quad;
    const float4 ga = reinterpret_cast<const float4*>(gamma)[q], be = reinterpret_cast<const float4*>(beta)[q];
    const int r0 = rb * s.rpb, r1 = min(s.rows, r0 + s.rpb);
    const long base = ((long)grp * s.rows) * c4 + q;
    const float4* xp = reinterpret_cast<const float4*>(x) + base;
    float4 pre[kTilePre];
    tile_bn_prefetch(xp, r0, r1, rl, c4, pre);
    const bool moves = (moving_mean != nullptr && rb == 0 && grp == 0);
    float4 mm = make_float4(0.f, 0.f, 0.f, 0.f), mv = mm;
    if (moves) {
        mm = reinterpret_cast<const float4*>(moving_mean)[q];
        mv = reinterpret_cast<const float4*>(moving_var)[q];
    }
    float4 mean, var;
    tile_bn_moments(partial, s, grp, ch0, quad, rl, red, mean, var);
    const float4 rstd = make_float4(1.0f / sqrtf(var.x + eps), 1.0f / sqrtf(var.y + eps), 1.0f / sqrtf(var.z + eps),
                                    1.0f / sqrtf(var.w + eps));
    if (rb == 0 && rl == 0) {
        reinterpret_cast<float4*>(save_mean + (long)grp * s.c)[q] = mean;
        reinterpret_cast<float4*>(save_rstd + (long)grp * s.c)[q] = rstd;
    }
    if (moves) {
        // the first workgroup of a channel tile moves the averages once per group, in group order
        const float n = (float)s.rows;
        const float corr = unbiased ? n / fmaxf(n - 1.f, 1.f) : 1.f;
        const float keep = momentum, take = 1.f - momentum;
        for (int g = 0; g < s.groups; ++g) {
            float4 gm = mean, gv = var;
            if (g > 0) tile_bn_moments(partial, s, g, ch0, quad, rl, red, gm, gv);
            mm = make_float4(mm.x * keep + gm.x * take, mm.y * keep + gm.y * take, mm.z * keep + gm.z * take, mm.w * keep + gm.w * take);
            mv = make_float4(mv.x * keep + gv.x * corr * take, mv.y * keep + gv.y * corr * take, mv.z * keep + gv.z * corr * take,
                             mv.w * keep + gv.w * corr * take);
        }
        if (rl == 0) {
            reinterpret_cast<float4*>(moving_mean)[q] = mm;
            reinterpret_cast<float4*>(moving_var)[q] = mv;
        }
    }
    const float4* rp = residual ? reinterpret_cast<const float4*>(residual) + base : nullptr;
    float4* yp = reinterpret_cast<float4*>(y) + base;
    GAN_ACT_DISPATCH(act, (tile_bn_apply_rows<ACT>(xp, rp, yp, r0, r1, rl, c4, mean, rstd, ga, be, act_param, pre)));
}

template <int ACT>
__device__ __forceinline__ float4 dz4(float4 xv, float4 dv, float4 mu, float4 rs, float4 ga, float4 be, float act_param,
                                      float4& xh) {
    float4 dz;
    dz.x = dz_of(xv.x, dv.x, mu.x, rs.x, ga.x, be.x, ACT, act_param, xh.x);
    dz.y = dz_of(xv.y, dv.y, mu.y, rs.y, ga.y, be.y, ACT, act_param, xh.y);
    dz.z = dz_of(xv.z, dv.z, mu.z, rs.z, ga.z, be.z, ACT, act_param, xh.z);
    dz.w = dz_of(xv.w, dv.w, mu.w, rs.w, ga.w, be.w, ACT, act_param, xh.w);
    return dz;
}

// (dp = the float4 view of dy, or unused when the rows of dy come from slabs; base4 = float4 index of row 0 of the
// group for this thread's quad q)
template <int ACT, int KIND>
__device__ __forceinline__ void tile_bn_bwd_sum_rows(const float4* xp, const float4* dp, const RowSource& src, long base4,
                                                     int q, int r0, int r1, int rl, int c4, float4 mu, float4 rs,
                                                     float4 ga, float4 be, float act_param, float4& s0, float4& s1) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 16) {
        float4 xh;
        const float4 dv = row_quad<KIND>(src, dp, base4 + (long)r * c4, q, true);
        const float4 dz = dz4<ACT>(xp[(long)r * c4], dv, mu, rs, ga, be, act_param, xh);
        s0 = f4_add(s0, dz);
        s1 = f4_fma(dz, xh, s1);
    }
}

template <int ACT>
__device__ __forceinline__ void tile_bn_bwd_apply_row(float4 xv, float4 dv, float4* op, long at, float4 mu, float4 rs, float4 ga,
                                                      float4 be, float act_param, float4 a0, float4 a1, float inv_rows,
                                                      float4& colsum) {
    float4 xh, o;
    const float4 dz = dz4<ACT>(xv, dv, mu, rs, ga, be, act_param, xh);
    o.x = ga.x * rs.x * (dz.x - a0.x * inv_rows - xh.x * (a1.x * inv_rows));
    o.y = ga.y * rs.y * (dz.y - a0.y * inv_rows - xh.y * (a1.y * inv_rows));
    o.z = ga.z * rs.z * (dz.z - a0.z * inv_rows - xh.z * (a1.z * inv_rows));
    o.w = ga.w * rs.w * (dz.w - a0.w * inv_rows - xh.w * (a1.w * inv_rows));
    op[at] = o;
    colsum = f4_add(colsum, o);
}

// (pre_x / pre_d: the first kTilePre rows of this row lane, requested before the block sums were merged)
template <int ACT>
__device__ __forceinline__ void tile_bn_bwd_apply_rows(const float4* xp, const float4* dp, float4* op, int r0, int r1, int rl,
                                                       int c4, float4 mu, float4 rs, float4 ga, float4 be, float act_param,
                                                       float4 a0, float4 a1, float inv_rows, float4& colsum,
                                                       const float4 (&pre_x)[kTilePre], const float4 (&pre_d)[kTilePre]) {
#pragma unroll
    for (int j = 0; j < kTilePre; ++j) {
        const int r = r0 + rl + 16 * j;
        if (r < r1) tile_bn_bwd_apply_row<ACT>(pre_x[j], pre_d[j], op, (long)r * c4, mu, rs, ga, be, act_param, a0, a1, inv_rows, colsum);
    }
#pragma unroll 4
    for (int r = r0 + rl + 16 * kTilePre; r < r1; r += 16)
        tile_bn_bwd_apply_row<ACT>(xp[(long)r * c4], dp[(long)r * c4], op, (long)r * c4, mu, rs, ga, be, act_param, a0, a1, inv_rows,
                                   colsum);
}

template <int KIND>
__global__ __launch_bounds__(256) void bn_tile_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float2* __restrict__ partial, TileBn s, int act, float act_param,
                                                               const RowSource src) {
    __shared__ float4 red[2][256];
    const int quad = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int rb = blockIdx.x, ch0 = blockIdx.y * 64, grp = blockIdx.z;
    const int c4 = s.c >> 2, q = (ch0 >> 2) + quad;
    const int r0 = rb * s.rpb, r1 = min(s.rows, r0 + s.rpb);
    const long base = ((long)grp * s.rows) * c4 + q;
    const float4* xp = reinterpret_cast<const float4*>(x) + base;
    const float4* dp = reinterpret_cast<const float4*>(dy);
    const float4 mu = reinterpret_cast<const float4*>(mean + (long)grp * s.c)[q];
    const float4 rs = reinterpret_cast<const float4*>(rstd + (long)grp * s.c)[q];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[q], be = reinterpret_cast<const float4*>(beta)[q];
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    GAN_ACT_DISPATCH(act, (tile_bn_bwd_sum_rows<ACT, KIND>(xp, dp, src, base, q, r0, r1, rl, c4, mu, rs, ga, be, act_param, s0, s1)));
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < 16; ++j) {
            s0 = f4_add(s0, red[0][quad + 16 * j]);
            s1 = f4_add(s1, red[1][quad + 16 * j]);
        }
        store_partial_quad(partial, ((long)grp * s.nb + rb) * s.c + ch0 + quad * 4, s0, s1);
    }
}

__global__ __launch_bounds__(256) void bn_tile_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float2* __restrict__ partial, float* __restrict__ dx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int accumulate, TileBn s, int act, float act_param,
                                                                float* __restrict__ colsum_partial) {
    __shared__ float4 red[2][256];
    const int quad = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int rb = blockIdx.x, ch0 = blockIdx.y * 64, grp = blockIdx.z;
    const int c4 = s.c >> 2, q = (ch0 >> 2) + quad;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 mu = reinterpret_cast<const float4*>(mean + (long)grp * s.c)[q];
    const float4 rs = reinterpret_cast<const float4*>(rstd + (long)grp * s.c)[q];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[q], be = reinterpret_cast<const float4*>(beta)[q];
    const int r0 = rb * s.rpb, r1 = min(s.rows, r0 + s.rpb);
    const long base = ((long)grp * s.rows) * c4 + q;
    const float4* xp = reinterpret_cast<const float4*>(x) + base;
    const float4* dp = reinterpret_cast<const float4*>(dy) + base;
    float4 pre_x[kTilePre], pre_d[kTilePre];
    if (dx != nullptr) {        // (uniform over the launch)
        tile_bn_prefetch(xp, r0, r1, rl, c4, pre_x);
        tile_bn_prefetch(dp, r0, r1, rl, c4, pre_d);
    }
    float4 a0, a1;
    tile_bn_sums<false>(partial, s, grp, ch0, quad, rl, zero, red, a0, a1);
    if (rb == 0 && grp == 0 && (dgamma != nullptr || dbeta != nullptr)) {
        // dgamma / dbeta are sums over the groups, taken in group order by the first workgroup of the channel tile
        float4 tb = a0, tg = a1;
        for (int g = 1; g < s.groups; ++g) {
            float4 b0, b1;
            tile_bn_sums<false>(partial, s, g, ch0, quad, rl, zero, red, b0, b1);
            tb = f4_add(tb, b0);
            tg = f4_add(tg, b1);
        }
        if (rl == 0) {
            if (dgamma) reinterpret_cast<float4*>(dgamma)[q] = f4_add(accumulate ? reinterpret_cast<const float4*>(dgamma)[q] : zero, tg);
            if (dbeta) reinterpret_cast<float4*>(dbeta)[q] = f4_add(accumulate ? reinterpret_cast<const float4*>(dbeta)[q] : zero, tb);
        }
    }
    if (dx == nullptr) return;
    float4* op = reinterpret_cast<float4*>(dx) + base;
    float4 cs = zero;
    GAN_ACT_DISPATCH(act, (tile_bn_bwd_apply_rows<ACT>(xp, dp, op, r0, r1, rl, c4, mu, rs, ga, be, act_param, a0, a1,
                                                       1.0f / (float)s.rows, cs, pre_x, pre_d)));
    if (colsum_partial == nullptr) return;      // (uniform over the launch)
    // the column sum of the rows this workgroup stored: first stage of the bias gradient of the layer in front
    __syncthreads();
    red[0][threadIdx.x] = cs;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < 16; ++j) cs = f4_add(cs, red[0][quad + 16 * j]);
        reinterpret_cast<float4*>(colsum_partial + ((long)grp * s.nb + rb) * s.c)[q] = cs;
    }
}

// The two-launch plan of a tensor, or false.
static bool tile_bn_plan(int groups, int rows, int c, TileBn& s, int min_rows = kFusedMaxRows + 1) {
    if (knobs().three_launch_bn || !knobs().two_launch_bn) return false;
    if (c % 64 != 0 || rows < min_rows) return false;
    const long ctiles = c / 64;
    if (groups * ctiles > 256) return false;
    // (256 workgroups per launch, one per CU, measured best in the DCGAN-64 step: 128 -0.8 %, 512 +1.7 %, 1024 +2.6 %)
    long nb = 256 / (groups * ctiles);          // <= 256 workgroups per launch ...
    const long most = (rows + 63) / 64;         // ... of at least 64 rows each
    if (nb > most) nb = most;
    if (nb < 1) nb = 1;
    s.groups = groups; s.rows = rows; s.c = c;
    s.rpb = (int)((rows + nb - 1) / nb);
    s.nb = (rows + s.rpb - 1) / s.rpb;
    return true;
}

static bool all16(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15) != 0) return false;
    return true;
}

static int grid_for(long work_items) {
    long b = (work_items + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

static RowSource tensor_rows() { return RowSource{nullptr, nullptr, nullptr, 0, 0}; }

BnRide& pending_bn_ride() {
    static thread_local BnRide r = BnRide{0, nullptr, TileBn{0, 0, 0, 0, 0}, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    return r;
}

bool bn_ride_plan(int groups, int rows, int c, bool backward, TileBn* s) {
    if (knobs().three_launch_bn) return false;
    if (backward) return tile_bn_plan(groups, rows, c, *s, groups > 1 ? 256 : kFusedMaxRows + 1);
    return tile_bn_plan(groups, rows, c, *s);
}

int launch_bn_ride(const BnRide& r, const RowSource& src, hipStream_t st, const char* what) {
    const TileBn& s = r.s;
    const int kind = src.slabs == nullptr ? 0 : 1;
    GAN_CHECK_ARG(r.partial != nullptr && src.out != nullptr, "%s: batch-norm reduction without a target", what);
    GAN_CHECK_ARG(all16({src.slabs, src.out, src.bias, r.partial, r.x, r.gamma, r.beta, r.mean, r.rstd}) &&
                  (kind == 0 || (src.slab_elems & 3) == 0),
                  "%s: the fused batch-norm reduction needs 16-byte aligned tensors", what);
    const dim3 grid(s.nb, s.c / 64, s.groups);
    if (r.kind == 1) {
        if (kind == 0) hipLaunchKernelGGL(bn_tile_stats_kernel<0>, grid, dim3(256), 0, st, (const float*)src.out, r.partial, s, src);
        else hipLaunchKernelGGL(bn_tile_stats_kernel<1>, grid, dim3(256), 0, st, (const float*)nullptr, r.partial, s, src);
    } else {
        GAN_CHECK_ARG(r.x && r.gamma && r.beta && r.mean && r.rstd, "%s: batch-norm backward sums: NULL tensor", what);
        if (kind == 0)
            hipLaunchKernelGGL(bn_tile_bwd_sums_kernel<0>, grid, dim3(256), 0, st, r.x, (const float*)src.out, r.gamma, r.beta, r.mean,
                               r.rstd, r.partial, s, r.act, r.act_param, src);
        else
            hipLaunchKernelGGL(bn_tile_bwd_sums_kernel<1>, grid, dim3(256), 0, st, r.x, (const float*)nullptr, r.gamma, r.beta, r.mean,
                               r.rstd, r.partial, s, r.act, r.act_param, src);
    }
    return check_launch(what);
}

size_t bn_ws_bytes(int groups, int rows, int c) {
    const BnShape s = bn_shape(groups, rows, c);
    size_t blocks = (size_t)s.rblocks;
    TileBn ts;
    if (tile_bn_plan(groups, rows, c, ts, groups > 1 ? 256 : kFusedMaxRows + 1) && (size_t)ts.nb > blocks) blocks = (size_t)ts.nb;
    return ((size_t)groups * blocks * c + (size_t)groups * c) * sizeof(float2);
}

}  // namespace gan

using namespace gan;

extern "C" size_t gan_bn_workspace_bytes(int groups, int rows, int c) {
    if (groups <= 0 || rows <= 0 || c <= 0) return 0;
    return bn_ws_bytes(groups, rows, c);
}

static int bn_args_check(const char* what, int groups, int rows, int c, int act) {
    GAN_CHECK_ARG(groups > 0 && rows > 0 && c > 0, "%s: groups, rows and c must be positive", what);
    GAN_CHECK_ARG(groups <= 65535, "%s: at most 65535 groups", what);
    GAN_CHECK_ARG((long)groups * rows * c < (1L << 31), "%s: tensors of 2^31 elements or more are not supported", what);
    GAN_CHECK_ARG(act >= GAN_ACT_NONE && act <= GAN_ACT_SIGMOID, "%s: unknown activation %d", what, act);
    return 0;
}

extern "C" int gan_bn_fwd_train(const float* x, const float* gamma, const float* beta, float* moving_mean,
                                float* moving_var, const float* residual, float* y, float* save_mean,
                                float* save_rstd, int groups, int rows, int c, float eps, float momentum,
                                int unbiased_moving_var, int act, float act_param, void* ws, size_t ws_bytes,
                                gan_stream_t stream) {
    if (int rc = bn_args_check("gan_bn_fwd_train", groups, rows, c, act)) return rc;
    GAN_CHECK_ARG(x && gamma && beta && y && save_mean && save_rstd, "gan_bn_fwd_train: NULL tensor");
    GAN_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "gan_bn_fwd_train: moving_mean and moving_var must be given together");
    const size_t need = bn_ws_bytes(groups, rows, c);
    GAN_CHECK_ARG(ws && ws_bytes >= need, "gan_bn_fwd_train: workspace of %zu bytes needed, %zu given", need, ws_bytes);
    const BnShape s = bn_shape(groups, rows, c);
    hipStream_t st = as_stream(stream);
    if (fused_bn_ok(rows, c, x, y, residual, save_mean) && fused_bn_ok(rows, c, save_rstd, gamma, beta, moving_mean) &&
        fused_bn_ok(rows, c, moving_var, nullptr, nullptr, nullptr)) {
        // stacked passes move the averages once per group, in order: one workgroup takes the groups in turn
        const int all_groups = (moving_mean != nullptr && groups > 1) ? 1 : 0;
        GAN_FUSED_BN_DISPATCH(bn_fwd_fused_kernel, rows, dim3(c / 4, all_groups ? 1 : groups), st, x, gamma, beta, moving_mean,
                              moving_var, residual, y, save_mean, save_rstd, groups, rows, c, eps, momentum,
                              unbiased_moving_var, act, act_param, all_groups);
        return check_launch("gan_bn_fwd_train");
    }
    float2* partial = static_cast<float2*>(ws);
    {
        TileBn ts;
        if (tile_bn_plan(groups, rows, c, ts) &&
            all16({x, y, residual, save_mean, save_rstd, gamma, beta, moving_mean, moving_var, ws})) {
            const dim3 grid(ts.nb, c / 64, groups);
            hipLaunchKernelGGL(bn_tile_stats_kernel<0>, grid, dim3(256), 0, st, x, partial, ts, tensor_rows());
            hipLaunchKernelGGL(bn_tile_apply_kernel, grid, dim3(256), 0, st, x, gamma, beta, moving_mean, moving_var, residual, y,
                               save_mean, save_rstd, (const float2*)partial, ts, eps, momentum, unbiased_moving_var, act, act_param);
            return check_launch("gan_bn_fwd_train");
        }
    }
    if (s.vec4)
        hipLaunchKernelGGL(bn_partial_stats_v4_kernel, dim3(s.rblocks, groups), dim3(256), 0, st, x, partial, s);
    else
        hipLaunchKernelGGL(bn_partial_stats_kernel, dim3(s.rblocks, groups, (c + kChanTile - 1) / kChanTile), dim3(256), 0,
                           st, x, partial, s);
    {
        // several passes through one layer stacked along the batch axis: each group is one pass and moves
        // the averages in turn, inside one launch
        const int chain = (moving_mean != nullptr && groups > 1) ? 1 : 0;
        const int waves = chain ? c : groups * c;
        hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, partial, save_mean,
                           save_rstd, moving_mean, moving_var, s, eps, momentum, unbiased_moving_var, chain);
    }
    const long total = (long)groups * rows * c;
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid_for((c & 3) ? total : total / 4)), dim3(256), 0, st, x, gamma,
                       beta, save_mean, save_rstd, residual, y, groups, rows, c, eps, act, act_param);
    return check_launch("gan_bn_fwd_train");
}

// moving <- momentum*moving + (1-momentum)*batch from the statistics a training forward saved
__global__ void bn_moving_update_kernel(float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                        const float* __restrict__ save_mean, const float* __restrict__ save_rstd, int c,
                                        float n, float eps, float momentum, int unbiased) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float rs = save_rstd[ch];
    float var = fmaxf(1.0f / (rs * rs) - eps, 0.f);
    if (unbiased) var *= n / fmaxf(n - 1.f, 1.f);
    moving_mean[ch] = moving_mean[ch] * momentum + save_mean[ch] * (1.f - momentum);
    moving_var[ch] = moving_var[ch] * momentum + var * (1.f - momentum);
}

extern "C" int gan_bn_moving_update(float* moving_mean, float* moving_var, const float* save_mean,
                                    const float* save_rstd, int rows, int c, float eps, float momentum,
                                    int unbiased_moving_var, gan_stream_t stream) {
    GAN_CHECK_ARG(rows > 0 && c > 0, "gan_bn_moving_update: rows and c must be positive");
    GAN_CHECK_ARG(moving_mean && moving_var && save_mean && save_rstd, "gan_bn_moving_update: NULL tensor");
    hipLaunchKernelGGL(bn_moving_update_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), moving_mean,
                       moving_var, save_mean, save_rstd, c, (float)rows, eps, momentum, unbiased_moving_var);
    return check_launch("gan_bn_moving_update");
}

// The same for up to kMovingJobs layers in ONE launch (a generator's replay is one such update per batch-norm layer,
// each a few hundred channels): the jobs travel by value in the kernel-argument segment, blockIdx.y picks the job.
constexpr int kMovingJobs = 64;
struct MovingJobs {
    gan_bn_moving_job job[kMovingJobs];
};

__global__ void bn_moving_update_multi_kernel(const MovingJobs jobs, float eps, float momentum, int unbiased) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (a runtime index into the by-value argument would copy it to scratch: read the one job from the kernel-argument
    // segment instead)
    gan_bn_moving_job j;
    const MovingJobs* ka = (const MovingJobs*)__builtin_amdgcn_kernarg_segment_ptr();
    __builtin_memcpy(&j, &ka->job[blockIdx.y], sizeof(j));
#else
    const gan_bn_moving_job j = jobs.job[0];
#endif
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= j.c) return;
    const float n = (float)j.rows;
    const float rs = j.save_rstd[ch];
    float var = fmaxf(1.0f / (rs * rs) - eps, 0.f);
    if (unbiased) var *= n / fmaxf(n - 1.f, 1.f);
    j.moving_mean[ch] = j.moving_mean[ch] * momentum + j.save_mean[ch] * (1.f - momentum);
    j.moving_var[ch] = j.moving_var[ch] * momentum + var * (1.f - momentum);
}

extern "C" int gan_bn_moving_update_multi(const gan_bn_moving_job* jobs, int count, float eps, float momentum,
                                          int unbiased_moving_var, gan_stream_t stream) {
    GAN_CHECK_ARG(jobs != nullptr && count > 0, "gan_bn_moving_update_multi: no jobs");
    for (int i = 0; i < count; ++i) {
        GAN_CHECK_ARG(jobs[i].rows > 0 && jobs[i].c > 0, "gan_bn_moving_update_multi: job %d: rows and c must be positive", i);
        GAN_CHECK_ARG(jobs[i].moving_mean && jobs[i].moving_var && jobs[i].save_mean && jobs[i].save_rstd,
                      "gan_bn_moving_update_multi: job %d: NULL tensor", i);
    }
    // the jobs of one launch run concurrently: two of them on the same averages would lose one update
    for (int first = 0; first < count; first += kMovingJobs) {
        const int nj = count - first < kMovingJobs ? count - first : kMovingJobs;
        for (int i = 1; i < nj; ++i)
            for (int j = 0; j < i; ++j)
                GAN_CHECK_ARG(jobs[first + i].moving_mean != jobs[first + j].moving_mean &&
                              jobs[first + i].moving_var != jobs[first + j].moving_var,
                              "gan_bn_moving_update_multi: jobs %d and %d update the same moving averages; a layer that moves "
                              "twice needs two calls", first + j, first + i);
    }
    for (int first = 0; first < count; first += kMovingJobs) {
        const int nj = count - first < kMovingJobs ? count - first : kMovingJobs;
        MovingJobs m;
        int cmax = 0;
        for (int i = 0; i < kMovingJobs; ++i) {
            m.job[i] = jobs[first + (i < nj ? i : 0)];
            if (i < nj && m.job[i].c > cmax) cmax = m.job[i].c;
        }
        hipLaunchKernelGGL(bn_moving_update_multi_kernel, dim3((cmax + 255) / 256, nj), dim3(256), 0, as_stream(stream), m, eps,
                           momentum, unbiased_moving_var);
    }
    return check_launch("gan_bn_moving_update_multi");
}

// ---------------------------------------------------------------------------- batch norm around a contraction's launches
extern "C" int gan_bn_tile_blocks(int groups, int rows, int c, int backward) {
    if (groups <= 0 || rows <= 0 || c <= 0 || groups > 65535) return 0;
    TileBn ts;
    if (knobs().no_bn_ride == 1) return 0;
    return bn_ride_plan(groups, rows, c, backward != 0, &ts) ? ts.nb : 0;
}

extern "C" int gan_bn_fwd_train_from_stats(const float* x, const float* gamma, const float* beta, float* moving_mean,
                                           float* moving_var, const float* residual, float* y, float* save_mean,
                                           float* save_rstd, int groups, int rows, int c, float eps, float momentum,
                                           int unbiased_moving_var, int act, float act_param, const float* bn_partial,
                                           gan_stream_t stream) {
    const char* what = "gan_bn_fwd_train_from_stats";
    if (int rc = bn_args_check(what, groups, rows, c, act)) return rc;
    GAN_CHECK_ARG(x && gamma && beta && y && save_mean && save_rstd && bn_partial, "%s: NULL tensor", what);
    GAN_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "%s: moving_mean and moving_var must be given together", what);
    TileBn ts;
    GAN_CHECK_ARG(bn_ride_plan(groups, rows, c, false, &ts), "%s: no block statistics exist for %d x %d x %d (gan_bn_tile_blocks "
                  "reports 0): use gan_bn_fwd_train", what, groups, rows, c);
    GAN_CHECK_ARG(all16({x, y, residual, save_mean, save_rstd, gamma, beta, moving_mean, moving_var, bn_partial}),
                  "%s: tensors must be 16-byte aligned", what);
    const dim3 grid(ts.nb, c / 64, groups);
    hipLaunchKernelGGL(bn_tile_apply_kernel, grid, dim3(256), 0, as_stream(stream), x, gamma, beta, moving_mean, moving_var, residual,
                       y, save_mean, save_rstd, reinterpret_cast<const float2*>(bn_partial), ts, eps, momentum, unbiased_moving_var,
                       act, act_param);
    return check_launch(what);
}

extern "C" int gan_bn_bwd_tiled(const float* x, const float* gamma, const float* beta, const float* save_mean,
                                const float* save_rstd, const float* dy, float* dx, float* dgamma, float* dbeta,
                                int accumulate, int groups, int rows, int c, int act, float act_param, float* sums_partial,
                                int sums_ready, float* colsum_partial, gan_stream_t stream) {
    const char* what = "gan_bn_bwd_tiled";
    if (int rc = bn_args_check(what, groups, rows, c, act)) return rc;
    GAN_CHECK_ARG(x && gamma && beta && save_mean && save_rstd && dy && sums_partial, "%s: NULL tensor", what);
    GAN_CHECK_ARG(colsum_partial == nullptr || dx != nullptr, "%s: column sums are those of dx", what);
    TileBn ts;
    GAN_CHECK_ARG(bn_ride_plan(groups, rows, c, true, &ts), "%s: no block decomposition exists for %d x %d x %d (gan_bn_tile_blocks "
                  "reports 0): use gan_bn_bwd", what, groups, rows, c);
    GAN_CHECK_ARG(all16({x, dy, dx, save_mean, save_rstd, gamma, beta, dgamma, dbeta, sums_partial, colsum_partial}),
                  "%s: tensors must be 16-byte aligned", what);
    hipStream_t st = as_stream(stream);
    const dim3 grid(ts.nb, c / 64, groups);
    float2* partial = reinterpret_cast<float2*>(sums_partial);
    if (!sums_ready)
        hipLaunchKernelGGL(bn_tile_bwd_sums_kernel<0>, grid, dim3(256), 0, st, x, dy, gamma, beta, save_mean, save_rstd, partial, ts,
                           act, act_param, tensor_rows());
    hipLaunchKernelGGL(bn_tile_bwd_apply_kernel, grid, dim3(256), 0, st, x, dy, gamma, beta, save_mean, save_rstd,
                       (const float2*)partial, dx, dgamma, dbeta, accumulate, ts, act, act_param, colsum_partial);
    return check_launch(what);
}

extern "C" int gan_bn_fwd_eval(const float* x, const float* gamma, const float* beta, const float* moving_mean,
                               const float* moving_var, float* y, int rows, int c, float eps, int act,
                               float act_param, gan_stream_t stream) {
    if (int rc = bn_args_check("gan_bn_fwd_eval", 1, rows, c, act)) return rc;
    GAN_CHECK_ARG(x && gamma && beta && moving_mean && moving_var && y, "gan_bn_fwd_eval: NULL tensor");
    const long total = (long)rows * c;
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(grid_for((c & 3) ? total : total / 4)), dim3(256), 0, as_stream(stream),
                       x, gamma, beta, moving_mean, moving_var, (const float*)nullptr, y, 1, rows, c, eps, act, act_param);
    return check_launch("gan_bn_fwd_eval");
}

extern "C" int gan_bn_bwd(const float* x, const float* gamma, const float* beta, const float* save_mean,
                          const float* save_rstd, const float* dy, float* dx, float* dgamma, float* dbeta,
                          int accumulate, int groups, int rows, int c, int act, float act_param, void* ws,
                          size_t ws_bytes, gan_stream_t stream) {
    if (int rc = bn_args_check("gan_bn_bwd", groups, rows, c, act)) return rc;
    GAN_CHECK_ARG(x && gamma && beta && save_mean && save_rstd && dy, "gan_bn_bwd: NULL tensor");
    const size_t need = bn_ws_bytes(groups, rows, c);
    GAN_CHECK_ARG(ws && ws_bytes >= need, "gan_bn_bwd: workspace of %zu bytes needed, %zu given", need, ws_bytes);
    const BnShape s = bn_shape(groups, rows, c);
    hipStream_t st = as_stream(stream);
    if (groups == 1 && fused_bn_ok(rows, c, x, dy, dx, save_mean) && fused_bn_ok(rows, c, save_rstd, gamma, beta, dgamma) &&
        fused_bn_ok(rows, c, dbeta, nullptr, nullptr, nullptr)) {
        GAN_FUSED_BN_DISPATCH(bn_bwd_fused_kernel, rows, dim3(c / 4), st, x, gamma, beta, save_mean, save_rstd, dy, dx, dgamma,
                              dbeta, accumulate, groups, rows, c, act, act_param);
        return check_launch("gan_bn_bwd");
    }
    float2* partial = static_cast<float2*>(ws);
    {
        TileBn ts;
        // (stacked passes of a small layer have no one-launch backward kernel: they come here too)
        if (tile_bn_plan(groups, rows, c, ts, groups > 1 ? 256 : kFusedMaxRows + 1) &&
            all16({x, dy, dx, save_mean, save_rstd, gamma, beta, dgamma, dbeta, ws})) {
            const dim3 grid(ts.nb, c / 64, groups);
            hipLaunchKernelGGL(bn_tile_bwd_sums_kernel<0>, grid, dim3(256), 0, st, x, dy, gamma, beta, save_mean, save_rstd, partial,
                               ts, act, act_param, tensor_rows());
            hipLaunchKernelGGL(bn_tile_bwd_apply_kernel, grid, dim3(256), 0, st, x, dy, gamma, beta, save_mean, save_rstd,
                               (const float2*)partial, dx, dgamma, dbeta, accumulate, ts, act, act_param, (float*)nullptr);
            return check_launch("gan_bn_bwd");
        }
    }
    float2* sums = partial + (size_t)groups * s.rblocks * c;
    if (s.vec4)
        hipLaunchKernelGGL(bn_bwd_partial_v4_kernel, dim3(s.rblocks, groups), dim3(256), 0, st, x, dy, gamma, beta, save_mean,
                           save_rstd, partial, s, act, act_param);
    else
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(s.rblocks, groups, (c + kChanTile - 1) / kChanTile), dim3(256), 0, st, x,
                           dy, gamma, beta, save_mean, save_rstd, partial, s, act, act_param);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + 3) / 4), dim3(256), 0, st, partial, sums, dgamma, dbeta, s,
                       accumulate);
    if (dx != nullptr) {
        const long total = (long)groups * rows * c;
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((c & 3) ? total : total / 4)), dim3(256), 0, st, x, dy, gamma,
                           beta, save_mean, save_rstd, sums, dx, groups, rows, c, act, act_param);
    }
    return check_launch("gan_bn_bwd");
}
