// Batch / instance normalisation fused with the activation that follows it (gfx950).
//
// Every kernel here is bandwidth bound.  x is viewed as [groups][rows][c] with c contiguous;
// one wave reads 64 consecutive channels of one row per load (256 B coalesced), four waves of a
// block walk four rows at a time, and partial statistics are combined through LDS, then across
// blocks by a small finalize kernel in a fixed order (deterministic, no atomics).
//
// Training forward: (1) per-block two-pass mean / sum of squared deviations over a row range
// (the second pass re-reads a range that is still L1/L2 resident), (2) Chan-merge of the block
// partials, moving-average update, (3) normalise + affine + activation (+ residual).
// Backward: (1) per-block sums of dz and dz*xhat, (2) merge, dgamma/dbeta, (3) dx.
#include "common.h"

namespace gan {

constexpr int kChanTile = 64;  // channels per block (one wave-width)
constexpr int kRowLanes = 4;   // waves per block, each on its own row

struct BnShape {
    int groups, rows, c;
    int rblocks, rows_per_block;
};

static BnShape bn_shape(int groups, int rows, int c) {
    BnShape s;
    s.groups = groups; s.rows = rows; s.c = c;
    const int ctiles = (c + kChanTile - 1) / kChanTile;
    long target = 2048 / ((long)ctiles * groups);
    if (target < 1) target = 1;
    int rb = (rows + 31) / 32;  // at least 32 rows per block
    if (rb > target) rb = (int)target;
    if (rb < 1) rb = 1;
    s.rows_per_block = (rows + rb - 1) / rb;
    s.rblocks = (rows + s.rows_per_block - 1) / s.rows_per_block;
    return s;
}

// partial[(g*rblocks + rb)*c + ch] = {mean, m2} over the block's rows
__global__ __launch_bounds__(256) void bn_partial_stats_kernel(const float* __restrict__ x, float2* __restrict__ partial,
                                                               BnShape s) {
    __shared__ float red[kRowLanes][kChanTile];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ch = blockIdx.x * kChanTile + tx;
    const int rb = blockIdx.y, grp = blockIdx.z;
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

// One thread per (group, channel): merge block partials, emit mean / rstd, update moving stats.
__global__ void bn_finalize_stats_kernel(const float2* __restrict__ partial, float* __restrict__ save_mean,
                                         float* __restrict__ save_rstd, float* __restrict__ moving_mean,
                                         float* __restrict__ moving_var, BnShape s, float eps, float momentum,
                                         int unbiased) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= s.groups * s.c) return;
    const int grp = idx / s.c, ch = idx - grp * s.c;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int rb = 0; rb < s.rblocks; ++rb) {
        const float2 pr = partial[((long)grp * s.rblocks + rb) * s.c + ch];
        const int r0 = rb * s.rows_per_block;
        const float nb = (float)(min(s.rows, r0 + s.rows_per_block) - r0);
        const float delta = pr.x - mean;
        const float nn = n + nb;
        mean += delta * (nb / nn);
        m2 += pr.y + delta * delta * (n * nb / nn);
        n = nn;
    }
    const float var = m2 / n;
    save_mean[idx] = mean;
    save_rstd[idx] = 1.0f / sqrtf(var + eps);
    if (moving_mean != nullptr) {
        const float v = unbiased ? var * (n / fmaxf(n - 1.f, 1.f)) : var;
        moving_mean[ch] = moving_mean[ch] * momentum + mean * (1.f - momentum);
        moving_var[ch] = moving_var[ch] * momentum + v * (1.f - momentum);
    }
}

// y = act((x-mean)*rstd*gamma+beta) + residual.  mean/rstd are per (group, channel); for the
// inference path the caller passes groups = 1 with mean = moving_mean and var in place of rstd.
template <bool EVAL>
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ mean,
                                const float* __restrict__ rstd_or_var, const float* __restrict__ residual,
                                float* __restrict__ y, int groups, int rows, int c, float eps, int act, float act_param) {
    const long total = (long)groups * rows * c;
    const long stride = (long)gridDim.x * blockDim.x;
    const long per_group = (long)rows * c;
    if ((c & 3) == 0) {
        for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += stride * 4) {
            const int ch = (int)(i % c);
            const int grp = (int)(i / per_group);
            const float4 xv = *reinterpret_cast<const float4*>(x + i);
            const float4 mu = *reinterpret_cast<const float4*>(mean + grp * c + ch);
            float4 rs = *reinterpret_cast<const float4*>(rstd_or_var + grp * c + ch);
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
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int ch = (int)(i % c);
            const int grp = (int)(i / per_group);
            float rs = rstd_or_var[grp * c + ch];
            if (EVAL) rs = 1.0f / sqrtf(rs + eps);
            float o = apply_act((x[i] - mean[grp * c + ch]) * rs * gamma[ch] + beta[ch], act, act_param);
            if (residual) o += residual[i];
            y[i] = o;
        }
    }
}

// partial[(g*rblocks+rb)*c + ch] = {sum dz, sum dz*xhat},  dz = dy * act'(z)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float2* __restrict__ partial, BnShape s, int act, float act_param) {
    __shared__ float red0[kRowLanes][kChanTile];
    __shared__ float red1[kRowLanes][kChanTile];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ch = blockIdx.x * kChanTile + tx;
    const int rb = blockIdx.y, grp = blockIdx.z;
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
            const float xhat = (x[i] - mu) * rs;
            float dz = dy[i];
            if (act != GAN_ACT_NONE) {
                const float z = xhat * ga + be;
                dz *= act_grad(z, apply_act(z, act, act_param), act, act_param);
            }
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

// sums[(g*c+ch)] = {sum dz, sum dz*xhat} per group; dgamma/dbeta = totals over groups.
__global__ void bn_bwd_finalize_kernel(const float2* __restrict__ partial, float2* __restrict__ sums,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, BnShape s,
                                       int accumulate) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= s.c) return;
    float tg = 0.f, tb = 0.f;
    for (int grp = 0; grp < s.groups; ++grp) {
        float a0 = 0.f, a1 = 0.f;
        for (int rb = 0; rb < s.rblocks; ++rb) {
            const float2 pr = partial[((long)grp * s.rblocks + rb) * s.c + ch];
            a0 += pr.x;
            a1 += pr.y;
        }
        sums[grp * s.c + ch] = make_float2(a0, a1);
        tb += a0;
        tg += a1;
    }
    if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + tg;
    if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + tb;
}

// dx = gamma*rstd*(dz - sum_dz/rows - xhat*sum_dz_xhat/rows)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float2* __restrict__ sums, float* __restrict__ dx, int groups, int rows,
                                    int c, int act, float act_param) {
    const long total = (long)groups * rows * c;
    const long stride = (long)gridDim.x * blockDim.x;
    const long per_group = (long)rows * c;
    const float inv_rows = 1.0f / (float)rows;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)(i % c);
        const int gi = (int)(i / per_group) * c + ch;
        const float rs = rstd[gi], ga = gamma[ch];
        const float xhat = (x[i] - mean[gi]) * rs;
        float dz = dy[i];
        if (act != GAN_ACT_NONE) {
            const float z = xhat * ga + beta[ch];
            dz *= act_grad(z, apply_act(z, act, act_param), act, act_param);
        }
        const float2 sm = sums[gi];
        dx[i] = ga * rs * (dz - sm.x * inv_rows - xhat * (sm.y * inv_rows));
    }
}

static int grid_for(long work_items) {
    long b = (work_items + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

size_t bn_ws_bytes(int groups, int rows, int c) {
    const BnShape s = bn_shape(groups, rows, c);
    return ((size_t)groups * s.rblocks * c + (size_t)groups * c) * sizeof(float2);
}

}  // namespace gan

using namespace gan;

extern "C" size_t gan_bn_workspace_bytes(int groups, int rows, int c) {
    if (groups <= 0 || rows <= 0 || c <= 0) return 0;
    return bn_ws_bytes(groups, rows, c);
}

static int bn_args_check(const char* what, int groups, int rows, int c, int act) {
    GAN_CHECK_ARG(groups > 0 && rows > 0 && c > 0, "%s: groups, rows and c must be positive", what);
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
    GAN_CHECK_ARG(moving_mean == nullptr || groups == 1, "gan_bn_fwd_train: moving statistics require groups == 1");
    const size_t need = bn_ws_bytes(groups, rows, c);
    GAN_CHECK_ARG(ws && ws_bytes >= need, "gan_bn_fwd_train: workspace of %zu bytes needed, %zu given", need, ws_bytes);
    const BnShape s = bn_shape(groups, rows, c);
    hipStream_t st = as_stream(stream);
    float2* partial = static_cast<float2*>(ws);
    dim3 grid((c + kChanTile - 1) / kChanTile, s.rblocks, groups);
    hipLaunchKernelGGL(bn_partial_stats_kernel, grid, dim3(256), 0, st, x, partial, s);
    hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3((groups * c + 255) / 256), dim3(256), 0, st, partial, save_mean,
                       save_rstd, moving_mean, moving_var, s, eps, momentum, unbiased_moving_var);
    const long total = (long)groups * rows * c;
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid_for((c & 3) ? total : total / 4)), dim3(256), 0, st, x, gamma,
                       beta, save_mean, save_rstd, residual, y, groups, rows, c, eps, act, act_param);
    return check_launch("gan_bn_fwd_train");
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
    float2* partial = static_cast<float2*>(ws);
    float2* sums = partial + (size_t)groups * s.rblocks * c;
    dim3 grid((c + kChanTile - 1) / kChanTile, s.rblocks, groups);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, grid, dim3(256), 0, st, x, dy, gamma, beta, save_mean, save_rstd, partial, s,
                       act, act_param);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, st, partial, sums, dgamma, dbeta, s,
                       accumulate);
    if (dx != nullptr) {
        const long total = (long)groups * rows * c;
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, dy, gamma, beta, save_mean,
                           save_rstd, sums, dx, groups, rows, c, act, act_param);
    }
    return check_launch("gan_bn_bwd");
}
