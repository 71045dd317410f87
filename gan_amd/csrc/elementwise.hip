// Bandwidth-bound elementwise / reduction kernels of the GAN hot path (gfx950):
// activations, residual add, channel concat, reflect pad, the three losses fused with
// their gradients, per-channel column sums (bias gradients) and the fused TF-style Adam step.
#include "common.h"

namespace gan {

static int grid_for(long work_items, int cap = 2048) {
    long b = (work_items + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------ activations
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act, float param,
                               float scale, float shift) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 o;
        o.x = apply_act(v.x, act, param) * scale + shift;
        o.y = apply_act(v.y, act, param) * scale + shift;
        o.z = apply_act(v.z, act, param) * scale + shift;
        o.w = apply_act(v.w, act, param) * scale + shift;
        reinterpret_cast<float4*>(y)[i] = o;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = apply_act(x[i], act, param) * scale + shift;
}

__global__ void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                               long n, int act, float param, float scale, float shift) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        const float4 g = reinterpret_cast<const float4*>(dy)[i];
        float4 o;
        o.x = act_bwd_one(v.x, g.x, act, param, scale, shift);
        o.y = act_bwd_one(v.y, g.y, act, param, scale, shift);
        o.z = act_bwd_one(v.z, g.z, act, param, scale, shift);
        o.w = act_bwd_one(v.w, g.w, act, param, scale, shift);
        reinterpret_cast<float4*>(dx)[i] = o;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dx[i] = act_bwd_one(y[i], dy[i], act, param, scale, shift);
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 u = reinterpret_cast<const float4*>(a)[i];
        const float4 v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(y)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = a[i] + b[i];
}

__global__ void fill_kernel(float* __restrict__ x, long n, float v) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = v;
}

// ------------------------------------------------------------------ concat
__global__ void concat_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                  long rows, int ca, int cb, int rows_per_b) {
    const int ct = ca + cb;
    const long total = rows * ct;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long r = i / ct;
        const int ch = (int)(i - r * ct);
        out[i] = ch < ca ? a[r * ca + ch] : b[(r / rows_per_b) * cb + (ch - ca)];
    }
}

__global__ void concat_bwd_a_kernel(const float* __restrict__ dout, float* __restrict__ da, long rows, int ca, int cb) {
    const int ct = ca + cb;
    const long total = rows * ca;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long r = i / ca;
        const int ch = (int)(i - r * ca);
        da[i] = dout[r * ct + ch];
    }
}

// db[g][ch] = sum over the group's rows of dout[row][ca+ch]; one block per group, fixed order.
__global__ __launch_bounds__(256) void concat_bwd_b_kernel(const float* __restrict__ dout, float* __restrict__ db,
                                                           int ca, int cb, int rows_per_b) {
    __shared__ float red[256];
    const int ct = ca + cb;
    const long g = blockIdx.x;
    for (int ch0 = 0; ch0 < cb; ch0 += 64) {
        const int ch = ch0 + (threadIdx.x & 63);
        float s = 0.f;
        if (ch < cb)
            for (int r = threadIdx.x >> 6; r < rows_per_b; r += 4) s += dout[(g * rows_per_b + r) * ct + ca + ch];
        red[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < 64 && ch < cb)
            db[g * cb + ch] = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
        __syncthreads();
    }
}

// ------------------------------------------------------------------ reflect pad
__device__ __forceinline__ int reflect_index(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

__global__ void reflect_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w, int c,
                                       int p) {
    const int hp = h + 2 * p, wp = w + 2 * p;
    const long total = (long)n * hp * wp * c;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)(i % c);
        long r = i / c;
        const int xx = (int)(r % wp); r /= wp;
        const int yy = (int)(r % hp);
        const int img = (int)(r / hp);
        const int sy = reflect_index(yy - p, h), sx = reflect_index(xx - p, w);
        y[i] = x[(((long)img * h + sy) * w + sx) * c + ch];
    }
}

// Gather form of the adjoint: every source pixel sums the (at most 3 x 3) padded positions that
// mirror onto it, in a fixed order.
__global__ void reflect_pad_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int n, int h, int w, int c,
                                       int p) {
    const int hp = h + 2 * p, wp = w + 2 * p;
    const long total = (long)n * h * w * c;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)(i % c);
        long r = i / c;
        const int sx = (int)(r % w); r /= w;
        const int sy = (int)(r % h);
        const int img = (int)(r / h);
        // padded rows that map to sy: sy+p always; p-sy if 1 <= sy <= p; 2(h-1)-sy+p if h-1-p <= sy <= h-2
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = sy + p;
        if (sy >= 1 && sy <= p) ys[ny++] = p - sy;
        if (sy <= h - 2 && sy >= h - 1 - p) ys[ny++] = 2 * (h - 1) - sy + p;
        xs[nx++] = sx + p;
        if (sx >= 1 && sx <= p) xs[nx++] = p - sx;
        if (sx <= w - 2 && sx >= w - 1 - p) xs[nx++] = 2 * (w - 1) - sx + p;
        float s = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) s += dy[(((long)img * hp + ys[a]) * wp + xs[b]) * c + ch];
        dx[i] = s;
    }
}

// ------------------------------------------------------------------ losses (single block: n is small)
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;  // valid in thread 0
}

// kind 0: sigmoid BCE(x, label=t)   grad -> g1 = weight*(sigmoid(x)-t)/n
// kind 1: (x-t)^2                   grad -> g1 = weight*2(x-t)/n
// kind 2: |a-b| (a = x, b = x2)     grad w.r.t. b -> g1 = weight*(-sign(a-b))/n
__device__ __forceinline__ void loss_term(int kind, float v, float v2, float t, float& per, float& g) {
    if (kind == 0) {
        per = fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
        const float sg = v >= 0.f ? 1.f / (1.f + expf(-v)) : expf(v) / (1.f + expf(v));
        g = (sg - t);
    } else if (kind == 1) {
        const float d = v - t;
        per = d * d;
        g = 2.f * d;
    } else {
        const float d = v - v2;
        per = fabsf(d);
        g = d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f);
    }
}

// One block: the mean needs every element and the entry points take no scratch.  `vec` (n % 4 == 0 and
// 16-byte aligned tensors) moves 16 bytes per lane, which matters for the image-sized L1 terms.
__global__ __launch_bounds__(1024) void loss_kernel(int kind, const float* __restrict__ x, const float* __restrict__ x2,
                                                    long n, float t, float weight, int accumulate, int vec,
                                                    float* __restrict__ loss_out, float* __restrict__ g1) {
    __shared__ float red[16];
    const float inv_n = 1.0f / (float)n;
    const float gs = weight * inv_n;
    float s = 0.f;
    if (vec) {
        const long n4 = n >> 2;
#pragma unroll 4
        for (long i = threadIdx.x; i < n4; i += blockDim.x) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kind == 2) w = reinterpret_cast<const float4*>(x2)[i];
            float p0, p1, p2, p3;
            float4 g;
            loss_term(kind, v.x, w.x, t, p0, g.x);
            loss_term(kind, v.y, w.y, t, p1, g.y);
            loss_term(kind, v.z, w.z, t, p2, g.z);
            loss_term(kind, v.w, w.w, t, p3, g.w);
            s += (p0 + p1) + (p2 + p3);
            if (g1) reinterpret_cast<float4*>(g1)[i] = make_float4(gs * g.x, gs * g.y, gs * g.z, gs * g.w);
        }
    } else {
        for (long i = threadIdx.x; i < n; i += blockDim.x) {
            float per, g;
            loss_term(kind, x[i], kind == 2 ? x2[i] : 0.f, t, per, g);
            s += per;
            if (g1) g1[i] = gs * g;
        }
    }
    const float tot = block_sum_1024(s, red);
    if (threadIdx.x == 0) loss_out[0] = (accumulate ? loss_out[0] : 0.f) + weight * tot * inv_n;
}

// Image-sized terms (CycleGAN's cycle loss over 256 x 256 x 3 values): one block is one CU's worth of memory
// bandwidth, so with a workspace from the caller the elementwise work is spread over up to 256 blocks, each leaves
// the sum of its contiguous range in the workspace, and a second launch adds the block sums in block order.
constexpr int kLossMaxBlocks = 256;

__global__ __launch_bounds__(256) void loss_partial_kernel(int kind, const float* __restrict__ x, const float* __restrict__ x2,
                                                           long n, long per_block, float t, float weight, int vec,
                                                           float* __restrict__ partial, float* __restrict__ g1) {
    __shared__ float red[4];
    const float gs = weight * (1.0f / (float)n);
    const long e0 = (long)blockIdx.x * per_block;            // per_block is a multiple of 4
    const long e1 = e0 + per_block < n ? e0 + per_block : n;
    float s = 0.f;
    if (vec) {
#pragma unroll 4
        for (long i = (e0 >> 2) + threadIdx.x; i < (e1 >> 2); i += 256) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kind == 2) w = reinterpret_cast<const float4*>(x2)[i];
            float p0, p1, p2, p3;
            float4 g;
            loss_term(kind, v.x, w.x, t, p0, g.x);
            loss_term(kind, v.y, w.y, t, p1, g.y);
            loss_term(kind, v.z, w.z, t, p2, g.z);
            loss_term(kind, v.w, w.w, t, p3, g.w);
            s += (p0 + p1) + (p2 + p3);
            if (g1) reinterpret_cast<float4*>(g1)[i] = make_float4(gs * g.x, gs * g.y, gs * g.z, gs * g.w);
        }
    } else {
#pragma unroll 4
        for (long i = e0 + threadIdx.x; i < e1; i += 256) {
            float per, g;
            loss_term(kind, x[i], kind == 2 ? x2[i] : 0.f, t, per, g);
            s += per;
            if (g1) g1[i] = gs * g;
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void loss_final_kernel(const float* __restrict__ partial, int blocks, long n, float weight,
                                                        int accumulate, float* __restrict__ loss_out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < blocks; i += 64) s += partial[i];   // at most 4 per lane
    s = wave_sum(s);
    if (threadIdx.x == 0) loss_out[0] = (accumulate ? loss_out[0] : 0.f) + weight * s * (1.0f / (float)n);
}

// ------------------------------------------------------------------ column sums (bias gradient)
// partial[rb][c] per row block, then one wave per channel sums the partials (fixed shuffle tree).
// float4 path (c % 4 == 0 and c/4 divides 256): thread = (row lane, 4-channel group).
__global__ __launch_bounds__(256) void colsum_partial_v4_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                                long rows, int c, long rows_per_block) {
    __shared__ float4 red[256];
    const int c4 = c >> 2;
    const int cg = threadIdx.x % c4, rl = threadIdx.x / c4, RL = 256 / c4;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const float4* xp = reinterpret_cast<const float4*>(x) + cg;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (long r = r0 + rl; r < r1; r += RL) {
        const float4 v = xp[r * c4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < RL; ++j) {
            const float4 v = red[cg + j * c4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(partial + (long)blockIdx.x * c)[cg] = s;
    }
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                             long rows, int c, long rows_per_block) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ch = blockIdx.y * 64 + tx;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s = 0.f;
    if (ch < c) {
#pragma unroll 8
        for (long r = r0 + ty; r < r1; r += 4) s += x[r * c + ch];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && ch < c) partial[(long)blockIdx.x * c + ch] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// c <= 4 (the bias gradient of an image-producing layer): a thread takes whole rows, so every lane works
// and a wave reads a contiguous run of rows.
template <int C>
__global__ __launch_bounds__(256) void colsum_partial_thin_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                                  long rows, long rows_per_block) {
    __shared__ float red[C][256];
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s[C];
#pragma unroll
    for (int k = 0; k < C; ++k) s[k] = 0.f;
#pragma unroll 4
    for (long r = r0 + threadIdx.x; r < r1; r += 256) {
#pragma unroll
        for (int k = 0; k < C; ++k) s[k] += x[r * C + k];
    }
#pragma unroll
    for (int k = 0; k < C; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int j = 0; j < 256; ++j) t += red[threadIdx.x][j];
        partial[(long)blockIdx.x * C + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const ColsumFinal job) { colsum_final_block(job, blockIdx.x); }

static bool colsum_vec4(int c) { return (c % 4 == 0) && (c / 4 <= 256) && (256 % (c / 4) == 0); }

static void colsum_plan(long rows, int c, int* rblocks, long* rows_per_block) {
    long rb;
    if (colsum_vec4(c)) {
        const int row_lanes = 256 / (c / 4);
        rb = (rows + row_lanes * 4 - 1) / (row_lanes * 4);
        if (rb > 1024) rb = 1024;
    } else if (c <= 3) {
        rb = (rows + 1023) / 1024;      // four rows per thread
        if (rb > 1024) rb = 1024;
    } else {
        const int ctiles = (c + 63) / 64;
        long target = 1024 / ctiles;
        if (target < 1) target = 1;
        rb = (rows + 63) / 64;
        if (rb > target) rb = target;
    }
    if (rb < 1) rb = 1;
    *rows_per_block = (rows + rb - 1) / rb;
    *rblocks = (int)((rows + *rows_per_block - 1) / *rows_per_block);
}

size_t colsum_ws_bytes(long rows, int c) {
    int rb;
    long rpb;
    colsum_plan(rows, c, &rb, &rpb);
    return (size_t)rb * c * sizeof(float);
}

// First stage: partial[rb][c] into ws.  Fills `job` with what the last stage needs.
int colsum_partial(const float* x, long rows, int c, void* ws, size_t ws_bytes, hipStream_t st, const char* what,
                   ColsumFinal* job) {
    int rb;
    long rpb;
    colsum_plan(rows, c, &rb, &rpb);
    const size_t need = (size_t)rb * c * sizeof(float);
    GAN_CHECK_ARG(ws && ws_bytes >= need, "%s: workspace of %zu bytes needed for the bias gradient, %zu given", what, need,
                  ws_bytes);
    if (colsum_vec4(c))
        hipLaunchKernelGGL(colsum_partial_v4_kernel, dim3(rb), dim3(256), 0, st, x, static_cast<float*>(ws), rows, c, rpb);
    else if (c == 3)
        hipLaunchKernelGGL(colsum_partial_thin_kernel<3>, dim3(rb), dim3(256), 0, st, x, static_cast<float*>(ws), rows, rpb);
    else if (c == 2)
        hipLaunchKernelGGL(colsum_partial_thin_kernel<2>, dim3(rb), dim3(256), 0, st, x, static_cast<float*>(ws), rows, rpb);
    else if (c == 1)
        hipLaunchKernelGGL(colsum_partial_thin_kernel<1>, dim3(rb), dim3(256), 0, st, x, static_cast<float*>(ws), rows, rpb);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(rb, (c + 63) / 64), dim3(256), 0, st, x, static_cast<float*>(ws), rows,
                           c, rpb);
    job->partial = static_cast<const float*>(ws);
    job->c = c;
    job->rblocks = rb;
    job->blocks = (c + 3) / 4;
    return check_launch(what);
}

// Last stage as a launch of its own.
int colsum_final(const ColsumFinal& job, hipStream_t st, const char* what) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(job.blocks), dim3(256), 0, st, job);
    return check_launch(what);
}

int colsum(const float* x, float* out, long rows, int c, int accumulate, void* ws, size_t ws_bytes, hipStream_t st,
           const char* what) {
    ColsumFinal job;
    if (int rc = colsum_partial(x, rows, c, ws, ws_bytes, st, what, &job)) return rc;
    job.out = out;
    job.accumulate = accumulate;
    return colsum_final(job, st, what);
}

// ------------------------------------------------------------------ Adam (TensorFlow 1.x form)
__global__ void adam_prepare_kernel(int* step, float* lr_out, float lr, float beta1, float beta2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int t = step[0] + 1;
        step[0] = t;
        const double b1t = pow((double)beta1, (double)t), b2t = pow((double)beta2, (double)t);
        lr_out[0] = (float)((double)lr * sqrt(1.0 - b2t) / (1.0 - b1t));
    }
}

__global__ void adam_apply_kernel(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                                  float* __restrict__ v, long n, const float* __restrict__ lr_dev, float beta1,
                                  float beta2, float eps, float grad_scale) {
    const float lr_t = lr_dev[0];
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = reinterpret_cast<const float4*>(grad)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float4 th = reinterpret_cast<float4*>(theta)[i];
        g.x *= grad_scale; g.y *= grad_scale; g.z *= grad_scale; g.w *= grad_scale;
        mm.x = beta1 * mm.x + omb1 * g.x; mm.y = beta1 * mm.y + omb1 * g.y;
        mm.z = beta1 * mm.z + omb1 * g.z; mm.w = beta1 * mm.w + omb1 * g.w;
        vv.x = beta2 * vv.x + omb2 * g.x * g.x; vv.y = beta2 * vv.y + omb2 * g.y * g.y;
        vv.z = beta2 * vv.z + omb2 * g.z * g.z; vv.w = beta2 * vv.w + omb2 * g.w * g.w;
        th.x -= lr_t * mm.x / (sqrtf(vv.x) + eps); th.y -= lr_t * mm.y / (sqrtf(vv.y) + eps);
        th.z -= lr_t * mm.z / (sqrtf(vv.z) + eps); th.w -= lr_t * mm.w / (sqrtf(vv.w) + eps);
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        reinterpret_cast<float4*>(theta)[i] = th;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float g = grad[i] * grad_scale;
        const float mm = beta1 * m[i] + omb1 * g;
        const float vv = beta2 * v[i] + omb2 * g * g;
        m[i] = mm;
        v[i] = vv;
        theta[i] -= lr_t * mm / (sqrtf(vv) + eps);
    }
}

}  // namespace gan

using namespace gan;

static int act_check(const char* what, int act) {
    GAN_CHECK_ARG(act >= GAN_ACT_NONE && act <= GAN_ACT_SIGMOID, "%s: unknown activation %d", what, act);
    return 0;
}

extern "C" int gan_act_fwd(const float* x, float* y, long n, int act, float act_param, float scale, float shift,
                           gan_stream_t stream) {
    if (int rc = act_check("gan_act_fwd", act)) return rc;
    GAN_CHECK_ARG(x && y && n > 0, "gan_act_fwd: NULL tensor or n <= 0");
    GAN_CHECK_ARG((scale == 1.f && shift == 0.f) || (act != GAN_ACT_RELU && act != GAN_ACT_LRELU),
                  "gan_act_fwd: scale/shift are not supported with relu / lrelu (the backward could not recover the sign)");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, as_stream(stream), x, y, n, act, act_param,
                       scale, shift);
    return check_launch("gan_act_fwd");
}

extern "C" int gan_act_bwd(const float* y, const float* dy, float* dx, long n, int act, float act_param, float scale,
                           float shift, gan_stream_t stream) {
    if (int rc = act_check("gan_act_bwd", act)) return rc;
    GAN_CHECK_ARG(y && dy && dx && n > 0, "gan_act_bwd: NULL tensor or n <= 0");
    GAN_CHECK_ARG(scale != 0.f, "gan_act_bwd: scale must be non-zero");
    GAN_CHECK_ARG((scale == 1.f && shift == 0.f) || (act != GAN_ACT_RELU && act != GAN_ACT_LRELU),
                  "gan_act_bwd: scale/shift are not supported with relu / lrelu");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, as_stream(stream), y, dy, dx, n, act,
                       act_param, scale, shift);
    return check_launch("gan_act_bwd");
}

extern "C" int gan_add_fwd(const float* a, const float* b, float* y, long n, gan_stream_t stream) {
    GAN_CHECK_ARG(a && b && y && n > 0, "gan_add_fwd: NULL tensor or n <= 0");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, as_stream(stream), a, b, y, n);
    return check_launch("gan_add_fwd");
}

extern "C" int gan_fill(float* x, long n, float value, gan_stream_t stream) {
    GAN_CHECK_ARG(x && n > 0, "gan_fill: NULL tensor or n <= 0");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, n, value);
    return check_launch("gan_fill");
}

extern "C" int gan_concat_fwd(const float* a, const float* b, float* out, long rows, int ca, int cb, int rows_per_b,
                              gan_stream_t stream) {
    GAN_CHECK_ARG(a && b && out, "gan_concat_fwd: NULL tensor");
    GAN_CHECK_ARG(rows > 0 && ca > 0 && cb > 0 && rows_per_b > 0 && rows % rows_per_b == 0,
                  "gan_concat_fwd: rows must be a positive multiple of rows_per_b, channel counts positive");
    hipLaunchKernelGGL(concat_fwd_kernel, dim3(grid_for(rows * (ca + cb))), dim3(256), 0, as_stream(stream), a, b, out, rows,
                       ca, cb, rows_per_b);
    return check_launch("gan_concat_fwd");
}

extern "C" int gan_concat_bwd(const float* dout, float* da, float* db, long rows, int ca, int cb, int rows_per_b,
                              gan_stream_t stream) {
    GAN_CHECK_ARG(dout, "gan_concat_bwd: NULL tensor");
    GAN_CHECK_ARG(rows > 0 && ca > 0 && cb > 0 && rows_per_b > 0 && rows % rows_per_b == 0,
                  "gan_concat_bwd: rows must be a positive multiple of rows_per_b, channel counts positive");
    hipStream_t st = as_stream(stream);
    if (da) hipLaunchKernelGGL(concat_bwd_a_kernel, dim3(grid_for(rows * ca)), dim3(256), 0, st, dout, da, rows, ca, cb);
    if (db) hipLaunchKernelGGL(concat_bwd_b_kernel, dim3((unsigned)(rows / rows_per_b)), dim3(256), 0, st, dout, db, ca, cb,
                               rows_per_b);
    return check_launch("gan_concat_bwd");
}

extern "C" int gan_reflect_pad_fwd(const float* x, float* y, int n, int h, int w, int c, int p, gan_stream_t stream) {
    GAN_CHECK_ARG(x && y, "gan_reflect_pad_fwd: NULL tensor");
    GAN_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && p >= 0 && p < h && p < w,
                  "gan_reflect_pad_fwd: pad must be smaller than the image (REFLECT), sizes positive");
    const long total = (long)n * (h + 2 * p) * (w + 2 * p) * c;
    hipLaunchKernelGGL(reflect_pad_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, y, n, h, w, c, p);
    return check_launch("gan_reflect_pad_fwd");
}

extern "C" int gan_reflect_pad_bwd(const float* dy, float* dx, int n, int h, int w, int c, int p, gan_stream_t stream) {
    GAN_CHECK_ARG(dy && dx, "gan_reflect_pad_bwd: NULL tensor");
    GAN_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && p >= 0 && p < h && p < w,
                  "gan_reflect_pad_bwd: pad must be smaller than the image (REFLECT), sizes positive");
    hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3(grid_for((long)n * h * w * c)), dim3(256), 0, as_stream(stream), dy, dx,
                       n, h, w, c, p);
    return check_launch("gan_reflect_pad_bwd");
}

static int launch_loss(const char* what, int kind, const float* x, const float* x2, long n, float t, float weight,
                       int accumulate, float* loss_out, float* g1, gan_stream_t stream) {
    GAN_CHECK_ARG(x && loss_out && n > 0, "%s: NULL tensor or n <= 0", what);
    const auto al = [](const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; };
    const int vec = (n >= 4096 && (n & 3) == 0 && al(x) && al(x2) && al(g1)) ? 1 : 0;
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), kind, x, x2, n, t, weight, accumulate, vec,
                       loss_out, g1);
    return check_launch(what);
}

extern "C" int gan_sigmoid_bce_mean(const float* logits, long n, float label, float weight, int accumulate,
                                    float* loss_out, float* dlogits, gan_stream_t stream) {
    return launch_loss("gan_sigmoid_bce_mean", 0, logits, nullptr, n, label, weight, accumulate, loss_out, dlogits, stream);
}

extern "C" int gan_mse_const_mean(const float* x, long n, float target, float weight, int accumulate, float* loss_out,
                                  float* dx, gan_stream_t stream) {
    return launch_loss("gan_mse_const_mean", 1, x, nullptr, n, target, weight, accumulate, loss_out, dx, stream);
}

extern "C" int gan_l1_mean(const float* a, const float* b, long n, float weight, int accumulate, float* loss_out,
                           float* db, gan_stream_t stream) {
    GAN_CHECK_ARG(b, "gan_l1_mean: NULL tensor");
    return launch_loss("gan_l1_mean", 2, a, b, n, 0.f, weight, accumulate, loss_out, db, stream);
}

extern "C" size_t gan_l1_mean_workspace_bytes(long n) { return n > 0 ? kLossMaxBlocks * sizeof(float) : 0; }

extern "C" int gan_l1_mean_ws(const float* a, const float* b, long n, float weight, int accumulate, float* loss_out,
                              float* db, void* ws, size_t ws_bytes, gan_stream_t stream) {
    GAN_CHECK_ARG(a && b && loss_out && n > 0, "gan_l1_mean_ws: NULL tensor or n <= 0");
    GAN_CHECK_ARG(ws != nullptr && ws_bytes >= gan_l1_mean_workspace_bytes(n), "gan_l1_mean_ws: workspace of %zu bytes needed, %zu given",
                  gan_l1_mean_workspace_bytes(n), ws_bytes);
    GAN_CHECK_ARG((reinterpret_cast<size_t>(ws) & 3) == 0, "gan_l1_mean_ws: workspace must be 4-byte aligned");
    const auto al = [](const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; };
    const int vec = ((n & 3) == 0 && al(a) && al(b) && al(db)) ? 1 : 0;
    // contiguous ranges of at least 4096 values, a multiple of 4 each, at most kLossMaxBlocks of them
    long per_block = (n + kLossMaxBlocks - 1) / kLossMaxBlocks;
    if (per_block < 4096) per_block = 4096;
    per_block = (per_block + 3) & ~3L;
    const int blocks = (int)((n + per_block - 1) / per_block);
    float* partial = static_cast<float*>(ws);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(loss_partial_kernel, dim3(blocks), dim3(256), 0, st, 2, a, b, n, per_block, 0.f, weight, vec, partial, db);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, partial, blocks, n, weight, accumulate, loss_out);
    return check_launch("gan_l1_mean_ws");
}

extern "C" int gan_adam_tf_step(float* theta, const float* grad, float* m, float* v, long n, float lr, float beta1,
                                float beta2, float eps, float grad_scale, int* step_dev, float* lr_scratch,
                                gan_stream_t stream) {
    GAN_CHECK_ARG(theta && grad && m && v && step_dev && lr_scratch && n > 0, "gan_adam_tf_step: NULL tensor or n <= 0");
    GAN_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "gan_adam_tf_step: betas must be in [0,1)");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, st, step_dev, lr_scratch, lr, beta1, beta2);
    hipLaunchKernelGGL(adam_apply_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, theta, grad, m, v, n,
                       (const float*)lr_scratch, beta1, beta2, eps, grad_scale);
    return check_launch("gan_adam_tf_step");
}
