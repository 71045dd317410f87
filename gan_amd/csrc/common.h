// Shared host/device helpers for libgan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/gan_hip.h"

namespace gan {

// Thread-local error text behind gan_last_error().
char* error_buffer();
int fail(const char* fmt, ...);

#define GAN_CHECK_ARG(cond, ...) \
    do {                         \
        if (!(cond)) return ::gan::fail(__VA_ARGS__); \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

// Test and tuning knobs.  Their initial values come from the environment (GAN_HIP_*), read ONCE at first
// use, so that a launch costs no getenv and a plan cannot change between a workspace query and the run it
// sizes; gan_knob_set changes them afterwards (tests, tuners).
struct Knobs {
    int force_cfg;      // GAN_HIP_FORCE_CFG: tile configuration of every contraction, -1 = unset
    int force_splits;   // GAN_HIP_FORCE_SPLITS: split-K factor of every contraction, 0 = unset
    int no_thin;        // GAN_HIP_NO_THIN: thin outputs stay on the MFMA path
    int slow_gather;    // GAN_HIP_SLOW_GATHER: general gather arithmetic in the split kernel
    int scalar_epilogue;  // GAN_HIP_SCALAR_EPILOGUE: dword stores instead of the 16-byte epilogue
    int three_launch_bn;  // GAN_HIP_THREE_LAUNCH_BN: every batch norm on the general three-launch path
    int two_launch_bn;    // GAN_HIP_TWO_LAUNCH_BN: 0 keeps batch norms above 1024 rows per group on the three-launch path
    int separate_colsum;  // GAN_HIP_SEPARATE_COLSUM: bias gradients as launches of their own
    int row_gather_any;   // GAN_HIP_ROW_GATHER_ANY: the filter-row gather for every channel count that is no multiple of 4 (default 1)
    int rgb_gather;       // GAN_HIP_RGB_GATHER: filter-row gather for 3-channel images: 1 FPROP and WGRAD, 2 WGRAD only, 0 off
    int tile_order;       // GAN_HIP_TILE_ORDER: 0 chosen per launch, 1 launch order, 2 by row tile, 3 by k range
    int exact_f32;      // GAN_HIP_EXACT_F32: default of gan_set_exact_f32
    int no_bn_ride;     // GAN_HIP_NO_BN_RIDE: 1 gan_bn_tile_blocks reports 0 (callers keep the separate batch-norm launches); 2 the
                        // reductions asked for run as launches of their own behind the contraction instead of riding
    int rgb_direct;     // GAN_HIP_RGB_DIRECT: 0 keeps forward passes over 3-channel images on the general kernels
    int no_carry;       // GAN_HIP_NO_CARRY: the last stage of a bias-gradient column sum never rides on a contraction launch
};
Knobs& knobs();

inline hipStream_t as_stream(gan_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Last stage of a bias-gradient column sum: out[ch] (+)= sum over row blocks of partial[rb][ch], one wave per
// channel with a fixed shuffle tree.  It depends only on the partial sums, so besides its own launch it can be
// carried by `blocks` extra workgroups at the end of another launch's grid (the split-K slab reduction of the
// weight gradient it belongs to).
struct ColsumFinal {
    const float* partial;
    float* out;
    int c, rblocks, accumulate;
    int blocks;  // workgroups of 256 threads that cover c channels; 0 = no job
};

__device__ __forceinline__ void colsum_final_block(const ColsumFinal& j, int block) {
    const int ch = (block * 256 + (int)threadIdx.x) >> 6;  // one wave per channel
    const int lane = threadIdx.x & 63;
    if (ch >= j.c) return;
    float s = 0.f;
    for (int rb = lane; rb < j.rblocks; rb += 64) s += j.partial[(long)rb * j.c + ch];
    s = wave_sum(s);
    if (lane == 0) j.out[ch] = (j.accumulate ? j.out[ch] : 0.f) + s;
}

__device__ __forceinline__ float apply_act(float z, int act, float param) {
    switch (act) {
        case GAN_ACT_RELU: return z > 0.f ? z : 0.f;
        case GAN_ACT_LRELU: return fmaxf(z, param * z);
        case GAN_ACT_TANH: return tanhf(z);
        case GAN_ACT_SIGMOID: return z >= 0.f ? 1.f / (1.f + expf(-z)) : expf(z) / (1.f + expf(z));
        default: return z;
    }
}

// d act / dz given the pre-activation z and the activation value a = act(z).
// relu: 0 at z == 0; lrelu (= max(z, leak*z)): slope 1 at z == 0 (first argument wins ties).
__device__ __forceinline__ float act_grad(float z, float a, int act, float param) {
    switch (act) {
        case GAN_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case GAN_ACT_LRELU: return z >= 0.f ? 1.f : param;
        case GAN_ACT_TANH: return 1.f - a * a;
        case GAN_ACT_SIGMOID: return a * (1.f - a);
        default: return 1.f;
    }
}

// From a saved output y = act(x)*scale+shift: a = (y-shift)/scale has the sign of x for relu / lrelu and equals
// tanh(x) / sigmoid(x), which is all the derivative needs.  Returns dy * d y / d x.
__device__ __forceinline__ float act_bwd_one(float y, float dy, int act, float param, float scale, float shift) {
    const float a = (scale == 1.f && shift == 0.f) ? y : (y - shift) / scale;
    return dy * scale * act_grad(a, a, act, param);
}

// What a contraction does to a finished output element v (bias already added) before storing it:
//   gate == nullptr:  v <- act(v) * scale + shift            (activation fused behind conv / deconv)
//   gate != nullptr:  v <- v * d act / d x at gate[same index] (the output is the gradient w.r.t. an activated tensor
//                     `gate` = act(x)*scale+shift; the stored value is the gradient w.r.t. x)
// act == GAN_ACT_NONE with scale 1, shift 0 and no gate is the identity.
struct Epilogue {
    int act;
    float param, scale, shift;
    const float* gate;
};

__host__ __device__ __forceinline__ bool epi_active(const Epilogue& e) {
    return e.gate != nullptr || e.act != GAN_ACT_NONE || e.scale != 1.f || e.shift != 0.f;
}

__device__ __forceinline__ float epi_apply(const Epilogue& e, float v, long idx) {
    if (e.gate != nullptr) return act_bwd_one(e.gate[idx], v, e.act, e.param, e.scale, e.shift);
    return apply_act(v, e.act, e.param) * e.scale + e.shift;
}

// Four consecutive elements at once: ONE dispatch on the activation instead of one per element (the epilogues of
// the contraction kernels store 16 bytes per lane).
__device__ __forceinline__ float4 epi_apply4(const Epilogue& e, float4 v, long idx) {
    if (e.gate != nullptr) {
        const float4 g = *reinterpret_cast<const float4*>(e.gate + idx);
        const bool plain = (e.scale == 1.f && e.shift == 0.f);
        const float inv = 1.f / e.scale;
        const float a0 = plain ? g.x : (g.x - e.shift) * inv, a1 = plain ? g.y : (g.y - e.shift) * inv;
        const float a2 = plain ? g.z : (g.z - e.shift) * inv, a3 = plain ? g.w : (g.w - e.shift) * inv;
        switch (e.act) {
            case GAN_ACT_RELU:
                v.x = a0 > 0.f ? v.x : 0.f; v.y = a1 > 0.f ? v.y : 0.f; v.z = a2 > 0.f ? v.z : 0.f; v.w = a3 > 0.f ? v.w : 0.f;
                break;
            case GAN_ACT_LRELU:
                v.x *= a0 >= 0.f ? 1.f : e.param; v.y *= a1 >= 0.f ? 1.f : e.param;
                v.z *= a2 >= 0.f ? 1.f : e.param; v.w *= a3 >= 0.f ? 1.f : e.param;
                break;
            case GAN_ACT_TANH:
                v.x *= 1.f - a0 * a0; v.y *= 1.f - a1 * a1; v.z *= 1.f - a2 * a2; v.w *= 1.f - a3 * a3;
                break;
            case GAN_ACT_SIGMOID:
                v.x *= a0 * (1.f - a0); v.y *= a1 * (1.f - a1); v.z *= a2 * (1.f - a2); v.w *= a3 * (1.f - a3);
                break;
            default: break;
        }
        v.x *= e.scale; v.y *= e.scale; v.z *= e.scale; v.w *= e.scale;
        return v;
    }
    switch (e.act) {
        case GAN_ACT_RELU:
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            break;
        case GAN_ACT_LRELU:
            v.x = fmaxf(v.x, e.param * v.x); v.y = fmaxf(v.y, e.param * v.y);
            v.z = fmaxf(v.z, e.param * v.z); v.w = fmaxf(v.w, e.param * v.w);
            break;
        case GAN_ACT_TANH: v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); break;
        case GAN_ACT_SIGMOID:
            v.x = apply_act(v.x, GAN_ACT_SIGMOID, 0.f); v.y = apply_act(v.y, GAN_ACT_SIGMOID, 0.f);
            v.z = apply_act(v.z, GAN_ACT_SIGMOID, 0.f); v.w = apply_act(v.w, GAN_ACT_SIGMOID, 0.f);
            break;
        default: break;
    }
    v.x = v.x * e.scale + e.shift; v.y = v.y * e.scale + e.shift;
    v.z = v.z * e.scale + e.shift; v.w = v.w * e.scale + e.shift;
    return v;
}

inline Epilogue no_epilogue() { return Epilogue{GAN_ACT_NONE, 0.f, 1.f, 0.f, nullptr}; }

}  // namespace gan
