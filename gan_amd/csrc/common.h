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

inline hipStream_t as_stream(gan_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
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

}  // namespace gan
