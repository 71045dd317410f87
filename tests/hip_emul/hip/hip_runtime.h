// A few lines of HIP for the host compiler: enough to run the batch-norm kernels of gan_amd/csrc/norm.hip on the CPU,
// one OS thread per work-item and one workgroup at a time, so that AddressSanitizer sees every access they make.
// Test infrastructure only (tests/test_bn_tile_emulated.py); nothing here is part of the library.
#pragma once
#include <pthread.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
typedef void* hipStream_t;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }

struct EmulIdx { unsigned x, y, z; };
extern thread_local EmulIdx threadIdx, blockIdx;
extern EmulIdx blockDim, gridDim;
extern pthread_barrier_t* emul_barrier;
inline void __syncthreads() { pthread_barrier_wait(emul_barrier); }
// (shuffles serve the one-launch kernels and the column-sum tail, which this harness does not run)
inline float __shfl_down(float v, int, int) { return v; }
using std::max;
using std::min;

template <typename K, typename... A>
void emul_launch(K kernel, dim3 grid, dim3 block, A... args) {
    blockDim = EmulIdx{block.x, block.y, block.z};
    gridDim = EmulIdx{grid.x, grid.y, grid.z};
    const unsigned nt = block.x * block.y * block.z;
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, nt);
    emul_barrier = &bar;
    // one OS thread per work-item, reused for every workgroup of the launch; the workgroups run one after the other
    // (every work-item of a workgroup must reach the same barriers, as on the device)
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([=, &bar]() {
            threadIdx = EmulIdx{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        blockIdx = EmulIdx{bx, by, bz};
                        kernel(args...);
                        pthread_barrier_wait(&bar);
                    }
        });
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&bar);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emul_launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)
