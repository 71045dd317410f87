// Micro-probe: what does one wave per SIMD sustain on v_mfma_f32_32x32x2_f32 when the loop around
// it grows from "registers only" to the structure of the implicit-GEMM kernel?  Tuning aid.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LD = 130;
constexpr int BK = 16;

template <int VARIANT>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ out, long long* cycles,
                                             int iters) {
    __shared__ float smem[2 * BK * (LD + LD + 4)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    float* As = smem;
    float* Bs = smem + BK * LD;
    for (int i = tid; i < 2 * BK * (LD + LD + 4); i += 256) smem[i] = src[i & 1023];
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fa = wm * 64 + (lane & 31), fb = wn * 64 + (lane & 31), kh2 = lane >> 5;
    float a0 = src[tid], a1 = src[tid + 256], b0 = src[tid + 512], b1 = src[tid + 768];
    float4 g0 = make_float4(0, 0, 0, 0), g1 = g0, g2 = g0, g3 = g0;
    const float4* gsrc = reinterpret_cast<const float4*>(src) + blockIdx.x * 4096 + tid;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (VARIANT >= 5) {
        const float4* g5 = reinterpret_cast<const float4*>(src) + (VARIANT == 5 ? 0 : blockIdx.x * 4096) + tid;
        auto frag = [&](int kk, float (&af)[2], float (&bf)[2]) {
            af[0] = As[(kk + kh2) * LD + fa]; af[1] = As[(kk + kh2) * LD + fa + 32];
            bf[0] = Bs[(kk + kh2) * LD + fb]; bf[1] = Bs[(kk + kh2) * LD + fb + 32];
        };
        auto mma = [&](const float (&af)[2], const float (&bf)[2]) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc[1][1], 0, 0, 0);
        };
        g0 = g5[0]; g1 = g5[2048]; g2 = g5[1024]; g3 = g5[3072];
        for (int it = 0; it < iters; ++it) {
            float af0[2], bf0[2], af1[2], bf1[2];
            float* Aw = smem + ((it & 1) ^ 1) * BK * (LD + LD + 4);
            float* Bw = Aw + BK * LD;
            const int chunk = tid & 3, row = tid >> 2;
            frag(0, af0, bf0);
            frag(2, af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mma(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            frag(4, af0, bf0);
            if (VARIANT != 7 && VARIANT != 9) {
                Aw[(chunk * 4 + 0) * LD + row] = g0.x; Aw[(chunk * 4 + 1) * LD + row] = g0.y;
                Aw[(chunk * 4 + 2) * LD + row] = g0.z; Aw[(chunk * 4 + 3) * LD + row] = g0.w;
                Aw[(chunk * 4 + 0) * LD + row + 64] = g1.x; Aw[(chunk * 4 + 1) * LD + row + 64] = g1.y;
                Aw[(chunk * 4 + 2) * LD + row + 64] = g1.z; Aw[(chunk * 4 + 3) * LD + row + 64] = g1.w;
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            frag(6, af1, bf1);
            if (VARIANT != 7 && VARIANT != 8) {
                *reinterpret_cast<float4*>(&Bw[(tid >> 5) * (LD + 2) + (tid & 31) * 4]) = g2;
                *reinterpret_cast<float4*>(&Bw[((tid >> 5) + 8) * (LD + 2) + (tid & 31) * 4]) = g3;
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            frag(8, af0, bf0);
            g0 = g5[(it & 7) * 256]; g1 = g5[(it & 7) * 256 + 2048];
            __builtin_amdgcn_sched_barrier(0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            frag(10, af1, bf1);
            g2 = g5[(it & 7) * 256 + 1024]; g3 = g5[(it & 7) * 256 + 3072];
            __builtin_amdgcn_sched_barrier(0);
            mma(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            frag(12, af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            frag(14, af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mma(af0, bf0);
            mma(af1, bf1);
            __syncthreads();
            As = Aw; Bs = Bw;
        }
    } else
    for (int it = 0; it < iters; ++it) {
        if (VARIANT >= 4) {  // global loads for a later slice
            g0 = gsrc[(it & 7) * 256];
            g1 = gsrc[(it & 7) * 256 + 2048];
            g2 = gsrc[(it & 7) * 256 + 1024];
            g3 = gsrc[(it & 7) * 256 + 3072];
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[2], bf[2];
            if (VARIANT >= 1) {
                af[0] = As[(kk + kh2) * LD + fa];
                af[1] = As[(kk + kh2) * LD + fa + 32];
                bf[0] = Bs[(kk + kh2) * LD + fb];
                bf[1] = Bs[(kk + kh2) * LD + fb + 32];
            } else {
                af[0] = a0; af[1] = a1; bf[0] = b0; bf[1] = b1;
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc[1][1], 0, 0, 0);
        }
        if (VARIANT >= 3) {  // stage the next slice (as the real kernel: 8 scalar + 2 vector LDS writes)
            float* Aw = smem + ((it & 1) ^ 1) * BK * (LD + LD + 4);
            const int chunk = tid & 3, row = tid >> 2;
            const float4 va = VARIANT >= 4 ? g0 : make_float4(a0, a1, b0, b1);
            const float4 vb = VARIANT >= 4 ? g1 : make_float4(a1, a0, b1, b0);
            Aw[(chunk * 4 + 0) * LD + row] = va.x; Aw[(chunk * 4 + 1) * LD + row] = va.y;
            Aw[(chunk * 4 + 2) * LD + row] = va.z; Aw[(chunk * 4 + 3) * LD + row] = va.w;
            Aw[(chunk * 4 + 0) * LD + row + 64] = vb.x; Aw[(chunk * 4 + 1) * LD + row + 64] = vb.y;
            Aw[(chunk * 4 + 2) * LD + row + 64] = vb.z; Aw[(chunk * 4 + 3) * LD + row + 64] = vb.w;
            float* Bw = Aw + BK * LD;
            *reinterpret_cast<float4*>(&Bw[(tid >> 5) * (LD + 2) + (tid & 31) * 4]) = VARIANT >= 4 ? g2 : va;
            *reinterpret_cast<float4*>(&Bw[((tid >> 5) + 8) * (LD + 2) + (tid & 31) * 4]) = VARIANT >= 4 ? g3 : vb;
        }
        if (VARIANT >= 2) __syncthreads();
        if (VARIANT >= 3) {
            As = smem + ((it & 1) ^ 1) * BK * (LD + LD + 4);
            Bs = As + BK * LD;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* name, const float* src, float* out, long long* cyc, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto c : h) avg += (double)c;
    avg /= blocks;
    const double flops = 2.0 * 128 * 128 * BK * (double)iters * blocks;
    printf("%-34s blocks %4d: %7.1f cycles / slice (ideal 2048), wall %.3f ms, %.1f TFLOP/s\n", name, blocks,
           avg / iters, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *src, *out;
    long long* cyc;
    const size_t n = 64u << 20;
    hipMalloc(&src, n * sizeof(float));
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipMalloc(&cyc, 4096 * sizeof(long long));
    hipMemset(src, 0, n * sizeof(float));
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    for (size_t off = 0; off < n; off += h.size()) hipMemcpy(src + off, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 200;
    for (int blocks : {256, 512, 768}) {
        run<0>("V0 registers only", src, out, cyc, blocks, iters);
        run<1>("V1 + LDS fragment reads", src, out, cyc, blocks, iters);
        run<2>("V2 + barrier per slice", src, out, cyc, blocks, iters);
        run<3>("V3 + LDS staging writes", src, out, cyc, blocks, iters);
        run<4>("V4 + global loads", src, out, cyc, blocks, iters);
        run<5>("V5 pinned pipeline, L2 loads", src, out, cyc, blocks, iters);
        run<6>("V6 pinned pipeline, streamed loads", src, out, cyc, blocks, iters);
        run<7>("V7 = V6 without LDS writes", src, out, cyc, blocks, iters);
        run<8>("V8 = V6, A (scalar) writes only", src, out, cyc, blocks, iters);
        run<9>("V9 = V6, B (b128) writes only", src, out, cyc, blocks, iters);
    }
    return 0;
}
