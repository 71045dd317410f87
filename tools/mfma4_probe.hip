#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ short bf(float x) { return (short)(__float_as_uint(x) >> 16); }
// one-hot A at (lane la, element ka), B all ones; or one-hot B, A all ones
__global__ void k(int which, int la, int ka, float* out) {
    const int l = threadIdx.x;
    s4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = bf(which == 0 ? 0.f : 1.f); b[i] = bf(which == 1 ? 0.f : 1.f); }
    if (l == la) { if (which == 0) a[ka] = bf(1.f); else b[ka] = bf(1.f); }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    std::vector<float> h(256);
    for (int which = 0; which < 2; ++which) {
        printf("one-hot %s:\n", which == 0 ? "A" : "B");
        for (int la = 0; la < 64; la += (la < 8 ? 1 : 7)) for (int ka = 0; ka < 4; ++ka) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, which, la, ka, d);
            hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
            printf("  lane %2d elem %d ->", la, ka);
            int n = 0;
            for (int i = 0; i < 256; ++i) if (h[i] != 0.f) { if (n < 10) printf(" (l%d,r%d)=%g", i / 4, i % 4, h[i]); ++n; }
            printf("  [%d nonzero]\n", n);
        }
    }
    return 0;
}
