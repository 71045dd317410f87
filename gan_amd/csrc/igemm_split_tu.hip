// The bf16 matrix-pipe contraction kernels (igemm_split.h), one translation unit per (contraction mode, tile family)
// so that the instantiations compile side by side: built nine times with
//   -DGAN_SPLIT_MODE=0|1|2   (FPROP, TPROP, WGRAD)
//   -DGAN_SPLIT_PART=0|1|2   (tile configurations {4, 7}: 128x128, {5, 8}: 128x64, {6, 9}: 64x64 and 128x32)
#include "igemm_common.h"

#ifndef GAN_SPLIT_MODE
#error "compile with -DGAN_SPLIT_MODE=0|1|2 -DGAN_SPLIT_PART=0|1|2"
#endif

namespace gan {

#include "igemm_split.h"

// The cheap gather arithmetic needs 16-channel alignment of the gathered image (FPROP: big side, TPROP: small
// side) or, for WGRAD, a small-image width that is a multiple of 4.  Mirrored taps (a.reflect) exist for the cheap
// arithmetic and for the filter-row gather of 3-channel images only: the planner keeps other geometries on the fp32
// kernel.
template <int MODE, int BM, int BN, int KG = 1, int WM = 2, int WN = 2>
static void launch_split(const IgemmArgs& a, bool slow_gather, dim3 grid, hipStream_t s) {
    const gan_conv_geom& g = a.g;
    const bool fast = (MODE == MODE_FPROP) ? (g.cb % 16 == 0)
                    : (MODE == MODE_TPROP) ? (g.cs % 16 == 0) : (g.ws % 4 == 0);
    if constexpr (MODE != MODE_TPROP) {
        if (a.row3_spr > 0) {
            if (a.reflect)
                hipLaunchKernelGGL((igemm_split_kernel<MODE, BM, BN, WM, WN, false, KG, true, true>), grid, dim3(256 * KG), 0, s, a);
            else
                hipLaunchKernelGGL((igemm_split_kernel<MODE, BM, BN, WM, WN, false, KG, true, false>), grid, dim3(256 * KG), 0, s, a);
            return;
        }
        if (a.reflect) {   // (the planner has checked `fast`)
            hipLaunchKernelGGL((igemm_split_kernel<MODE, BM, BN, WM, WN, true, KG, false, true>), grid, dim3(256 * KG), 0, s, a);
            return;
        }
    }
    if (fast && !slow_gather)
        hipLaunchKernelGGL((igemm_split_kernel<MODE, BM, BN, WM, WN, true, KG>), grid, dim3(256 * KG), 0, s, a);
    else
        hipLaunchKernelGGL((igemm_split_kernel<MODE, BM, BN, WM, WN, false, KG>), grid, dim3(256 * KG), 0, s, a);
}

#define GAN_CAT2(a, b, c) a##b##_p##c
#define GAN_CAT(a, b, c) GAN_CAT2(a, b, c)

void GAN_CAT(launch_split_m, GAN_SPLIT_MODE, GAN_SPLIT_PART)(const IgemmArgs& a, int cfg, bool slow_gather, dim3 grid, hipStream_t s) {
    constexpr int MODE = GAN_SPLIT_MODE;
#if GAN_SPLIT_PART == 0
    if (cfg == 4) launch_split<MODE, 128, 128>(a, slow_gather, grid, s);
    else launch_split<MODE, 128, 128, 2>(a, slow_gather, grid, s);
#elif GAN_SPLIT_PART == 1
    if (cfg == 5) launch_split<MODE, 128, 64>(a, slow_gather, grid, s);
    else launch_split<MODE, 128, 64, 2>(a, slow_gather, grid, s);
#else
#if GAN_SPLIT_MODE != 1
    if (cfg == 9) { launch_split<MODE, 128, 32, 1, 4, 1>(a, slow_gather, grid, s); return; }
#endif
    launch_split<MODE, 64, 64>(a, slow_gather, grid, s);
#endif
}

}  // namespace gan
