#include "launchers.hpp"
namespace hbmpc {
void launch_gao_u29(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s) {
    using F = U29;
    // lanes per chunk: polynomials have up to n + 1 coefficients
    if (n <= 15) {
        hipLaunchKernelGGL((k_gao<F, 64, 16>), dim3(grid), dim3(64), 4 * gao_group_words(16, F::NL) * 4, s, ga);
    } else if (n <= 31) {
        hipLaunchKernelGGL((k_gao<F, 64, 32>), dim3(grid), dim3(64), 2 * gao_group_words(32, F::NL) * 4, s, ga);
    } else if (n <= 63) {
        hipLaunchKernelGGL((k_gao<F, 64, 64>), dim3(grid), dim3(64), gao_group_words(64, F::NL) * 4, s, ga);
    } else if (n <= 127) {
        hipLaunchKernelGGL((k_gao<F, 128, 128>), dim3(grid), dim3(128), gao_group_words(128, F::NL) * 4, s, ga);
    } else {
        hipLaunchKernelGGL((k_gao<F, 256, 256>), dim3(grid), dim3(256), gao_group_words(256, F::NL) * 4, s, ga);
    }
    // un-scale the accepted quotients: one lane per eight flagged chunks
    const size_t lanes = (ga.G + 7) / 8;
    hipLaunchKernelGGL((k_unscale<F>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, ga);
}
}
