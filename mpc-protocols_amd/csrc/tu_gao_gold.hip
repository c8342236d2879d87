#include "fr_gold.hpp"
#include "launchers.hpp"
namespace hbmpc {
void launch_gao_gold(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s) {
    using F = Gold;
    if (n <= 63) {
        const size_t lds = (size_t)(4 * 64 + 4) * F::NL * 4 + 64;
        hipLaunchKernelGGL((k_gao<F, 64>), dim3(grid), dim3(64), lds, s, ga);
    } else if (n <= 127) {
        const size_t lds = (size_t)(4 * 128 + 4) * F::NL * 4 + 64;
        hipLaunchKernelGGL((k_gao<F, 128>), dim3(grid), dim3(128), lds, s, ga);
    } else {
        const size_t lds = (size_t)(4 * 256 + 4) * F::NL * 4 + 64;
        hipLaunchKernelGGL((k_gao<F, 256>), dim3(grid), dim3(256), lds, s, ga);
    }
    // un-scale the accepted quotients: one lane per eight flagged chunks
    const size_t lanes = (ga.G + 7) / 8;
    hipLaunchKernelGGL((k_unscale<F>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, ga);
}
}
