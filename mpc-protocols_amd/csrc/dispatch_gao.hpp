// dispatch_gao.hpp -- launch of the OEC/Gao kernels for one field (instantiated per field in tu_gao_*.hip)
#pragma once
#include "launchers.hpp"

namespace hbmpc {

template <class F, bool INLINE>
inline void launch_gao_shape(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s) {
    // lanes per chunk: polynomials have up to n + 1 coefficients
    if (n <= 15) {
        hipLaunchKernelGGL((k_gao<F, 64, 16, INLINE>), dim3(grid), dim3(64), 4 * gao_group_words(16, F::NL) * 4, s, ga);
    } else if (n <= 31) {
        hipLaunchKernelGGL((k_gao<F, 64, 32, INLINE>), dim3(grid), dim3(64), 2 * gao_group_words(32, F::NL) * 4, s, ga);
    } else if (n <= 63) {
        hipLaunchKernelGGL((k_gao<F, 64, 64, INLINE>), dim3(grid), dim3(64), gao_group_words(64, F::NL) * 4, s, ga);
    } else if (n <= 127) {
        hipLaunchKernelGGL((k_gao<F, 128, 128, INLINE>), dim3(grid), dim3(128), gao_group_words(128, F::NL) * 4, s, ga);
    } else {
        hipLaunchKernelGGL((k_gao<F, 256, 256, INLINE>), dim3(grid), dim3(256), gao_group_words(256, F::NL) * 4, s, ga);
    }
}
// inline_unscale: small batches -- k_gao un-scales its own results and ends the call; otherwise k_unscale follows
// (one inversion per eight flagged chunks)
template <class F>
inline void launch_gao_t(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s, bool inline_unscale) {
    if (inline_unscale) {
        launch_gao_shape<F, true>(ga, n, grid, s);
        return;
    }
    launch_gao_shape<F, false>(ga, n, grid, s);
    const size_t lanes = (ga.G + 7) / 8;
    hipLaunchKernelGGL((k_unscale<F>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, ga);
}

}  // namespace hbmpc
