// dispatch_eval.hpp -- compile-time ranges of the pruned-FFT kernels
#pragma once
#include <utility>

#include "kernels_eval.hpp"

namespace hbmpc {

// number of independent [G][d+1] -> [n][G] problems of the launch being issued (blockIdx.y); set by the C ABI layer
// around the launcher calls of one thread (hbmpc_capi.hip)
extern thread_local unsigned g_eval_parties;
// elements between consecutive output rows of that launch (0: dense, = G); same mechanism
extern thread_local size_t g_eval_ystride;

template <class F, int LOG, int CNT>
inline void launch_fft1_one(const uint32_t* x, size_t G, int n, const uint32_t* tw, uint32_t* y, hipStream_t s) {
    const unsigned grid = (unsigned)((G + EVAL_TILE - 1) / EVAL_TILE);
    const size_t lds = (size_t)EVAL_TILE * (CNT * F::EW + TILE_PAD<F::EW>) * 4;
    hipLaunchKernelGGL((k_eval_fft1<F, LOG, CNT>), dim3(grid, g_eval_parties), dim3(EVAL_TILE), lds, s, x, G, n, tw, y, g_eval_ystride ? g_eval_ystride : G);
}
// cnt in [LO, LO + sizeof...(I))
template <class F, int LOG, int LO, int... I>
inline bool dispatch_fft1_range(int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, uint32_t* y,
                                hipStream_t s, std::integer_sequence<int, I...>) {
    bool hit = false;
    ((cnt == LO + I ? (launch_fft1_one<F, LOG, LO + I>(x, G, n, tw, y, s), hit = true) : false), ...);
    return hit;
}
template <class F, int CNT16, bool FOLD>
inline void launch_fftP_one(const uint32_t* x, size_t G, int n, int dp1, int P, const uint32_t* tw16,
                            const uint32_t* twist, uint32_t* y, hipStream_t s) {
    const unsigned grid = (unsigned)((G + EVAL_TILE - 1) / EVAL_TILE);
    const size_t lds = (size_t)EVAL_TILE * (dp1 * F::EW + TILE_PAD<F::EW>) * 4;
    hipLaunchKernelGGL((k_eval_fftP<F, CNT16, FOLD>), dim3(grid, g_eval_parties), dim3(EVAL_TILE), lds, s, x, G, n, dp1, P, tw16, twist,
                       y, g_eval_ystride ? g_eval_ystride : G);
}
template <class F, int LO, int... I>
inline bool dispatch_fftP_range(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16,
                                const uint32_t* twist, uint32_t* y, hipStream_t s, std::integer_sequence<int, I...>) {
    bool hit = false;
    ((dp1 == LO + I ? (launch_fftP_one<F, LO + I, false>(x, G, n, dp1, P, tw16, twist, y, s), hit = true) : false), ...);
    return hit;
}

}  // namespace hbmpc
