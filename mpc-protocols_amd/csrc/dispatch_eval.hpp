// dispatch_eval.hpp -- compile-time ranges of the pruned-FFT kernels
#pragma once
#include <utility>

#include "eval_out.hpp"
#include "kernels_eval.hpp"

namespace hbmpc {

template <class F, int LOG, int CNT>
inline void launch_fft1_one(const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s) {
    const unsigned grid = (unsigned)((G + EVAL_TILE - 1) / EVAL_TILE);
    const size_t lds = (size_t)EVAL_TILE * (CNT * F::EW + TILE_PAD<F::EW>) * 4;
    hipLaunchKernelGGL((k_eval_fft1<F, LOG, CNT>), dim3(grid, y.parties), dim3(EVAL_TILE), lds, s, x, G, n, tw, y.y, y.ys ? y.ys : G);
}
// cnt in [LO, LO + sizeof...(I))
template <class F, int LOG, int LO, int... I>
inline bool dispatch_fft1_range(int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y,
                                hipStream_t s, std::integer_sequence<int, I...>) {
    bool hit = false;
    ((cnt == LO + I ? (launch_fft1_one<F, LOG, LO + I>(x, G, n, tw, y, s), hit = true) : false), ...);
    return hit;
}
template <class F, int CNT16, bool FOLD>
inline void launch_fftP_one(const uint32_t* x, size_t G, int n, int dp1, int P, const uint32_t* tw16,
                            const uint32_t* twist, EvalOut y, hipStream_t s) {
    const unsigned grid = (unsigned)((G + EVAL_TILE - 1) / EVAL_TILE);
    const size_t lds = (size_t)EVAL_TILE * (dp1 * F::EW + TILE_PAD<F::EW>) * 4;
    hipLaunchKernelGGL((k_eval_fftP<F, CNT16, FOLD>), dim3(grid, y.parties), dim3(EVAL_TILE), lds, s, x, G, n, dp1, P, tw16, twist,
                       y.y, y.ys ? y.ys : G);
}
template <class F, int LO, int... I>
inline bool dispatch_fftP_range(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16,
                                const uint32_t* twist, EvalOut y, hipStream_t s, std::integer_sequence<int, I...>) {
    bool hit = false;
    ((dp1 == LO + I ? (launch_fftP_one<F, LO + I, false>(x, G, n, dp1, P, tw16, twist, y, s), hit = true) : false), ...);
    return hit;
}

}  // namespace hbmpc
