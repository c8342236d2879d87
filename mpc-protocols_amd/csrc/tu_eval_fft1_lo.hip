#include "dispatch_eval.hpp"
#include "launchers.hpp"
namespace hbmpc {
bool launch_fft1_lo(int log, int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s) {
    switch (log) {
        case 0: return dispatch_fft1_range<U29, 0, 1>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 1>{});
        case 1: return dispatch_fft1_range<U29, 1, 1>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 2>{});
        case 2: return dispatch_fft1_range<U29, 2, 1>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 4>{});
        case 3: return dispatch_fft1_range<U29, 3, 1>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 8>{});
    }
    return false;
}
// the producers' mixing step on domains of 4 and 8 points (3 .. 8 parties): CNT = n inputs as rows, lists and party-major rows written by the
// kernel (kernels_eval.hpp: k_eval_fft1_mix) -- where the point-pair matrix-core kernel does not cover the shape
template <int LOG, int CNT>
static void mix_one(const uint32_t* x, size_t xs, size_t G, int n, const uint32_t* tw, const MixOut& o, hipStream_t s) {
    const unsigned grid = (unsigned)((G + EVAL_TILE - 1) / EVAL_TILE);
    hipLaunchKernelGGL((k_eval_fft1_mix<U29, LOG, CNT>), dim3(grid), dim3(EVAL_TILE), 0, s, x, xs, G, n, tw, o);
}
template <int LOG, int LO, int... I>
static bool mix_range(int cnt, const uint32_t* x, size_t xs, size_t G, int n, const uint32_t* tw, const MixOut& o, hipStream_t s,
                      std::integer_sequence<int, I...>) {
    bool hit = false;
    ((cnt == LO + I ? (mix_one<LOG, LO + I>(x, xs, G, n, tw, o, s), hit = true) : false), ...);
    return hit;
}
bool launch_fft1_mix_lo(int log, int cnt, const uint32_t* x, size_t xs, size_t G, int n, const uint32_t* tw, const MixOut& o, hipStream_t s) {
    switch (log) {
        case 2: return mix_range<2, 3>(cnt, x, xs, G, n, tw, o, s, std::make_integer_sequence<int, 2>{});   // 3, 4 parties
        case 3: return mix_range<3, 5>(cnt, x, xs, G, n, tw, o, s, std::make_integer_sequence<int, 4>{});   // 5 .. 8
    }
    return false;
}
}
