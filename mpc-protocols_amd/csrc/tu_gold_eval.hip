// Goldilocks instantiations of the pruned-FFT evaluation kernels
#include "fr_gold.hpp"
#include "dispatch_eval.hpp"
#include "launchers.hpp"
namespace hbmpc {
bool launch_gold_fft1(int log, int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s) {
    switch (log) {
        case 0: return dispatch_fft1_range<Gold, 0, 1>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 1>{});
        case 1: return dispatch_fft1_range<Gold, 1, 1>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 2>{});
        case 2: return dispatch_fft1_range<Gold, 2, 1>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 4>{});
        case 3: return dispatch_fft1_range<Gold, 3, 1>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 8>{});
        case 4: return dispatch_fft1_range<Gold, 4, 1>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 16>{});
    }
    return false;
}
bool launch_gold_fftP(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16, const uint32_t* twist,
                      EvalOut y, hipStream_t s) {
    if (dp1 > 16 && dp1 <= 32) {
        launch_fftP_one<Gold, 16, true>(x, G, n, dp1, P, tw16, twist, y, s);
        return true;
    }
    return dispatch_fftP_range<Gold, 1>(dp1, x, G, n, P, tw16, twist, y, s, std::make_integer_sequence<int, 16>{});
}
}
