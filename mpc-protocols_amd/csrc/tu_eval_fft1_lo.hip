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
}
