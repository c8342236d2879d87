#include "dispatch_eval.hpp"
#include "launchers.hpp"
namespace hbmpc {
bool launch_fft1_16c(int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s) {
    return dispatch_fft1_range<U29, 4, 9>(cnt, x, G, n, tw, y, s, std::make_integer_sequence<int, 4>{});
}
}
