#include "dispatch_eval.hpp"
#include "launchers.hpp"
namespace hbmpc {
bool launch_fftP_b(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16, const uint32_t* twist,
                   EvalOut y, hipStream_t s) {
    return dispatch_fftP_range<U29, 5>(dp1, x, G, n, P, tw16, twist, y, s, std::make_integer_sequence<int, 4>{});
}
}
