#include "dispatch_eval.hpp"
#include "launchers.hpp"
namespace hbmpc {
bool launch_fftP_fold(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16, const uint32_t* twist,
                      EvalOut y, hipStream_t s) {
    if (dp1 <= 16 || dp1 > 32) return false;
    launch_fftP_one<U29, 16, true>(x, G, n, dp1, P, tw16, twist, y, s);
    return true;
}
}
