#include "dispatch_gao.hpp"
namespace hbmpc {
void launch_gao_sat(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s, bool inline_unscale) {
    launch_gao_t<Sat32>(ga, n, grid, s, inline_unscale);
}
}
