#include "dispatch_gao.hpp"
namespace hbmpc {
void launch_gao_u29(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s, bool inline_unscale) {
    launch_gao_t<U29>(ga, n, grid, s, inline_unscale);
}
}
