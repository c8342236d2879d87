#include "fr_gold.hpp"
#include "dispatch_gao.hpp"
namespace hbmpc {
void launch_gao_gold(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s, bool inline_unscale) {
    launch_gao_t<Gold>(ga, n, grid, s, inline_unscale);
}
}
