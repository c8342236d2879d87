// Goldilocks instantiations of the register-resident batch-recover kernel
#include <utility>

#include "fr_gold.hpp"
#include "launchers.hpp"
namespace hbmpc {
template <int M, bool P0>
static void one(const RecoverArgs& ra, unsigned grid, hipStream_t s) {
    const size_t lds = (size_t)((ra.needed - M) + (P0 ? 1 : M)) * M * Gold::NL * 4;
    hipLaunchKernelGGL((k_batch_recover<Gold, M, P0>), dim3(grid), dim3(256), lds, s, ra);
}
template <int LO, int... I>
static bool range(int m, bool p0, const RecoverArgs& ra, unsigned grid, hipStream_t s, std::integer_sequence<int, I...>) {
    bool hit = false;
    ((m == LO + I ? (p0 ? one<LO + I, true>(ra, grid, s) : one<LO + I, false>(ra, grid, s), hit = true) : false), ...);
    return hit;
}
bool launch_gold_recover(int m, bool p0, const RecoverArgs& ra, unsigned grid, hipStream_t s) {
    return range<1>(m, p0, ra, grid, s, std::make_integer_sequence<int, 16>{});
}
}
