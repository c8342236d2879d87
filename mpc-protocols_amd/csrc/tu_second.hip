// tu_second.hip -- k_second_chance_m<U29, M> for M = 2 .. 16 (long flagged lists: kernels_recover.hpp)
#include <utility>

#include "launchers.hpp"
namespace hbmpc {
namespace {
template <int M>
void one(const SecondArgs& a, unsigned grid, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((k_second_chance_m<U29, M>), dim3(grid), dim3(256), lds, s, a);
}
template <int LO, int... I>
bool range(int m, const SecondArgs& a, unsigned grid, size_t lds, hipStream_t s, std::integer_sequence<int, I...>) {
    bool hit = false;
    ((m == LO + I ? (one<LO + I>(a, grid, lds, s), hit = true) : false), ...);
    return hit;
}
}  // namespace
bool launch_second_chance_m(int m, const SecondArgs& a, unsigned grid, hipStream_t s) {
    const size_t lds = (size_t)a.P * m * U29::NL * 4;
    if (lds > 64 * 1024) return false;
    return range<2>(m, a, grid, lds, s, std::make_integer_sequence<int, 15>{});
}
}  // namespace hbmpc
