// TripleGenNode for all parties of a small batch in one launch (kernels_triplegen_wg.hpp)
#include <hip/hip_runtime.h>

#include "fr_gold.hpp"
#include "fr_u29.hpp"
#include "kernels_triplegen_wg.hpp"
#include "launchers.hpp"

namespace hbmpc {
void launch_triplegen_wg(int impl, const TripleGenWgArgs& a, hipStream_t s) {
    const bool gold = impl == 2;
    const TripleGenWgLds L(a.n, a.t, gold ? 2 : 12, gold ? 2 : 9);
    if (gold) hipLaunchKernelGGL((k_triplegen_wg<Gold>), dim3((unsigned)a.G), dim3(256), (size_t)L.total * 4, s, a);
    else hipLaunchKernelGGL((k_triplegen_wg<U29>), dim3((unsigned)a.G), dim3(256), (size_t)L.total * 4, s, a);
}
}  // namespace hbmpc
