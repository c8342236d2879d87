// TripleGenNode for all parties of a small batch in one launch (kernels_triplegen_wg.hpp)
#include <hip/hip_runtime.h>

#include "fr_u29.hpp"
#include "kernels_triplegen_wg.hpp"
#include "launchers.hpp"

namespace hbmpc {
void launch_triplegen_wg(const TripleGenWgArgs& a, hipStream_t s) {
    const TripleGenWgLds L(a.n, a.t);
    hipLaunchKernelGGL((k_triplegen_wg<U29>), dim3((unsigned)a.G), dim3(256), (size_t)L.total * 4, s, a);
}
}  // namespace hbmpc
