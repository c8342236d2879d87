// FPMulNode for all parties of a small batch in one launch (kernels_fpmul_wave.hpp)
#include <hip/hip_runtime.h>

#include "fr_u29.hpp"
#include "kernels_fpmul_wave.hpp"
#include "launchers.hpp"

namespace hbmpc {
bool launch_fpmul_wave(const FpmulWaveArgs& a, hipStream_t s, bool dry_run) {
    const int nv = a.needed - a.M;
    const FpmulWaveLds L(a.needed, a.parties, a.m, (nv + 2) * a.M * 9);
    if ((size_t)L.total * 4 > 64 * 1024) return false;
    if (!dry_run) hipLaunchKernelGGL((k_fpmul_wave<U29>), dim3((unsigned)((a.N + 3) / 4)), dim3(256), (size_t)L.total * 4, s, a);
    return true;
}
}  // namespace hbmpc
