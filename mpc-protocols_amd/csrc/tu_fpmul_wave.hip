// FPMulNode for all parties of a small batch in one launch (kernels_fpmul_wave.hpp)
#include <hip/hip_runtime.h>

#include "fr_u29.hpp"
#include "kernels_fpmul_wave.hpp"
#include "launchers.hpp"

namespace hbmpc {
bool launch_fpmul_wave(const FpmulWaveArgs& a, int device, hipStream_t s, bool dry_run) {
    const int nv = a.needed - a.M;
    const FpmulWaveLds L(a.needed, a.parties, a.m, (nv + 2) * a.M * 9);
    const size_t lds = (size_t)L.total * 4;
    if (lds > 160 * 1024) return false;
    static std::atomic<bool> attr_set[HBMPC_MAX_DEVICES];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&k_fpmul_wave<U29>), attr_set, device, lds)) return false;
    if (!dry_run) hipLaunchKernelGGL((k_fpmul_wave<U29>), dim3((unsigned)((a.N + 3) / 4)), dim3(256), lds, s, a);
    return true;
}
}  // namespace hbmpc
