#include <cstring>
#include "fr_gold.hpp"
#include "kernels_eval.hpp"
#include "launchers.hpp"
namespace hbmpc {
void launch_eval_generic(int impl, const uint32_t* x, size_t G, int n, int dp1, const uint32_t* alpha, EvalOut y,
                         hipStream_t s) {
    const unsigned grid = (unsigned)((G + 255) / 256);
    if (impl == 0) hipLaunchKernelGGL((k_eval_generic<U29>), dim3(grid, y.parties), dim3(256), 0, s, x, G, n, dp1, alpha, y.y, y.ys ? y.ys : G);
    else if (impl == 1) hipLaunchKernelGGL((k_eval_generic<Sat32>), dim3(grid, y.parties), dim3(256), 0, s, x, G, n, dp1, alpha, y.y, y.ys ? y.ys : G);
    else hipLaunchKernelGGL((k_eval_generic<Gold>), dim3(grid, y.parties), dim3(256), 0, s, x, G, n, dp1, alpha, y.y, y.ys ? y.ys : G);
}
void launch_eval_wide(int impl, const uint32_t* x, size_t G, int n, int dp1, const uint32_t* alpha, EvalOut y, hipStream_t s) {
    const unsigned grid = (unsigned)((G + 3) / 4);
    if (impl == 0) hipLaunchKernelGGL((k_eval_wide<U29>), dim3(grid, y.parties), dim3(256), 0, s, x, G, n, dp1, alpha, y.y, y.ys ? y.ys : G);
    else if (impl == 1) hipLaunchKernelGGL((k_eval_wide<Sat32>), dim3(grid, y.parties), dim3(256), 0, s, x, G, n, dp1, alpha, y.y, y.ys ? y.ys : G);
    else hipLaunchKernelGGL((k_eval_wide<Gold>), dim3(grid, y.parties), dim3(256), 0, s, x, G, n, dp1, alpha, y.y, y.ys ? y.ys : G);
}
void launch_recover_generic(int impl, bool p0, const RecoverArgs& ra, unsigned grid, hipStream_t s) {
    if (impl == 0) {
        if (p0) hipLaunchKernelGGL((k_batch_recover_generic<U29, true>), dim3(grid), dim3(256), 0, s, ra);
        else hipLaunchKernelGGL((k_batch_recover_generic<U29, false>), dim3(grid), dim3(256), 0, s, ra);
    } else if (impl == 1) {
        if (p0) hipLaunchKernelGGL((k_batch_recover_generic<Sat32, true>), dim3(grid), dim3(256), 0, s, ra);
        else hipLaunchKernelGGL((k_batch_recover_generic<Sat32, false>), dim3(grid), dim3(256), 0, s, ra);
    } else {
        if (p0) hipLaunchKernelGGL((k_batch_recover_generic<Gold, true>), dim3(grid), dim3(256), 0, s, ra);
        else hipLaunchKernelGGL((k_batch_recover_generic<Gold, false>), dim3(grid), dim3(256), 0, s, ra);
    }
}
void launch_recover_wide(int impl, bool p0, const RecoverArgs& ra, const SecondArgs* sc, hipStream_t s) {
    const unsigned grid = (unsigned)((ra.G + 3) / 4);
    const size_t ew = impl == 2 ? 2 : 8, lds = 4 * (size_t)ra.needed * ew * 4;  // one chunk's sender values per wave
    WideArgs wa;
    wa.r = ra;
    wa.fused = sc != nullptr;
    if (sc) wa.sc = *sc;
    else memset(&wa.sc, 0, sizeof wa.sc);
    if (impl == 0) {
        if (p0) hipLaunchKernelGGL((k_batch_recover_wide<U29, true>), dim3(grid), dim3(256), lds, s, wa);
        else hipLaunchKernelGGL((k_batch_recover_wide<U29, false>), dim3(grid), dim3(256), lds, s, wa);
    } else if (impl == 1) {
        if (p0) hipLaunchKernelGGL((k_batch_recover_wide<Sat32, true>), dim3(grid), dim3(256), lds, s, wa);
        else hipLaunchKernelGGL((k_batch_recover_wide<Sat32, false>), dim3(grid), dim3(256), lds, s, wa);
    } else {
        if (p0) hipLaunchKernelGGL((k_batch_recover_wide<Gold, true>), dim3(grid), dim3(256), lds, s, wa);
        else hipLaunchKernelGGL((k_batch_recover_wide<Gold, false>), dim3(grid), dim3(256), lds, s, wa);
    }
}
void launch_second_chance(int impl, const SecondArgs& a, unsigned grid, hipStream_t s) {
    if (impl == 0) hipLaunchKernelGGL((k_second_chance<U29>), dim3(grid), dim3(256), 0, s, a);
    else if (impl == 1) hipLaunchKernelGGL((k_second_chance<Sat32>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_second_chance<Gold>), dim3(grid), dim3(256), 0, s, a);
}
void launch_matvec(int impl, const uint32_t* lb, const uint32_t* y, int S, uint32_t* out, hipStream_t s) {
    const unsigned grid = (unsigned)((S + 255) / 256);
    if (impl == 0) hipLaunchKernelGGL((k_matvec<U29>), dim3(grid), dim3(256), 0, s, lb, y, S, out);
    else if (impl == 1) hipLaunchKernelGGL((k_matvec<Sat32>), dim3(grid), dim3(256), 0, s, lb, y, S, out);
    else hipLaunchKernelGGL((k_matvec<Gold>), dim3(grid), dim3(256), 0, s, lb, y, S, out);
}
}
