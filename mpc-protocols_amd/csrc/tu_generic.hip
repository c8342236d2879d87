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
void launch_eval_wide_dot(const uint32_t* x, size_t G, int n, int dp1, const uint32_t* vmat, EvalOut y, hipStream_t s) {
    const unsigned grid = (unsigned)((G + 3) / 4);
    int lk = 0;
    while (lk < 2 && (n << (lk + 1)) <= 64 && (2 << lk) <= dp1) ++lk;
    hipLaunchKernelGGL((k_eval_wide_dot<U29>), dim3(grid, y.parties), dim3(256), (size_t)n * dp1 * U29::NL * 4, s, x, G, n, dp1, vmat, y.y, y.ys ? y.ys : G, lk);
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
// LDS of one workgroup of k_batch_recover_wide; *tab_words = 0 when the call's table cannot be staged (not contiguous, or
// beyond what a launch may ask for without raising the function's limit)
static size_t wide_lds(int impl, const RecoverArgs& ra, bool p0, int ow_sel, int* tab_words, int* split) {
    const size_t ew = impl == 2 ? 2 : 8, nl = impl == 0 ? 9 : impl == 1 ? 8 : 2;
    const size_t nv = (size_t)(ra.needed - ra.m), ow = p0 ? 1 : ow_sel ? (size_t)ow_sel : (size_t)ra.m;
    const size_t front = 4 * (size_t)ra.needed * ew * 4;
    const size_t tw = (nv + ow) * (size_t)ra.m * nl;
    const bool fits = front + tw * 4 <= 64 * 1024;
    const bool contiguous = ra.bc == ra.vm + nv * (size_t)ra.m * nl && ((uintptr_t)ra.vm & 15) == 0;
    const bool staged = fits && (contiguous || p0);  // a single output row elsewhere in the table (one coefficient): staged from both ranges
    *split = staged && !contiguous ? 1 : 0;
    *tab_words = staged ? (int)tw : 0;
    return front + (staged ? tw * 4 : 0);
}
void launch_recover_wide(int impl, bool p0, const RecoverArgs& ra, const SecondArgs* sc, hipStream_t s, int ow_sel) {
    const unsigned grid = (unsigned)((ra.G + 3) / 4);
    WideArgs wa;
    wa.r = ra;
    wa.fused = sc != nullptr;
    if (sc) wa.sc = *sc;
    else memset(&wa.sc, 0, sizeof wa.sc);
    wa.ow = p0 ? 0 : ow_sel;
    const size_t lds = wide_lds(impl, ra, p0, ow_sel, &wa.tab_words, &wa.split);
    const bool tab = wa.tab_words != 0;
    wa.lk = 0;
    if (tab && impl == 0) {  // U29: a row's products shared by up to four lanes while every row still fits the wave (dot_shared)
        const int rows = (ra.needed - ra.m) + (p0 ? 1 : ow_sel ? ow_sel : ra.m);
        while (wa.lk < 2 && (rows << (wa.lk + 1)) <= 64 && (2 << wa.lk) <= ra.m) ++wa.lk;
    }
#define HBMPC_WIDE(F, P0) \
    do { \
        if (tab) hipLaunchKernelGGL((k_batch_recover_wide<F, P0, true>), dim3(grid), dim3(256), lds, s, wa); \
        else hipLaunchKernelGGL((k_batch_recover_wide<F, P0, false>), dim3(grid), dim3(256), lds, s, wa); \
    } while (0)
    if (impl == 0) {
        if (p0) HBMPC_WIDE(U29, true);
        else HBMPC_WIDE(U29, false);
    } else if (impl == 1) {
        if (p0) HBMPC_WIDE(Sat32, true);
        else HBMPC_WIDE(Sat32, false);
    } else {
        if (p0) HBMPC_WIDE(Gold, true);
        else HBMPC_WIDE(Gold, false);
    }
#undef HBMPC_WIDE
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
