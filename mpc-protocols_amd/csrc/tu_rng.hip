#include "kernels_rng.hpp"
#include "launchers.hpp"
namespace hbmpc {
void launch_fill_coeffs(int ew, const uint32_t seed[8], const uint32_t* secrets, size_t B, uint64_t first_index, int dp1,
                        uint32_t* coeffs, hipStream_t s) {
    SeedArg k;
    for (int i = 0; i < 8; ++i) k.k[i] = seed[i];
    const size_t total = B * (size_t)dp1;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (ew == 8) hipLaunchKernelGGL((k_fill_coeffs<8>), dim3(grid), dim3(256), 0, s, k, secrets, B, first_index, dp1, coeffs);
    else hipLaunchKernelGGL((k_fill_coeffs<2>), dim3(grid), dim3(256), 0, s, k, secrets, B, first_index, dp1, coeffs);
}
}
