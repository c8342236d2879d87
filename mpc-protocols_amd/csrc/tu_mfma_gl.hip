// tu_mfma_gl.hip -- instantiations of the Goldilocks matrix-core kernel (kernels_mfma_gl.hpp) for 1..4 K-steps
#include "kernels_mfma_gl.hpp"
#include "launchers.hpp"
namespace hbmpc {
namespace {
template <int KS, bool ENCODE>
bool launch2(const mf::MfmaGlArgs& a, unsigned grid, size_t lds, int device, hipStream_t s) {
    static std::atomic<bool> attr_set[HBMPC_MAX_DEVICES];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&mf::k_mfma_rows_gl<KS, ENCODE>), attr_set, device, lds)) return false;
    hipLaunchKernelGGL((mf::k_mfma_rows_gl<KS, ENCODE>), dim3(grid), dim3(256), lds, s, a);
    return true;
}
template <int KS>
bool launch(const mf::MfmaGlArgs& a, unsigned grid, size_t lds, int device, hipStream_t s) {
    // encode: chunk-major input, party-major output, no verify rows; decode: sender rows in, chunk-major coefficients out
    return a.in_chunk_major ? launch2<KS, true>(a, grid, lds, device, s) : launch2<KS, false>(a, grid, lds, device, s);
}
}  // namespace
bool launch_mfma_rows_gl(const mf::MfmaGlArgs& a, unsigned grid, int device, hipStream_t s) {
    const int ks = (8 * a.m + 31) / 32, nt = (a.nrows + 3) / 4;
    const size_t lds = (size_t)nt * (ks * 1024 + 128) + 256 * 8;  // table + the row offsets
    if (lds > 160 * 1024) return false;
    switch (ks) {
        case 1: return launch<1>(a, grid, lds, device, s);
        case 2: return launch<2>(a, grid, lds, device, s);
        case 3: return launch<3>(a, grid, lds, device, s);
        case 4: return launch<4>(a, grid, lds, device, s);
    }
    return false;
}
}  // namespace hbmpc
