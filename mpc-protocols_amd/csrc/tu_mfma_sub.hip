// k_mfma_rows<.., SUB>: the matrix-core decode whose sender values are differences formed after loading (kernels_mfma.hpp), m = 2 .. 11
#include <utility>

#include "kernels_mfma.hpp"
#include "launchers.hpp"
namespace hbmpc {
namespace {
// the raw operands of the next tile (2 m registers of 4) wait beside the current tile's m: fewer waves than the plain kernel
template <int M>
constexpr int sub_waves() { return M <= 6 ? 12 : 8; }
template <int M>
bool one(const mf::MfmaRowsArgs& a, int device, hipStream_t s) {
    constexpr int W = sub_waves<M>();
    const size_t lds = (size_t)mf::mf_max_role_rows(a) * (M * 1024 + 128);
    static std::atomic<bool> attr_set[HBMPC_MAX_DEVICES];
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&mf::k_mfma_rows<M, 1, W, 0, true>), attr_set, device, lds)) return false;
    hipLaunchKernelGGL((mf::k_mfma_rows<M, 1, W, 0, true>), dim3((unsigned)mf::mf_grid(a)), dim3(64 * W), lds, s, a);
    return true;
}
template <int LO, int... I>
bool range(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s, std::integer_sequence<int, I...>) {
    bool hit = false;
    ((m == LO + I ? (hit = one<LO + I>(a, device, s)) : false), ...);
    return hit;
}
}  // namespace
bool mfma_sub_covers(int m) { return m >= 2 && m <= 11; }
bool launch_mfma_rows_sub(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s) {
    return range<2>(m, a, device, s, std::make_integer_sequence<int, 10>{});
}
}  // namespace hbmpc
