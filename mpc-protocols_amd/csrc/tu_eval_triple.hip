// k_eval_fft1_triple: the shapes TripleGenNode runs (d = 2t, n >= 3t + 1, domains up to 16 points)
#include "dispatch_eval.hpp"
#include "launchers.hpp"
namespace hbmpc {
namespace {
template <int LOG, int CNT>
void one(const uint32_t* a, const uint32_t* b, const uint32_t* r2t, size_t G, int n, const uint32_t* tw, EvalOut y,
         const TripleConsts& cs, hipStream_t s) {
    const unsigned grid = (unsigned)((G + EVAL_TILE - 1) / EVAL_TILE);
    const size_t lds = (size_t)EVAL_TILE * (CNT * U29::EW + TILE_PAD<U29::EW>) * 4;
    hipLaunchKernelGGL((k_eval_fft1_triple<U29, LOG, CNT>), dim3(grid, y.parties), dim3(EVAL_TILE), lds, s, a, b, r2t, G, n, tw, y.y,
                       y.ys ? y.ys : G, cs);
}
}  // namespace
// (log2 of the domain size, d + 1): t = 1 .. 5 with the smallest domains that hold n = 3t + 1 .. 16 parties
bool launch_fft1_triple(int lg, int cnt, const uint32_t* a, const uint32_t* b, const uint32_t* r2t, size_t G, int n,
                        const uint32_t* tw, EvalOut y, const uint32_t r2[9], hipStream_t s) {
    TripleConsts cs;
    for (int i = 0; i < 9; ++i) cs.r2[i] = r2[i];
    if (lg == 2 && cnt == 3) return one<2, 3>(a, b, r2t, G, n, tw, y, cs, s), true;
    if (lg == 3 && cnt == 3) return one<3, 3>(a, b, r2t, G, n, tw, y, cs, s), true;
    if (lg == 3 && cnt == 5) return one<3, 5>(a, b, r2t, G, n, tw, y, cs, s), true;
    if (lg == 4 && cnt == 5) return one<4, 5>(a, b, r2t, G, n, tw, y, cs, s), true;
    if (lg == 4 && cnt == 7) return one<4, 7>(a, b, r2t, G, n, tw, y, cs, s), true;
    if (lg == 4 && cnt == 9) return one<4, 9>(a, b, r2t, G, n, tw, y, cs, s), true;
    if (lg == 4 && cnt == 11) return one<4, 11>(a, b, r2t, G, n, tw, y, cs, s), true;
    return false;
}
}  // namespace hbmpc
