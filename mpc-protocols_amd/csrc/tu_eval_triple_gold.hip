// k_eval_fft1_triple over Goldilocks: the small-field TripleGenNode (honeybadger/mod.rs:316-324) shapes, d = 2t,
// domains up to 16 points
#include "fr_gold.hpp"
#include "dispatch_eval.hpp"
#include "launchers.hpp"
namespace hbmpc {
namespace {
template <int LOG, int CNT>
void one(const uint32_t* a, const uint32_t* b, const uint32_t* r2t, size_t G, int n, const uint32_t* tw, EvalOut y,
         const TripleConsts& cs, hipStream_t s) {
    const unsigned grid = (unsigned)((G + EVAL_TILE - 1) / EVAL_TILE);
    const size_t lds = (size_t)EVAL_TILE * (CNT * Gold::EW + TILE_PAD<Gold::EW>) * 4;
    hipLaunchKernelGGL((k_eval_fft1_triple<Gold, LOG, CNT>), dim3(grid, y.parties), dim3(EVAL_TILE), lds, s, a, b, r2t, G, n, tw, y.y,
                       y.ys ? y.ys : G, cs);
}
}  // namespace
bool launch_fft1_triple_gold(int lg, int cnt, const uint32_t* a, const uint32_t* b, const uint32_t* r2t, size_t G, int n,
                             const uint32_t* tw, EvalOut y, hipStream_t s) {
    TripleConsts cs = {};
    cs.r2[0] = 1;  // no Montgomery form: mulc(x, 1) = x
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
