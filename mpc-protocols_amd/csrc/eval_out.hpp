// eval_out.hpp -- where an evaluation launch writes (shared by the launcher declarations and the dispatch templates)
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace hbmpc {

// where a launch writes: output row j of party p at y + ((p * n + j) * ys + g) elements
struct EvalOut {
    uint32_t* y;
    size_t ys;         // elements between consecutive output rows (0: dense, = G)
    unsigned parties;  // independent [G][d+1] -> [n][G] problems of the launch (blockIdx.y)
};

// where the producers' mixing step writes (kernels_eval.hpp: k_eval_fft1_mix)
struct MixOut {
    uint32_t *y, *others;  // y[row][G] (others == nullptr) or the party-major block
    size_t K;
    int row0, rows;
    struct Slice {
        uint32_t* dst;
        size_t stride;  // elements between the lists of consecutive parties
        size_t k0, count;
    } list[2];
};

}  // namespace hbmpc
