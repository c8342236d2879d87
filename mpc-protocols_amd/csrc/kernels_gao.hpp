// kernels_gao.hpp -- the Byzantine fallback on the device: oec_decode + gao_rs_decode
// (reference row a8, honeybadger/robust_interpolate/robust_interpolate.rs:456-628), run for the
// chunks the optimistic verify flagged (robust_interpolate.rs:433-438).
//
// One workgroup per flagged chunk, one lane per polynomial coefficient, polynomials in LDS.
// What depends only on positions is shared by all chunks and prepared on the host per OEC round
// r (required = d+t+1+r lowest sender ids are "known", every other position is an erasure):
//   g0_r  = prod_{i known} (x - alpha_i)            (= prod_{i<n}(x-alpha_i) / s(x), :474-495)
//   LB_r  = Lagrange basis of the known points       (g1 = sum_j y_j LB_r[.][j], :483-491)
// Per chunk: g1 by `required` dot products (one lane each), then the extended Euclidean algorithm
// on (g0, g1) until deg < (required + k)/2 (:498-522), f = g / v (:527-537), and the acceptance
// count (:613-622).  The EEA is run FRACTION-FREE (each elimination step is
// lead(r1)*r0 - lead(r0)*x^s*r1, applied to the t-sequence too): (g, v) come out scaled by a common
// nonzero factor, which changes neither degrees, nor g/v, nor whether the remainder is zero -- so
// the only field inversion per chunk is the one that un-scales an ACCEPTED quotient (the division itself is
// fraction-free too, so rejected rounds never invert).
// All LDS-resident coefficients are kept canonical (of Montgomery-form values), so "is zero" and
// degrees are plain limb tests.
#pragma once
#include <algorithm>
#include "../../include/hbmpc_hip.h"
#include "fr_sat.hpp"
#include "fr_u29.hpp"
#include "kernels_recover.hpp"  // RowsArg

namespace hbmpc {

struct GaoRound {
    int required;        // number of known points (lowest `required` sorted sender ids)
    int threshold;       // (required + k) / 2
    uint32_t g0_off;     // offset (in u32) into `tables`: g0 coefficients [required+1], Montgomery limb form
    uint32_t lb_off;     // LB[k][j] * R in device-constant form, [required][required]
};
struct GaoArgs {
    const uint32_t* evals;     // sender rows, canonical (row s at evals + rows[s] * row_stride * 8 words)
    size_t G;
    size_t row_stride;
    RowsArg rows;              // rows[s] = position of the s-th lowest sender id (kernels_recover.hpp)
    const uint32_t* alpha_s;   // [S] alpha of the s-th lowest sender id, device-constant form
    const GaoRound* rounds;    // [n_rounds]
    int n_rounds;
    const uint32_t* tables;
    int k;                     // message length d+1
    int accept_min;            // d+t+1 (0: standalone gao_rs_decode, no acceptance count)
    int out_width;             // coefficients written per chunk (d+1, or 1 for P(0)-only)
    const uint32_t* flagged;   // compact list (null: chunk = block index)
    const uint32_t* counters;  // [0] = number of flagged chunks (null: G)
    uint32_t* out;             // [G][out_width]
    uint32_t* ncoeffs;         // [G] or null
    uint8_t* status;           // [G] or null
    uint32_t* summary;         // hbmpc_recover_summary or null: {n_fallback, n_failed, first_failed, first_error}
    const uint32_t* one_plain; // limbs of the integer 1 (mont(x, 1) leaves Montgomery form)
    const uint32_t* r2;        // device-constant form of R (mont(x, r2) = x*R)
    const uint32_t* inv_exp;   // r - 2 as 8 x u32
    uint32_t* reset;           // batch path: the call's four counters ([3] = ticket); the last working block of k_unscale
                               // zeroes them for the next call
    uint32_t* scales;          // [count][NL + 1]: word 0 = 1 when the chunk's output still carries the factor l^N held in
                               // words 1..NL (Montgomery form); k_unscale divides it out with a batched inversion
};

template <int BLOCK>
HB_DEV int block_max(int v, int* scratch) {  // max over the block of a per-lane int (>= -1)
    if constexpr (BLOCK == 64) {
        (void)scratch;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int w = __shfl_xor(v, o);
            v = w > v ? w : v;
        }
        return v;
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int w = __shfl_xor(v, o);
            v = w > v ? w : v;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
        __syncthreads();
        int m = scratch[0];
#pragma unroll
        for (int i = 1; i < BLOCK / 64; ++i) m = scratch[i] > m ? scratch[i] : m;
        return m;
    }
}

template <class F>
HB_DEV void lds_put(uint32_t* p, const typename F::E& x) {
#pragma unroll
    for (int i = 0; i < F::NL; ++i) p[i] = x.l[i];
}
template <class F>
HB_DEV typename F::E lds_get(const uint32_t* p) {
    typename F::E x;
#pragma unroll
    for (int i = 0; i < F::NL; ++i) x.l[i] = p[i];
    return x;
}

// Fermat inverse of a canonical Montgomery-form value a (a*R): returns a^-1 * R, canonical.
template <class F>
__device__ __noinline__ typename F::E fr_inverse_mont(typename F::E a, const uint32_t* __restrict__ exp_words,
                                                      const uint32_t* __restrict__ r2, const uint32_t* __restrict__ one_plain) {
    using E = typename F::E;
    E one;  // 1 in Montgomery form = mont(1_plain as data, r2) -> use load_const(one_plain) as data
    one = F::cond_sub_r(F::mulc(F::load_const(one_plain), r2));
    E acc = one;
    for (int i = 255; i >= 0; --i) {
        acc = F::cond_sub_r(F::mont(acc, acc));
        if ((exp_words[i >> 5] >> (i & 31)) & 1) acc = F::cond_sub_r(F::mont(acc, a));
    }
    return acc;
}

// A GROUP of SUB lanes decodes one chunk (lane = coefficient index, so SUB > n); a workgroup holds BLOCK / SUB
// groups.  For n < 32 a 64-lane wave would otherwise run a quarter or half empty, so it carries 4 or 2
// independent chunks: the groups of a wave diverge freely (their EEA / division trip counts differ), every
// exchange is inside a group, and group synchronisation inside one wave is a fence -- LDS operations of a wave
// execute in order.  With SUB == BLOCK (n >= 32) a group is the whole workgroup and the barriers are real.
constexpr size_t gao_group_words(int sub, int nl) { return (size_t)(4 * sub + 4) * nl + 16; }
template <int BLOCK, int SUB>
HB_DEV void group_sync() {
    if constexpr (SUB == BLOCK) {
        __syncthreads();
    } else {
        static_assert(BLOCK == 64, "several groups only within one wave");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}
template <int BLOCK, int SUB>
HB_DEV int group_max(int v, int* scratch) {
    if constexpr (SUB == BLOCK) {
        return block_max<BLOCK>(v, scratch);
    } else {
        (void)scratch;
#pragma unroll
        for (int o = SUB / 2; o > 0; o >>= 1) {
            const int w = __shfl_xor(v, o);
            v = w > v ? w : v;
        }
        return v;
    }
}
// Groups stride over the flagged list.
// INLINE (small batches): the accepted quotient is un-scaled right here -- one inversion per chunk instead of one per
// eight, which only matters for long lists -- and this kernel is the last of the call: no k_unscale launch.
template <class F, int BLOCK, int SUB, bool INLINE = false>
__global__ __launch_bounds__(BLOCK) void k_gao(GaoArgs a) {
    using E = typename F::E;
    constexpr int NL = F::NL, NSUB = BLOCK / SUB;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    const int tid = threadIdx.x % SUB, sub = threadIdx.x / SUB;
    uint32_t* lds = lds_all + (size_t)sub * gao_group_words(SUB, NL);
    // polynomials: SUB coefficients each
    uint32_t* R0 = lds;
    uint32_t* R1 = R0 + SUB * NL;
    uint32_t* T0 = R1 + SUB * NL;
    uint32_t* T1 = T0 + SUB * NL;
    uint32_t* BC = T1 + SUB * NL;             // broadcast slots: 4 elements
    int* iscr = reinterpret_cast<int*>(BC + 4 * NL);  // small int scratch (16 ints)
    const size_t count = a.counters ? handoff_count(a.counters, a.G) : a.G;

    for (size_t fi = (size_t)blockIdx.x * NSUB + sub; fi < count; fi += (size_t)gridDim.x * NSUB) {
        const size_t g = a.flagged ? (size_t)a.flagged[fi] : fi;
        int result = DecodingError;  // oec_decode's error when no round succeeds (:625)
        int out_len = 0;
        bool done = false;
        for (int rd = 0; rd < a.n_rounds && !done; ++rd) {
            const GaoRound R = a.rounds[rd];
            const int req = R.required;
            group_sync<BLOCK, SUB>();
            // ---- r0 = g0, r1 = g1 = sum_j y_j LB[.][j], t0 = 0, t1 = 1 (Montgomery forms) --------
            {
                E v0 = F::zero(), v1 = F::zero();
                if (tid <= req) v0 = F::load_const(a.tables + R.g0_off + (size_t)tid * NL);
                if (tid < req) {
                    typename F::Acc acc;
                    F::acc_zero(acc);
                    const uint32_t* row = a.tables + R.lb_off + (size_t)tid * req * NL;
                    int pending = 0;
                    for (int j = 0; j < req; ++j) {
                        if (pending == F::MAX_DOT_TERMS) {
                            F::acc_fold(acc);
                            pending = 1;
                        }
                        F::acc_mac(acc, F::load(a.evals + ((size_t)a.rows[j] * a.row_stride + g) * F::EW), row + (size_t)j * NL);
                        ++pending;
                    }
                    F::acc_fold(acc);
                    v1 = F::canon_loose(F::acc_reduce(acc));
                }
                lds_put<F>(R0 + tid * NL, v0);
                lds_put<F>(R1 + tid * NL, v1);
                lds_put<F>(T0 + tid * NL, F::zero());
                E one = F::zero();
                if (tid == 0) one = F::cond_sub_r(F::mulc(F::load_const(a.one_plain), a.r2));
                lds_put<F>(T1 + tid * NL, one);
            }
            group_sync<BLOCK, SUB>();
            uint32_t *r0 = R0, *r1 = R1, *t0 = T0, *t1 = T1;
            int d0 = req;  // deg g0
            int d1 = group_max<BLOCK, SUB>(F::is_zero_canon(lds_get<F>(r1 + tid * NL)) ? -1 : tid, iscr);
            // ---- EEA: while r1.degree() >= threshold (degree of the zero polynomial is 0) ---------
            while ((d1 < 0 ? 0 : d1) >= R.threshold) {
                // r0 <- r0 mod r1 (fraction-free), t0 <- t0 - q t1 (same combination); then swap
                while (d0 >= d1) {
                    const int sh = d0 - d1;
                    const E lam = lds_get<F>(r1 + d1 * NL);  // lead(r1)
                    const E mu = lds_get<F>(r0 + d0 * NL);   // lead(r0)
                    E nr = F::mont(lds_get<F>(r0 + tid * NL), lam);
                    E nt = F::mont(lds_get<F>(t0 + tid * NL), lam);
                    if (tid >= sh) {
                        nr = F::template sub<4>(nr, F::mont(lds_get<F>(r1 + (tid - sh) * NL), mu));
                        nt = F::template sub<4>(nt, F::mont(lds_get<F>(t1 + (tid - sh) * NL), mu));
                    }
                    nr = F::canon_loose(nr);
                    nt = F::canon_loose(nt);
                    group_sync<BLOCK, SUB>();
                    lds_put<F>(r0 + tid * NL, nr);
                    lds_put<F>(t0 + tid * NL, nt);
                    group_sync<BLOCK, SUB>();
                    d0 = group_max<BLOCK, SUB>(F::is_zero_canon(nr) ? -1 : tid, iscr);
                }
                uint32_t* tmp = r0;
                r0 = r1;
                r1 = tmp;
                tmp = t0;
                t0 = t1;
                t1 = tmp;
                const int td = d0;
                d0 = d1;
                d1 = td;
            }
            // ---- f = g / v with g = r1, v = t1 (:524-537) ------------------------------------------
            const int dg = d1;
            const int dv = group_max<BLOCK, SUB>(F::is_zero_canon(lds_get<F>(t1 + tid * NL)) ? -1 : tid, iscr);
            bool ok = true;
            bool scaled = false;  // fq holds l^N * quotient, BC[1] = l^N (Montgomery form)
            int df = -1;  // degree of the quotient (-1: zero polynomial)
            uint32_t* fq = t0;  // quotient coefficients are written over t0 (no longer needed)
            if (dg < 0) {
                // g == 0: quotient and remainder are zero -> the zero polynomial (degree() = 0 < k)
                group_sync<BLOCK, SUB>();
                lds_put<F>(fq + tid * NL, F::zero());
                group_sync<BLOCK, SUB>();
            } else if (dg < dv) {
                ok = false;  // quotient 0, remainder g != 0
            } else {
                // Fraction-free long division first: l^N g = Q' v + R' with l = lead(v), N = dg - dv + 1
                // (each step: R <- l R - lead(R) x^sh v, Q' <- l Q' + lead(R) x^sh).  Whether the remainder is
                // zero and the degree of the quotient do not depend on the scaling, so a round that fails here
                // (the usual fate of an OEC round with too few points for the errors present) costs NO field
                // inversion; only an accepted division pays for one (a single-lane Fermat chain, ~100k
                // instructions -- it used to dominate the fallback when several rounds were needed).
                const E l = lds_get<F>(t1 + dv * NL);
                group_sync<BLOCK, SUB>();
                lds_put<F>(fq + tid * NL, F::zero());
                group_sync<BLOCK, SUB>();
                for (int top = dg; top >= dv; --top) {
                    const E lead = lds_get<F>(r1 + top * NL);
                    const int sh = top - dv;
                    E nr = F::mont(lds_get<F>(r1 + tid * NL), l);
                    if (tid >= sh && tid <= top) nr = F::template sub<4>(nr, F::mont(lds_get<F>(t1 + (tid - sh) * NL), lead));
                    nr = F::canon_loose(nr);
                    E nq = F::mont(lds_get<F>(fq + tid * NL), l);
                    if (tid == sh) nq = F::add(nq, lead);
                    nq = F::canon_loose(nq);
                    group_sync<BLOCK, SUB>();
                    lds_put<F>(r1 + tid * NL, nr);
                    lds_put<F>(fq + tid * NL, nq);
                    group_sync<BLOCK, SUB>();
                }
                const int drem = group_max<BLOCK, SUB>(F::is_zero_canon(lds_get<F>(r1 + tid * NL)) ? -1 : tid, iscr);
                if (drem >= 0) ok = false;  // remainder must be zero
                df = group_max<BLOCK, SUB>(F::is_zero_canon(lds_get<F>(fq + tid * NL)) ? -1 : tid, iscr);
                if ((df < 0 ? 0 : df) >= a.k) ok = false;  // quotient.degree() < k
                if (ok) {  // l^N, needed by the acceptance count and by the final un-scaling
                    if (tid == 0) {
                        E sc = l;
                        for (int i = dv; i < dg; ++i) sc = F::cond_sub_r(F::mont(sc, l));
                        lds_put<F>(BC + NL, sc);
                    }
                    scaled = true;
                    group_sync<BLOCK, SUB>();
                }
            }
            if (ok && a.accept_min > 0) {
                // matched = #{known j : f(alpha_j) == y_j} >= d+t+1 (:613-620)
                int hit = 0;
                if (tid < req) {
                    const uint32_t* al = a.alpha_s + (size_t)tid * NL;
                    E acc = F::zero();
                    for (int kx = df; kx >= 0; --kx) {
                        acc = F::mont(acc, al);
                        acc = F::add(acc, lds_get<F>(fq + kx * NL));
                    }
                    const E val = F::cond_sub_r(F::mont(F::canon_loose(acc), a.one_plain));  // leave Montgomery form
                    E ys = F::load(a.evals + ((size_t)a.rows[tid] * a.row_stride + g) * F::EW);
                    // compare l^N f(alpha_j) with l^N y_j: still no inversion for a round that is not accepted
                    if (scaled) ys = F::cond_sub_r(F::mont(ys, lds_get<F>(BC + NL)));
                    hit = F::eq_canon(val, ys) ? 1 : 0;
                }
                // block-wide sum via max of prefix counts is overkill: use LDS atomics
                group_sync<BLOCK, SUB>();
                if (tid == 0) iscr[8] = 0;
                group_sync<BLOCK, SUB>();
                if (hit) atomicAdd(&iscr[8], 1);
                group_sync<BLOCK, SUB>();
                if (iscr[8] < a.accept_min) ok = false;
                group_sync<BLOCK, SUB>();
            }
            if (ok) {
                // Write the coefficients, zero padded to out_width.  An accepted division leaves Q' = l^N Q: the
                // Montgomery-form residues of Q' go out as they are together with l^N, and k_unscale finishes
                // them -- ONE Fermat inversion per eight chunks there (Montgomery's batching trick) instead of
                // one single-lane, ~100k-instruction inversion per chunk here, which dominated the fallback.
                if constexpr (INLINE) {
                    if (scaled) {  // BC + NL holds l^N (Montgomery form): replace it by its inverse, once per chunk
                        group_sync<BLOCK, SUB>();
                        if (tid == 0) lds_put<F>(BC + NL, fr_inverse_mont<F>(lds_get<F>(BC + NL), a.inv_exp, a.r2, a.one_plain));
                        group_sync<BLOCK, SUB>();
                    }
                }
                if (tid < a.out_width) {
                    E c = F::zero();
                    if (tid <= df) {
                        c = lds_get<F>(fq + tid * NL);
                        if (INLINE && scaled) c = F::cond_sub_r(F::mont(c, lds_get<F>(BC + NL)));  // Q'_k / l^N
                        if (INLINE || !scaled) c = F::cond_sub_r(F::mont(c, a.one_plain));          // leave Montgomery form
                    }
                    F::store_lt2r(a.out + (g * (size_t)a.out_width + tid) * F::EW, c);
                }
                if (!INLINE && tid == 0) {
                    uint32_t* sc = a.scales + fi * (size_t)(NL + 1);
                    sc[0] = scaled ? 1u : 0u;
                    if (scaled) lds_put<F>(sc + 1, lds_get<F>(BC + NL));
                }
                out_len = df + 1;
                result = ShareSuccess;
                done = true;
            }
        }
        if (result != ShareSuccess && tid < a.out_width) {
            F::store_lt2r(a.out + (g * (size_t)a.out_width + tid) * F::EW, F::zero());
        }
        if (tid == 0) {
            if (!INLINE && result != ShareSuccess) a.scales[fi * (size_t)(NL + 1)] = 0u;
            if (a.ncoeffs) a.ncoeffs[g] = result == ShareSuccess ? (uint32_t)out_len : 0u;
            if (a.status) a.status[g] = result == ShareSuccess ? 1 : (uint8_t)result;
            if (a.summary) {
                atomicAdd(&a.summary[0], 1u);
                if (result != ShareSuccess) {
                    atomicAdd(&a.summary[1], 1u);
                    atomicMin(&a.summary[2], (uint32_t)g);
                    a.summary[3] = (uint32_t)result;
                }
            }
        }
        group_sync<BLOCK, SUB>();
    }
    if constexpr (INLINE) {
        // last kernel of the call: the last block that owns list entries leaves the counters at zero (see k_unscale)
        __syncthreads();
        if (a.reset && threadIdx.x == 0) {
            const unsigned quorum = count ? (unsigned)std::min<size_t>((count + NSUB - 1) / NSUB, gridDim.x) : 1u;
            if (blockIdx.x < quorum && atomicAdd(&a.reset[3], 1u) == quorum - 1) {
                const uint32_t tally = load_handoff(&a.reset[2]);
                if (a.summary && tally) atomicAdd(&a.summary[0], tally);
#pragma unroll
                for (int k = 0; k < 4; ++k) store_handoff(&a.reset[k], 0u);
            }
        }
    }
}

// Second half of the fallback: divide the factor l^N out of the chunks k_gao marked.  One lane takes eight
// consecutive entries of the flagged list, multiplies their factors together, inverts the product once
// (Fermat) and peels the individual inverses off again (Montgomery's trick: 3 multiplications per entry).
template <class F>
HB_DEV void unscale_lane(const GaoArgs& a) {
    using E = typename F::E;
    constexpr int NL = F::NL, B = 8;
    const size_t count = a.counters ? handoff_count(a.counters, a.G) : a.G;
    const size_t f0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * B;
    if (f0 >= count) return;
    E one = F::cond_sub_r(F::mulc(F::load_const(a.one_plain), a.r2));  // 1 in Montgomery form
    E sc[B], pre[B];
    bool pend[B];
    E run = one;
#pragma unroll
    for (int i = 0; i < B; ++i) {
        const size_t fi = f0 + i;
        pend[i] = fi < count && a.scales[fi * (size_t)(NL + 1)] != 0u;
        sc[i] = one;
        if (pend[i]) sc[i] = lds_get<F>(a.scales + fi * (size_t)(NL + 1) + 1);
        pre[i] = run;                                   // product of the factors before entry i
        run = F::cond_sub_r(F::mont(run, sc[i]));
    }
    bool any = false;
#pragma unroll
    for (int i = 0; i < B; ++i) any = any || pend[i];
    if (!any) return;
    E inv = fr_inverse_mont<F>(run, a.inv_exp, a.r2, a.one_plain);  // (prod sc)^-1
#pragma unroll
    for (int i = B - 1; i >= 0; --i) {
        const E inv_i = F::cond_sub_r(F::mont(inv, pre[i]));        // sc_i^-1
        inv = F::cond_sub_r(F::mont(inv, sc[i]));
        if (!pend[i]) continue;
        const size_t fi = f0 + i;
        const size_t g = a.flagged ? (size_t)a.flagged[fi] : fi;
        for (int k = 0; k < a.out_width; ++k) {
            uint32_t* p = a.out + (g * (size_t)a.out_width + k) * F::EW;
            const E q = F::cond_sub_r(F::mont(F::load(p), inv_i));             // Q'_k / l^N, Montgomery form
            F::store_lt2r(p, F::cond_sub_r(F::mont(q, a.one_plain)));          // leave Montgomery form
        }
    }
}
template <class F>
__global__ __launch_bounds__(64) void k_unscale(GaoArgs a) {
    unscale_lane<F>(a);
    // Last kernel of a batch_recover call: the block that finishes last leaves the call's counters at zero, which is
    // what the first kernel of the NEXT call on this stream expects (there is no init launch).  Every lane of a block
    // -- one wave -- has read its count in the first instruction of unscale_lane, before this lane gets here.
    if (a.reset && threadIdx.x == 0) {
        // Only the blocks that own entries of the list (and block 0, so that the quorum is never empty) take a ticket:
        // an idle block has nothing the reset could disturb -- if it runs late and reads a count that is already
        // zero it computes quorum 1 and is still idle.  No fence: nothing this block wrote needs publishing before
        // the ticket, and a working block's reads of the count precede its ticket in program order.
        const size_t count = a.counters ? handoff_count(a.counters, a.G) : a.G;
        const size_t per_block = (size_t)blockDim.x * 8;
        const unsigned quorum = count ? (unsigned)((count + per_block - 1) / per_block) : 1u;
        if (blockIdx.x < quorum && atomicAdd(&a.reset[3], 1u) == quorum - 1) {
            // chunks the fused small-batch kernel repaired were tallied in reset[2] (the summary was still being
            // initialised then); every other writer of the summary finished with the previous kernel
            const uint32_t tally = load_handoff(&a.reset[2]);
            if (a.summary && tally) atomicAdd(&a.summary[0], tally);
#pragma unroll
            for (int k = 0; k < 4; ++k) store_handoff(&a.reset[k], 0u);
        }
    }
}

// coefficients = LB * y for ONE instance (NonRobustShare::recover_secret: plain Lagrange through all
// supplied shares, common/share/shamir.rs:199-239): lane k computes coefficient k.
template <class F>
__global__ __launch_bounds__(256) void k_matvec(const uint32_t* __restrict__ lb, const uint32_t* __restrict__ y,
                                                int S, uint32_t* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S) return;
    typename F::Acc acc;
    F::acc_zero(acc);
    int pending = 0;
    for (int j = 0; j < S; ++j) {
        if (pending == F::MAX_DOT_TERMS) {
            F::acc_fold(acc);
            pending = 1;
        }
        F::acc_mac(acc, F::load(y + (size_t)j * F::EW), lb + ((size_t)k * S + j) * F::NL);
        ++pending;
    }
    F::acc_fold(acc);
    F::store_loose(out + (size_t)k * F::EW, F::acc_reduce(acc));
}

}  // namespace hbmpc
