// kernels_recover.hpp -- batch_recover_secret, optimistic path (reference row a7,
// honeybadger/robust_interpolate/robust_interpolate.rs:284-443).
//
// One lane per chunk.  The tables depend only on the (sorted) sender ids, so they are built once on
// the host and shared by every chunk -- exactly the reference's structure (:343-399):
//   vm[s - m][i] = L_i(x_s)  for the verify points s = m .. needed-1   (rows s < m of the reference's
//                  verify matrix are the identity: L_i(x_s) = [i == s]; they always pass in exact
//                  arithmetic and are skipped -- same boolean)
//   bc[k][i]     = coefficient k of L_i                                   (:423)
// Per chunk: needed-m verify dot products + m (or 1, P(0)-only) recover dot products of length m,
// each a lazy accumulation (81 carry-free mads per term) with ONE Montgomery reduction per dot.
// Chunks that fail verification are flagged and appended to a compact list with one wave-aggregated
// atomic per wave (ballot + popcount); the OEC/Gao kernel (kernels_gao.hpp) then decodes those.
// Layout: evals[S][G] party-major -> a wave reads 2 KiB contiguous per sender row; outputs
// chunk-major coeffs[G][m] (or secrets[G]).
#pragma once
#include "fr_sat.hpp"
#include "../../include/hbmpc_hip.h"
#include "fr_u29.hpp"

namespace hbmpc {

// The row permutation (rows[s] = position, in the caller's arrays, of the s-th lowest sender id; S <= n <= 255)
// travels in the kernel ARGUMENTS of every kernel of a call: no host buffer an async copy would depend on, no cache
// entry per arrival order (arrival orders change with every reconstruction), graph-capturable, and no init launch.
struct RowsArg {
    uint32_t w[64];  // 256 one-byte positions, four to a word: a dynamic index is one scalar dword load + a shift
    __host__ __device__ int operator[](int s) const { return (int)((w[s >> 2] >> (8 * (s & 3))) & 0xffu); }
    void set(size_t s, unsigned v) { w[s >> 2] = (w[s >> 2] & ~(0xffu << (8 * (s & 3)))) | ((v & 0xffu) << (8 * (s & 3))); }
};

struct RecoverArgs {
    const uint32_t* evals;   // sender rows, canonical; row s starts at evals + rows[s] * row_stride * 8 words
    size_t G;
    size_t row_stride;       // elements between consecutive sender rows (G when the rows are dense)
    RowsArg rows;            // rows[s] = position (in the caller's arrays) of the s-th lowest sender id
    int needed;              // d + t + 1
    int m;                   // d + 1
    const uint32_t* vm;      // [(needed - m)][m] device-constant form
    const uint32_t* bc;      // [m][m]           device-constant form
    uint32_t* out;           // [G][m] or [G]
    uint32_t* ncoeffs;       // [G] or null
    uint8_t* status;         // [G] or null
    uint32_t* flagged;       // [G] compact list of failing chunks
    uint32_t* counters;      // [0] = number of flagged chunks (zero when the call starts: the previous call's last kernel
                             // leaves it so, k_unscale)
    uint32_t* summary;       // {n_fallback, n_failed, first_failed, first_error}: initialised by block 0 of this kernel
    int direct;              // 1: the call has no OEC round and this kernel is all of it (see fail_chunk / finish_direct)
    // k_batch_recover_wide only: the G chunks are `group` consecutive ones per GROUP, and group q's values start q * group_stride
    // elements further on in every sender row (several verifiers' columns in one launch: each verifier has its own block of sender
    // rows); 0: one dense range
    size_t group, group_stride;
};
// Words that one kernel of a call hands to the next and that are WRITTEN WITH ATOMICS (the flagged-chunk counters,
// the summary) must be read with an agent-scope atomic load (global_load ... sc1), never with a plain or scalar load:
// device-scope atomics execute at the memory side and neither update nor invalidate a copy of the line that another
// XCD's L2 or the scalar cache still holds, so `a.counters[0]` as an s_load_dword can return the value from before the
// previous kernel's atomicAdd.  Measured (tools/repro_stale.hip, profiles/r02_repro_stale_pool_memory.txt): with the
// word in hipMallocAsync memory, 6 of 2400 first sequences after an allocation served stale values to s_load_dword /
// plain global_load readers of an atomically incremented word -- every workgroup of the reading kernel, all of which
// started after the writer had finished; never to sc1 loads or atomics, never when the writer was a plain store, never
// in hipMalloc memory.  That is the round-1 failure ("k_gao dropped flagged chunks with pool scratch") -- the scratch
// stays in hipMalloc memory AND the reads are sc1, so caller buffers from any allocator are safe too.
HB_DEV uint32_t load_handoff(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
HB_DEV void store_handoff(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// number of entries of a flagged list: never more than the batch (a counter that did not start at zero -- a replayed
// graph after a failed call -- must not index past the list)
HB_DEV size_t handoff_count(const uint32_t* counter, size_t G) {
    const size_t c = load_handoff(counter);
    return c < G ? c : G;
}
// first kernel of a call: nothing else touches the summary before this kernel has finished
HB_DEV void init_summary(const RecoverArgs& a) {
    if (!a.direct && blockIdx.x == 0 && threadIdx.x < 4) a.summary[threadIdx.x] = threadIdx.x == 2 ? 0xffffffffu : 0u;  // direct: finish_direct writes it
}

template <class F>
HB_DEV void flag_chunks(bool bad, size_t g, const RecoverArgs& a) {
    const unsigned long long mask = __ballot(bad);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(a.counters, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    const size_t slot = (size_t)base + __popcll(mask & ((1ull << lane) - 1ull));
    if (bad && slot < a.G) a.flagged[slot] = (uint32_t)g;  // the list has G entries (see handoff_count)
}

// A call with NO OEC round (S == d + t + 1 senders -- what BatchRecon passes, it decodes as soon as that many have
// arrived, batch_recon.rs:371-389): a chunk that fails the verification can only fail (oec_decode's DecodingError,
// robust_interpolate.rs:625), so the first kernel is the whole call ("direct").  A failing chunk writes its own failure
// (what k_gao would have written: status, zero coefficients, length 0); failures are counted in counters[0] -- zero at
// the start like every call's -- and the lowest failing chunk kept in counters[1] as 0xffffffff - g (zero = none); the
// LAST block to finish (ticket counters[3]) turns the two into the summary and leaves the counters at zero.
template <class F>
HB_DEV void fail_chunk(size_t g, uint32_t* out, int out_width, uint32_t* ncoeffs, uint8_t* status) {
    if (status) status[g] = (uint8_t)DecodingError;
    if (ncoeffs) ncoeffs[g] = 0u;
    for (int k = 0; k < out_width; ++k) F::store_lt2r(out + (g * (size_t)out_width + k) * F::EW, F::zero());
}
// g ascends with the lane: the lowest failing chunk of the wave is its first failing lane's
HB_DEV void count_failures(bool bad, size_t g, uint32_t* counters) {
    const unsigned long long mask = __ballot(bad);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    if (lane == __ffsll((long long)mask) - 1) {
        atomicAdd(counters, (uint32_t)__popcll(mask));
        atomicMax(counters + 1, 0xffffffffu - (uint32_t)g);
        __threadfence();  // performed before this block takes its ticket
    }
}
// every thread of every block, at the end of a direct kernel.  The ticket is two-level (16 sub-tickets in counters[8..24),
// then counters[3]): thousands of blocks taking one ticket word serialise on it -- 2048 blocks of the wave-per-chunk
// kernel added 11 us to a 20 us call.
constexpr unsigned DIRECT_FAN = 16;
HB_DEV void finish_direct(uint32_t* counters, uint32_t* summary) {
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned nblocks = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned sub = b % DIRECT_FAN, quota = nblocks / DIRECT_FAN + (sub < nblocks % DIRECT_FAN ? 1u : 0u);  // blocks with this residue
    __threadfence();  // release: this block's failure counts (any lane's atomics, joined by the barrier above) before its ticket
    if (atomicAdd(counters + 8 + sub, 1u) != quota - 1) return;
    const unsigned groups = nblocks < DIRECT_FAN ? nblocks : DIRECT_FAN;
    if (atomicAdd(counters + 3, 1u) != groups - 1) return;
    __threadfence();  // acquire: every other block's counts before the summary is read
#pragma unroll
    for (unsigned k = 0; k < DIRECT_FAN; ++k) store_handoff(counters + 8 + k, 0u);
    const uint32_t failed = load_handoff(counters), low = load_handoff(counters + 1);
    if (summary) {
        summary[0] = failed, summary[1] = failed;
        summary[2] = failed ? 0xffffffffu - low : 0xffffffffu;
        summary[3] = failed ? (uint32_t)DecodingError : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) store_handoff(counters + k, 0u);
}

// M = d + 1 known at compile time: the chunk's m interpolation inputs live in registers (9 M VGPRs).
// The table constants are wave-uniform, so they are read through the scalar cache into SGPRs and fed
// to v_mad_u64_u32 as its scalar operand: zero VGPRs, zero vector/LDS instructions.  The loads are
// double-buffered by hand (constant i+1 is requested before term i's 81 mads) and fenced with
// sched_barrier so that hipcc does not hoist a whole row into the scalar file (it then spills SGPRs
// through v_writelane/v_readlane, ~700 extra instructions per dot product).
template <class F>
struct ConstRegs {
    uint32_t w[F::NL];
};
HB_DEV const uint32_t* make_uniform(const uint32_t* p) {  // tells hipcc the address is wave-uniform -> s_load
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (const uint32_t*)(((uint64_t)hi << 32) | lo);
}
template <class F>
HB_DEV ConstRegs<F> load_uniform_const(const uint32_t* __restrict__ p) {
    ConstRegs<F> c;
#pragma unroll
    for (int j = 0; j < F::NL; ++j) c.w[j] = p[j];
    return c;
}
// one dot product of the register-resident y[0..M) with the constant row `row`
template <class F, int M>
HB_DEV typename F::E dot_row(const typename F::E (&y)[M], const uint32_t* __restrict__ row) {
    typename F::Acc acc;
    F::acc_zero(acc);
    ConstRegs<F> cur = load_uniform_const<F>(row);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        ConstRegs<F> nxt = cur;
        if (i + 1 < M) nxt = load_uniform_const<F>(row + (i + 1) * F::NL);
        __builtin_amdgcn_sched_barrier(0);
        if (i > 0 && i % F::MAX_DOT_TERMS == 0) F::template acc_fold_needed<M>(acc);  // 64-bit column headroom
        F::acc_mac_pinned(acc, y[i], cur.w);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    return F::acc_reduce(acc);
}

// The shares Multiply opens (mul/multiplication.rs:417-426) as the input of a decode that forms them itself: chunk g < N is a - x of
// element g, chunk N + g is b - y, from a, b, x, y [party][N] (kernels_mfma.hpp, k_mfma_rows<.., SUB>; hbmpc_dev_fpmul_parties)
struct PairInput {
    const uint32_t *a, *b, *x, *y;
    size_t N;
};
template <class F, int M, bool P0_ONLY>
__global__ __launch_bounds__(256, 2) void k_batch_recover(RecoverArgs a) {
    using E = typename F::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];
    init_summary(a);
    const uint32_t *vm, *bc;  // the tables, staged in LDS (the arguments themselves stay untouched in the kernarg segment)
    {
        const int vm_words = (a.needed - M) * M * F::NL, bc_words = (P0_ONLY ? 1 : M) * M * F::NL;
        for (int w = threadIdx.x; w < vm_words; w += 256) tab[w] = a.vm[w];
        for (int w = threadIdx.x; w < bc_words; w += 256) tab[vm_words + w] = a.bc[w];
        vm = tab;
        bc = tab + vm_words;
    }
    __syncthreads();
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < a.G;
    const size_t gg = live ? g : a.G - 1;
    E y[M];
#pragma unroll
    for (int i = 0; i < M; ++i) y[i] = F::load(a.evals + ((size_t)a.rows[i] * a.row_stride + gg) * F::EW);
    bool ok = true;
    for (int s = M; s < a.needed; ++s) {
        const E p = F::cond_sub_r(dot_row<F, M>(y, vm + (size_t)(s - M) * M * F::NL));
        const E ys = F::load(a.evals + ((size_t)a.rows[s] * a.row_stride + gg) * F::EW);
        ok = ok && F::eq_canon(p, ys);
    }
    constexpr int OW = P0_ONLY ? 1 : M;
    if (a.direct) count_failures(live && !ok, g, a.counters);
    else flag_chunks<F>(live && !ok, g, a);
    if (live && !ok) {
        if (a.direct) fail_chunk<F>(g, a.out, OW, a.ncoeffs, a.status);
        else if (a.status) a.status[g] = 0xff;  // 0xff: pending, rewritten by the OEC/Gao kernel
    }
    if (live && ok) {
        if (a.status) a.status[g] = 0;
        for (int k = 0; k < OW; ++k)
            F::store_lt2r(a.out + (g * OW + k) * F::EW, dot_row<F, M>(y, bc + (size_t)k * M * F::NL));
        if (a.ncoeffs) a.ncoeffs[g] = M;
    }
    if (a.direct) finish_direct(a.counters, a.summary);
}

// any m: inputs are re-read from global memory inside the dot products (they stay L2-resident).
template <class F, bool P0_ONLY>
__global__ __launch_bounds__(256) void k_batch_recover_generic(RecoverArgs a) {
    using E = typename F::E;
    init_summary(a);
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < a.G;
    const size_t gg = live ? g : a.G - 1;
    const int M = a.m;
    bool ok = true;
    for (int s = M; s < a.needed; ++s) {
        typename F::Acc acc;
        F::acc_zero(acc);
        const uint32_t* row = a.vm + (size_t)(s - M) * M * F::NL;
        int pending = 0;
        for (int i = 0; i < M; ++i) {
            if (pending == F::MAX_DOT_TERMS) {
                F::acc_fold(acc);
                pending = 1;  // the folded columns count as less than one term
            }
            F::acc_mac(acc, F::load(a.evals + ((size_t)a.rows[i] * a.row_stride + gg) * F::EW), row + i * F::NL);
            ++pending;
        }
        F::acc_fold(acc);
        const E p = F::canon_loose(F::acc_reduce(acc));
        const E ys = F::load(a.evals + ((size_t)a.rows[s] * a.row_stride + gg) * F::EW);
        ok = ok && F::eq_canon(p, ys);
    }
    const int OW = P0_ONLY ? 1 : M;
    if (a.direct) count_failures(live && !ok, g, a.counters);
    else flag_chunks<F>(live && !ok, g, a);
    if (live && !ok) {
        if (a.direct) fail_chunk<F>(g, a.out, OW, a.ncoeffs, a.status);
        else if (a.status) a.status[g] = 0xff;
    }
    if (live && ok && a.status) a.status[g] = 0;
    for (int k = 0; k < (live && ok ? OW : 0); ++k) {
        typename F::Acc acc;
        F::acc_zero(acc);
        const uint32_t* row = a.bc + (size_t)k * M * F::NL;
        int pending = 0;
        for (int i = 0; i < M; ++i) {
            if (pending == F::MAX_DOT_TERMS) {
                F::acc_fold(acc);
                pending = 1;
            }
            F::acc_mac(acc, F::load(a.evals + ((size_t)a.rows[i] * a.row_stride + gg) * F::EW), row + i * F::NL);
            ++pending;
        }
        F::acc_fold(acc);
        F::store_loose(a.out + (g * (size_t)OW + k) * F::EW, F::acc_reduce(acc));
    }
    if (live && ok && a.ncoeffs) a.ncoeffs[g] = (uint32_t)M;
    if (a.direct) finish_direct(a.counters, a.summary);
}

// ---------------------------------------------------------------------------------------------------------------
// Second chance for flagged chunks, before the OEC/Gao kernel.  The reference's fallback (oec_decode,
// robust_interpolate.rs:579-628) returns THE polynomial of degree <= d that agrees with at least d+t+1 of the first
// needed + r sorted shares for some round r <= min(t, S - needed) -- it is unique across rounds (two such polynomials
// share >= d+1 points), Gao's decoder finds it whenever it exists (<= r errors, capacity (t + r) / 2), and agreement
// on a prefix only grows with r.  So a candidate that interpolates ANY d+1 of those shares and disagrees with at
// most rmax of the first P = needed + rmax IS that polynomial.  Up to four candidates cost a few dot products each:
// window A = the lowest d+1 senders (right whenever the liars sit in the verify rows), window B = the next d+1
// (right whenever they sit in A), the last d+1 of the prefix and one straddling A and B (tables.hpp,
// second_windows).  A single Byzantine sender -- the case that otherwise sends EVERY chunk of a
// batch down a path 20x slower than the optimistic one -- is always caught by one of the two.  What neither
// resolves goes on to k_gao through a second list.  Lane per flagged chunk; inputs are re-read through the cache.
struct SecondArgs {
    const uint32_t* evals;
    size_t G, row_stride;
    RowsArg rows;
    int m, P, rmax, n_windows, out_width;
    int win_start[4];
    const uint32_t* ev[4];       // [(P - m)][m] rows L_i(x_s), s ascending over the positions outside the window
    const uint32_t* bc[4];       // [m][m]
    const uint32_t* flagged;     // chunks the optimistic kernel flagged; counters[0] of them
    uint32_t* flagged2;          // chunks left for k_gao; counters[1] of them
    uint32_t* counters;
    uint32_t* out;
    uint32_t* ncoeffs;
    uint8_t* status;
    uint32_t* summary;
};
template <class F>
HB_DEV typename F::E second_dot(const SecondArgs& a, size_t g, int ws, const uint32_t* __restrict__ row) {
    typename F::Acc acc;
    F::acc_zero(acc);
    int pending = 0;
    for (int i = 0; i < a.m; ++i) {
        if (pending == F::MAX_DOT_TERMS) {
            F::acc_fold(acc);
            pending = 1;
        }
        F::acc_mac(acc, F::load(a.evals + ((size_t)a.rows[ws + i] * a.row_stride + g) * F::EW), row + i * F::NL);
        ++pending;
    }
    F::acc_fold(acc);
    return F::canon_loose(F::acc_reduce(acc));
}
// true: resolved (coefficients, length and status written); false: the chunk goes on to OEC/Gao.  No atomics in here: the
// caller tallies a wave's results with ONE atomic each (a batch in which every chunk is flagged made 2^20 atomic
// increments of one word)
template <class F>
HB_DEV bool second_chance_one(const SecondArgs& a, size_t g, int& first) {
    using E = typename F::E;
    const int M = a.m;
    // liars are usually the same senders in every chunk of a batch: start with the window that resolved this lane's
    // previous chunk (the order of the candidates cannot change the result -- an accepted one is THE polynomial)
    for (int i = 0; i < a.n_windows; ++i) {
        const int w = first + i < a.n_windows ? first + i : first + i - a.n_windows;
        const int ws = a.win_start[w];
        int mism = 0;
        for (int e = 0; e < a.P - M && mism <= a.rmax; ++e) {
            const int s = e < ws ? e : e + M;  // e-th position outside [ws, ws + M)
            const E p = second_dot<F>(a, g, ws, a.ev[w] + (size_t)e * M * F::NL);
            const E ys = F::load(a.evals + ((size_t)a.rows[s] * a.row_stride + g) * F::EW);
            mism += F::eq_canon(p, ys) ? 0 : 1;
        }
        if (mism > a.rmax) continue;
        // accepted: coefficients, zero padded by construction; DensePolynomial length for the fallback's trimmed row
        int len = 0;
        for (int k = 0; k < M; ++k) {
            if (k >= a.out_width && !a.ncoeffs) break;
            const E c = second_dot<F>(a, g, ws, a.bc[w] + (size_t)k * M * F::NL);
            if (!F::is_zero_canon(c)) len = k + 1;
            if (k < a.out_width) F::store_lt2r(a.out + (g * (size_t)a.out_width + k) * F::EW, c);
        }
        if (a.ncoeffs) a.ncoeffs[g] = (uint32_t)len;
        if (a.status) a.status[g] = 1;
        first = w;
        return true;
    }
    return false;
}
// The same decision with a whole wave per flagged chunk (lane = table row): used when the flagged list is short
// enough to give every chunk a wave -- one lying share in a one-polynomial recover_secret then costs two dot
// products of latency per candidate instead of a serial walk over (P - m) + m of them.
template <class F>
HB_DEV void second_chance_wave(const SecondArgs& a, size_t g, uint32_t* tally = nullptr) {
    using E = typename F::E;
    const int M = a.m, lane = threadIdx.x & 63;
    for (int w = 0; w < a.n_windows; ++w) {
        const int ws = a.win_start[w];
        bool bad = false;
        if (lane < a.P - M) {
            const int s = lane < ws ? lane : lane + M;
            const E p = second_dot<F>(a, g, ws, a.ev[w] + (size_t)lane * M * F::NL);
            bad = !F::eq_canon(p, F::load(a.evals + ((size_t)a.rows[s] * a.row_stride + g) * F::EW));
        }
        if (__popcll(__ballot(bad)) > a.rmax) continue;
        bool nonzero = false;
        if (lane < M && (lane < a.out_width || a.ncoeffs)) {
            const E c = second_dot<F>(a, g, ws, a.bc[w] + (size_t)lane * M * F::NL);
            nonzero = !F::is_zero_canon(c);
            if (lane < a.out_width) F::store_lt2r(a.out + (g * (size_t)a.out_width + lane) * F::EW, c);
        }
        const unsigned long long nz = __ballot(nonzero);
        if (lane == 0) {
            if (a.ncoeffs) a.ncoeffs[g] = nz ? 64u - (uint32_t)__clzll((long long)nz) : 0u;
            if (a.status) a.status[g] = 1;
            // fused into the first kernel of the call, the summary is still being initialised by block 0: count in
            // the scratch tally instead (k_unscale adds it to the summary at the end of the call)
            atomicAdd(tally ? tally : &a.summary[0], 1u);
        }
        return;
    }
    if (lane == 0) {
        const uint32_t slot = atomicAdd(&a.counters[1], 1u);
        if (slot < a.G) a.flagged2[slot] = (uint32_t)g;
    }
}
template <class F>
__global__ __launch_bounds__(256) void k_second_chance(SecondArgs a) {
    // grid-stride over the flagged list: the launch is sized for a modest list and costs next to nothing when the
    // list is empty (the normal case)
    const size_t count = handoff_count(a.counters, a.G);
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
    // A wave per chunk spreads the P - m check rows over its lanes: (P - m) times shorter per chunk, on 64 lanes instead
    // of one.  While the list is shorter than the chip is wide in lanes, that is a pure gain in latency: one lane per
    // chunk leaves most SIMDs idle behind a few hundred serial products (10 485 flagged chunks of config 3: 0.35 ms in
    // 164 waves).  From about nwaves (P - m) chunks on, the lane form has the better throughput.
    const size_t spread = (size_t)(a.P - a.m > 2 ? (a.P - a.m) / 2 : 1);
    if (count <= nwaves * spread && a.P - a.m <= 64 && a.m <= 64) {
        for (size_t fi = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); fi < count; fi += nwaves)
            second_chance_wave<F>(a, a.flagged[fi]);
        return;
    }
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    int first = 0;
    uint32_t resolved = 0;
    // wave-uniform trip count (the ballots below need every lane of the wave in the loop)
    for (size_t base = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < count; base += step) {
        const size_t fi = base + lane;
        const bool valid = fi < count;
        const uint32_t g = valid ? a.flagged[fi] : 0u;
        const bool done = valid && second_chance_one<F>(a, g, first);
        resolved += done ? 1u : 0u;
        const unsigned long long left = __ballot(valid && !done);  // on to OEC/Gao: one slot allocation per wave
        if (left != 0) {
            const int leader = __ffsll((long long)left) - 1;
            uint32_t slot0 = 0;
            if (lane == leader) slot0 = atomicAdd(&a.counters[1], (uint32_t)__popcll(left));
            slot0 = __shfl(slot0, leader);
            const size_t slot = (size_t)slot0 + __popcll(left & ((1ull << lane) - 1ull));
            if (valid && !done && slot < a.G) a.flagged2[slot] = g;
        }
    }
    // one tally per wave
    for (int off = 32; off > 0; off >>= 1) resolved += __shfl_down(resolved, off);
    if (lane == 0 && resolved) atomicAdd(&a.summary[0], resolved);
}

// Long lists (a Byzantine sender flags EVERY chunk of a batch), U29, m known at compile time: the lane-per-chunk walk with
// the machinery of k_batch_recover -- the window's m inputs in registers, its table staged in LDS once per block and
// window, constants fed to the multiply-accumulates as scalars.  For that the candidates are tried in the SAME order by
// every lane of a block (the generic form lets each lane start with the window that resolved its previous chunk and
// re-reads inputs and constants per term: 2.65 ms for 2^20 chunks of config 3 against 0.875 ms for the optimistic kernel
// that does two thirds of the products).  The order of the candidates cannot change the result -- an accepted one is THE
// polynomial -- so a block starts with the window that last accepted something.  ev[w] and bc[w] are contiguous
// (tables.hpp, SecondTables::layout): (P - m) check rows, then m coefficient rows.
template <class F, int M>
__global__ __launch_bounds__(256, 2) void k_second_chance_m(SecondArgs a) {
    using E = typename F::E;
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];
    const size_t count = handoff_count(a.counters, a.G);
    const int lane = threadIdx.x & 63;
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
    // this lane form overtakes the wave form early: 10 485 flagged chunks of config 3 take 0.15 ms as waves, 0.11 ms as lanes
    if (count <= nwaves && a.P - M <= 64) {  // short list: a wave per chunk (k_second_chance)
        for (size_t fi = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); fi < count; fi += nwaves)
            second_chance_wave<F>(a, a.flagged[fi]);
        return;
    }
    const int checks = a.P - M, tab_words = a.P * M * F::NL;
    int first = 0;
    uint32_t resolved = 0;
    for (size_t base = (size_t)blockIdx.x * 256; base < count; base += (size_t)gridDim.x * 256) {  // block-uniform
        const size_t fi = base + threadIdx.x;
        const bool valid = fi < count;
        const uint32_t g = a.flagged[valid ? fi : count - 1];
        bool pending = valid;
        int next_first = first;
        bool have_next = false;
        for (int i = 0; i < a.n_windows; ++i) {
            if (!__syncthreads_or(pending)) break;  // also: everybody is done with the previous window's table
            const int w = first + i < a.n_windows ? first + i : first + i - a.n_windows;
            const int ws = a.win_start[w];
            for (int k = threadIdx.x; k < tab_words; k += 256) tab[k] = a.ev[w][k];
            __syncthreads();
            E y[M];
#pragma unroll
            for (int j = 0; j < M; ++j) y[j] = F::load(a.evals + ((size_t)a.rows[ws + j] * a.row_stride + g) * F::EW);
            int mism = 0;
            for (int e = 0; e < checks; ++e) {
                const int s = e < ws ? e : e + M;  // e-th position outside [ws, ws + M)
                const E p = F::cond_sub_r(dot_row<F, M>(y, tab + (size_t)e * M * F::NL));
                const E ys = F::load(a.evals + ((size_t)a.rows[s] * a.row_stride + g) * F::EW);
                mism += F::eq_canon(p, ys) ? 0 : 1;
                if (__ballot(pending && mism <= a.rmax) == 0) break;  // nobody in this wave can still accept this window
            }
            const bool acc = pending && mism <= a.rmax;
            if (__ballot(acc) != 0) {  // coefficients, zero padded by construction; DensePolynomial length for the trimmed row
                int len = 0;
                const int kmax = a.ncoeffs ? M : (a.out_width < M ? a.out_width : M);
                for (int k = 0; k < kmax; ++k) {
                    const E c = F::cond_sub_r(dot_row<F, M>(y, tab + (size_t)(checks + k) * M * F::NL));
                    if (!F::is_zero_canon(c)) len = k + 1;
                    if (acc && k < a.out_width) F::store_lt2r(a.out + (g * (size_t)a.out_width + k) * F::EW, c);
                }
                if (acc) {
                    if (a.ncoeffs) a.ncoeffs[g] = (uint32_t)len;
                    if (a.status) a.status[g] = 1;
                    ++resolved;
                    pending = false;
                }
            }
            if (!have_next && __syncthreads_or(acc)) next_first = w, have_next = true;
        }
        first = next_first;
        const unsigned long long left = __ballot(pending);  // on to OEC/Gao: one slot allocation per wave
        if (left != 0) {
            const int leader = __ffsll((long long)left) - 1;
            uint32_t slot0 = 0;
            if (lane == leader) slot0 = atomicAdd(&a.counters[1], (uint32_t)__popcll(left));
            slot0 = __shfl(slot0, leader);
            const size_t slot = (size_t)slot0 + __popcll(left & ((1ull << lane) - 1ull));
            if (pending && slot < a.G) a.flagged2[slot] = g;
        }
    }
    for (int off = 32; off > 0; off >>= 1) resolved += __shfl_down(resolved, off);
    if (lane == 0 && resolved) atomicAdd(&a.summary[0], resolved);
}

// Small batches (the regime the protocols really run in: one reconstruction, a few hundred elements per message):
// with one lane per chunk a call of G chunks occupies G/64 waves and each lane walks all (needed - m + out_width) m
// products serially -- 55 us for ONE chunk at n = 31.  Here a chunk gets a whole wave and every lane owns one ROW of
// the tables (a verify row or an output row): latency m products instead of (needed - m + out_width) m, at the price
// of idle lanes, which is free while the chip is not full.  Same tables, same arithmetic, same results; the chunk's
// sender values are staged once in the wave's slice of LDS, constants are read per lane.
// One table row times the M values of a chunk with the products SHARED by 2^lk adjacent lanes (lk = 0, 1, 2; U29): lane s of
// them takes the terms i = s, s + 2^lk, ..; the carry-folded partial sums (columns < 2^30) are added across the quad by DPP
// (quad_perm [1,0,3,2], then [2,3,0,1]) and every lane reduces the total.  Only lane 0's partial sum carries the bias of
// acc_zero.  A lone wave per SIMD issues a v_mad_u64_u32 every ~10 cycles (profiles/r01_isa_rates.txt): the 81 m of a row's products are the latency of a
// small decode, and the lanes that would idle take three quarters of them (profiles/r04_small_batch_fpmul.txt).
// The lanes that share a row must be active together (whole pairs / quads).
template <class F, class Y>
HB_DEV typename F::E dot_shared(Y&& value_of, const uint32_t* row, int M, int lk, int sidx) {
    static_assert(F::NL == 9, "the column layout of U29's accumulator");
    typename F::Acc acc;
    F::acc_zero(acc);
    if (sidx != 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) acc.c[i] = 0;
    }
    int pending = 0;
    for (int i = sidx; i < M; i += 1 << lk) {
        if (pending == F::MAX_DOT_TERMS) {
            F::acc_fold(acc);
            pending = 1;
        }
        F::acc_mac(acc, value_of(i), row + i * F::NL);
        ++pending;
    }
    F::acc_fold(acc);
    if (lk >= 1) {
#pragma unroll
        for (int i = 0; i < 17; ++i) acc.c[i] = (uint32_t)acc.c[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc.c[i], 0xB1, 0xf, 0xf, false);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc.c[17], 0xB1, 0xf, 0xf, false);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(acc.c[17] >> 32), 0xB1, 0xf, 0xf, false);
        acc.c[17] += ((uint64_t)hi << 32) | lo;
    }
    if (lk >= 2) {
#pragma unroll
        for (int i = 0; i < 17; ++i) acc.c[i] = (uint32_t)acc.c[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc.c[i], 0x4E, 0xf, 0xf, false);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc.c[17], 0x4E, 0xf, 0xf, false);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(acc.c[17] >> 32), 0x4E, 0xf, 0xf, false);
        acc.c[17] += ((uint64_t)hi << 32) | lo;
    }
    return F::acc_reduce(acc);
}

struct WideArgs {
    RecoverArgs r;
    SecondArgs sc;  // used when fused != 0
    int fused;      // a chunk that fails the verification tries the second-chance candidates right here (same wave)
    int tab_words;  // TAB instances: (needed - m + out_width) * m constants, r.vm and r.bc contiguous, staged in LDS
    int lk;         // U29 TAB instances: log2 of the lanes that share a row's products (dot_shared); > 0 only when every row fits the wave
    int split;      // TAB instances: r.bc does not follow r.vm (a single coefficient row other than row 0): staged word by word from both
    int ow;         // !P0_ONLY instances: output rows per chunk when not all m (a table of selected coefficient rows); 0 = m
};
// rows[i], i < 64, from the scalar side: a per-lane index into the argument struct compiles to a VECTOR load from the
// argument segment -- a full memory round trip in front of the loads that depend on it; sixteen scalar words and a
// select per word are not
HB_DEV int row_of_lane(const RowsArg& rows, int i) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if ((i >> 2) == k) w = rows.w[k];
    return (int)((w >> (8 * (i & 3))) & 0xffu);
}
// LDS of one workgroup (4 chunks): [4][needed] sender values | TAB: the verify and output rows' constants, staged once per
// workgroup: both loads of a chunk -- table and sender values -- are in flight together and the m products of a row run
// from LDS, instead of m dependent round trips for the constants and a fourth of the table traffic.  (At n = 16 a decode
// of 1024 chunks takes 13 us either way: a lone wave per SIMD spends them in its ~570 v_mad_u64_u32 at ~10 cycles each and one
// round trip; profiles/r04_small_batch_fpmul.txt.)
template <class F, bool P0_ONLY, bool TAB = true>
__global__ __launch_bounds__(256) void k_batch_recover_wide(WideArgs wa) {
    using E = typename F::E;
    const RecoverArgs& a = wa.r;
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
    init_summary(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t g_raw = (size_t)blockIdx.x * 4 + wave;
    const bool live = g_raw < a.G;
    const size_t g = live ? g_raw : a.G - 1;
    const int M = a.m, nv = a.needed - M, ow = P0_ONLY ? 1 : wa.ow ? wa.ow : M;
    const size_t gin = a.group ? (g / a.group) * a.group_stride + g % a.group : g;  // where the chunk's values are in a sender row
    uint32_t* ys = tile + (size_t)wave * a.needed * F::EW;
    uint32_t* tab = tile + (size_t)4 * a.needed * F::EW;
    // --- every global load first ---
    const uint4* tsrc = reinterpret_cast<const uint4*>(a.vm);
    const int tq = wa.tab_words >> 2;
    uint4 t0 = make_uint4(0, 0, 0, 0);
    const int vm_words = nv * M * F::NL;
    if (TAB && wa.split) {  // one word per thread and pass: the verify rows from r.vm, the output rows from r.bc
        if ((int)threadIdx.x < wa.tab_words) t0.x = (int)threadIdx.x < vm_words ? a.vm[threadIdx.x] : a.bc[threadIdx.x - vm_words];
    } else if (TAB && (int)threadIdx.x < tq) {
        t0 = tsrc[threadIdx.x];
    }
    {
        uint32_t first[F::EW];
        if (lane < a.needed) {
            const uint32_t* src = a.evals + ((size_t)row_of_lane(a.rows, lane) * a.row_stride + gin) * F::EW;
#pragma unroll
            for (int w = 0; w < F::EW; ++w) first[w] = src[w];
        }
        if (TAB && wa.split) {
            if ((int)threadIdx.x < wa.tab_words) tab[threadIdx.x] = t0.x;
        } else if (TAB && (int)threadIdx.x < tq) {
            reinterpret_cast<uint4*>(tab)[threadIdx.x] = t0;
        }
        if (lane < a.needed) {
#pragma unroll
            for (int w = 0; w < F::EW; ++w) ys[lane * F::EW + w] = first[w];
        }
        for (int i = lane + 64; i < a.needed; i += 64) {
            const uint32_t* src = a.evals + ((size_t)a.rows[i] * a.row_stride + gin) * F::EW;
#pragma unroll
            for (int w = 0; w < F::EW; ++w) ys[i * F::EW + w] = src[w];
        }
    }
    if constexpr (TAB) {  // what one pass of the workgroup did not cover (tables beyond 4 KB), and the words past the last 16 bytes
        if (wa.split) {
            for (int w = threadIdx.x + 256; w < wa.tab_words; w += 256) tab[w] = w < vm_words ? a.vm[w] : a.bc[w - vm_words];
        } else {
            for (int q = threadIdx.x + 256; q < tq; q += 256) reinterpret_cast<uint4*>(tab)[q] = tsrc[q];
            for (int w = (tq << 2) + threadIdx.x; w < wa.tab_words; w += 256) tab[w] = a.vm[w];
        }
    }
    __syncthreads();
    auto chunk = [&]() __attribute__((always_inline)) {
    if (!live) return;
    auto dot = [&](int r) -> E {  // r < nv: verify row r; else output row r - nv
        const uint32_t* row = TAB ? tab + (size_t)r * M * F::NL : r < nv ? a.vm + (size_t)r * M * F::NL : a.bc + (size_t)(r - nv) * M * F::NL;
        typename F::Acc acc;
        F::acc_zero(acc);
        int pending = 0;
        for (int i = 0; i < M; ++i) {
            if (pending == F::MAX_DOT_TERMS) {
                F::acc_fold(acc);
                pending = 1;
            }
            F::acc_mac(acc, F::load(ys + i * F::EW), row + i * F::NL);
            ++pending;
        }
        F::acc_fold(acc);
        return F::acc_reduce(acc);
    };
    auto expect = [&](int r) -> E { return F::load(ys + (size_t)(M + r) * F::EW); };  // the value verify row r is compared with
    // first sweep: rows lane, of verify and output rows alike (an output row's value waits in registers for the vote)
    bool bad = false, have = false;
    E kept = F::zero();
    int row_of_mine = lane;
    if constexpr (TAB && F::NL == 9) {
        if (wa.lk > 0) {  // every row fits the wave with 2^lk lanes each (host): this sweep is the only one
            const int sidx = lane & ((1 << wa.lk) - 1);
            row_of_mine = lane >> wa.lk;
            if (row_of_mine < nv + ow) {
                kept = dot_shared<F>([&](int i) { return F::load(ys + i * F::EW); }, tab + (size_t)row_of_mine * M * F::NL, M, wa.lk, sidx);
                if (row_of_mine < nv) bad = !F::eq_canon(F::canon_loose(kept), expect(row_of_mine));
                else have = sidx == 0;
            }
            row_of_mine = sidx == 0 ? row_of_mine : nv + ow;  // the other lanes of a row own nothing
        }
    }
    if ((!(TAB && F::NL == 9) || wa.lk == 0) && lane < nv + ow) {
        kept = dot(lane);
        if (lane < nv) bad = !F::eq_canon(F::canon_loose(kept), expect(lane));
        else have = true;
    }
    for (int r = lane + 64; r < nv; r += 64) bad = bad || !F::eq_canon(F::canon_loose(dot(r)), expect(r));
    const bool ok = __ballot(bad) == 0;
    if (!ok && a.direct) {  // no OEC round: the failure is final (fail_chunk, one lane per output element)
        if (lane < ow) F::store_lt2r(a.out + (g * (size_t)ow + lane) * F::EW, F::zero());
        for (int k = lane + 64; k < ow; k += 64) F::store_lt2r(a.out + (g * (size_t)ow + k) * F::EW, F::zero());
        if (lane == 0) {
            if (a.status) a.status[g] = (uint8_t)DecodingError;
            if (a.ncoeffs) a.ncoeffs[g] = 0u;
            atomicAdd(a.counters, 1u);
            atomicMax(a.counters + 1, 0xffffffffu - (uint32_t)g);
            __threadfence();
        }
        return;
    }
    if (lane == 0) {
        if (a.status) a.status[g] = ok ? 0 : 0xff;  // 0xff: pending, rewritten by the OEC/Gao kernel
        if (!ok && !wa.fused) {
            const uint32_t slot = atomicAdd(a.counters, 1u);
            if (slot < a.G) a.flagged[slot] = (uint32_t)g;
        }
        else if (ok && a.ncoeffs) a.ncoeffs[g] = (uint32_t)M;
    }
    if (!ok) {
        // one launch less for a small batch: resolved chunks are tallied in counters[2], the others go to the
        // OEC/Gao list (counters[1]) exactly as the separate kernel would leave them
        if (wa.fused) second_chance_wave<F>(wa.sc, g, a.counters + 2);
        return;
    }
    if (have) F::store_loose(a.out + (g * (size_t)ow + (row_of_mine - nv)) * F::EW, kept);
    for (int r = lane + 64; r < nv + ow; r += 64)
        if (r >= nv) F::store_loose(a.out + (g * (size_t)ow + (r - nv)) * F::EW, dot(r));
    };
    chunk();
    if (a.direct) finish_direct(a.counters, a.summary);
}


}  // namespace hbmpc
