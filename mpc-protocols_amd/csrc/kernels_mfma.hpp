// kernels_mfma.hpp -- the constant-matrix maps of the path on the gfx950 matrix cores (v_mfma_i32_32x32x32_i8).
//
//   decode  (batch_recover_secret, robust_interpolate.rs:391-427): verify rows  Y * VM  and coefficient rows  Y * BC
//   encode  (apply_vandermonde, common/share/mod.rs:50-76; compute_shares = the same map, robust_interpolate.rs:52-82)
//
// Formulation and table layout: tables_mfma.hpp.  One MFMA tile = 32 chunks (columns, B operand = the chunk's canonical
// bytes straight from HBM, sign-flipped with one XOR per dword) x 32 result digits (rows, A operand = table slab);
// K = 32 bytes of one input element, so an output element of a chunk costs m MFMAs per 32 chunks.  A lane pair
// (c, h = 0 / 1) holds the 32 digit sums of chunk c: digits 16 h .. 16 h + 15 in the 16 accumulator registers, i.e.
// each lane owns one 128-bit half of the 256-bit result.  The epilogue stays in registers:
//   gather    digits (< 2^24, spaced 8 bits) -> 4 x 32-bit words + carry per half (8 v_lshl_add_u32, 4 v_mad_u64_u32,
//             one 128-bit add)
//   verify    r = 1 (mod 2^32), so  S = y + q r  <=>  q = (S - y) mod 2^32  and  S + q (2^256 - r) = y + q 2^256:
//             one 4-word multiply-add chain per half, exact, no quotient estimate and no conditional subtraction
//   reduce    q' = floor(top 49 bits / (r >> 224) + 1) <= q, R = S - q' r; R < r whenever word 8 cancels and the top
//             word is below r's top word; the (rare) rest takes a wave-uniform slow path of conditional subtractions
// Carries cross from the low half to the high half once per chain (v_permlane32_swap).
// Table rows are RESIDENT in LDS for the lifetime of a workgroup (one workgroup per CU, up to 160 KB = 13 rows at
// m = 11): the rows of a call are cut into row groups ("roles"), the grid is divided among the roles in proportion to
// their rows, and every wave walks 32-chunk tiles in a grid-stride loop with no barrier after the table load.  A first
// version streamed the current row through a double-buffered LDS slot with one barrier per row: every interval then
// waited for one L2 round trip of the next row's loads (measured 2.3 us per interval, 0.41 ms for config 3 whatever
// was removed from the interval -- profiles/r02_ubench_mfma_v2_streaming_ablation.txt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fr_u29.hpp"
#include "kernels_recover.hpp"

namespace hbmpc {
namespace mf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// words of r and of 2^256 - r, least significant first
__device__ static constexpr uint32_t R_W[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                               0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
__device__ static constexpr uint32_t NR_W[8] = {0xffffffffu, 0x00000000u, 0x0001a401u, 0xac425bfdu,
                                                0xf65e27fau, 0xccc627f7u, 0xd66282b7u, 0x8c1258acu};
constexpr uint32_t R_TOP = 0x73eda753u;   // r >> 224
constexpr uint32_t Q_RECIP = 0x8d54253au; // floor(2^62 / (R_TOP + 1))

struct Half {
    uint32_t nr[4], rw[4];  // this lane half's words of 2^256 - r and of r
    uint32_t hmask;         // all ones in the high half
    uint32_t k16;           // 2^16 kept opaque so that the gather stays v_mad_u64_u32
};
HB_DEV Half make_half(int h) {
    Half H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        H.nr[j] = h ? NR_W[4 + j] : NR_W[j];
        H.rw[j] = h ? R_W[4 + j] : R_W[j];
    }
    H.hmask = h ? 0xffffffffu : 0u;
    uint32_t b = 1u << 16;
    asm volatile("" : "+s"(b));
    H.k16 = b;
    return H;
}
// v_permlane32_swap vdst, src0 exchanges lanes 32..63 of vdst with lanes 0..31 of src0; with both = x the first
// result is the low half's value in every lane, the second the high half's
HB_DEV uint32_t low_bcast(uint32_t x) { return (uint32_t)__builtin_amdgcn_permlane32_swap(x, x, false, false)[0]; }
HB_DEV uint32_t high_bcast(uint32_t x) { return (uint32_t)__builtin_amdgcn_permlane32_swap(x, x, false, false)[1]; }

// The epilogue works on UN-NORMALISED 64-bit words: T[j] = digits 4j .. 4j+3 of this half combined (two digits fit
// 32 bits, v_lshl_add_u32; two such pairs fit one v_mad_u64_u32), so T[j] < 2^48 overlaps T[j+1] by its high word.
// The quotient product is added on top with the words as the 64-bit addend of the multiply-add, and carries are
// resolved ONCE, by one v_add_co / v_addc chain over the low and high words.
HB_DEV void gather(const v16i& acc, uint64_t (&T)[4], const Half& H) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p0 = ((uint32_t)acc[4 * j + 1] << 8) + (uint32_t)acc[4 * j];
        const uint32_t p1 = ((uint32_t)acc[4 * j + 3] << 8) + (uint32_t)acc[4 * j + 2];
        T[j] = (uint64_t)p1 * H.k16 + p0;
    }
}
// the digit pairs alone (kernels_mfma_bfly.hpp adds / subtracts two accumulators at this level: 8 values instead of 16)
HB_DEV void gather_pairs(const v16i& acc, uint32_t (&p)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = ((uint32_t)acc[2 * j + 1] << 8) + (uint32_t)acc[2 * j];
}
// U = S + q (2^256 - r) for S = sum T[j] 2^(32 j) of both halves.  On return the high half holds words 4..7 of U and
// `top` = everything above 2^256; the low half words 0..3 (its `top` already handed to the high half).
HB_DEV void add_q_nr(uint32_t q, const uint64_t (&T)[4], uint32_t (&U)[4], uint32_t& top, const Half& H) {
    uint64_t Q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Q[j] = (uint64_t)q * H.nr[j] + T[j];
    uint32_t c;
    U[0] = (uint32_t)Q[0];
    U[1] = __builtin_addc((uint32_t)Q[1], (uint32_t)(Q[0] >> 32), 0u, &c);
    U[2] = __builtin_addc((uint32_t)Q[2], (uint32_t)(Q[1] >> 32), c, &c);
    U[3] = __builtin_addc((uint32_t)Q[3], (uint32_t)(Q[2] >> 32), c, &c);
    top = (uint32_t)(Q[3] >> 32) + c;
    const uint32_t cin = low_bcast(top) & H.hmask;
    U[0] = __builtin_addc(U[0], cin, 0u, &c);
    U[1] = __builtin_addc(U[1], 0u, c, &c);
    U[2] = __builtin_addc(U[2], 0u, c, &c);
    U[3] = __builtin_addc(U[3], 0u, c, &c);
    top += c;
}
// nonzero in some lane of the pair iff  sum != ys (mod r);  ys = this half's 4 words of the claimed value (canonical)
HB_DEV uint32_t verify_tile(const v16i& acc, const v4i& ys, const Half& H) {
    uint64_t T[4];
    uint32_t U[4], top;
    gather(acc, T, H);
    const uint32_t q = low_bcast((uint32_t)T[0] - (uint32_t)ys[0]);  // word 0 of the sum has no carry-in
    add_q_nr(q, T, U, top, H);
    uint32_t bad = (U[0] ^ (uint32_t)ys[0]) | (U[1] ^ (uint32_t)ys[1]) | (U[2] ^ (uint32_t)ys[2]) | (U[3] ^ (uint32_t)ys[3]);
    bad |= (top ^ q) & H.hmask;
    return bad;
}
// canonical residue of the digit sums: this half's 4 words in Rw
HB_DEV void reduce_words(const uint64_t (&T)[4], uint32_t (&Rw)[4], const Half& H);
HB_DEV void reduce_tile(const v16i& acc, uint32_t (&Rw)[4], const Half& H) {
    uint64_t T[4];
    gather(acc, T, H);
    reduce_words(T, Rw, H);
}
HB_DEV void reduce_words(const uint64_t (&T)[4], uint32_t (&Rw)[4], const Half& H) {
    uint32_t top;
    // the high half estimates the quotient from its top word alone: T[3] >> 17 <= (sum >> 241), so never too large
    const uint32_t xq = (uint32_t)(T[3] >> 17);
    const uint32_t q = high_bcast(__umulhi(xq, Q_RECIP) >> 13);
    add_q_nr(q, T, Rw, top, H);
    // exact when word 8 cancels (R = S - q r fits 256 bits) and R's top word is below r's
    const bool fast = H.hmask == 0 || (top == q && Rw[3] < R_TOP);
    if (__builtin_expect(__any(!fast) != 0, 0)) {
        // e = what is left above 2^256 (0 <= e, small); subtract r while e 2^256 + R >= r
        uint32_t e = high_bcast(top - q);
        for (int it = 0; it < 5; ++it) {
            uint32_t D[4];
            uint64_t b = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t d = (uint64_t)Rw[j] - H.rw[j] - b;
                D[j] = (uint32_t)d;
                b = (d >> 32) & 1;
            }
            const uint32_t bin = low_bcast((uint32_t)b) & H.hmask;
            uint64_t d = (uint64_t)D[0] - bin;
            D[0] = (uint32_t)d;
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                d = (uint64_t)D[j] - ((d >> 32) & 1);
                D[j] = (uint32_t)d;
            }
            const uint32_t bout = high_bcast((uint32_t)b + (uint32_t)((d >> 32) & 1));  // 0 or 1 (never both)
            const bool take = e >= bout;  // e 2^256 + R - r >= 0
            if (take) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Rw[j] = D[j];
                e -= bout;
            }
        }
    }
}

// a - x (mod r) of two canonical elements held as lane-pair halves: one 128-bit subtract per half, the low half's borrow (and, in
// the same word, the carry its "+ r" would produce) to the high half, the sign back, r added where the difference was negative
HB_DEV v4i sub_mod_r(const v4i& a, const v4i& x, const Half& H) {
    uint32_t d[4], e[4], b, c;
    d[0] = __builtin_subc((uint32_t)a[0], (uint32_t)x[0], 0u, &b);
    d[1] = __builtin_subc((uint32_t)a[1], (uint32_t)x[1], b, &b);
    d[2] = __builtin_subc((uint32_t)a[2], (uint32_t)x[2], b, &b);
    d[3] = __builtin_subc((uint32_t)a[3], (uint32_t)x[3], b, &b);
    // the low half: d + r's low words and that sum's carry (what the high half needs when the difference is negative)
    e[0] = __builtin_addc(d[0], H.rw[0], 0u, &c);
    e[1] = __builtin_addc(d[1], H.rw[1], c, &c);
    e[2] = __builtin_addc(d[2], H.rw[2], c, &c);
    e[3] = __builtin_addc(d[3], H.rw[3], c, &c);
    const uint32_t lo = low_bcast(b | (c << 1));  // bit 0: the low half's borrow, bit 1: its "+ r" carry
    // the high half takes the borrow in
    uint32_t bb, b2;
    const uint32_t bin = lo & 1u & H.hmask;
    uint32_t f[4];
    f[0] = __builtin_subc(d[0], bin, 0u, &bb);
    f[1] = __builtin_subc(d[1], 0u, bb, &bb);
    f[2] = __builtin_subc(d[2], 0u, bb, &bb);
    f[3] = __builtin_subc(d[3], 0u, bb, &b2);
    const uint32_t neg = high_bcast(b | b2);  // the 256-bit difference is negative (b and b2 are never both set)
    // high half, negative: f + r's high words + the low half's carry
    uint32_t g[4], cc;
    const uint32_t cin = (lo >> 1) & H.hmask;
    g[0] = __builtin_addc(f[0], H.rw[0], cin, &cc);
    g[1] = __builtin_addc(f[1], H.rw[1], cc, &cc);
    g[2] = __builtin_addc(f[2], H.rw[2], cc, &cc);
    g[3] = __builtin_addc(f[3], H.rw[3], cc, &cc);
    v4i r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (int)(neg ? (H.hmask ? g[j] : e[j]) : f[j]);
    return r;
}

HB_DEV v4i flip(v4i x) {
    x[0] ^= 0x80808080, x[1] ^= 0x80808080, x[2] ^= 0x80808080, x[3] ^= 0x80808080;
    return x;
}

// buffer descriptors for stores that drop what must not be written (an offset beyond the buffer; a buffer of zero bytes): the
// unrolled tile loops of kernels_mfma_bfly.hpp have no conditional stores, so that hipcc can count the memory operations
// between a load and its use
HB_DEV __amdgpu_buffer_rsrc_t rt_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
constexpr uint32_t RT_OOB = 0xfffffff0u;  // beyond every buffer: the store is dropped

// the M MFMAs of one output row for CG tiles; the A operand (table slab) is read from LDS three slabs ahead of its use
template <int M, int CG>
HB_DEV void mfma_row(const uint8_t* tab_lane, const v4i (&data)[CG][M], v16i (&acc)[CG]) {
    constexpr int D = 3;
    v4i av[D];
#pragma unroll
    for (int i = 0; i < D - 1 && i < M; ++i) av[i] = *reinterpret_cast<const v4i*>(tab_lane + i * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        if (i + D - 1 < M) av[(i + D - 1) % D] = *reinterpret_cast<const v4i*>(tab_lane + (i + D - 1) * 1024);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) acc[cg] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i % D], data[cg][i], acc[cg], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks every read to just before its MFMA
    }
}

struct MfmaRole {
    int row0, nrows;  // rows [row0, row0 + nrows) of the table
};
constexpr int MF_MAX_ROLES = 4;
struct MfmaRowsArgs {
    // input: M elements per chunk
    const uint8_t* in;
    size_t G;
    int in_chunk_major;   // 1: x[G][M] (encode);  0: sender rows, row s at in + rows[s] * row_stride * 32 (decode)
    size_t row_stride;    // elements
    RowsArg rows;
    // table: rows [0, nv) are verify rows (claimed value = sender row M + r), rows >= nv produce output k = r - nv
    const uint8_t* table;
    int nv;
    // output k of chunk g: party-major out + (k * out_stride + g) * 32, or chunk-major out + (g * out_stride + k) * 32
    uint8_t* out;
    int out_party_major;
    size_t out_stride;
    uint32_t* ncoeffs;    // decode only (nullable): M for accepted chunks
    uint8_t* status;      // decode only (nullable)
    uint32_t* flagged;
    uint32_t* counters;
    uint32_t* summary;    // decode only (nullable): initialised by workgroup 0
    // Roles: the rows of the call cut into groups that fit the LDS -- the verify rows, then the output rows (kept
    // together as far as they fit, so that all coefficients of a chunk are written through one XCD's L2).  Workgroups
    // are dealt to roles in blocks of 8, in proportion to the roles' rows and interleaved: block j serves role
    // blk_role[j] as its blk_idx[j]-th block.  Workgroups b and b + 8 sit on the same XCD under round-robin placement, so
    // a role often finds the input rows of a tile in that XCD's L2 (speed only, never correctness).
    int nroles;
    int nblocks;          // grid = 8 * nblocks
    int role_nwg[MF_MAX_ROLES];
    uint8_t blk_role[64], blk_idx[64];
    MfmaRole role[MF_MAX_ROLES];
    int direct;           // 1 (decode, ONE role): no OEC round exists, a chunk that fails the verification fails for good and this
                          // kernel is the whole call (kernels_recover.hpp: fail_chunk / count_failures / finish_direct)
    int half, nout;       // kernels_mfma_bfly.hpp only: table rows are point PAIRS (k, k + half), outputs k + half >= nout do not exist
    // k_mfma_bfly<.., TRIPLE>: the inputs are (a b - r2t) / 2^261 of `in` (a), in_b and in_r, x[parties][G][M] each, and the
    // table rows carry the factor 2^261
    const uint8_t* in_b;
    const uint8_t* in_r;
    int parties;
    // kernels_mfma_bfly.hpp, DEG instances: only output rows below store_rows are stored (0: all) -- the degree still looks at every
    // coefficient, so store_rows = 1 with out_stride = 1 leaves c_0[G] and degree[G] (the RanDouSha verifier's two tests)
    int store_rows;
    // kernels_mfma_bfly.hpp: the producers' mixing step writes the parties' OUTPUT rows in the reference's list order.  Output rows
    // [list_row0, list_row0 + list_rows) of chunk g = j list_K + k (party j, batch element k) go, 32 bytes each, to
    //   list[s].dst + ((j list[s].stride + (k - list[s].k0) list_rows + (row - list_row0)) * 32,   s: the slice whose [k0, k0 + count) holds k
    // (share_gen.rs:199-203, ran_dou_sha/mod.rs:314-331: per party [k][row]); list_rows = 0: every row is party-major
    int list_row0, list_rows;
    uint32_t list_K;
    // with lists: the OTHER rows (what the parties send the verifiers) party-major, other_stride != 0: row number r' among them (r' = row
    // below the list rows, row - list_rows above) of chunk (j, k) goes to out + ((j other_stride + r' list_K + k) * 32, other_stride =
    // (n - list_rows) list_K -- every verifier's column block of a sender is then one contiguous row of (verifier, k) chunks
    uint32_t other_stride;
    // k_mfma_rows<.., SUB> (decode): the senders' values are DIFFERENCES formed as they are loaded (the shares Multiply opens,
    // mul/multiplication.rs:417-426): chunk g < sub_half is in[row][g] - sub_x[row][g], chunk g >= sub_half is
    // in2[row][g - sub_half] - sub_x2[row][g - sub_half]  (mod r); sub_half is a multiple of 32, rows are row_stride elements apart
    const uint8_t *sub_x, *in2, *sub_x2;
    size_t sub_half;
    struct ListSlice {
        uint8_t* dst;
        uint64_t stride;  // elements between the lists of consecutive parties
        uint32_t k0, count;
    } list[2];
};

// NR > 0: every role of the launch has at most NR rows and the row loop is unrolled NR times with a compile-time trip
// count -- hipcc can then count the stores issued after the next tile's loads and wait with vmcnt(#stores) at the tile
// boundary; with a run-time trip count it waits for vmcnt(0), i.e. for every store of the tile to complete.
// (The timing-only ablations and the software-pipelined row loop of rounds 2 and 3 live in tools/kernels_mfma_lab.hpp.)
// SUB (decode, CG = 1, one role): the sender values are differences of two arrays formed after loading (MfmaRowsArgs::sub_x); the
// raw operands of the NEXT tile wait in registers while the current tile is processed.
template <int M, int CG, int WAVES, int NR = 0, bool SUB = false>
__global__ __launch_bounds__(64 * WAVES) void k_mfma_rows(MfmaRowsArgs a) {
    static_assert(!SUB || CG == 1, "SUB instances walk single tiles");
    static_assert(M <= 15, "digit sums must stay below 0xff0000 (tables_mfma.hpp) and the sum below 2^273");
    constexpr int ROWB = M * 1024 + 128;
    constexpr int NT = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // role.nrows * ROWB
    if (a.summary && !a.direct && blockIdx.x == 0 && threadIdx.x < 4) a.summary[threadIdx.x] = threadIdx.x == 2 ? 0xffffffffu : 0u;
    const int blk8 = (int)blockIdx.x >> 3, role_id = a.blk_role[blk8];
    const int wg_in_role = (int)a.blk_idx[blk8] * 8 + ((int)blockIdx.x & 7);
    MfmaRole role = a.role[0];
    int role_wgs = a.role_nwg[0];
#pragma unroll
    for (int k = 1; k < MF_MAX_ROLES; ++k)
        if (k == role_id) role = a.role[k], role_wgs = a.role_nwg[k];
    {
        const uint8_t* src = a.table + (size_t)role.row0 * ROWB;
        const int pieces = role.nrows * (ROWB / 16);
        for (int p = threadIdx.x; p < pieces; p += NT)
            *reinterpret_cast<v4i*>(lds + (size_t)p * 16) = *reinterpret_cast<const v4i*>(src + (size_t)p * 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const Half H = make_half(h);
    const size_t ntiles = (a.G + 32 * CG - 1) / (32 * CG);
    const int nver = role.row0 < a.nv ? (a.nv - role.row0 < role.nrows ? a.nv - role.row0 : role.nrows) : 0;  // verify rows of this role
    // Addresses are a wave-uniform 64-bit row base plus a 32-bit lane offset (the host keeps G * 32 * max(M, out
    // width) below 2^32).
    // HBM latency: a wave that loads the M input rows of its tile and then computes on them for ~20 us leaves too few
    // bytes in flight per CU (measured: +0.145 ms on config 3 over the same kernel with its inputs in L2,
    // profiles/r02_ubench_mfma_v3_resident_ablation.txt).  So a wave keeps TWO input register sets and the tile loop is
    // unrolled by two: at the start of a tile all M loads of its NEXT tile are issued into the other set, a whole tile
    // ahead of their use.  (Tried and dropped: spreading those loads over the output rows -- register indices that
    // depend on the row index made hipcc peel the loop and copy the set at every merge; pulling the next tile into L2
    // with LDS-DMA loads into a scratch slot -- vmcnt is in-order, so the next wait for a claimed value then also waits
    // for the prefetch issued just before it: 0.44 ms instead of 0.36.)
    const size_t tstep = (size_t)role_wgs * WAVES;
    const uint32_t in_lane_stride = a.in_chunk_major ? M * 32u : 32u;
    auto tile_chunks = [&](size_t t, uint32_t (&gg)[CG]) {
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            const size_t gi = (t * CG + cg) * 32 + c;
            gg[cg] = (uint32_t)(gi < a.G ? gi : a.G - 1);
        }
    };
    auto load_inputs = [&](size_t t, v4i (&dst)[CG][M]) {
        uint32_t gg[CG];
        tile_chunks(t, gg);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const uint8_t* base = a.in_chunk_major ? a.in + (size_t)i * 32 : a.in + (size_t)a.rows[i] * a.row_stride * 32;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) dst[cg][i] = *reinterpret_cast<const v4i*>(base + (gg[cg] * in_lane_stride + 16u * h));
        }
    };
    // the role that owns the verify rows (all of them: the host never splits them) gives the verdict per chunk;
    // with no verify rows at all (needed == M) the role of table row 0 accepts every chunk
    auto give_verdict = [&](const uint32_t (&bad)[CG], const uint32_t (&g)[CG], const bool (&live)[CG]) {
        if (a.status != nullptr || a.flagged != nullptr) {
            if (nver > 0 || (a.nv == 0 && role.row0 == 0)) {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    const unsigned long long m = __ballot(bad[cg] != 0);
                    const uint32_t m32 = (uint32_t)m | (uint32_t)(m >> 32);
                    const bool ok = ((m32 >> c) & 1u) == 0;
                    const bool flag = live[cg] && !ok && h == 0;
                    const unsigned long long fm = __ballot(flag);
                    if (fm != 0 && a.direct) {  // count_failures: chunks ascend with the lane
                        if (lane == __ffsll((long long)fm) - 1) {
                            atomicAdd(a.counters, (uint32_t)__popcll(fm));
                            atomicMax(a.counters + 1, 0xffffffffu - g[cg]);
                            __threadfence();
                        }
                    } else if (fm != 0) {
                        const int leader = __ffsll((long long)fm) - 1;
                        uint32_t base = 0;
                        if (lane == leader) base = atomicAdd(a.counters, (uint32_t)__popcll(fm));
                        base = __shfl(base, leader);
                        const size_t slot = (size_t)base + __popcll(fm & ((1ull << lane) - 1ull));
                        if (flag && slot < a.G) a.flagged[slot] = g[cg];  // the list has G entries (handoff_count)
                    }
                    if (live[cg] && h == 0) {
                        if (a.status) a.status[g[cg]] = ok ? 0 : a.direct ? (uint8_t)DecodingError : 0xff;  // 0xff: pending, rewritten by the fallback kernels
                        if (a.ncoeffs && (ok || a.direct)) a.ncoeffs[g[cg]] = ok ? M : 0;
                    }
                }
            }
        }
    };
    // one tile from the register set `data` (raw bytes on entry; sign-flipped in place)
    auto process_tile = [&](size_t t, v4i (&data)[CG][M]) {
        uint32_t g[CG];
        bool live[CG];
        tile_chunks(t, g);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            live[cg] = ((t * CG + cg) * 32 + c) < a.G;
#pragma unroll
            for (int i = 0; i < M; ++i) data[cg][i] = flip(data[cg][i]);
        }
        const bool second = SUB && t * 32 >= a.sub_half;  // wave-uniform: sub_half is a multiple of the tile
        auto load_ys = [&](int r, v4i (&ys)[CG]) {  // claimed values of verify row r (table row index)
            uint32_t ri = (uint32_t)a.rows[M + r];
            asm volatile("" : "+s"(ri));  // recomputed at every use: hoisted out of the tile loop, the row bases of an
                                          // unrolled row loop (NR > 0) would take two SGPRs each and spill the scalar file
            if constexpr (SUB) {
                const size_t ro = (size_t)ri * a.row_stride * 32;
                const uint32_t lo = (g[0] - (second ? (uint32_t)a.sub_half : 0u)) * 32u + 16u * h;
                const v4i va = *reinterpret_cast<const v4i*>((second ? a.in2 : a.in) + ro + lo);
                const v4i vx = *reinterpret_cast<const v4i*>((second ? a.sub_x2 : a.sub_x) + ro + lo);
                ys[0] = sub_mod_r(va, vx, H);
            } else {
                const uint8_t* base = a.in + (size_t)ri * a.row_stride * 32;
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) ys[cg] = *reinterpret_cast<const v4i*>(base + (g[cg] * 32u + 16u * h));
            }
        };
        v4i ys_cur[CG], ys_next[CG];
        if (nver > 0) load_ys(role.row0, ys_cur);
        uint32_t bad[CG];
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) bad[cg] = 0;
#pragma unroll
        for (int r = 0; r < (NR > 0 ? NR : role.nrows); ++r) {
            if (NR > 0 && r >= role.nrows) break;
            const uint8_t* cur = lds + (size_t)r * ROWB;
            const int rho = role.row0 + r;
            if (r + 1 < nver) load_ys(rho + 1, ys_next);
            v16i acc[CG];
            {
                const v4i* bp = reinterpret_cast<const v4i*>(cur + M * 1024 + h * 64);
                const v4i b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
                v16i bias;
#pragma unroll
                for (int k = 0; k < 4; ++k) bias[k] = b0[k], bias[4 + k] = b1[k], bias[8 + k] = b2[k], bias[12 + k] = b3[k];
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) acc[cg] = bias;
                mfma_row<M, CG>(cur + lane * 16, data, acc);
            }
            if (r < nver) {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) bad[cg] |= verify_tile(acc[cg], ys_cur[cg], H);
                if (r + 1 < nver) {
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) ys_cur[cg] = ys_next[cg];
                }
            } else {
                uint32_t k32 = (uint32_t)(rho - a.nv);
                asm volatile("" : "+s"(k32));  // as above: the output row base is recomputed, not kept per unrolled row
                const size_t k = k32;
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    uint32_t Rw[4];
                    reduce_tile(acc[cg], Rw, H);
                    if (a.direct) {  // one role, the verify rows are behind us: a chunk that failed them gets zeros
                        const unsigned long long mb = __ballot(bad[cg] != 0);
                        if ((((uint32_t)mb | (uint32_t)(mb >> 32)) >> c) & 1u) Rw[0] = Rw[1] = Rw[2] = Rw[3] = 0u;
                    }
                    uint8_t* qb = a.out_party_major ? a.out + k * a.out_stride * 32 : a.out + k * 32;  // wave-uniform
                    const uint32_t qo = g[cg] * (a.out_party_major ? 32u : (uint32_t)a.out_stride * 32u) + 16u * h;
                    if (live[cg]) *reinterpret_cast<uint4*>(qb + qo) = make_uint4(Rw[0], Rw[1], Rw[2], Rw[3]);
                }
            }
        }
        give_verdict(bad, g, live);
    };
    size_t t = (size_t)wg_in_role * WAVES + wave;
    if constexpr (SUB) {
        // raw operands of the next tile in flight (2 M registers of 4), the current tile's differences in `cur`
        v4i ra[M], rx[M], cur[1][M];
        auto load_raw = [&](size_t tt) {
            const bool sec = tt * 32 >= a.sub_half;
            const size_t gi = tt * 32 + c, gc = gi < a.G ? gi : a.G - 1;
            const uint32_t lo = (uint32_t)(gc - (sec ? a.sub_half : 0)) * 32u + 16u * h;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const size_t ro = (size_t)a.rows[i] * a.row_stride * 32;
                ra[i] = *reinterpret_cast<const v4i*>((sec ? a.in2 : a.in) + ro + lo);
                rx[i] = *reinterpret_cast<const v4i*>((sec ? a.sub_x2 : a.sub_x) + ro + lo);
            }
        };
        if (t < ntiles) load_raw(t);
        while (t < ntiles) {
#pragma unroll
            for (int i = 0; i < M; ++i) cur[0][i] = sub_mod_r(ra[i], rx[i], H);
            if (t + tstep < ntiles) load_raw(t + tstep);
            process_tile(t, cur);
            t += tstep;
        }
        if (a.direct) finish_direct(a.counters, a.summary);
        return;
    }
    v4i setA[CG][M], setB[CG][M];
    if (t < ntiles) load_inputs(t, setA);
    // the tile loop: the input sets alternate, the loads of a wave's NEXT tile are issued before it starts on the current one
#define HBMPC_MF_WALK(TILE)                                                \
    while (t < ntiles) {                                                   \
        if (t + tstep < ntiles) load_inputs(t + tstep, setB);              \
        TILE(t, setA);                                                     \
        t += tstep;                                                        \
        if (t >= ntiles) break;                                            \
        if (t + tstep < ntiles) load_inputs(t + tstep, setA);              \
        TILE(t, setB);                                                     \
        t += tstep;                                                        \
    }
    HBMPC_MF_WALK(process_tile)
#undef HBMPC_MF_WALK
    if (a.direct) finish_direct(a.counters, a.summary);
}

// nwg workgroups (rounded down to blocks of 8, at most 64 blocks) shared among the roles already in a->role[0 .. a->nroles)
// in proportion to their rows
inline bool mf_deal_blocks(int rows, int nwg, MfmaRowsArgs* a) {
    const int nroles = a->nroles;
    int nblocks = nwg / 8;
    nblocks = nblocks > 64 ? 64 : nblocks < nroles ? nroles : nblocks;
    a->nblocks = nblocks;
    // blocks per role in proportion to its rows (at least one), dealt out by largest remaining deficit
    int have[MF_MAX_ROLES] = {0, 0, 0, 0};
    for (int j = 0; j < nblocks; ++j) {
        int best = 0;
        double bestd = -1e30;
        for (int k = 0; k < nroles; ++k) {
            const double want = (double)(j + 1) * a->role[k].nrows / rows;
            const double dfc = have[k] == 0 && nblocks - j <= nroles ? 1e9 : want - have[k];  // nobody is left without a block
            if (dfc > bestd) bestd = dfc, best = k;
        }
        a->blk_role[j] = (uint8_t)best;
        a->blk_idx[j] = (uint8_t)have[best]++;
    }
    for (int k = 0; k < nroles; ++k) {
        if (have[k] == 0) return false;
        a->role_nwg[k] = have[k] * 8;
    }
    return true;
}
// Host side: cut `rows` table rows (the first nv of them verify rows) into roles of at most `cap` rows.  Everything in one
// role when it fits; otherwise the verify rows form role 0 and the output rows are cut evenly into as few roles as
// possible.  nwg workgroups (rounded down to blocks of 8, at most 64 blocks) are shared in proportion to the rows.
// Returns false when the verify rows do not fit one role (the caller then uses the lane-per-chunk kernels).
inline bool mf_plan_roles(int rows, int nv, int cap, int nwg, MfmaRowsArgs* a) {
    if (cap < 1 || nv > cap || rows < 1) return false;
    int nroles = 0;
    if (rows <= cap) {
        a->role[nroles++] = MfmaRole{0, rows};
    } else {
        if (nv > 0) a->role[nroles++] = MfmaRole{0, nv};
        const int ow = rows - nv, parts = (ow + cap - 1) / cap, per = (ow + parts - 1) / parts;
        for (int r = nv; r < rows; r += per) {
            if (nroles == MF_MAX_ROLES) return false;
            a->role[nroles++] = MfmaRole{r, rows - r < per ? rows - r : per};
        }
    }
    a->nroles = nroles;
    return mf_deal_blocks(rows, nwg, a);
}
// the point pairs of kernels_mfma_bfly.hpp: `pairs` (a power of two) table rows in roles of EQUAL size, the largest power
// of two that fits `cap` rows -- the kernel's unrolled pair loop has one trip count for every workgroup of a launch
inline bool mf_plan_pairs(int pairs, int cap, int nwg, MfmaRowsArgs* a) {
    if (cap < 1 || pairs < 1 || (pairs & (pairs - 1)) != 0) return false;
    int per = pairs;
    while (per > cap) per >>= 1;
    if (per < 1 || pairs / per > MF_MAX_ROLES) return false;
    a->nroles = pairs / per;
    for (int k = 0; k < a->nroles; ++k) a->role[k] = MfmaRole{k * per, per};
    return mf_deal_blocks(pairs, nwg, a);
}
inline int mf_grid(const MfmaRowsArgs& a) { return 8 * a.nblocks; }
inline int mf_max_role_rows(const MfmaRowsArgs& a) {
    int m = 0;
    for (int k = 0; k < a.nroles; ++k) m = a.role[k].nrows > m ? a.role[k].nrows : m;
    return m;
}

}  // namespace mf
}  // namespace hbmpc
