// kernels_elem.hpp -- element-wise share arithmetic of one party (reference rows a9, a11, a12, a13).
// One lane per element, 32-byte coalesced loads/stores; all of these are HBM-bound.
// Data x data products need Montgomery form on one side: mont(mont(a, R^2), b) = a*b.
#pragma once
#include "fr_sat.hpp"
#include "fr_u29.hpp"

namespace hbmpc {

// device-constant-form scalars every element-wise kernel may need
struct ElemConsts {
    uint32_t r2[9];     // R^2 mod r  (mont(x, r2) = x*R: canonical -> Montgomery)
    uint32_t c0[9];     // kernel-specific constant 0 (e.g. 2^m, (2^m)^-1) in device-constant form
    uint32_t c1[9];     // kernel-specific constant 1 (e.g. 2^(k-1) as a canonical element in limb form)
};

// Party-batched launches have gridDim.y = parties; per-party arrays are [party][N] (index ip), public operands -- the
// opened values every party shares -- are [N] (index i).  Blocks are dispatched x-fastest, so the linear block id is
// re-read as (element block, party) with the PARTY fastest: the parties of one element range run back to back and the
// public operand is fetched from HBM once, not once per party (config 4's triple_finalize: 16 x 134 MB of re-reads that
// neither the L2 nor the 256 MB memory-side cache held across a 400 MB party pass).  gridDim.y = 1: i as always.
#define HB_GID                                                                 \
    const size_t lin_ = (size_t)blockIdx.y * gridDim.x + blockIdx.x;           \
    const size_t i = (lin_ / gridDim.y) * blockDim.x + threadIdx.x;            \
    if (i >= N) return;
#define HB_PID const size_t ip = (lin_ % gridDim.y) * N + i;

enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2 };

// a + b and a - b (mod r) of canonical 256-bit elements on the STORED words (8 little-endian u32): one carry chain and one conditional
// chain -- for the kernels that only add or subtract, a limb conversion each way costs more instructions than the work itself and
// these kernels should run at what their bytes cost (k_triple_finalize: 0.97 -> 0.89 ms per 2^22 x 16 elements)
HB_DEV void raw_load8(const uint32_t* __restrict__ p, uint32_t (&w)[8]) {
    const uint4 lo = *reinterpret_cast<const uint4*>(p), hi = *reinterpret_cast<const uint4*>(p + 4);
    w[0] = lo.x, w[1] = lo.y, w[2] = lo.z, w[3] = lo.w, w[4] = hi.x, w[5] = hi.y, w[6] = hi.z, w[7] = hi.w;
}
HB_DEV void raw_store8(uint32_t* __restrict__ p, const uint32_t (&w)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4*>(p + 4) = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ static constexpr uint32_t RAW_R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
HB_DEV void raw_add_mod(const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t (&o)[8]) {
    uint32_t sum[8], dif[8], c = 0, bo = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum[k] = __builtin_addc(a[k], b[k], c, &c);  // < 2 r < 2^256: no carry out
#pragma unroll
    for (int k = 0; k < 8; ++k) dif[k] = __builtin_subc(sum[k], RAW_R[k], bo, &bo);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = bo ? sum[k] : dif[k];
}
HB_DEV void raw_sub_mod(const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t (&o)[8]) {
    uint32_t dif[8], fix[8], bo = 0, c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) dif[k] = __builtin_subc(a[k], b[k], bo, &bo);
#pragma unroll
    for (int k = 0; k < 8; ++k) fix[k] = __builtin_addc(dif[k], RAW_R[k], c, &c);  // negative: + r (the carry out cancels the borrow)
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = bo ? fix[k] : dif[k];
}

// generic a (+,-,*) b  (common/mod.rs:167-300: share + share, share - share, share_mul)
template <class F, int OP>
__global__ __launch_bounds__(256) void k_binop(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                               size_t N, ElemConsts cs, uint32_t* __restrict__ out) {
    using E = typename F::E;
    HB_GID
    if constexpr (F::EW == 8 && (OP == OP_ADD || OP == OP_SUB)) {
        uint32_t x8[8], y8[8], o8[8];
        raw_load8(a + i * 8, x8), raw_load8(b + i * 8, y8);
        if constexpr (OP == OP_ADD) raw_add_mod(x8, y8, o8);
        else raw_sub_mod(x8, y8, o8);
        raw_store8(out + i * 8, o8);
        return;
    }
    const E x = F::load(a + i * F::EW), y = F::load(b + i * F::EW);
    if constexpr (OP == OP_ADD) {
        F::store_loose(out + i * F::EW, F::add(x, y));
    } else if constexpr (OP == OP_SUB) {
        F::store_loose(out + i * F::EW, F::template sub<2>(x, y));
    } else {
        const E xm = F::mulc(x, cs.r2);
        F::store_lt2r(out + i * F::EW, F::mont(y, xm));
    }
}

// share (+,-,*) ONE field element, and element - share (common/mod.rs:205-280: Add<F>, Sub<F>, Mul<F>,
// from_scalar_sub): the scalar travels in the kernel arguments, nobody materialises N copies of it
enum { OP_RSUB = 3 };
struct ScalarArg {
    alignas(16) uint32_t w[8];  // canonical element (Goldilocks: the first two words)
};
template <class F, int OP>
__global__ __launch_bounds__(256) void k_scalarop(const uint32_t* __restrict__ a, ScalarArg sc, size_t N, ElemConsts cs,
                                                  uint32_t* __restrict__ out) {
    using E = typename F::E;
    HB_GID
    const E x = F::load(a + i * F::EW), y = F::load(sc.w);
    if constexpr (OP == OP_ADD) {
        F::store_loose(out + i * F::EW, F::add(x, y));
    } else if constexpr (OP == OP_SUB) {
        F::store_loose(out + i * F::EW, F::template sub<2>(x, y));
    } else if constexpr (OP == OP_RSUB) {
        F::store_loose(out + i * F::EW, F::template sub<2>(y, x));
    } else {
        const E ym = F::mulc(y, cs.r2);
        F::store_lt2r(out + i * F::EW, F::mont(x, ym));
    }
}

// triple_gen/triple_generation.rs:333-340:  out = a*b - r2t
template <class F>
__global__ __launch_bounds__(256) void k_triple_local(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                      const uint32_t* __restrict__ r2t, size_t N, ElemConsts cs,
                                                      uint32_t* __restrict__ out) {
    using E = typename F::E;
    HB_GID
    const E am = F::mulc(F::load(a + i * F::EW), cs.r2);
    const E p = F::mont(F::load(b + i * F::EW), am);  // a*b, < 2r
    F::store_loose(out + i * F::EW, F::template sub<2>(p, F::load(r2t + i * F::EW)));
}
// triple_generation.rs:196-208:  c = rt + opened
template <class F>
__global__ __launch_bounds__(256) void k_triple_finalize(const uint32_t* __restrict__ rt,
                                                         const uint32_t* __restrict__ opened, size_t N,
                                                         uint32_t* __restrict__ out) {
    HB_GID
    HB_PID
    if constexpr (F::EW == 8) {
        uint32_t a8[8], b8[8], o8[8];
        raw_load8(rt + ip * 8, a8), raw_load8(opened + i * 8, b8);
        raw_add_mod(a8, b8, o8);
        raw_store8(out + ip * 8, o8);
    } else {
        F::store_loose(out + ip * F::EW, F::add(F::load(rt + ip * F::EW), F::load(opened + i * F::EW)));
    }
}
// mul/multiplication.rs:417-426:  d_sh = a - x, e_sh = b - y
template <class F>
__global__ __launch_bounds__(256) void k_beaver_open(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                     const uint32_t* __restrict__ x, const uint32_t* __restrict__ y,
                                                     size_t N, uint32_t* __restrict__ d_sh,
                                                     uint32_t* __restrict__ e_sh) {
    HB_GID
    if constexpr (F::EW == 8) {
        uint32_t p8[8], q8[8], o8[8];
        raw_load8(a + i * 8, p8), raw_load8(x + i * 8, q8);
        raw_sub_mod(p8, q8, o8);
        raw_store8(d_sh + i * 8, o8);
        raw_load8(b + i * 8, p8), raw_load8(y + i * 8, q8);
        raw_sub_mod(p8, q8, o8);
        raw_store8(e_sh + i * 8, o8);
        return;
    }
    F::store_loose(d_sh + i * F::EW, F::template sub<2>(F::load(a + i * F::EW), F::load(x + i * F::EW)));
    F::store_loose(e_sh + i * F::EW, F::template sub<2>(F::load(b + i * F::EW), F::load(y + i * F::EW)));
}
// the same for `parties` parties at once ([party][N] inputs), with the two results of a party side by side:
// de[party][0][N] = a - x, de[party][1][N] = b - y -- ONE robust-interpolation call over 2 N "chunks" then opens both (the
// sender rows of that call are the parties' rows of 2 N elements), instead of two calls over N
template <class F>
__global__ __launch_bounds__(256) void k_beaver_open_pair(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                          const uint32_t* __restrict__ x, const uint32_t* __restrict__ y,
                                                          size_t N, uint32_t* __restrict__ de) {
    HB_GID
    HB_PID
    const size_t o = (lin_ % gridDim.y) * 2 * N + i;
    if constexpr (F::EW == 8) {
        uint32_t p8[8], q8[8], o8[8];
        raw_load8(a + ip * 8, p8), raw_load8(x + ip * 8, q8);
        raw_sub_mod(p8, q8, o8);
        raw_store8(de + o * 8, o8);
        raw_load8(b + ip * 8, p8), raw_load8(y + ip * 8, q8);
        raw_sub_mod(p8, q8, o8);
        raw_store8(de + (o + N) * 8, o8);
        return;
    }
    F::store_loose(de + o * F::EW, F::template sub<2>(F::load(a + ip * F::EW), F::load(x + ip * F::EW)));
    F::store_loose(de + (o + N) * F::EW, F::template sub<2>(F::load(b + ip * F::EW), F::load(y + ip * F::EW)));
}
// multiplication.rs:57-100 finalize_mul:  z = c - d*e - d*y - e*x  =  c - d*(e + [y]) - e*[x]
// d, e are public (the opened values, [N]); c, x, y, z are [party][N].  One thread serves element i of EVERY party: the
// Montgomery forms of d and e are formed once and each party costs two products (the literal form costs five per party
// -- two conversions and d*e again for every party -- and left the kernel on the vector ALU at 3.7 TB/s).
template <class F>
__global__ __launch_bounds__(256) void k_beaver_finalize(const uint32_t* __restrict__ c, const uint32_t* __restrict__ x,
                                                         const uint32_t* __restrict__ y, const uint32_t* __restrict__ d,
                                                         const uint32_t* __restrict__ e, size_t N, ElemConsts cs,
                                                         uint32_t* __restrict__ z, unsigned parties) {
    using E = typename F::E;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const E ev = F::load(e + i * F::EW);
    const E dm = F::mulc(F::load(d + i * F::EW), cs.r2), em = F::mulc(ev, cs.r2);  // Montgomery forms, normalised, < 2r
    // gridDim.y = 1: this thread serves every party (large N); gridDim.y = parties: one party each (small N, where sixteen
    // dependent load -> multiply -> store rounds in one thread are latency the chip has idle lanes to hide)
    for (unsigned p = blockIdx.y; p < parties; p += gridDim.y) {
        const size_t ip = (size_t)p * N + i;
        const E dey = F::mont(F::add(ev, F::load(y + ip * F::EW)), dm);  // d*(e + [y]); the lazy sum is a legal first operand
        const E ex = F::mont(F::load(x + ip * F::EW), em);               // e*[x]
        E acc = F::template sub<4>(F::load(c + ip * F::EW), dey);
        acc = F::template sub<4>(acc, ex);  // < 9 r
        F::store_loose(z + ip * F::EW, acc);
    }
}
// fpmul/truncpr.rs:277-283:  r_dash[i] = sum_{j<m} 2^j * r_bits[j][i];  pow2[j] = 2^j device-constant form
template <class F>
__global__ __launch_bounds__(256) void k_truncpr_rdash(const uint32_t* __restrict__ r_bits, int m, size_t N,
                                                       const uint32_t* __restrict__ pow2,
                                                       uint32_t* __restrict__ out) {
    using E = typename F::E;
    HB_GID
    typename F::Acc acc;
    F::acc_zero(acc);
    int pending = 0;
    for (int j = 0; j < m; ++j) {
        if (pending == F::MAX_DOT_TERMS) {
            F::acc_fold(acc);
            pending = 1;
        }
        F::acc_mac(acc, F::load(r_bits + (((lin_ % gridDim.y) * m + j) * N + i) * F::EW), pow2 + (size_t)j * F::NL);
        ++pending;
    }
    F::acc_fold(acc);
    const E r = F::acc_reduce(acc);
    HB_PID
    F::store_loose(out + ip * F::EW, r);
}
// truncpr.rs:275,294-297:  open = (a + 2^(k-1)) + (2^m * r_int + r_dash);  cs.c0 = 2^m (const form), cs.c1 = 2^(k-1) limbs
template <class F>
__global__ __launch_bounds__(256) void k_truncpr_open(const uint32_t* __restrict__ a, const uint32_t* __restrict__ r_dash,
                                                      const uint32_t* __restrict__ r_int, size_t N, ElemConsts cs,
                                                      uint32_t* __restrict__ out) {
    using E = typename F::E;
    HB_GID
    E acc = F::add(F::load(a + i * F::EW), F::load_const(cs.c1));
    acc = F::add(acc, F::mulc(F::load(r_int + i * F::EW), cs.c0));
    acc = F::add(acc, F::load(r_dash + i * F::EW));
    F::store_loose(out + i * F::EW, acc);
}
// FPMulNode between its two rounds of opens, in ONE launch (at the batch sizes the protocols use every launch is ~4 us of
// a ~50 us multiplication): finalize_mul (multiplication.rs:57-100), r' (truncpr.rs:277-283) and the share that TruncPr opens
// (truncpr.rs:275,294-297) --
//   z = c - d (e + [y]) - e [x];   r' = sum_j 2^j r_bits[j];   open = (z + 2^(k-1)) + (2^m r_int + r')
// d, e public [N]; everything else [party][N] (r_bits [party][m][N]).  cs.r2 = R^2, cs.c0 = 2^m (const form), cs.c1 = 2^(k-1).
// gridDim.y = 1: the thread serves every party; gridDim.y = parties: one party each (k_beaver_finalize).
template <class F>
__global__ __launch_bounds__(256) void k_fpmul_middle(const uint32_t* __restrict__ c, const uint32_t* __restrict__ x,
                                                      const uint32_t* __restrict__ y, const uint32_t* __restrict__ d,
                                                      const uint32_t* __restrict__ e, const uint32_t* __restrict__ r_bits,
                                                      const uint32_t* __restrict__ r_int, int m, size_t N, ElemConsts cs,
                                                      const uint32_t* __restrict__ pow2, uint32_t* __restrict__ z,
                                                      uint32_t* __restrict__ r_dash, uint32_t* __restrict__ open_out,
                                                      unsigned parties) {
    using E = typename F::E;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const E ev = F::load(e + i * F::EW);
    const E dm = F::mulc(F::load(d + i * F::EW), cs.r2), em = F::mulc(ev, cs.r2);
    for (unsigned p = blockIdx.y; p < parties; p += gridDim.y) {
        const size_t ip = (size_t)p * N + i;
        const E dey = F::mont(F::add(ev, F::load(y + ip * F::EW)), dm);
        const E ex = F::mont(F::load(x + ip * F::EW), em);
        E acc = F::template sub<4>(F::load(c + ip * F::EW), dey);
        acc = F::template sub<4>(acc, ex);
        const E zc = F::canon_loose(acc);  // the canonical z: what k_truncpr_open would load
        F::store_lt2r(z + ip * F::EW, zc);
        typename F::Acc ra;
        F::acc_zero(ra);
        int pending = 0;
        for (int j = 0; j < m; ++j) {
            if (pending == F::MAX_DOT_TERMS) {
                F::acc_fold(ra);
                pending = 1;
            }
            F::acc_mac(ra, F::load(r_bits + (((size_t)p * m + j) * N + i) * F::EW), pow2 + (size_t)j * F::NL);
            ++pending;
        }
        F::acc_fold(ra);
        const E rd = F::canon_loose(F::acc_reduce(ra));
        F::store_lt2r(r_dash + ip * F::EW, rd);
        E o = F::add(zc, F::load_const(cs.c1));
        o = F::add(o, F::mulc(F::load(r_int + ip * F::EW), cs.c0));
        o = F::add(o, rd);
        F::store_loose(open_out + ip * F::EW, o);
    }
}
// truncpr.rs:215-220 + fpmul/mod.rs:381-406:  d = (a - ((c mod 2^m) - r_dash)) * (2^m)^-1;  cs.c0 = (2^m)^-1
template <class F>
__global__ __launch_bounds__(256) void k_truncpr_finalize(const uint32_t* __restrict__ a,
                                                          const uint32_t* __restrict__ r_dash,
                                                          const uint32_t* __restrict__ c_open, int m, size_t N,
                                                          ElemConsts cs, uint32_t* __restrict__ out) {
    using E = typename F::E;
    HB_GID
    // low m bits of the canonical integer: mask the 8 little-endian words
    const uint4 lo = *reinterpret_cast<const uint4*>(c_open + i * F::EW);
    const uint4 hi = *reinterpret_cast<const uint4*>(c_open + i * F::EW + 4);
    uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int lo_bit = 32 * q;
        uint32_t mask = m >= lo_bit + 32 ? 0xffffffffu : (m <= lo_bit ? 0u : ((1u << (m - lo_bit)) - 1u));
        w[q] &= mask;
    }
    __attribute__((aligned(16))) uint32_t tmp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) tmp[q] = w[q];
    // c_mod < 2^m <= c < r for m < 256; m >= 256 keeps c itself.
    E cm;
    if constexpr (F::NL == 9) {
        cm = U29::from_words(w);
    } else {
        cm = F::load(tmp);
    }
    HB_PID
    E t = F::template sub<2>(F::load(r_dash + ip * F::EW), cm);  // r_dash - c_mod  (= -(c_mod - r_dash)), + 2r
    t = F::add(t, F::load(a + ip * F::EW));                       // a - a'
    F::store_lt2r(out + ip * F::EW, F::mulc(t, cs.c0));
}

// register-resident Montgomery-multiply chain: the integer-ALU ceiling of the field implementation
template <class F>
__global__ __launch_bounds__(256) void k_modmul_ubench(uint32_t* __restrict__ out, uint32_t iters, ElemConsts cs) {
    using E = typename F::E;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = (uint32_t)(i * 2654435761u + q * 40503u + 1u);
    w[7] &= 0x3fffffffu;
    __attribute__((aligned(16))) uint32_t tmp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) tmp[q] = w[q];
    E x = F::load(tmp), y = F::mulc(x, cs.r2);
    for (uint32_t k = 0; k < iters; ++k) {
        x = F::mont(x, y);
        y = F::mont(y, x);
    }
    F::store_lt2r(out + i * F::EW, F::mont(x, y));
}

// measurement aid: the memory traffic of an encode with NO arithmetic -- x[G][m] chunk-major in, y[n][G] party-major out, in the
// access shapes of kernels_mfma_bfly.hpp (a lane pair per chunk: 16-byte loads at a stride of 32 m bytes, 1 KiB contiguous per wave
// store; a wave walks 32-chunk tiles in a grid-stride loop, the next tile's loads issued before the current tile's stores).  Every
// loaded dword reaches every stored value, so nothing is dropped as dead.
__global__ __launch_bounds__(768) void k_traffic_ubench(const uint4* __restrict__ x, size_t G, int m, uint4* __restrict__ y, int n) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const size_t ntiles = (G + 31) / 32, tstep = (size_t)gridDim.x * (blockDim.x >> 6);
    size_t t = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    auto load = [&](size_t tile) {
        const size_t gi = tile * 32 + c, g = gi < G ? gi : G - 1;
        uint4 acc = make_uint4(0, 0, 0, 0);
        for (int i = 0; i < m; ++i) {
            const uint4 v = x[(g * m + i) * 2 + h];
            acc.x ^= v.x, acc.y += v.y, acc.z ^= v.z, acc.w += v.w;
        }
        return acc;
    };
    if (t >= ntiles) return;
    uint4 cur = load(t);
    for (; t < ntiles; t += tstep) {
        const uint4 nxt = load(t + tstep < ntiles ? t + tstep : t);
        const size_t g = t * 32 + c;
        if (g < G) {
            for (int j = 0; j < n; ++j) {
                uint4 v = cur;
                v.x += (uint32_t)j;
                y[((size_t)j * G + g) * 2 + h] = v;
            }
        }
        cur = nxt;
    }
}

}  // namespace hbmpc
