// tables.hpp -- host-side construction of the small constant tables the kernels stage
// (all depend only on (n, d, t, sender ids), never on batch data).
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/hbmpc_hip.h"
#include "host_fr.hpp"

namespace hbmpc {

enum FieldImpl { IMPL_U29 = 0, IMPL_SAT32 = 1, IMPL_GOLD = 2 };
inline int impl_nl(int impl) { return impl == IMPL_U29 ? 9 : impl == IMPL_SAT32 ? 8 : 2; }
inline size_t impl_ebytes(int impl) { return impl == IMPL_GOLD ? 8 : 32; }  // bytes per stored element

inline void put_const(std::vector<uint32_t>& out, const HFr& v, int impl) {
    uint32_t tmp[9];
    if (impl == IMPL_U29) {
        v.to_u29(tmp);
        out.insert(out.end(), tmp, tmp + 9);
    } else {
        v.to_sat32(tmp);
        out.insert(out.end(), tmp, tmp + 8);
    }
}
// canonical integer in the implementation's LIMB form (not Montgomery): for load_const of plain values
inline void put_plain(std::vector<uint32_t>& out, const HFr& v, int impl) {
    uint64_t c[4];
    v.to_canon(c);
    if (impl == IMPL_U29) {
        for (int i = 0; i < 9; ++i) {
            const int o = 29 * i, q = o >> 6, s = o & 63;
            uint64_t x = c[q] >> s;
            if (s > 64 - 29 && q + 1 < 4) x |= c[q + 1] << (64 - s);
            out.push_back((uint32_t)(x & 0x1fffffffu));
        }
    } else {
        for (int i = 0; i < 4; ++i) {
            out.push_back((uint32_t)c[i]);
            out.push_back((uint32_t)(c[i] >> 32));
        }
    }
}

inline void put_const(std::vector<uint32_t>& out, const HGl& v, int) {
    out.push_back((uint32_t)v.v);
    out.push_back((uint32_t)(v.v >> 32));
}
inline void put_plain(std::vector<uint32_t>& out, const HGl& v, int impl) { put_const(out, v, impl); }

inline size_t domain_size(size_t n) {
    size_t s = 1;
    while (s < n) s <<= 1;
    return s;
}
inline int ilog2(size_t s) {
    int l = 0;
    while (((size_t)1 << l) < s) ++l;
    return l;
}
// omega_size = 7^((r-1)/2^32) ^ (2^32/size)   (common/mod.rs:51-68 -> GeneralEvaluationDomain::new)
template <class H>
inline bool domain_omega(size_t size, H* w) {
    const int lg = ilog2(size);
    if (lg > 32) return false;
    H root = H::two_adic_root();
    for (int i = lg; i < 32; ++i) root = root * root;
    *w = root;
    return true;
}
template <class H = HFr>
inline std::vector<H> domain_elements(size_t n, size_t count) {
    H w;
    domain_omega(domain_size(n), &w);
    std::vector<H> el(count);
    H p = H::one();
    for (size_t j = 0; j < count; ++j) {
        el[j] = p;
        p = p * w;
    }
    return el;
}

// tw[q] = omega_S^q, q < max(S/2, 1)
template <class H = HFr>
inline std::vector<uint32_t> build_twiddles(size_t S, int impl) {
    std::vector<uint32_t> out;
    H w;
    domain_omega(S, &w);
    H p = H::one();
    for (size_t q = 0; q < std::max<size_t>(S / 2, 1); ++q) {
        put_const(out, p, impl);
        p = p * w;
    }
    return out;
}
// twist[r][k] = omega_size^(r k), r < P, k < dp1
template <class H = HFr>
inline std::vector<uint32_t> build_twist(size_t size, size_t P, size_t dp1, int impl) {
    std::vector<uint32_t> out;
    H w;
    domain_omega(size, &w);
    H wr = H::one();
    for (size_t r = 0; r < P; ++r) {
        H p = H::one();
        for (size_t k = 0; k < dp1; ++k) {
            put_const(out, p, impl);
            p = p * wr;
        }
        wr = wr * w;
    }
    return out;
}
template <class H = HFr>
inline std::vector<uint32_t> build_alpha(size_t n, int impl) {
    std::vector<uint32_t> out;
    for (const H& a : domain_elements<H>(n, n)) put_const(out, a, impl);
    return out;
}
inline std::vector<uint32_t> build_pow2(size_t m, int impl) {
    std::vector<uint32_t> out;
    HFr p = HFr::one(), two = HFr::from_u64(2);
    for (size_t j = 0; j < m; ++j) {
        put_const(out, p, impl);
        p = p * two;
    }
    return out;
}

// Everything a sender set's tables need from the evaluation domain, computed ONCE per n: the size elements alpha_k and
// E[k] = 1 / (alpha_k - 1), k >= 1 (one field inversion for all of them).  A difference of two domain elements is
// alpha_a - alpha_b = alpha_b (alpha_{a-b} - 1), so its inverse is alpha_{size-b} E[a-b]: no table of a sender set needs a
// field inversion (a Fermat inversion is ~380 multiplications -- it used to be most of a small set's Lagrange basis).
template <class H>
struct DomainInv {
    size_t n = 0, size = 0;
    std::vector<H> el, E;
    explicit DomainInv(size_t n_) : n(n_), size(domain_size(n_)) {
        el = domain_elements<H>(n_, size);
        E.assign(size, H::zero());
        if (size < 2) return;
        std::vector<H> pre(size);
        H run = H::one();
        for (size_t k = 1; k < size; ++k) {
            pre[k] = run;
            run = run * (el[k] - H::one());
        }
        H all = run.inv();
        for (size_t k = size; k-- > 1;) {
            E[k] = all * pre[k];
            all = all * (el[k] - H::one());
        }
    }
    // 1 / (alpha_a - alpha_b), a != b
    H dinv(size_t a, size_t b) const { return el[(size - b) % size] * E[(a + size - b) % size]; }
};

// Lagrange basis over the domain points with the given ids (distinct): basis[i][k] = coefficient k of
// L_i(x) = A(x) / ((x - x_i) A'(x_i)),  A = prod (x - x_j)     (robust_interpolate.rs:351-376).
// A_out (optional): the m + 1 coefficients of A; invden_out (optional): 1 / A'(x_i).
template <class H>
inline std::vector<std::vector<H>> lagrange_basis_ids(const DomainInv<H>& D, const size_t* ids, size_t m, std::vector<H>* A_out = nullptr,
                                                      std::vector<H>* invden_out = nullptr) {
    std::vector<H> A(m + 1, H::zero());
    A[0] = H::one();
    size_t deg = 0;
    for (size_t j = 0; j < m; ++j) {  // A *= (x - x_j)
        const H nx = D.el[ids[j]].neg();
        A[deg + 1] = A[deg];
        for (size_t k = deg; k > 0; --k) A[k] = A[k] * nx + A[k - 1];
        A[0] = A[0] * nx;
        ++deg;
    }
    std::vector<H> inv(m);
    for (size_t i = 0; i < m; ++i) {  // 1 / A'(x_i) = prod_{j != i} 1 / (x_i - x_j)
        H dd = H::one();
        for (size_t j = 0; j < m; ++j)
            if (j != i) dd = dd * D.dinv(ids[i], ids[j]);
        inv[i] = dd;
    }
    std::vector<std::vector<H>> basis(m, std::vector<H>(m, H::zero()));
    std::vector<H> q(m);
    for (size_t i = 0; i < m; ++i) {
        // synthetic division of A by (x - x_i): q[m-1] = A[m], q[k-1] = A[k] + x_i q[k]
        const H xi = D.el[ids[i]];
        q[m - 1] = A[m];
        for (size_t k = m - 1; k > 0; --k) q[k - 1] = A[k] + xi * q[k];
        for (size_t k = 0; k < m; ++k) basis[i][k] = q[k] * inv[i];
    }
    if (A_out) *A_out = A;
    if (invden_out) *invden_out = inv;
    return basis;
}
// L_i(x_s) for a domain point s OUTSIDE the set: A(x_s) / ((x_s - x_i) A'(x_i)) -- two products per value once
// A(x_s) = prod_j (x_s - x_j) is known, instead of a Horner walk over the basis polynomial
template <class H>
inline void lagrange_eval_row(const DomainInv<H>& D, const size_t* ids, size_t m, const std::vector<H>& invden, size_t s_id, H* row) {
    H As = H::one();
    const H xs = D.el[s_id];
    for (size_t j = 0; j < m; ++j) As = As * (xs - D.el[ids[j]]);
    for (size_t i = 0; i < m; ++i) row[i] = As * invden[i] * D.dinv(s_id, ids[i]);
}
// the general form over arbitrary distinct points (one field inversion): what the tests of the host tables pin
template <class H>
inline std::vector<std::vector<H>> lagrange_basis(const std::vector<H>& xs) {
    const size_t m = xs.size();
    std::vector<H> A(m + 1, H::zero());
    A[0] = H::one();
    size_t deg = 0;
    for (const H& x : xs) {  // A *= (x - x_j)
        const H nx = x.neg();
        A[deg + 1] = A[deg];
        for (size_t k = deg; k > 0; --k) A[k] = A[k] * nx + A[k - 1];
        A[0] = A[0] * nx;
        ++deg;
    }
    std::vector<H> den(m), pre(m), inv(m);
    for (size_t i = 0; i < m; ++i) {
        H dd = H::one();
        for (size_t j = 0; j < m; ++j)
            if (j != i) dd = dd * (xs[i] - xs[j]);
        den[i] = dd;
    }
    H run = H::one();
    for (size_t i = 0; i < m; ++i) {
        pre[i] = run;
        run = run * den[i];
    }
    H all = run.inv();
    for (size_t i = m; i-- > 0;) {
        inv[i] = all * pre[i];
        all = all * den[i];
    }
    std::vector<std::vector<H>> basis(m, std::vector<H>(m, H::zero()));
    std::vector<H> q(m);
    for (size_t i = 0; i < m; ++i) {
        q[m - 1] = A[m];
        for (size_t k = m - 1; k > 0; --k) q[k - 1] = A[k] + xs[i] * q[k];
        for (size_t k = 0; k < m; ++k) basis[i][k] = q[k] * inv[i];
    }
    return basis;
}
template <class H>
inline H horner(const std::vector<H>& p, const H& x) {
    H acc = H::zero();
    for (size_t k = p.size(); k-- > 0;) acc = acc * x + p[k];
    return acc;
}

// batch_recover tables for id-sorted senders: vm[(needed-m)][m], bc[m][m]
struct RecoverTables {
    std::vector<uint32_t> vm, bc;
};
// the coefficient rows of batch_recover for id-sorted senders (robust_interpolate.rs:343-399, 423):
//   rows [0, needed - m): verify rows, row s - m = (L_i(x_s))_i for the verify points s = m .. needed-1
//   rows [needed - m, needed): coefficient rows, row k = (coefficient k of L_i)_i
template <class H>
inline std::vector<std::vector<H>> recover_coeff_rows(const DomainInv<H>& D, const std::vector<size_t>& sorted_ids, size_t d, size_t t) {
    const size_t m = d + 1, needed = d + t + 1;
    std::vector<H> invden;
    auto basis = lagrange_basis_ids<H>(D, sorted_ids.data(), m, nullptr, &invden);
    std::vector<std::vector<H>> rows;
    for (size_t s = m; s < needed; ++s) {
        std::vector<H> row(m);
        lagrange_eval_row(D, sorted_ids.data(), m, invden, sorted_ids[s], row.data());
        rows.push_back(row);
    }
    for (size_t k = 0; k < m; ++k) {
        std::vector<H> row(m);
        for (size_t i = 0; i < m; ++i) row[i] = basis[i][k];
        rows.push_back(row);
    }
    return rows;
}
template <class H>
inline std::vector<std::vector<H>> recover_coeff_rows(const std::vector<size_t>& sorted_ids, size_t n, size_t d, size_t t) {
    return recover_coeff_rows<H>(DomainInv<H>(n), sorted_ids, d, t);
}
template <class H = HFr>
inline RecoverTables build_recover_tables(const DomainInv<H>& D, const std::vector<size_t>& sorted_ids, size_t d, size_t t, int impl) {
    const size_t m = d + 1, needed = d + t + 1;
    const auto rows = recover_coeff_rows<H>(D, sorted_ids, d, t);
    RecoverTables T;
    for (size_t r = 0; r < needed - m; ++r)
        for (size_t i = 0; i < m; ++i) put_const(T.vm, rows[r][i], impl);
    for (size_t k = 0; k < m; ++k)
        for (size_t i = 0; i < m; ++i) put_const(T.bc, rows[needed - m + k][i], impl);
    return T;
}
template <class H = HFr>
inline RecoverTables build_recover_tables(const std::vector<size_t>& sorted_ids, size_t n, size_t d, size_t t, int impl) {
    return build_recover_tables<H>(DomainInv<H>(n), sorted_ids, d, t, impl);
}

// "Second chance" tables (kernels_recover.hpp, k_second_chance): for a window W of m consecutive sorted senders,
// the rows L_i^W(x_s) for every other position s of the OEC prefix [0, P), ascending, followed by the coefficient
// rows of the basis.  Window A = [0, m) is the optimistic path's interpolation set again (its verify rows extended
// from `needed` to P); the others exist when they fit the prefix.
constexpr int SECOND_MAX_WINDOWS = 4;
// candidate windows of m consecutive positions inside the prefix [0, P): the lowest m, the next m, the last m and one
// straddling the first two -- any window free of liars yields the answer, so more (overlapping) windows only widen the
// set of liar patterns that never reach OEC/Gao
inline std::vector<int> second_windows(size_t m, size_t P) {
    std::vector<int> ws;
    auto add = [&](size_t o) {
        if (o + m > P) return;
        for (int x : ws)
            if ((size_t)x == o) return;
        if ((int)ws.size() < SECOND_MAX_WINDOWS) ws.push_back((int)o);
    };
    add(0);
    add(m);
    add(P >= m ? P - m : 0);
    add(m / 2 + (m & 1));
    return ws;
}
struct SecondTables {
    std::vector<uint32_t> words;
    int n_windows = 0;
    int win_start[SECOND_MAX_WINDOWS] = {0, 0, 0, 0};
    uint32_t ev_off[SECOND_MAX_WINDOWS] = {0, 0, 0, 0}, bc_off[SECOND_MAX_WINDOWS] = {0, 0, 0, 0};
    // offsets are a pure function of the shape (m, P, limbs per constant)
    void layout(size_t m, size_t P, size_t nl) {
        n_windows = 0;
        uint32_t off = 0;
        for (int ws : second_windows(m, P)) {
            const int w = n_windows++;
            win_start[w] = ws;
            ev_off[w] = off;
            off += (uint32_t)((P - m) * m * nl);
            bc_off[w] = off;
            off += (uint32_t)(m * m * nl);
        }
    }
};
template <class H = HFr>
inline SecondTables build_second_tables(const DomainInv<H>& D, const std::vector<size_t>& sorted_ids, size_t d, size_t P, int impl) {
    const size_t m = d + 1;
    SecondTables T;
    std::vector<H> row(m), invden;
    for (int wsi : second_windows(m, P)) {
        const size_t ws = (size_t)wsi;
        auto basis = lagrange_basis_ids<H>(D, sorted_ids.data() + ws, m, nullptr, &invden);
        const int w = T.n_windows++;
        T.win_start[w] = wsi;
        T.ev_off[w] = (uint32_t)T.words.size();
        for (size_t s = 0; s < P; ++s) {
            if (s >= ws && s < ws + m) continue;
            lagrange_eval_row(D, sorted_ids.data() + ws, m, invden, sorted_ids[s], row.data());
            for (size_t i = 0; i < m; ++i) put_const(T.words, row[i], impl);
        }
        T.bc_off[w] = (uint32_t)T.words.size();
        for (size_t k = 0; k < m; ++k)
            for (size_t i = 0; i < m; ++i) put_const(T.words, basis[i][k], impl);
    }
    return T;
}
template <class H = HFr>
inline SecondTables build_second_tables(const std::vector<size_t>& sorted_ids, size_t n, size_t d, size_t P, int impl) {
    return build_second_tables<H>(DomainInv<H>(n), sorted_ids, d, P, impl);
}

}  // namespace hbmpc
