// The reference's own unit tests for the path, restated against the C++ mirror (include/hbmpc_shares.hpp):
// same shapes, same assertions, every arithmetic step on the GPU through the C ABI.
// Sources (relative to /root/reference/mpc/src): common/share/mod.rs:78-173, common/share/shamir.rs:242-459,
// honeybadger/robust_interpolate/robust_interpolate.rs:629-968, ffi/tests/secret_share.c:64-118.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <functional>
#include <memory>
#include <vector>

#include "hbmpc_shares.hpp"

using namespace hbmpc;
using RS = ShamirShare<Robust>;
using NS = ShamirShare<NonRobust>;

static int g_failed = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
            ++g_failed;                                                          \
        }                                                                        \
    } while (0)
#define RUN(t)                     \
    do {                           \
        std::printf("%s\n", #t);   \
        t();                       \
    } while (0)

// ark_std::test_rng() stand-in: a fixed-seed SplitMix64 stream, rejection-sampled below r
static Rng test_rng(uint64_t seed = 0x5EED) {
    auto state = std::make_shared<uint64_t>(seed);
    return [state]() {
        static const uint64_t R[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
        for (;;) {
            U256 u;
            for (int k = 0; k < 4; ++k) {
                uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
                u.data[k] = z ^ (z >> 31);
            }
            u.data[3] &= 0x7fffffffffffffffULL;
            bool lt = false;
            for (int k = 3; k >= 0; --k)
                if (u.data[k] != R[k]) {
                    lt = u.data[k] < R[k];
                    break;
                }
            if (lt) return Fr(u);
        }
    };
}
static Fr evaluate(const std::vector<Fr>& coeffs, const Fr& x) {  // DensePolynomial::evaluate
    Fr acc = Fr::zero();
    for (size_t k = coeffs.size(); k-- > 0;) acc = acc * x + coeffs[k];
    return acc;
}
template <class F>
static void combinations(size_t n, size_t k, F f) {  // itertools::combinations
    std::vector<size_t> idx(k);
    std::function<void(size_t, size_t)> rec = [&](size_t pos, size_t start) {
        if (pos == k) return f(idx);
        for (size_t i = start; i < n; ++i) {
            idx[pos] = i;
            rec(pos + 1, i + 1);
        }
    };
    rec(0, 0);
}

// ---- common/share/mod.rs -------------------------------------------------------------------------------
static void test_make_vandermonde_basic() {  // :87-136
    const size_t n = 4, t = 2;
    auto vandermonde = make_vandermonde(n, t).unwrap();
    CHECK(vandermonde.size() == n);
    for (auto& row : vandermonde) CHECK(row.size() == t + 1);
    CHECK(vandermonde[0][0] == Fr::one() && vandermonde[0][1] == Fr::one() && vandermonde[0][2] == Fr::one());
    const Fr alpha_1 = domain_element(n, 1);
    CHECK(vandermonde[1][0] == Fr::one() && vandermonde[1][1] == alpha_1 && vandermonde[1][2] == alpha_1 * alpha_1);
    CHECK(vandermonde[2][1] == domain_element(n, 2).pow(1));
    CHECK(vandermonde[3][2] == domain_element(n, 3).pow(2));
}
static void test_apply_vandermonde_basic() {  // :138-172
    const size_t n = 4, t = 2;
    auto vandermonde = make_vandermonde(n, t).unwrap();
    std::vector<RS> shares = {RS::make(Fr::from(1), 0, 2), RS::make(Fr::from(2), 0, 2), RS::make(Fr::from(3), 0, 2)};
    auto y_values = apply_vandermonde(vandermonde, shares).unwrap();
    CHECK(y_values.size() == n);
    for (size_t j = 0; j < n; ++j) {
        const Fr a = domain_element(n, j);
        const Fr expected = shares[0].share * a.pow(0) + shares[1].share * a.pow(1) + shares[2].share * a.pow(2);
        CHECK(y_values[j].share == expected);
        CHECK(y_values[j].id == 0 && y_values[j].degree == 2);  // keeps the INPUT id / degree
    }
    shares.pop_back();
    CHECK(apply_vandermonde(vandermonde, shares).unwrap_err() == InvalidInput);  // :59-64
}

// ---- common/share/shamir.rs ----------------------------------------------------------------------------
static void general_evaluation_domain_does_not_contain_zero() {  // :453-458
    auto v = make_vandermonde(100, 1).unwrap();
    for (auto& row : v) CHECK(!row[1].is_zero());
}
static void should_recover_secret() {  // :250-258
    const Fr secret = Fr::from(918520);
    Rng rng = test_rng();
    auto shares = NonRobustShare::compute_shares(secret, 6, 5, nullptr, rng).unwrap();
    CHECK(NonRobustShare::recover_secret(shares, 6, 0).unwrap().second == secret);
}
static void should_add_shares() {  // :260-275
    Rng rng = test_rng();
    auto s1 = NonRobustShare::compute_shares(Fr::from(10), 6, 5, nullptr, rng).unwrap();
    auto s2 = NonRobustShare::compute_shares(Fr::from(20), 6, 5, nullptr, rng).unwrap();
    std::vector<NS> added;
    for (size_t i = 0; i < 6; ++i) added.push_back((s1[i] + s2[i]).unwrap());
    CHECK(NonRobustShare::recover_secret(added, 6, 0).unwrap().second == Fr::from(30));
}
static void should_multiply_scalar() {  // :277-290
    Rng rng = test_rng();
    auto shares = NonRobustShare::compute_shares(Fr::from(55), 8, 5, nullptr, rng).unwrap();
    std::vector<NS> tripled;
    for (auto& s : shares) tripled.push_back((s * Fr::from(3)).unwrap());
    CHECK(NonRobustShare::recover_secret(tripled, 8, 0).unwrap().second == Fr::from(165));
}
static void test_degree_mismatch() {  // :292-309
    Rng rng = test_rng();
    auto shares = NonRobustShare::compute_shares(Fr::from(918520), 6, 5, nullptr, rng).unwrap();
    shares[2].degree = 4;
    CHECK(NonRobustShare::recover_secret(shares, 6, 0).unwrap_err() == DegreeMismatch);
}
static void test_insufficient_shares() {  // :311-326
    Rng rng = test_rng();
    auto shares = NonRobustShare::compute_shares(Fr::from(918520), 3, 2, nullptr, rng).unwrap();
    std::vector<NS> fewer(shares.begin() + 1, shares.end());
    CHECK(NonRobustShare::recover_secret(fewer, 3, 0).unwrap_err() == InsufficientShares);
}
static void test_id_mis_match() {  // :328-348
    Rng rng = test_rng();
    auto s1 = NonRobustShare::compute_shares(Fr::from(10), 6, 5, nullptr, rng).unwrap();
    auto s2 = NonRobustShare::compute_shares(Fr::from(20), 6, 5, nullptr, rng).unwrap();
    std::vector<size_t> ids2 = {7, 8, 9, 4, 5, 6};
    for (auto& s : s2) {
        s.id = ids2.back();
        ids2.pop_back();
    }
    CHECK((s1[0] + s2[0]).unwrap_err() == IdMismatch);
}

// ---- honeybadger/robust_interpolate/robust_interpolate.rs ----------------------------------------------
static void test_robust_interpolate_fnt_optimistic_case() {  // :645-680 (through recover_secret: fnt is private)
    const size_t n = 16, t = 2;
    const std::vector<Fr> coeffs = {Fr::from(7), Fr::from(3), Fr::from(5)};
    std::vector<RS> shares;
    for (size_t i = 0; i < n; ++i) shares.push_back(RS::make(evaluate(coeffs, domain_element(n, i)), i, t));
    std::vector<RS> used(shares.begin(), shares.begin() + 2 * t + 1);
    auto rec = RobustShare::recover_secret(used, n, t);
    CHECK(rec.is_ok());
    CHECK(rec.unwrap().first.size() == coeffs.size());
    for (size_t k = 0; k < coeffs.size(); ++k) CHECK(rec.unwrap().first[k] == coeffs[k]);
}
static void test_reed_solomon_erasure() {  // :682-704
    Rng rng = test_rng();
    const size_t t = 2, n = 8;
    const Fr secret = Fr::from(42);
    auto shares = RobustShare::compute_shares(secret, n, t, nullptr, rng).unwrap();
    std::vector<Fr> erased;
    for (auto& s : shares) erased.push_back(s.share);
    const std::vector<size_t> erasures = {1, 2};
    for (size_t i : erasures) erased[i] = Fr::zero();
    CHECK(gao_rs_decode(erased, t + 1, n, erasures).unwrap()[0] == secret);
}
static void test_reed_solomon_error() {  // :705-726
    Rng rng = test_rng();
    const size_t t = 2, n = 10;
    const Fr secret = Fr::from(42);
    auto shares = RobustShare::compute_shares(secret, n, t, nullptr, rng).unwrap();
    std::vector<Fr> corrupted;
    for (auto& s : shares) corrupted.push_back(s.share);
    corrupted[2] += Fr::from(5);
    corrupted[4] += Fr::from(3);
    CHECK(gao_rs_decode(corrupted, t + 1, n, {}).unwrap()[0] == secret);
}
static void test_reed_solomon_error_all_triples() {  // :727-756
    Rng rng = test_rng();
    const size_t t = 3, n = 10;
    const Fr secret = Fr::from(42);
    auto shares = RobustShare::compute_shares(secret, n, t, nullptr, rng).unwrap();
    combinations(n, 3, [&](const std::vector<size_t>& triple) {
        std::vector<Fr> corrupted;
        for (auto& s : shares) corrupted.push_back(s.share);
        corrupted[triple[0]] += Fr::from(5);
        corrupted[triple[1]] += Fr::from(3);
        corrupted[triple[2]] += Fr::from(3);
        CHECK(gao_rs_decode(corrupted, t + 1, n, {}).unwrap()[0] == secret);
    });
}
static void test_oec_protocol() {  // :757-789 (oec_decode is private: reached through recover_secret)
    Rng rng = test_rng();
    const size_t t = 2, n = 10;
    const Fr secret = Fr::from(42);
    auto shares = RobustShare::compute_shares(secret, n, t, nullptr, rng).unwrap();
    shares[0].share += Fr::from(999);
    shares[5].share += Fr::from(999);
    auto result = RobustShare::recover_secret(shares, n, t);
    CHECK(result.is_ok());
    CHECK(result.unwrap().second == secret);
}
static void test_robust_interpolate_full() {  // :790-826
    Rng rng = test_rng();
    const size_t t = 3, n = 10;
    const Fr secret = Fr::from(42);
    auto shares = RobustShare::compute_shares(secret, n, t, nullptr, rng).unwrap();
    for (size_t i : {size_t(1), size_t(4)}) shares[i] = (shares[i] + RS::make(Fr::from(7), i, t)).unwrap();
    auto result = RobustShare::recover_secret(shares, n, t);
    CHECK(result.is_ok());
    CHECK(result.unwrap().second == secret);
}
static void test_robust_interpolate_all_corruption_combinations() {  // :827-876
    Rng rng = test_rng();
    const size_t t = 2, n = 7;
    const Fr secret = Fr::from(42);
    auto base = RobustShare::compute_shares(secret, n, t, nullptr, rng).unwrap();
    for (size_t k = 1; k <= t; ++k)
        combinations(n, k, [&](const std::vector<size_t>& idx) {
            auto shares = base;
            for (size_t i : idx) shares[i].share += Fr::from(999);
            auto result = RobustShare::recover_secret(shares, n, t);
            CHECK(result.is_ok());
            if (result.is_ok()) CHECK(result.unwrap().second == secret);
        });
}
static void test_batch_recover_secret_matches_per_chunk() {  // :880-927
    Rng rng = test_rng();
    const size_t n = 10, t = 3, degree = t, batch_len = 16;
    std::vector<std::vector<Fr>> polys(batch_len);
    for (auto& p : polys)
        for (size_t k = 0; k <= degree; ++k) p.push_back(rng());
    std::vector<std::pair<size_t, std::vector<Fr>>> evals_by_sender;
    for (size_t id = 0; id < n; ++id) {
        const Fr x = domain_element(n, id);
        std::vector<Fr> v;
        for (auto& p : polys) v.push_back(evaluate(p, x));
        evals_by_sender.emplace_back(id, v);
    }
    std::reverse(evals_by_sender.begin(), evals_by_sender.end());
    auto batched = batch_recover_secret(evals_by_sender, n, degree, t).unwrap();
    CHECK(batched.size() == batch_len);
    for (size_t c = 0; c < batch_len; ++c) {
        std::vector<RS> shares;
        for (auto& e : evals_by_sender) shares.push_back(RS::make(e.second[c], e.first, degree));
        auto per_chunk = RobustShare::recover_secret(shares, n, t).unwrap().first;
        per_chunk.resize(degree + 1, Fr::zero());
        CHECK(batched[c] == per_chunk);
        CHECK(batched[c][0] == polys[c][0]);
    }
}
static void test_batch_recover_secret_with_corruption() {  // :931-967
    Rng rng = test_rng();
    const size_t n = 10, t = 3, degree = t, batch_len = 8;
    std::vector<std::vector<Fr>> polys(batch_len);
    for (auto& p : polys)
        for (size_t k = 0; k <= degree; ++k) p.push_back(rng());
    std::vector<std::pair<size_t, std::vector<Fr>>> evals_by_sender;
    for (size_t id = 0; id < n; ++id) {
        const Fr x = domain_element(n, id);
        std::vector<Fr> v;
        for (auto& p : polys) v.push_back(evaluate(p, x));
        evals_by_sender.emplace_back(id, v);
    }
    for (size_t bad = 0; bad < t; ++bad)
        for (size_t c = 0; c < batch_len; ++c) evals_by_sender[bad].second[c] += Fr::from((c + 1) * 7 + bad);
    auto batched = batch_recover_secret(evals_by_sender, n, degree, t).unwrap();
    for (size_t c = 0; c < batch_len; ++c) CHECK(batched[c][0] == polys[c][0]);
}

// ---- ffi/tests/secret_share.c:64-118 against this ABI ---------------------------------------------------
static void c_abi_secret_share_roundtrip() {
    Rng rng = test_rng();
    const Fr secret(U256{{3, 3, 22, 22}});
    auto shares = RobustShare::compute_shares(secret, 6, 2, nullptr, rng).unwrap();
    CHECK(shares.size() == 6);
    auto rec = RobustShare::recover_secret(shares, 6, 1).unwrap();
    CHECK(rec.second == secret && rec.first.size() == 3);
    CHECK(RobustShare::compute_shares(secret, 2, 2, nullptr, rng).unwrap_err() == InvalidInput);  // n <= degree
}


// ---- the same properties over the generic Scheme<FieldTraits>, instantiated for BOTH fields ----------------
// (the reference's functions are generic over F; its small-field nodes run them over GoldilocksField)
template <class S>
static typename S::Rng field_rng(uint64_t seed);
template <>
FrScheme::Rng field_rng<FrScheme>(uint64_t seed) { return test_rng(seed); }
template <>
GlScheme::Rng field_rng<GlScheme>(uint64_t seed) {
    auto state = std::make_shared<uint64_t>(seed);
    return [state]() {
        for (;;) {  // rejection sampling below p = 2^64 - 2^32 + 1
            uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            if (z < 0xFFFFFFFF00000001ULL) return GlScheme::F(z);
        }
    };
}
template <class S>
static void generic_share_recover_and_operators() {
    using F = typename S::F;
    using RSh = typename S::template ShamirShare<Robust>;
    auto rng = field_rng<S>(7);
    const size_t n = 10, t = 3;
    const F secret = F::from(918520);
    auto shares = S::RobustShare::compute_shares(secret, n, t, nullptr, rng).unwrap();
    CHECK(shares.size() == n && shares[4].id == 4 && shares[4].degree == t);
    auto rec = S::RobustShare::recover_secret(shares, n, t).unwrap();
    CHECK(rec.second == secret && rec.first.size() == t + 1);
    // up to t lies are corrected, t + 1 are not (n = 3t + 1)
    auto bad = shares;
    for (size_t i = 0; i < t; ++i) bad[2 * i].share += F::from(5 + i);
    CHECK(S::RobustShare::recover_secret(bad, n, t).unwrap().second == secret);
    bad[9].share += F::one();
    CHECK(S::RobustShare::recover_secret(bad, n, t).is_err());
    // share algebra: (a + b) and (a * c) open to the sums / products; mismatches report degree before id
    auto sh2 = S::RobustShare::compute_shares(F::from(42), n, t, nullptr, rng).unwrap();
    std::vector<RSh> sum, scaled;
    for (size_t i = 0; i < n; ++i) {
        sum.push_back((shares[i] + sh2[i]).unwrap());
        scaled.push_back((shares[i] * F::from(3)).unwrap());
    }
    CHECK(S::RobustShare::recover_secret(sum, n, t).unwrap().second == F::from(918520 + 42));
    CHECK(S::RobustShare::recover_secret(scaled, n, t).unwrap().second == F::from(3 * 918520));
    CHECK((shares[0] + sh2[1]).unwrap_err() == IdMismatch);
    CHECK((shares[0] + RSh(F::one(), 1, t + 1)).unwrap_err() == DegreeMismatch);
    CHECK(shares[0].share_mul(sh2[0]).unwrap().degree == 2 * t);
    CHECK(S::RobustShare::compute_shares(secret, 3, 3, nullptr, rng).unwrap_err() == InvalidInput);
    // non-robust: plain interpolation through all shares, degree check
    auto ns = S::NonRobustShare::compute_shares(secret, 6, 5, nullptr, rng).unwrap();
    CHECK(S::NonRobustShare::recover_secret(ns, 6, 0).unwrap().second == secret);
    ns.pop_back();
    CHECK(S::NonRobustShare::recover_secret(ns, 6, 0).unwrap_err() == InsufficientShares);
}
// seeded dealer: sharings of many secrets in one call with device-drawn coefficients ("hbmpc-chacha20-v1")
template <class S>
static void generic_seeded_sharing() {
    using F = typename S::F;
    using RSh = typename S::template ShamirShare<Robust>;
    const size_t n = 13, t = 4, B = 40;
    uint8_t seed[32], other[32];
    for (int i = 0; i < 32; ++i) seed[i] = (uint8_t)(3 * i + 1), other[i] = (uint8_t)(5 * i + 2);
    std::vector<F> secrets;
    for (size_t b = 0; b < B; ++b) secrets.push_back(F::from(1000 + 17 * b));
    auto sh = S::RobustShare::compute_shares_seeded(secrets, n, t, seed).unwrap();
    CHECK(sh.size() == n && sh[0].size() == B && sh[5][7].id == 5 && sh[5][7].degree == t);
    for (size_t b = 0; b < B; b += 7) {  // every column opens to its secret, with t lies corrected
        std::vector<RSh> col;
        for (size_t j = 0; j < n; ++j) col.push_back(sh[j][b]);
        for (size_t i = 0; i < t; ++i) col[3 * i].share += F::from(9 + i);
        auto rec = S::RobustShare::recover_secret(col, n, t).unwrap();
        CHECK(rec.second == secrets[b] && rec.first.size() == t + 1 && !(rec.first[t] == F::zero()));
    }
    // a function of (seed, index) only: same call -> same shares; the index window can be split over calls;
    // another seed or another index -> other coefficients
    auto again = S::RobustShare::compute_shares_seeded(secrets, n, t, seed).unwrap();
    std::vector<F> tail(secrets.begin() + 10, secrets.end());
    auto split = S::RobustShare::compute_shares_seeded(tail, n, t, seed, 10).unwrap();
    auto shifted = S::RobustShare::compute_shares_seeded(secrets, n, t, seed, 1).unwrap();
    auto reseeded = S::RobustShare::compute_shares_seeded(secrets, n, t, other).unwrap();
    for (size_t j = 0; j < n; ++j)
        for (size_t b = 0; b < B; ++b) {
            CHECK(again[j][b].share == sh[j][b].share);
            if (b >= 10) CHECK(split[j][b - 10].share == sh[j][b].share);
            if (j) CHECK(!(shifted[j][b].share == sh[j][b].share) && !(reseeded[j][b].share == sh[j][b].share));
        }
    CHECK(S::RobustShare::compute_shares_seeded(secrets, 4, 4, seed).unwrap_err() == InvalidInput);
}

template <class S>
static void generic_vandermonde_and_batch_recover() {
    using F = typename S::F;
    using RSh = typename S::template ShamirShare<Robust>;
    auto rng = field_rng<S>(11);
    const size_t n = 7, t = 2, d = 2, G = 5;
    auto vdm = S::make_vandermonde(n, d).unwrap();
    for (size_t j = 0; j < n; ++j) {
        CHECK(vdm[j][0] == F::one() && vdm[j][1] == S::domain_element(n, j) && vdm[j][2] == vdm[j][1] * vdm[j][1]);
        CHECK(!vdm[j][1].is_zero());
    }
    // BatchRecon's encode: chunks of d + 1 shares of one party -> one evaluation per recipient
    std::vector<std::vector<F>> polys(G, std::vector<F>(d + 1));
    std::vector<std::pair<size_t, std::vector<F>>> by_sender(n);
    for (size_t j = 0; j < n; ++j) by_sender[j].first = n - 1 - j;  // arrival order reversed
    for (size_t c = 0; c < G; ++c) {
        std::vector<RSh> chunk;
        for (size_t k = 0; k <= d; ++k) {
            polys[c][k] = rng();
            chunk.emplace_back(polys[c][k], 3, t);
        }
        auto y = S::template apply_vandermonde<Robust>(vdm, chunk).unwrap();
        CHECK(y.size() == n && y[0].id == 3 && y[0].degree == t);
        for (size_t j = 0; j < n; ++j) by_sender[n - 1 - j].second.push_back(y[j].share);
    }
    auto out = S::batch_recover_secret(by_sender, n, d, t).unwrap();
    for (size_t c = 0; c < G; ++c) CHECK(out[c] == polys[c]);
    // t corrupted senders: every chunk takes the OEC/Gao path and still opens to the same polynomials
    for (size_t b = 0; b < t; ++b)
        for (size_t c = 0; c < G; ++c) by_sender[b].second[c] += F::from(7 * (c + 1) + b);
    out = S::batch_recover_secret(by_sender, n, d, t).unwrap();
    for (size_t c = 0; c < G; ++c) CHECK(out[c][0] == polys[c][0]);
    by_sender.resize(d + t);  // not enough evaluations
    CHECK(S::batch_recover_secret(by_sender, n, d, t).unwrap_err() == InvalidInput);
    // Reed-Solomon with erasures and errors on its own
    std::vector<F> msg = {F::from(3), F::from(1), F::from(4)}, cw;
    for (size_t i = 0; i < 10; ++i) {
        F acc = F::zero();
        const F x = S::domain_element(10, i);
        for (size_t k = msg.size(); k-- > 0;) acc = acc * x + msg[k];
        cw.push_back(acc);
    }
    cw[1] += F::one();
    cw[8] += F::from(99);
    CHECK(S::gao_rs_decode(cw, 3, 10, {4, 6}).unwrap() == msg);
}

int main() {
    RUN(test_make_vandermonde_basic);
    RUN(test_apply_vandermonde_basic);
    RUN(general_evaluation_domain_does_not_contain_zero);
    RUN(should_recover_secret);
    RUN(should_add_shares);
    RUN(should_multiply_scalar);
    RUN(test_degree_mismatch);
    RUN(test_insufficient_shares);
    RUN(test_id_mis_match);
    RUN(test_robust_interpolate_fnt_optimistic_case);
    RUN(test_reed_solomon_erasure);
    RUN(test_reed_solomon_error);
    RUN(test_reed_solomon_error_all_triples);
    RUN(test_oec_protocol);
    RUN(test_robust_interpolate_full);
    RUN(test_robust_interpolate_all_corruption_combinations);
    RUN(test_batch_recover_secret_matches_per_chunk);
    RUN(test_batch_recover_secret_with_corruption);
    RUN(c_abi_secret_share_roundtrip);
    RUN(generic_share_recover_and_operators<FrScheme>);
    RUN(generic_share_recover_and_operators<GlScheme>);
    RUN(generic_vandermonde_and_batch_recover<FrScheme>);
    RUN(generic_vandermonde_and_batch_recover<GlScheme>);
    RUN(generic_seeded_sharing<FrScheme>);
    RUN(generic_seeded_sharing<GlScheme>);
    std::printf(g_failed ? "%d CHECKS FAILED\n" : "all reference unit tests passed (%d failures)\n", g_failed);
    return g_failed ? 1 : 0;
}
