// hbmpc_shares.hpp -- C++ host-side mirror of the reference's interface for the path, above the C ABI of
// hbmpc_hip.h.  (The reference's host language is Rust and there is no Rust toolchain in the build image,
// so the host side above the ABI is written in C++; INTEGRATION.md holds the Rust shim source.)
//
// Same names, argument meaning and error behaviour as the reference (paths relative to mpc/src/):
//   ShamirShare<P> {share, id, degree} and its operators          common/mod.rs:92-99, 167-300
//   SecretSharingScheme: compute_shares / recover_secret          common/mod.rs:101-128
//   RobustShare  = ShamirShare<Robust>                            honeybadger/robust_interpolate/robust_interpolate.rs:17-157
//   NonRobustShare                                                common/share/shamir.rs:131-240
//   make_vandermonde / apply_vandermonde                          common/share/mod.rs:31-76
//   batch_recover_secret                                          robust_interpolate.rs:284-443
//   gao_rs_decode (private there, called by its unit tests)       robust_interpolate.rs:456-538
// Errors: the reference returns Result<_, ShareError | InterpolateError>; here Result<T> carries the
// ShareErrorCode the reference's own C ABI maps those enums to (ffi/c_bindings/share/mod.rs:18-37).
// Every arithmetic step runs on the GPU through the C ABI; there is no CPU arithmetic in this header.
//
// The reference's code is generic over the field (`F: FftField`): it instantiates the same functions for
// bls12-381 Fr and for GoldilocksField (common/math/goldilocks.rs; honeybadger/mod.rs:316-324).  So is this
// mirror: Scheme<FieldTraits> holds every type and function; namespace hbmpc exposes the Fr instantiation under
// the reference's names, namespace hbmpc::gl the Goldilocks one (hbmpc_gl_* entry points).
#pragma once
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hbmpc_hip.h"

namespace hbmpc {

template <class T>
struct Result {
    ShareErrorCode code = ShareSuccess;
    T value{};
    bool is_ok() const { return code == ShareSuccess; }
    bool is_err() const { return code != ShareSuccess; }
    const T& unwrap() const {
        if (is_err()) throw std::runtime_error("unwrap on error " + std::to_string((int)code));
        return value;
    }
    ShareErrorCode unwrap_err() const {
        if (is_ok()) throw std::runtime_error("unwrap_err on Ok");
        return code;
    }
    static Result ok(T v) { return Result{ShareSuccess, std::move(v)}; }
    static Result err(ShareErrorCode c) { return Result{c, T{}}; }
};

struct Robust {};
struct NonRobust {};

namespace detail {
inline hbmpc_ctx* make_context(FieldKind kind) {
    hbmpc_ctx* c = nullptr;
    const char* dev = std::getenv("HBMPC_DEVICE");
    const ShareErrorCode rc = hbmpc_create(dev ? std::atoi(dev) : 0, kind, &c);
    if (rc != ShareSuccess) throw std::runtime_error(std::string("hbmpc_create failed: ") + hbmpc_last_error(nullptr));
    return c;
}
}  // namespace detail

// ---- the two fields: element representation at the ABI + the entry points that serve it -----------------
struct FrTraits {  // ark_bls12_381::Fr <-> U256 (ffi/c_bindings/mod.rs:37-49)
    using repr = U256;
    static hbmpc_ctx* context() {  // one hbmpc_ctx per process and field (device 0 unless HBMPC_DEVICE is set)
        static hbmpc_ctx* ctx = detail::make_context(Bls12_381Fr);
        return ctx;
    }
    static repr from_u64(uint64_t x) { return U256{{x, 0, 0, 0}}; }
    static bool is_zero(const repr& v) { return (v.data[0] | v.data[1] | v.data[2] | v.data[3]) == 0; }
    static bool eq(const repr& a, const repr& b) { return std::memcmp(a.data, b.data, 32) == 0; }
    static constexpr auto compute_shares = hbmpc_compute_shares;
    static constexpr auto compute_shares_seeded = hbmpc_compute_shares_seeded;
    static constexpr auto make_vandermonde = hbmpc_make_vandermonde;
    static constexpr auto batch_recover = hbmpc_batch_recover;
    static constexpr auto recover_secret = hbmpc_recover_secret;
    static constexpr auto nonrobust_recover_secret = hbmpc_nonrobust_recover_secret;
    static constexpr auto gao_rs_decode = hbmpc_gao_rs_decode;
    static constexpr auto fr_op = hbmpc_fr_op;
};
struct GlTraits {  // GoldilocksField = Fp64 <-> u64 (common/math/goldilocks.rs:4-13)
    using repr = uint64_t;
    static hbmpc_ctx* context() {
        static hbmpc_ctx* ctx = detail::make_context(Goldilocks64);
        return ctx;
    }
    static repr from_u64(uint64_t x) { return x % 0xFFFFFFFF00000001ull; }
    static bool is_zero(const repr& v) { return v == 0; }
    static bool eq(const repr& a, const repr& b) { return a == b; }
    static constexpr auto compute_shares = hbmpc_gl_compute_shares;
    static constexpr auto compute_shares_seeded = hbmpc_gl_compute_shares_seeded;
    static constexpr auto make_vandermonde = hbmpc_gl_make_vandermonde;
    static constexpr auto batch_recover = hbmpc_gl_batch_recover;
    static constexpr auto recover_secret = hbmpc_gl_recover_secret;
    static constexpr auto nonrobust_recover_secret = hbmpc_gl_nonrobust_recover_secret;
    static constexpr auto gao_rs_decode = hbmpc_gl_gao_rs_decode;
    static constexpr auto fr_op = hbmpc_gl_fr_op;
};

template <class FT>
struct Scheme {
    using repr = typename FT::repr;
    static hbmpc_ctx* context() { return FT::context(); }

    // ---- F: canonical element; arithmetic goes through the field's fr_op entry point (N = 1) -------------
    struct F {
        repr v = FT::from_u64(0);
        F() = default;
        explicit F(const repr& u) : v(u) {}
        static F from(uint64_t x) { return F(FT::from_u64(x)); }
        static F zero() { return F(); }
        static F one() { return from(1); }
        bool is_zero() const { return FT::is_zero(v); }
        bool operator==(const F& o) const { return FT::eq(v, o.v); }
        bool operator!=(const F& o) const { return !(*this == o); }
        static F op(int which, const F& a, const F& b) {
            F r;
            if (FT::fr_op(FT::context(), which, &a.v, &b.v, 1, &r.v) != ShareSuccess) throw std::runtime_error(hbmpc_last_error(FT::context()));
            return r;
        }
        F operator+(const F& o) const { return op(0, *this, o); }
        F operator-(const F& o) const { return op(1, *this, o); }
        F operator*(const F& o) const { return op(2, *this, o); }
        F& operator+=(const F& o) { return *this = *this + o; }
        F pow(uint64_t e) const {
            F acc = one(), b = *this;
            for (; e; e >>= 1, b = b * b)
                if (e & 1) acc = acc * b;
            return acc;
        }
    };
    using Rng = std::function<F()>;  // the mirror's `&mut impl Rng`: each call is one F::rand(rng)

    // ---- ShamirShare<P> (common/mod.rs:92-99) and its operators (:167-300) --------------------------------
    template <class P>
    struct ShamirShare {
        F share;  // share[0] of the reference's [F; 1]
        size_t id = 0;
        size_t degree = 0;
        ShamirShare() = default;
        ShamirShare(F s, size_t i, size_t d) : share(s), id(i), degree(d) {}
        static ShamirShare make(F s, size_t i, size_t d) { return ShamirShare(s, i, d); }  // RobustShare::new
        bool operator==(const ShamirShare& o) const { return share == o.share && id == o.id && degree == o.degree; }

        Result<ShamirShare> operator+(const ShamirShare& o) const {  // :167-188: degree first, then id
            if (degree != o.degree) return Result<ShamirShare>::err(DegreeMismatch);
            if (id != o.id) return Result<ShamirShare>::err(IdMismatch);
            return Result<ShamirShare>::ok(ShamirShare(share + o.share, id, degree));
        }
        Result<ShamirShare> operator-(const ShamirShare& o) const {  // :220-240
            if (degree != o.degree) return Result<ShamirShare>::err(DegreeMismatch);
            if (id != o.id) return Result<ShamirShare>::err(IdMismatch);
            return Result<ShamirShare>::ok(ShamirShare(share - o.share, id, degree));
        }
        Result<ShamirShare> operator+(const F& s) const { return Result<ShamirShare>::ok(ShamirShare(share + s, id, degree)); }  // :205-218
        Result<ShamirShare> operator-(const F& s) const { return Result<ShamirShare>::ok(ShamirShare(share - s, id, degree)); }  // :242-254
        Result<ShamirShare> operator*(const F& s) const { return Result<ShamirShare>::ok(ShamirShare(share * s, id, degree)); }  // :267-280
        static ShamirShare from_scalar_sub(const F& s, const ShamirShare& o) { return ShamirShare(s - o.share, o.id, o.degree); }  // :255-265
        Result<ShamirShare> share_mul(const ShamirShare& o) const {  // :282-300: only the id is checked, degrees add
            if (id != o.id) return Result<ShamirShare>::err(IdMismatch);
            return Result<ShamirShare>::ok(ShamirShare(share * o.share, id, degree + o.degree));
        }
    };

    template <class P>
    static Result<std::vector<ShamirShare<P>>> compute_shares_impl(const F& secret, size_t n, size_t degree, Rng& rng) {
        using V = std::vector<ShamirShare<P>>;
        if (n <= degree) return Result<V>::err(InvalidInput);  // robust_interpolate.rs:59-64 / shamir.rs:165-167
        // DensePolynomial::rand(degree, rng): degree + 1 draws; then coeffs[0] = secret (:68-69)
        std::vector<repr> coeffs(degree + 1);
        for (size_t k = 0; k <= degree; ++k) coeffs[k] = rng().v;
        coeffs[0] = secret.v;
        std::vector<repr> out(n);
        const ShareErrorCode rc = FT::compute_shares(FT::context(), coeffs.data(), 1, n, degree, out.data());
        if (rc != ShareSuccess) return Result<V>::err(rc);
        V shares;
        for (size_t i = 0; i < n; ++i) shares.emplace_back(F(out[i]), i, degree);
        return Result<V>::ok(std::move(shares));
    }

    // B sharings in one call with the random coefficients drawn on the device from `seed` (contract
    // "hbmpc-chacha20-v1", hbmpc_hip.h) -- for dealers that do not need the draws of a particular host rng.
    // Result [party j][secret b].  first_index keeps stream positions disjoint when one seed serves several calls.
    template <class P>
    static Result<std::vector<std::vector<ShamirShare<P>>>> compute_shares_seeded_impl(const std::vector<F>& secrets, size_t n,
                                                                                       size_t degree, const uint8_t (&seed)[32],
                                                                                       uint64_t first_index) {
        using V = std::vector<std::vector<ShamirShare<P>>>;
        if (n <= degree) return Result<V>::err(InvalidInput);
        const size_t B = secrets.size();
        std::vector<repr> sec(B), out(n * B);
        for (size_t b = 0; b < B; ++b) sec[b] = secrets[b].v;
        const ShareErrorCode rc = FT::compute_shares_seeded(FT::context(), seed, sec.data(), B, first_index, n, degree, out.data());
        if (rc != ShareSuccess) return Result<V>::err(rc);
        V shares(n);
        for (size_t j = 0; j < n; ++j)
            for (size_t b = 0; b < B; ++b) shares[j].emplace_back(F(out[j * B + b]), j, degree);
        return Result<V>::ok(std::move(shares));
    }

    using Recovered = std::pair<std::vector<F>, F>;  // (coefficients normalised like DensePolynomial, P(0))

    template <class P, class Fn>
    static Result<Recovered> recover_impl(const std::vector<ShamirShare<P>>& shares, size_t cap, Fn call) {
        std::vector<size_t> ids, degs;
        std::vector<repr> vals;
        for (const auto& s : shares) {
            ids.push_back(s.id);
            degs.push_back(s.degree);
            vals.push_back(s.share.v);
        }
        std::vector<repr> co(cap);
        size_t nco = 0;
        repr secret = FT::from_u64(0);
        const ShareErrorCode rc = call(ids.data(), degs.data(), vals.data(), co.data(), &nco, &secret);
        if (rc != ShareSuccess) return Result<Recovered>::err(rc);
        Recovered r;
        for (size_t i = 0; i < nco; ++i) r.first.emplace_back(co[i]);
        r.second = F(secret);
        return Result<Recovered>::ok(std::move(r));
    }

    struct RobustShare : ShamirShare<Robust> {
        using ShamirShare<Robust>::ShamirShare;
        RobustShare(const ShamirShare<Robust>& s) : ShamirShare<Robust>(s) {}
        // SecretSharingScheme::compute_shares (robust_interpolate.rs:52-82); `ids` is ignored there too
        static Result<std::vector<ShamirShare<Robust>>> compute_shares(const F& secret, size_t n, size_t degree,
                                                                       const std::vector<size_t>* /*ids*/, Rng& rng) {
            return compute_shares_impl<Robust>(secret, n, degree, rng);
        }
        static Result<std::vector<std::vector<ShamirShare<Robust>>>> compute_shares_seeded(const std::vector<F>& secrets, size_t n,
                                                                                           size_t degree, const uint8_t (&seed)[32],
                                                                                           uint64_t first_index = 0) {
            return compute_shares_seeded_impl<Robust>(secrets, n, degree, seed, first_index);
        }
        // SecretSharingScheme::recover_secret (robust_interpolate.rs:94-157)
        static Result<Recovered> recover_secret(const std::vector<ShamirShare<Robust>>& shares, size_t n, size_t t) {
            return recover_impl(shares, shares.empty() ? 1 : shares[0].degree + 1,
                                [&](const size_t* ids, const size_t* degs, const repr* vals, repr* co, size_t* nco, repr* secret) {
                                    return FT::recover_secret(FT::context(), ids, degs, vals, shares.size(), n, t, co, nco, secret);
                                });
        }
    };

    struct NonRobustShare : ShamirShare<NonRobust> {
        using ShamirShare<NonRobust>::ShamirShare;
        static Result<std::vector<ShamirShare<NonRobust>>> compute_shares(const F& secret, size_t n, size_t degree,
                                                                          const std::vector<size_t>* /*ids*/, Rng& rng) {
            return compute_shares_impl<NonRobust>(secret, n, degree, rng);  // shamir.rs:158-196
        }
        static Result<Recovered> recover_secret(const std::vector<ShamirShare<NonRobust>>& shares, size_t n, size_t /*t*/) {
            return recover_impl(shares, shares.size() + 1,  // shamir.rs:199-239
                                [&](const size_t* ids, const size_t* degs, const repr* vals, repr* co, size_t* nco, repr* secret) {
                                    return FT::nonrobust_recover_secret(FT::context(), ids, degs, vals, shares.size(), n, co, nco, secret);
                                });
        }
    };

    // ---- make_vandermonde / apply_vandermonde (common/share/mod.rs:31-76) ---------------------------------
    using Matrix = std::vector<std::vector<F>>;
    static Result<Matrix> make_vandermonde(size_t n, size_t t) {
        std::vector<repr> flat(n * (t + 1));
        const ShareErrorCode rc = FT::make_vandermonde(FT::context(), n, t, flat.data());
        if (rc != ShareSuccess) return Result<Matrix>::err(rc);
        Matrix m(n, std::vector<F>(t + 1));
        for (size_t j = 0; j < n; ++j)
            for (size_t k = 0; k <= t; ++k) m[j][k] = F(flat[j * (t + 1) + k]);
        return Result<Matrix>::ok(std::move(m));
    }
    // V * shares for an ARBITRARY matrix, like the reference: rows whose length differs from the number of shares
    // are InvalidInput (:59-64); IdMismatch / DegreeMismatch come out of the share additions (:69-72); the result
    // keeps the inputs' id and degree.
    template <class P>
    static Result<std::vector<ShamirShare<P>>> apply_vandermonde(const Matrix& vandermonde, const std::vector<ShamirShare<P>>& shares) {
        using V = std::vector<ShamirShare<P>>;
        for (const auto& row : vandermonde)
            if (row.size() != shares.size()) return Result<V>::err(InvalidInput);
        if (vandermonde.empty()) return Result<V>::ok(V{});
        if (shares.empty()) throw std::out_of_range("apply_vandermonde: shares[0] (the reference panics here too)");
        for (size_t k = 1; k < shares.size(); ++k) {  // what acc + term would report, in the reference's order
            if (shares[k].degree != shares[0].degree) return Result<V>::err(DegreeMismatch);
            if (shares[k].id != shares[0].id) return Result<V>::err(IdMismatch);
        }
        const size_t n = vandermonde.size(), m = shares.size();
        // element-wise on the device: prod[j][k] = V[j][k] * x[k], then a running sum over k
        std::vector<repr> a(n * m), b(n * m), prod(n * m);
        for (size_t j = 0; j < n; ++j)
            for (size_t k = 0; k < m; ++k) {
                a[j * m + k] = vandermonde[j][k].v;
                b[j * m + k] = shares[k].share.v;
            }
        ShareErrorCode rc = FT::fr_op(FT::context(), 2, a.data(), b.data(), n * m, prod.data());
        if (rc != ShareSuccess) return Result<V>::err(rc);
        std::vector<repr> acc(n), col(n);
        for (size_t j = 0; j < n; ++j) acc[j] = prod[j * m];
        for (size_t k = 1; k < m; ++k) {
            for (size_t j = 0; j < n; ++j) col[j] = prod[j * m + k];
            rc = FT::fr_op(FT::context(), 0, acc.data(), col.data(), n, acc.data());
            if (rc != ShareSuccess) return Result<V>::err(rc);
        }
        V out;
        for (size_t j = 0; j < n; ++j) out.emplace_back(F(acc[j]), shares[0].id, shares[0].degree);
        return Result<V>::ok(std::move(out));
    }

    // ---- batch_recover_secret (robust_interpolate.rs:284-443) ---------------------------------------------
    static Result<std::vector<std::vector<F>>> batch_recover_secret(const std::vector<std::pair<size_t, std::vector<F>>>& evals_by_sender,
                                                                    size_t n, size_t degree, size_t t) {
        using V = std::vector<std::vector<F>>;
        if (n < 3 * t + 1) return Result<V>::err(InvalidInput);            // :290
        if (evals_by_sender.empty()) return Result<V>::err(InvalidInput);  // :297
        const size_t G = evals_by_sender[0].second.size();
        if (G == 0) return Result<V>::err(InvalidInput);  // :303
        for (const auto& e : evals_by_sender)
            if (e.second.size() != G) return Result<V>::err(InvalidInput);  // :306
        const size_t S = evals_by_sender.size();
        std::vector<size_t> ids(S);
        std::vector<repr> flat(S * G);
        for (size_t i = 0; i < S; ++i) {
            ids[i] = evals_by_sender[i].first;
            for (size_t c = 0; c < G; ++c) flat[i * G + c] = evals_by_sender[i].second[c].v;
        }
        std::vector<repr> co(G * (degree + 1));
        std::vector<uint32_t> nco(G);
        const ShareErrorCode rc = FT::batch_recover(FT::context(), ids.data(), S, flat.data(), G, n, degree, t, co.data(), nco.data(), nullptr);
        if (rc != ShareSuccess) return Result<V>::err(rc);
        V out(G);
        for (size_t c = 0; c < G; ++c) {
            if (nco[c] > degree + 1) throw std::runtime_error("batch_recover_secret: ncoeffs " + std::to_string(nco[c]) + " for chunk " + std::to_string(c));
            for (uint32_t k = 0; k < nco[c]; ++k) out[c].emplace_back(co[c * (degree + 1) + k]);
        }
        return Result<V>::ok(std::move(out));
    }

    // ---- gao_rs_decode (robust_interpolate.rs:456-538) ----------------------------------------------------
    static Result<std::vector<F>> gao_rs_decode(const std::vector<F>& received, size_t k, size_t n, const std::vector<size_t>& erasure_positions) {
        std::vector<repr> rec(received.size());
        for (size_t i = 0; i < received.size(); ++i) rec[i] = received[i].v;
        std::vector<repr> co(k ? k : 1);
        size_t nco = 0;
        const ShareErrorCode rc = FT::gao_rs_decode(FT::context(), rec.data(), k, n, erasure_positions.data(), erasure_positions.size(), co.data(), &nco);
        if (rc != ShareSuccess) return Result<std::vector<F>>::err(rc);
        std::vector<F> out;
        for (size_t i = 0; i < nco; ++i) out.emplace_back(co[i]);
        return Result<std::vector<F>>::ok(std::move(out));
    }

    // GeneralEvaluationDomain::<F>::new(n).element(j) as the tests use it: row j of the Vandermonde matrix, column 1
    static F domain_element(size_t n, size_t j) {
        std::vector<repr> flat(n * 2);
        if (FT::make_vandermonde(FT::context(), n, 1, flat.data()) != ShareSuccess) throw std::runtime_error("domain");
        return F(flat[j * 2 + 1]);
    }
};

// ---- the Fr instantiation under the reference's names -------------------------------------------------------
using FrScheme = Scheme<FrTraits>;
inline hbmpc_ctx* context() { return FrTraits::context(); }
using Fr = FrScheme::F;
using Rng = FrScheme::Rng;
template <class P>
using ShamirShare = FrScheme::ShamirShare<P>;
using Recovered = FrScheme::Recovered;
using RobustShare = FrScheme::RobustShare;
using NonRobustShare = FrScheme::NonRobustShare;
using Matrix = FrScheme::Matrix;
inline Result<Matrix> make_vandermonde(size_t n, size_t t) { return FrScheme::make_vandermonde(n, t); }
template <class P>
inline Result<std::vector<ShamirShare<P>>> apply_vandermonde(const Matrix& v, const std::vector<ShamirShare<P>>& shares) {
    return FrScheme::apply_vandermonde<P>(v, shares);
}
inline Result<std::vector<std::vector<Fr>>> batch_recover_secret(const std::vector<std::pair<size_t, std::vector<Fr>>>& evals_by_sender,
                                                                 size_t n, size_t degree, size_t t) {
    return FrScheme::batch_recover_secret(evals_by_sender, n, degree, t);
}
inline Result<std::vector<Fr>> gao_rs_decode(const std::vector<Fr>& received, size_t k, size_t n, const std::vector<size_t>& erasure_positions) {
    return FrScheme::gao_rs_decode(received, k, n, erasure_positions);
}
inline Fr domain_element(size_t n, size_t j) { return FrScheme::domain_element(n, j); }

// ---- the Goldilocks instantiation: hbmpc::gl::RobustShare, hbmpc::gl::batch_recover_secret, ... ---------------
using GlScheme = Scheme<GlTraits>;

}  // namespace hbmpc
