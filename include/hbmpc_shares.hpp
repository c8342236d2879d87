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
#pragma once
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hbmpc_hip.h"

namespace hbmpc {

// ---- context: one hbmpc_ctx per process (device 0 unless HBMPC_DEVICE is set) -------------------------
inline hbmpc_ctx* context() {
    static hbmpc_ctx* ctx = [] {
        hbmpc_ctx* c = nullptr;
        const char* dev = std::getenv("HBMPC_DEVICE");
        const ShareErrorCode rc = hbmpc_create(dev ? std::atoi(dev) : 0, Bls12_381Fr, &c);
        if (rc != ShareSuccess) throw std::runtime_error(std::string("hbmpc_create failed: ") + hbmpc_last_error(nullptr));
        return c;
    }();
    return ctx;
}

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

// ---- Fr: canonical element; arithmetic goes through hbmpc_fr_op (N = 1) ---------------------------------
struct Fr {
    U256 v{{0, 0, 0, 0}};
    Fr() = default;
    explicit Fr(const U256& u) : v(u) {}
    static Fr from(uint64_t x) {
        Fr f;
        f.v.data[0] = x;
        return f;
    }
    static Fr zero() { return Fr(); }
    static Fr one() { return from(1); }
    bool is_zero() const { return (v.data[0] | v.data[1] | v.data[2] | v.data[3]) == 0; }
    bool operator==(const Fr& o) const { return std::memcmp(v.data, o.v.data, 32) == 0; }
    bool operator!=(const Fr& o) const { return !(*this == o); }
    static Fr op(int which, const Fr& a, const Fr& b) {
        Fr r;
        if (hbmpc_fr_op(context(), which, &a.v, &b.v, 1, &r.v) != ShareSuccess) throw std::runtime_error(hbmpc_last_error(context()));
        return r;
    }
    Fr operator+(const Fr& o) const { return op(0, *this, o); }
    Fr operator-(const Fr& o) const { return op(1, *this, o); }
    Fr operator*(const Fr& o) const { return op(2, *this, o); }
    Fr& operator+=(const Fr& o) { return *this = *this + o; }
    Fr pow(uint64_t e) const {
        Fr acc = one(), b = *this;
        for (; e; e >>= 1, b = b * b)
            if (e & 1) acc = acc * b;
        return acc;
    }
};
using Rng = std::function<Fr()>;  // the mirror's `&mut impl Rng`: each call is one F::rand(rng)

// ---- ShamirShare<P> (common/mod.rs:92-99) and its operators (:167-300) ----------------------------------
struct Robust {};
struct NonRobust {};

template <class P>
struct ShamirShare {
    Fr share;  // share[0] of the reference's [F; 1]
    size_t id = 0;
    size_t degree = 0;
    ShamirShare() = default;
    ShamirShare(Fr s, size_t i, size_t d) : share(s), id(i), degree(d) {}
    static ShamirShare make(Fr s, size_t i, size_t d) { return ShamirShare(s, i, d); }  // RobustShare::new
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
    Result<ShamirShare> operator+(const Fr& s) const { return Result<ShamirShare>::ok(ShamirShare(share + s, id, degree)); }  // :205-218
    Result<ShamirShare> operator-(const Fr& s) const { return Result<ShamirShare>::ok(ShamirShare(share - s, id, degree)); }  // :242-254
    Result<ShamirShare> operator*(const Fr& s) const { return Result<ShamirShare>::ok(ShamirShare(share * s, id, degree)); }  // :267-280
    static ShamirShare from_scalar_sub(const Fr& s, const ShamirShare& o) { return ShamirShare(s - o.share, o.id, o.degree); }  // :255-265
    Result<ShamirShare> share_mul(const ShamirShare& o) const {  // :282-300: only the id is checked, degrees add
        if (id != o.id) return Result<ShamirShare>::err(IdMismatch);
        return Result<ShamirShare>::ok(ShamirShare(share * o.share, id, degree + o.degree));
    }
};

namespace detail {
template <class P>
inline Result<std::vector<ShamirShare<P>>> compute_shares(const Fr& secret, size_t n, size_t degree, Rng& rng) {
    using V = std::vector<ShamirShare<P>>;
    if (n <= degree) return Result<V>::err(InvalidInput);  // robust_interpolate.rs:59-64 / shamir.rs:165-167
    // DensePolynomial::rand(degree, rng): degree + 1 draws; then coeffs[0] = secret (:68-69)
    std::vector<U256> coeffs(degree + 1);
    for (size_t k = 0; k <= degree; ++k) coeffs[k] = rng().v;
    coeffs[0] = secret.v;
    std::vector<U256> out(n);
    const ShareErrorCode rc = hbmpc_compute_shares(context(), coeffs.data(), 1, n, degree, out.data());
    if (rc != ShareSuccess) return Result<V>::err(rc);
    V shares;
    for (size_t i = 0; i < n; ++i) shares.emplace_back(Fr(out[i]), i, degree);
    return Result<V>::ok(std::move(shares));
}
}  // namespace detail

using Recovered = std::pair<std::vector<Fr>, Fr>;  // (coefficients normalised like DensePolynomial, P(0))

struct RobustShare : ShamirShare<Robust> {
    using ShamirShare<Robust>::ShamirShare;
    RobustShare(const ShamirShare<Robust>& s) : ShamirShare<Robust>(s) {}
    // SecretSharingScheme::compute_shares (robust_interpolate.rs:52-82); `ids` is ignored there too
    static Result<std::vector<ShamirShare<Robust>>> compute_shares(const Fr& secret, size_t n, size_t degree,
                                                                   const std::vector<size_t>* /*ids*/, Rng& rng) {
        return detail::compute_shares<Robust>(secret, n, degree, rng);
    }
    // SecretSharingScheme::recover_secret (robust_interpolate.rs:94-157)
    static Result<Recovered> recover_secret(const std::vector<ShamirShare<Robust>>& shares, size_t n, size_t t) {
        std::vector<size_t> ids, degs;
        std::vector<U256> vals;
        for (const auto& s : shares) {
            ids.push_back(s.id);
            degs.push_back(s.degree);
            vals.push_back(s.share.v);
        }
        const size_t cap = shares.empty() ? 1 : shares[0].degree + 1;
        std::vector<U256> co(cap);
        size_t nco = 0;
        U256 secret{};
        const ShareErrorCode rc = hbmpc_recover_secret(context(), ids.data(), degs.data(), vals.data(), shares.size(), n, t,
                                                       co.data(), &nco, &secret);
        if (rc != ShareSuccess) return Result<Recovered>::err(rc);
        Recovered r;
        for (size_t i = 0; i < nco; ++i) r.first.emplace_back(co[i]);
        r.second = Fr(secret);
        return Result<Recovered>::ok(std::move(r));
    }
};

struct NonRobustShare : ShamirShare<NonRobust> {
    using ShamirShare<NonRobust>::ShamirShare;
    static Result<std::vector<ShamirShare<NonRobust>>> compute_shares(const Fr& secret, size_t n, size_t degree,
                                                                      const std::vector<size_t>* /*ids*/, Rng& rng) {
        return detail::compute_shares<NonRobust>(secret, n, degree, rng);  // shamir.rs:158-196
    }
    static Result<Recovered> recover_secret(const std::vector<ShamirShare<NonRobust>>& shares, size_t n, size_t /*t*/) {
        std::vector<size_t> ids, degs;  // shamir.rs:199-239
        std::vector<U256> vals;
        for (const auto& s : shares) {
            ids.push_back(s.id);
            degs.push_back(s.degree);
            vals.push_back(s.share.v);
        }
        std::vector<U256> co(shares.size() + 1);
        size_t nco = 0;
        U256 secret{};
        const ShareErrorCode rc = hbmpc_nonrobust_recover_secret(context(), ids.data(), degs.data(), vals.data(), shares.size(),
                                                                 n, co.data(), &nco, &secret);
        if (rc != ShareSuccess) return Result<Recovered>::err(rc);
        Recovered r;
        for (size_t i = 0; i < nco; ++i) r.first.emplace_back(co[i]);
        r.second = Fr(secret);
        return Result<Recovered>::ok(std::move(r));
    }
};

// ---- make_vandermonde / apply_vandermonde (common/share/mod.rs:31-76) -----------------------------------
using Matrix = std::vector<std::vector<Fr>>;
inline Result<Matrix> make_vandermonde(size_t n, size_t t) {
    std::vector<U256> flat(n * (t + 1));
    const ShareErrorCode rc = hbmpc_make_vandermonde(context(), n, t, flat.data());
    if (rc != ShareSuccess) return Result<Matrix>::err(rc);
    Matrix m(n, std::vector<Fr>(t + 1));
    for (size_t j = 0; j < n; ++j)
        for (size_t k = 0; k <= t; ++k) m[j][k] = Fr(flat[j * (t + 1) + k]);
    return Result<Matrix>::ok(std::move(m));
}
// V * shares for an ARBITRARY matrix, like the reference: rows whose length differs from the number of shares
// are InvalidInput (:59-64); IdMismatch / DegreeMismatch come out of the share additions (:69-72); the result
// keeps the inputs' id and degree.
template <class P>
inline Result<std::vector<ShamirShare<P>>> apply_vandermonde(const Matrix& vandermonde, const std::vector<ShamirShare<P>>& shares) {
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
    // element-wise on the device: prod[j][k] = V[j][k] * x[k], then a log-free running sum over k
    std::vector<U256> a(n * m), b(n * m), prod(n * m);
    for (size_t j = 0; j < n; ++j)
        for (size_t k = 0; k < m; ++k) {
            a[j * m + k] = vandermonde[j][k].v;
            b[j * m + k] = shares[k].share.v;
        }
    ShareErrorCode rc = hbmpc_fr_op(context(), 2, a.data(), b.data(), n * m, prod.data());
    if (rc != ShareSuccess) return Result<V>::err(rc);
    std::vector<U256> acc(n), col(n);
    for (size_t j = 0; j < n; ++j) acc[j] = prod[j * m];
    for (size_t k = 1; k < m; ++k) {
        for (size_t j = 0; j < n; ++j) col[j] = prod[j * m + k];
        rc = hbmpc_fr_op(context(), 0, acc.data(), col.data(), n, acc.data());
        if (rc != ShareSuccess) return Result<V>::err(rc);
    }
    V out;
    for (size_t j = 0; j < n; ++j) out.emplace_back(Fr(acc[j]), shares[0].id, shares[0].degree);
    return Result<V>::ok(std::move(out));
}

// ---- batch_recover_secret (robust_interpolate.rs:284-443) -----------------------------------------------
inline Result<std::vector<std::vector<Fr>>> batch_recover_secret(const std::vector<std::pair<size_t, std::vector<Fr>>>& evals_by_sender,
                                                                 size_t n, size_t degree, size_t t) {
    using V = std::vector<std::vector<Fr>>;
    if (n < 3 * t + 1) return Result<V>::err(InvalidInput);          // :290
    if (evals_by_sender.empty()) return Result<V>::err(InvalidInput);  // :297
    const size_t G = evals_by_sender[0].second.size();
    if (G == 0) return Result<V>::err(InvalidInput);  // :303
    for (const auto& e : evals_by_sender)
        if (e.second.size() != G) return Result<V>::err(InvalidInput);  // :306
    const size_t S = evals_by_sender.size();
    std::vector<size_t> ids(S);
    std::vector<U256> flat(S * G);
    for (size_t i = 0; i < S; ++i) {
        ids[i] = evals_by_sender[i].first;
        for (size_t c = 0; c < G; ++c) flat[i * G + c] = evals_by_sender[i].second[c].v;
    }
    std::vector<U256> co(G * (degree + 1));
    std::vector<uint32_t> nco(G);
    const ShareErrorCode rc = hbmpc_batch_recover(context(), ids.data(), S, flat.data(), G, n, degree, t, co.data(), nco.data(), nullptr);
    if (rc != ShareSuccess) return Result<V>::err(rc);
    V out(G);
    for (size_t c = 0; c < G; ++c) {
        if (nco[c] > degree + 1) throw std::runtime_error("batch_recover_secret: ncoeffs " + std::to_string(nco[c]) + " for chunk " + std::to_string(c));
        for (uint32_t k = 0; k < nco[c]; ++k) out[c].emplace_back(co[c * (degree + 1) + k]);
    }
    return Result<V>::ok(std::move(out));
}

// ---- gao_rs_decode (robust_interpolate.rs:456-538) ------------------------------------------------------
inline Result<std::vector<Fr>> gao_rs_decode(const std::vector<Fr>& received, size_t k, size_t n, const std::vector<size_t>& erasure_positions) {
    std::vector<U256> rec(received.size());
    for (size_t i = 0; i < received.size(); ++i) rec[i] = received[i].v;
    std::vector<U256> co(k ? k : 1);
    size_t nco = 0;
    const ShareErrorCode rc = hbmpc_gao_rs_decode(context(), rec.data(), k, n, erasure_positions.data(), erasure_positions.size(), co.data(), &nco);
    if (rc != ShareSuccess) return Result<std::vector<Fr>>::err(rc);
    std::vector<Fr> out;
    for (size_t i = 0; i < nco; ++i) out.emplace_back(co[i]);
    return Result<std::vector<Fr>>::ok(std::move(out));
}

// GeneralEvaluationDomain::<Fr>::new(n).element(j) as the tests use it: row j of the Vandermonde matrix, column 1
inline Fr domain_element(size_t n, size_t j) {
    std::vector<U256> flat(n * 2);
    if (hbmpc_make_vandermonde(context(), n, 1, flat.data()) != ShareSuccess) throw std::runtime_error("domain");
    return Fr(flat[j * 2 + 1]);
}

}  // namespace hbmpc
