// include/hbmpc_pipelines.hpp (C++ host over the hbmpc_dev_* calls) against the protocol algebra:
// BASELINE config 4 (triple generation) and config 5 (fixed-point multiplication = Beaver + TruncPr) shapes at
// small sizes, for all n simulated parties; eager run, then the same sequence replayed as a HIP graph.
#include <cstdio>
#include <cstring>
#include <vector>

#include "hbmpc_pipelines.hpp"
#include "hbmpc_shares.hpp"

using namespace hbmpc;
static int g_failed = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
            ++g_failed;                                                          \
        }                                                                        \
    } while (0)

static uint64_t g_state = 0x0123456789ABCDEFull;
static uint64_t next64() {
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static U256 rand_fr() { return U256{{next64(), next64(), next64(), next64() % 0x73eda753299d7d48ULL}}; }
static U256 small(uint64_t v) { return U256{{v, 0, 0, 0}}; }

// [n][N] degree-d sharings of N secrets (random higher coefficients), by the library's own compute_shares
static std::vector<U256> share_all(const std::vector<U256>& secrets, size_t n, size_t d) {
    const size_t N = secrets.size();
    std::vector<U256> coeffs(N * (d + 1)), out(n * N);
    for (size_t i = 0; i < N; ++i) {
        coeffs[i * (d + 1)] = secrets[i];
        for (size_t k = 1; k <= d; ++k) coeffs[i * (d + 1) + k] = rand_fr();
    }
    pl_check(hbmpc_compute_shares(context(), coeffs.data(), N, n, d, out.data()), context(), "compute_shares");
    return out;
}
static std::vector<U256> open_all(const std::vector<U256>& shares, size_t n, size_t N, size_t d, size_t t) {
    std::vector<size_t> ids;
    for (size_t i = 0; i < n; ++i) ids.push_back(i);
    std::vector<U256> p0(N);
    pl_check(hbmpc_batch_recover_p0(context(), ids.data(), n, shares.data(), N, n, d, t, p0.data(), nullptr), context(), "open");
    return p0;
}
static std::vector<U256> mul_all(const std::vector<U256>& a, const std::vector<U256>& b) {
    std::vector<U256> c(a.size());
    pl_check(hbmpc_fr_op(context(), 2, a.data(), b.data(), a.size(), c.data()), context(), "mul");
    return c;
}
static bool same(const std::vector<U256>& a, const std::vector<U256>& b) {
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(U256)) == 0;
}

static void triple_gen(void* stream) {
    const size_t n = 7, t = 2, groups = 9, N = groups * (2 * t + 1);
    std::vector<U256> a(N), b(N), r(N);
    for (size_t i = 0; i < N; ++i) a[i] = rand_fr(), b[i] = rand_fr(), r[i] = rand_fr();
    TripleGen tg(context(), n, t, N, stream);
    tg.upload(tg.a, share_all(a, n, t).data(), n * N);
    tg.upload(tg.b, share_all(b, n, t).data(), n * N);
    tg.upload(tg.rt, share_all(r, n, t).data(), n * N);
    tg.upload(tg.r2t, share_all(r, n, 2 * t).data(), n * N);
    tg.run();
    std::vector<U256> c(n * N);
    tg.download(c.data(), tg.c, n * N);
    CHECK(tg.last_summary().n_failed == 0);
    CHECK(same(open_all(c, n, N, t, t), mul_all(a, b)));  // [c]_t opens to a * b
    tg.capture();
    std::vector<U256> zero(n * N, small(0)), c2(n * N);
    tg.upload(tg.c, zero.data(), n * N);
    tg.replay();
    tg.download(c2.data(), tg.c, n * N);
    CHECK(same(c, c2));
}

static void fpmul(void* stream) {
    const size_t n = 7, t = 2, N = 40, k = 16, m = 4;
    std::vector<U256> x(N), y(N), ta(N), tb(N), rint(N);
    std::vector<uint64_t> xs(N), ys(N);
    for (size_t i = 0; i < N; ++i) {
        xs[i] = next64() % 128, ys[i] = next64() % 128;
        x[i] = small(xs[i]), y[i] = small(ys[i]);
        ta[i] = rand_fr(), tb[i] = rand_fr();
        rint[i] = small(next64() % (1ull << 40));
    }
    FpMul fp(context(), n, t, N, k, m, stream);
    fp.upload(fp.x, share_all(x, n, t).data(), n * N);
    fp.upload(fp.y, share_all(y, n, t).data(), n * N);
    fp.upload(fp.ta, share_all(ta, n, t).data(), n * N);
    fp.upload(fp.tb, share_all(tb, n, t).data(), n * N);
    fp.upload(fp.tc, share_all(mul_all(ta, tb), n, t).data(), n * N);
    fp.upload(fp.rint, share_all(rint, n, t).data(), n * N);
    for (size_t j = 0; j < m; ++j) {  // r_bits[party][bit][N]
        std::vector<U256> bit(N);
        for (auto& v : bit) v = small(next64() & 1);
        const std::vector<U256> sh = share_all(bit, n, t);
        for (size_t p = 0; p < n; ++p) fp.upload(fp.rbits + (p * m + j) * N, sh.data() + p * N, N);
    }
    fp.run();
    std::vector<U256> z(n * N), out(n * N);
    fp.download(z.data(), fp.z, n * N);
    fp.download(out.data(), fp.out, n * N);
    CHECK(fp.last_summary().n_failed == 0);
    const std::vector<U256> zo = open_all(z, n, N, t, t), oo = open_all(out, n, N, t, t);
    for (size_t i = 0; i < N; ++i) {
        const uint64_t prod = xs[i] * ys[i];
        CHECK(zo[i].data[0] == prod && (zo[i].data[1] | zo[i].data[2] | zo[i].data[3]) == 0);        // Beaver product
        CHECK((oo[i].data[0] == (prod >> m) || oo[i].data[0] == (prod >> m) + 1) && oo[i].data[1] == 0);  // probabilistic truncation
    }
    fp.capture();
    std::vector<U256> zero(n * N, small(0)), out2(n * N);
    fp.upload(fp.out, zero.data(), n * N);
    fp.replay();
    fp.download(out2.data(), fp.out, n * N);
    CHECK(same(out, out2));
}

// dealers' polynomials -> RanSha / DouSha + RanDouSha -> TripleGen, all on the device; the opened c must equal a * b
// for the secrets the producers' outputs open to, and tampering with one dealt share must turn the verdict
static void preprocessing(void* stream, size_t groups) {  // groups = 3: N = 15 divides into whole batch elements (the slices go straight into TripleGen's arrays)
    const size_t n = 7, t = 2, N = groups * (2 * t + 1);
    Preprocessing pre(context(), n, t, N, stream);
    const size_t Krs = pre.rs.nout / (n - 2 * t), Krd = pre.rd.nout / (t + 1);
    std::vector<U256> co(n * Krs * (t + 1)), ct(n * Krd * (t + 1)), c2t(n * Krd * (2 * t + 1));
    for (auto& v : co) v = rand_fr();
    for (auto& v : ct) v = rand_fr();
    for (auto& v : c2t) v = rand_fr();
    for (size_t pk = 0; pk < n * Krd; ++pk) c2t[pk * (2 * t + 1)] = ct[pk * (t + 1)];  // the same secret in both sharings
    pre.rs.upload(pre.rs.coeffs, co.data(), co.size());
    pre.rd.upload(pre.rd.coeffs_t, ct.data(), ct.size());
    pre.rd.upload(pre.rd.coeffs_2t, c2t.data(), c2t.size());
    pre.run();
    uint32_t v1[2], v2[2];
    pre.rs.verdict(v1);
    pre.rd.verdict(v2);
    CHECK(v1[0] == 0 && v2[0] == 0);
    CHECK(pre.tg.last_summary().n_failed == 0);
    std::vector<U256> a(n * N), b(n * N), c(n * N);
    pre.tg.download(a.data(), pre.tg.a, n * N);
    pre.tg.download(b.data(), pre.tg.b, n * N);
    pre.tg.download(c.data(), pre.tg.c, n * N);
    CHECK(same(open_all(c, n, N, t, t), mul_all(open_all(a, n, N, t, t), open_all(b, n, N, t, t))));
    // one dealt share changed before the recipients mix: the verifiers must notice
    pre.rs.deal();
    std::vector<U256> one(1);
    pre.rs.download(one.data(), pre.rs.S + (3 * n + 1) * Krs + 2, 1);
    one[0].data[0] ^= 1;
    pre.rs.upload(pre.rs.S + (3 * n + 1) * Krs + 2, one.data(), 1);
    pre.rs.finish();
    pre.rs.verdict(v1);
    CHECK(v1[0] >= 1 && v1[1] == 2);
}

int main() {
    void* stream = nullptr;
    pl_check(hbmpc_stream_create(context(), &stream), context(), "stream_create");
    std::printf("triple_gen\n");
    triple_gen(stream);
    std::printf("fpmul\n");
    fpmul(stream);
    std::printf("preprocessing (RanSha, RanDouSha, TripleGen)\n");
    preprocessing(stream, 4);
    preprocessing(stream, 3);
    pl_check(hbmpc_stream_destroy(context(), stream), context(), "stream_destroy");
    std::printf(g_failed ? "%d CHECKS FAILED\n" : "pipelines passed (%d failures)\n", g_failed);
    return g_failed ? 1 : 0;
}
