// capi_pipelines.hip -- the device-resident pipelines behind the C ABI (hbmpc_pipe_* of include/hbmpc_hip.h): replays of the
// reference's arithmetic pipelines for ALL n simulated parties on one GPU -- how every reference test and bench runs (n
// parties in one process on FakeNetwork) -- so that a Rust node (or any C caller) gets the call sequencing, the arena
// layout and the capture rules from the library instead of re-deriving them.
//
//   triplegen      TripleGenNode::init_batch + BatchReconNode (degree 2t) + try_finalize_triple_gen
//                  triple_gen/triple_generation.rs:304-364,164-232; batch_recon/batch_recon.rs:144-185,332-481
//   fpmul          FPMulNode::init = Multiply (Beaver, RBC path) + TruncPrNode
//                  fpmul/fpmul.rs:61-110, mul/multiplication.rs:417-426,57-139, fpmul/truncpr.rs:185-318
//   ransha         RanShaNode: deal, n x n Vandermonde, verifier reconstruction + degree test, output slice
//                  share_gen/share_gen.rs:232-289,401-454,516-530,199-203
//   randousha      DouShaNode deal + RanDouShaNode: both Vandermonde products, verifier interpolations + tests, output slice
//                  double_share/double_share_generation.rs:151-215, ran_dou_sha/mod.rs:371-449,569-602,314-331
//   preprocessing  run_preprocessing's triple part (honeybadger/mod.rs:1239-1393): ransha -> a, b; randousha -> r; triplegen
//
// Host-side orchestration only: this file is a CLIENT of the hbmpc_dev_* entry points (it includes nothing but the public
// header); every arithmetic step is a device call, buffers never leave HBM, the parties' all-to-all is a layout.
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/hbmpc_hip.h"

namespace {

struct PipeError {
    ShareErrorCode rc;
};
#define PL(call)                              \
    do {                                      \
        const ShareErrorCode rc_ = (call);    \
        if (rc_ != ShareSuccess) throw PipeError{rc_}; \
    } while (0)

// the two fields' entry points behind one signature (elements are 32-byte U256 or 8-byte uint64_t)
struct Calls {
    bool gl;
    size_t eb;  // bytes per element
    ShareErrorCode compute_shares(hbmpc_ctx* c, const void* co, size_t B, size_t n, size_t d, void* out, void* s) const {
        return gl ? hbmpc_gl_dev_compute_shares(c, (const uint64_t*)co, B, n, d, (uint64_t*)out, s)
                  : hbmpc_dev_compute_shares(c, (const U256*)co, B, n, d, (U256*)out, s);
    }
    ShareErrorCode apply_rows(hbmpc_ctx* c, const void* x, size_t stride, size_t G, size_t n, size_t d, void* tmp, void* y, void* s) const {
        return gl ? hbmpc_gl_dev_vandermonde_apply_rows(c, (const uint64_t*)x, stride, G, n, d, (uint64_t*)tmp, (uint64_t*)y, s)
                  : hbmpc_dev_vandermonde_apply_rows(c, (const U256*)x, stride, G, n, d, (U256*)tmp, (U256*)y, s);
    }
    ShareErrorCode apply_rows_lists(hbmpc_ctx* c, const void* x, size_t stride, size_t G, size_t n, size_t d, void* tmp, void* y, size_t row0,
                                    size_t rows, size_t K, const hbmpc_list_slice* sl, size_t nsl, void* s) const {
        return gl ? hbmpc_gl_dev_vandermonde_apply_rows_lists(c, (const uint64_t*)x, stride, G, n, d, (uint64_t*)tmp, (uint64_t*)y, row0, rows, K, sl, nsl, s)
                  : hbmpc_dev_vandermonde_apply_rows_lists(c, (const U256*)x, stride, G, n, d, (U256*)tmp, (U256*)y, row0, rows, K, sl, nsl, s);
    }
    ShareErrorCode apply_rows_split(hbmpc_ctx* c, const void* x, size_t stride, size_t G, size_t n, size_t d, void* tmp, void* y, size_t row0,
                                    size_t rows, size_t K, const hbmpc_list_slice* sl, size_t nsl, void* others, void* s) const {
        return gl ? hbmpc_gl_dev_vandermonde_apply_rows_split(c, (const uint64_t*)x, stride, G, n, d, (uint64_t*)tmp, (uint64_t*)y, row0, rows, K, sl, nsl,
                                                              (uint64_t*)others, s)
                  : hbmpc_dev_vandermonde_apply_rows_split(c, (const U256*)x, stride, G, n, d, (U256*)tmp, (U256*)y, row0, rows, K, sl, nsl, (U256*)others, s);
    }
    ShareErrorCode interpolate_c0(hbmpc_ctx* c, const size_t* ids, size_t S, const void* ev, size_t stride, size_t G, size_t n, void* tmp,
                                  void* c0, uint32_t* deg, void* s) const {
        return gl ? hbmpc_gl_dev_batch_interpolate_c0(c, ids, S, (const uint64_t*)ev, stride, G, n, (uint64_t*)tmp, (uint64_t*)c0, deg, s)
                  : hbmpc_dev_batch_interpolate_c0(c, ids, S, (const U256*)ev, stride, G, n, (U256*)tmp, (U256*)c0, deg, s);
    }
    ShareErrorCode recover_strided(hbmpc_ctx* c, const size_t* ids, size_t S, const void* ev, size_t stride, size_t G, size_t n, size_t d,
                                   size_t t, int p0, void* out, uint32_t* nco, uint8_t* st, hbmpc_recover_summary* sm, void* s) const {
        return gl ? hbmpc_gl_dev_batch_recover_strided(c, ids, S, (const uint64_t*)ev, stride, G, n, d, t, p0, (uint64_t*)out, nco, st, sm, s)
                  : hbmpc_dev_batch_recover_strided(c, ids, S, (const U256*)ev, stride, G, n, d, t, p0, (U256*)out, nco, st, sm, s);
    }
    ShareErrorCode recover(hbmpc_ctx* c, const size_t* ids, size_t S, const void* ev, size_t G, size_t n, size_t d, size_t t, void* out,
                           uint32_t* nco, uint8_t* st, hbmpc_recover_summary* sm, void* s) const {
        return gl ? hbmpc_gl_dev_batch_recover(c, ids, S, (const uint64_t*)ev, G, n, d, t, (uint64_t*)out, nco, st, sm, s)
                  : hbmpc_dev_batch_recover(c, ids, S, (const U256*)ev, G, n, d, t, (U256*)out, nco, st, sm, s);
    }
    ShareErrorCode interpolate(hbmpc_ctx* c, const size_t* ids, size_t S, const void* ev, size_t stride, size_t G, size_t n, void* out,
                               uint32_t* deg, void* s) const {
        return gl ? hbmpc_gl_dev_batch_interpolate(c, ids, S, (const uint64_t*)ev, stride, G, n, (uint64_t*)out, deg, s)
                  : hbmpc_dev_batch_interpolate(c, ids, S, (const U256*)ev, stride, G, n, (U256*)out, deg, s);
    }
    ShareErrorCode triple_encode(hbmpc_ctx* c, const void* a, const void* b, const void* r, size_t G, size_t n, size_t d, size_t parties,
                                 void* tmp, void* y, void* s) const {
        return gl ? hbmpc_gl_dev_triple_encode_parties(c, (const uint64_t*)a, (const uint64_t*)b, (const uint64_t*)r, G, n, d, parties,
                                                       (uint64_t*)tmp, (uint64_t*)y, s)
                  : hbmpc_dev_triple_encode_parties(c, (const U256*)a, (const U256*)b, (const U256*)r, G, n, d, parties, (U256*)tmp, (U256*)y, s);
    }
    ShareErrorCode triple_finalize(hbmpc_ctx* c, const void* rt, const void* opened, size_t N, size_t parties, void* out, void* s) const {
        return gl ? hbmpc_gl_dev_triple_finalize_parties(c, (const uint64_t*)rt, (const uint64_t*)opened, N, parties, (uint64_t*)out, s)
                  : hbmpc_dev_triple_finalize_parties(c, (const U256*)rt, (const U256*)opened, N, parties, (U256*)out, s);
    }
};

struct Buffer {
    unsigned char* p;
    size_t elements;  // of the context's field (status / verdict buffers: bytes)
};
// where a slice of every party's output list goes instead of the producer's own buffer: hbmpc_list_slice {dst, party stride,
// k0, count} = batch elements [k0, k0 + count) of party p to dst + p * stride (elements)
using Slice = hbmpc_list_slice;

}  // namespace

struct hbmpc_pipe {
    hbmpc_ctx* ctx;
    void* stream;
    Calls f;
    bool checked = false;
    hbmpc_graph* graph = nullptr;
    unsigned char* base = nullptr;  // one hbmpc_dev_alloc block, bump-allocated (no per-step allocations)
    size_t size = 0, off = 0;
    std::map<std::string, Buffer> buffers;
    hbmpc_recover_summary* summ = nullptr;  // the summary of the last decode
    uint32_t* bad = nullptr;                // producers: {verifier checks that failed, first failing batch element}

    hbmpc_pipe(hbmpc_ctx* c, void* s) : ctx(c), stream(s) {
        f.gl = hbmpc_field_of(c) == Goldilocks64;
        f.eb = f.gl ? 8 : 32;
    }
    virtual ~hbmpc_pipe() {
        hbmpc_graph_destroy(graph);
        if (base) (void)hbmpc_dev_free(ctx, base);
    }
    void arena(size_t bytes) {
        void* p = nullptr;
        PL(hbmpc_dev_alloc(ctx, bytes, &p));
        base = static_cast<unsigned char*>(p), size = bytes;
    }
    unsigned char* take_bytes(const char* name, size_t bytes, size_t elements) {
        const size_t padded = (bytes + 255) & ~(size_t)255;
        if (off + padded > size) throw PipeError{InvalidInput};
        unsigned char* p = base + off;
        off += padded;
        if (name) buffers[name] = Buffer{p, elements};
        return p;
    }
    unsigned char* take(const char* name, size_t elements) { return take_bytes(name, elements * f.eb, elements); }
    virtual void run() = 0;  // enqueue only (checked: summaries are read back after every decode)
    virtual void deal() { throw PipeError{InvalidInput}; }
    virtual void finish() { throw PipeError{InvalidInput}; }
    virtual hbmpc_pipe* part(const std::string&) { return nullptr; }
    virtual void verdict(uint32_t out[2]) {
        if (!bad) throw PipeError{InvalidInput};
        PL(hbmpc_memcpy_d2h(ctx, out, bad, 8, stream));
        PL(hbmpc_stream_sync(ctx, stream));
    }
    void check_summary(const hbmpc_recover_summary* which = nullptr) {  // checked mode: a failed chunk ends the run where the reference's `?` would
        if (!checked) return;
        hbmpc_recover_summary s;
        PL(hbmpc_memcpy_d2h(ctx, &s, which ? which : summ, sizeof s, stream));
        PL(hbmpc_stream_sync(ctx, stream));
        if (s.n_failed != 0) throw PipeError{(ShareErrorCode)s.first_error};
    }
    void clear_bad() {
        static const uint32_t init[2] = {0u, 0xffffffffu};
        PL(hbmpc_memcpy_h2d(ctx, bad, init, sizeof init, stream));
    }
};

namespace {

// n parties, threshold t, N triples (a multiple of 2t+1); buffers [party][N]
struct TripleGen : hbmpc_pipe {
    size_t n, t, N, G;
    unsigned char *a, *b, *r2t, *rt, *c, *Y, *Z, *opened;
    uint8_t* status;
    hbmpc_recover_summary* summ_first;  // the recipients' decodes' (summ: the revealed values')
    std::vector<size_t> ids;
    TripleGen(hbmpc_ctx* cx, size_t n_, size_t t_, size_t N_, void* s) : hbmpc_pipe(cx, s), n(n_), t(t_), N(N_), G(N_ / (2 * t_ + 1)) {
        if (n == 0 || N == 0 || N % (2 * t + 1) != 0) throw PipeError{InvalidInput};
        arena((5 * n * N + n * n * G + n * G + N) * f.eb + (n + 2) * G + (1 << 14));
        a = take("a", n * N), b = take("b", n * N), r2t = take("r2t", n * N), rt = take("rt", n * N), c = take("c", n * N);
        Y = take("Y", n * n * G);   // Y[p][j][g]: party p's evaluation for recipient j
        Z = take("Z", n * G);       // Z[j][g]: recipient j's opened y_j (the broadcast RevealBatch)
        opened = take("opened", N); // [G][2t+1] == flat [N]
        status = take_bytes("status", n * G, n * G);
        summ = reinterpret_cast<hbmpc_recover_summary*>(take_bytes("summary", 64, 16));
        summ_first = reinterpret_cast<hbmpc_recover_summary*>(take_bytes("summary_first", 64, 16));
        for (size_t i = 0; i < n; ++i) ids.push_back(i);
    }
    void run() override {
        // one call for the whole step: one launch for a small batch, four otherwise (hbmpc_[gl_]dev_triplegen_parties)
        if (f.gl)
            PL(hbmpc_gl_dev_triplegen_parties(ctx, (const uint64_t*)a, (const uint64_t*)b, (const uint64_t*)r2t, (const uint64_t*)rt, N, n, t, (uint64_t*)Y,
                                              (uint64_t*)Z, (uint64_t*)opened, (uint64_t*)c, status, summ_first, summ, stream));
        else
            PL(hbmpc_dev_triplegen_parties(ctx, (const U256*)a, (const U256*)b, (const U256*)r2t, (const U256*)rt, N, n, t, (U256*)Y, (U256*)Z,
                                           (U256*)opened, (U256*)c, status, summ_first, summ, stream));
        check_summary(summ_first);
        check_summary(summ);
    }
};

// Fixed-point multiplication of N element pairs for n parties: Beaver mul (a-x, b-y opened by direct robust interpolation,
// the RBC path FPMulNode always takes) followed by TruncPr with k-bit values and m fractional bits.  open_senders: how many
// parties' shares an open interpolates from (0 = the default 2t+1: the reference opens as soon as that many have arrived,
// multiplication.rs:388,617, truncpr.rs:202 -- with d = t a decode with no OEC round, one launch).  Fr only, as in the reference.
struct FpMul : hbmpc_pipe {
    size_t n, t, N, k, m;
    U256 *x, *y, *ta, *tb, *tc, *rint, *z, *rdash, *osh, *out, *desh, *rbits, *dop, *eop, *cop;
    uint8_t* status;
    hbmpc_recover_summary* summ_first;  // the first open's (summ: the second's)
    std::vector<size_t> ids;
    FpMul(hbmpc_ctx* cx, size_t n_, size_t t_, size_t N_, size_t k_, size_t m_, size_t open_senders, void* s)
        : hbmpc_pipe(cx, s), n(n_), t(t_), N(N_), k(k_), m(m_) {
        if (f.gl) throw PipeError{TypeMismatch};
        if (open_senders == 0) open_senders = 2 * t + 1;
        if (n == 0 || N == 0 || open_senders < 2 * t + 1 || open_senders > n) throw PipeError{InvalidInput};
        arena(((12 + m) * n * N + 4 * N) * 32 + 8 * N + (1 << 14));
        const char* names[] = {"x", "y", "ta", "tb", "tc", "rint", "z", "rdash", "osh", "out"};
        U256** per_party[] = {&x, &y, &ta, &tb, &tc, &rint, &z, &rdash, &osh, &out};
        for (int i = 0; i < 10; ++i) *per_party[i] = reinterpret_cast<U256*>(take(names[i], n * N));
        desh = reinterpret_cast<U256*>(take("desh", 2 * n * N));    // [party][2][N]: a party's shares of a - x and of b - y side by side
        rbits = reinterpret_cast<U256*>(take("rbits", n * m * N));  // [party][bit][N]
        dop = reinterpret_cast<U256*>(take("deop", 2 * N)), eop = dop + N;  // the opened a - x [N], then the opened b - y [N]
        buffers["dop"] = Buffer{(unsigned char*)dop, N}, buffers["eop"] = Buffer{(unsigned char*)eop, N};
        cop = reinterpret_cast<U256*>(take("cop", N));
        status = take_bytes("status", 2 * N, 2 * N);
        summ = reinterpret_cast<hbmpc_recover_summary*>(take_bytes("summary", 64, 16));
        summ_first = reinterpret_cast<hbmpc_recover_summary*>(take_bytes("summary_first", 64, 16));
        for (size_t i = 0; i < open_senders; ++i) ids.push_back(i);
    }
    void run() override {
        // all parties at once (the [party][N] arrays are contiguous; opened values are broadcast): one launch for a small batch,
        // one per step otherwise (hbmpc_dev_fpmul_parties).  Checked mode looks at the two opens' summaries afterwards, in order.
        PL(hbmpc_dev_fpmul_parties(ctx, ids.data(), ids.size(), ta, tb, tc, x, y, rbits, rint, k, m, N, n, t, desh, dop, z, rdash, osh, cop, out,
                                   status, summ_first, summ, stream));
        check_summary(summ_first);
        check_summary(summ);
    }
};

// What RanSha and RanDouSha share.  The dealers' polynomials are the INPUT (coefficient rows [dealer][K][deg + 1], column 0 the
// secret: uploaded by the host or filled on the device by hbmpc_dev_fill_coeffs -- the reference draws them from each party's
// rng).  Layouts (all n parties on one device): dealt S[p][j, k] --n x n Vandermonde over the rows p--> y[i][j, k]; what party j
// sends verifier i is y[i][j K .. j K + K): a strided sender row, nothing is copied.
struct Producer : hbmpc_pipe {
    size_t n, t, K;
    // all verifiers' columns in ONE launch of the wave-per-chunk decode (groups): ahead of the other forms up to this many chunks -- at 6 144 and
    // 8 192 (n = 7 at the node's 1 536 / 2 048 columns) the party-major form below takes 0.070 / 0.037 ms against 0.121 / 0.071
    // (profiles/r04_protocol_batch_sizes.txt)
    size_t grouped_max = 4096;
    // above: a launch per dealer.  Measured (tools/time_dealers_together.py, n = 16, degree 5): 0.037 ms against 0.146 at 16 000 polynomials per
    // dealer, 0.159 / 0.295 at 65 536, 0.613 / 0.787 at 262 144 -- a launch of one dealer's batch spends a visible part of its time filling and
    // draining the chip until the batch has some 10^6 polynomials
    size_t dealers_together_max = ((size_t)1 << 20) - 1;
    Producer(hbmpc_ctx* cx, size_t n_, size_t t_, size_t K_, void* s) : hbmpc_pipe(cx, s), n(n_), t(t_), K(K_) {}
    void deal_one(const unsigned char* coeffs, size_t deg, unsigned char* S) {
        // compute_shares of every dealer's K polynomials: coeffs [dealer][K][deg + 1] -> S [dealer][recipient][K] is the party-batched
        // encode's layout (hbmpc_dev_vandermonde_apply_parties).  A launch per dealer is at the memory rate only from some 10^6 polynomials;
        // below that all dealers go into ONE launch over (dealer, polynomial), whatever kernel family the size takes: the wave-per-chunk
        // kernels up to 2 048 chunks, the point-pair matrix-core kernel on domains of 8 / 16 points from 16 384, the lane kernels elsewhere.
        // At the node's batch sizes (K ~ 7 000 .. 15 000, n = 16) 16 launches of 10 us each were 45 % of the producers' time
        // (profiles/r04_protocol_batch_sizes.txt); at small ones (n = 16, 367 polynomials per dealer) one launch takes 12.6 us against
        // 23.7 for slices that stay within the wave-per-chunk kernels' range, at 1 024 polynomials 13.2 against 48.7.
        auto together = [&](size_t p, size_t cnt) {
            const unsigned char* co = coeffs + p * K * (deg + 1) * f.eb;
            unsigned char* out = S + p * n * K * f.eb;
            PL(f.gl ? hbmpc_gl_dev_vandermonde_apply_parties(ctx, (const uint64_t*)co, K, n, deg, cnt, (uint64_t*)out, stream)
                    : hbmpc_dev_vandermonde_apply_parties(ctx, (const U256*)co, K, n, deg, cnt, (U256*)out, stream));
        };
        if (K <= dealers_together_max) {  // as many dealers per call as the one-launch form's 32-bit offsets allow (4 GiB of outputs, of inputs)
            const size_t lim = ((size_t)1 << 32) - 1;
            const size_t per = std::min(n, std::min(lim / (n * K * f.eb), lim / (K * (deg + 1) * f.eb)));
            if (per >= 2) {
                for (size_t p = 0; p < n; p += per) together(p, n - p < per ? n - p : per);
                return;
            }
        }
        for (size_t p = 0; p < n; ++p)  // dealer p: compute_shares of its K polynomials
            PL(f.compute_shares(ctx, coeffs + p * K * (deg + 1) * f.eb, K, n, deg, S + p * n * K * f.eb, stream));
    }
    // The share of dealer p for (recipient, element) is row p of S: the n x n map reads the dealers' outputs in place (x is the
    // workspace of the shapes that have to be transposed first).  Rows [row0, row0 + rows) of the result are the parties' OUTPUT:
    // they are written as the per-party lists [k][row] the reference returns, where `lists` says; the other rows -- what the
    // parties send the verifiers -- to y[row][party, k]
    // with `others`: the verifiers' rows party-major (hbmpc_dev_vandermonde_apply_rows_split): others[(party (n - rows) + r') K + k]
    void mix(const unsigned char* S, unsigned char* x, unsigned char* y, size_t row0, size_t rows, const std::vector<Slice>& lists,
             unsigned char* others = nullptr) {
        if (others) PL(f.apply_rows_split(ctx, S, n * K, n * K, n, n - 1, x, y, row0, rows, K, lists.data(), lists.size(), others, stream));
        else PL(f.apply_rows_lists(ctx, S, n * K, n * K, n, n - 1, x, y, row0, rows, K, lists.data(), lists.size(), stream));
    }
    // Mid-size batches (the reference's own: some thousands of columns per dealer): one verifier's decode is a launch of ~10 us that
    // does not fill the chip, and there are 2t (RanSha) or 2 (n - t - 1) (RanDouSha) of them.  Where the mixing kernel can write the
    // verifiers' rows party-major, all verifiers of a kind are ONE decode over (verifier, column) chunks (profiles/r04_protocol_batch_sizes.txt).
    // Measured ahead of a call per verifier up to 250 000 columns (n = 16: RanSha 0.742 -> 0.616 ms at 80 000, RanDouSha 4.39 -> 3.92 at 250 000), level
    // with it at 699 050 (config 4's producers); from 2^19 columns on -- and where the party-major block would pass 4 GiB -- a call each, without the
    // extra rows.  Shapes no list kernel covers (more than 16 parties; 9 .. 16 over Fr at small batches) pay one more pass for the party-major rows
    // (k_rows_party_major): from three verifiers on that is still fewer launches.
    bool verifiers_together(size_t nver) const {
        int yes = 0;
        if (nver < 2 || K >= ((size_t)1 << 19) || n * nver * K * f.eb >= ((size_t)1 << 32)) return false;
        if (hbmpc_dev_apply_rows_lists_in_kernel(ctx, n * K, n, n - 1, &yes) != ShareSuccess) return false;
        return yes != 0 || nver >= 3;
    }
    void run() override {
        deal();
        finish();
    }
};

// K batch elements per dealer -> (n - 2t) K random degree-t sharings per party, verified by parties 0 .. 2t - 1 from the
// shares of the first verify_senders parties (0 = the default 2t + 1: the handler fires when that many have arrived,
// share_gen.rs:497 -- with degree t a decode with no OEC round)
struct RanSha : Producer {
    size_t nout;
    unsigned char *coeffs, *S, *x, *y, *yv = nullptr, *poly, *out;
    uint8_t* status;
    bool grouped, together;
    std::vector<size_t> ids;
    std::vector<Slice> split;  // set by Preprocessing: where the output slices go instead of `out`
    static size_t checked_nout(size_t n, size_t t, size_t K) {  // before the arena is sized from n - 2t
        if (n <= 2 * t || K == 0) throw PipeError{InvalidInput};
        return (n - 2 * t) * K;
    }
    RanSha(hbmpc_ctx* cx, size_t n_, size_t t_, size_t K_, size_t verify_senders, void* s)
        : Producer(cx, n_, t_, K_, s), nout(checked_nout(n_, t_, K_)) {
        if (verify_senders == 0) verify_senders = 2 * t + 1;
        if (verify_senders < 2 * t + 1 || verify_senders > n) throw PipeError{InvalidInput};
        grouped = 2 * t * K <= grouped_max;  // the verifiers' columns fit one launch of the wave-per-chunk decode: workspace for all of them
        together = !grouped && verify_senders == 2 * t + 1 && verifiers_together(2 * t);
        const bool all = grouped || together;
        const size_t poly_el = K * (all && 2 * t > t + 1 ? 2 * t : t + 1), status_b = all ? 2 * t * K + 1 : K;
        arena((n * K * (t + 1) + 3 * n * n * K + (together ? 2 * t * n * K : 0) + poly_el + n * nout) * f.eb + status_b + (1 << 14));
        coeffs = take("coeffs", n * K * (t + 1));  // [dealer][K][t + 1]
        S = take("S", n * n * K);                  // [dealer][recipient][K]
        x = take("x", n * n * K);
        y = take("y", n * n * K);                  // [row i][party][K]
        if (together) yv = take("yv", 2 * t * n * K);  // [party][verifier i][K]
        poly = take("poly", poly_el);
        status = take_bytes("status", status_b, K);
        summ = reinterpret_cast<hbmpc_recover_summary*>(take_bytes("summary", 64, 16));
        bad = reinterpret_cast<uint32_t*>(take_bytes("bad", 64, 16));
        out = take("out", n * nout);               // [party][K][n - 2t]: the reference's output order (share_gen.rs:199-203)
        for (size_t i = 0; i < verify_senders; ++i) ids.push_back(i);
    }
    void deal() override { deal_one(coeffs, t, S); }
    void finish() override {  // everything after the dealers' messages have arrived
        // rows 2t .. n - 1 of every batch element are the output, per party in the order [k][i - 2t]  (share_gen.rs:199-203)
        mix(S, x, y, 2 * t, n - 2 * t, split.empty() ? std::vector<Slice>{{out, nout, 0, K}} : split, yv);
        clear_bad();
        // verifiers 0 .. 2t - 1: recover_secret of the K columns + exact-degree test (share_gen.rs:516-530); verifier i's sender rows are
        // y + i n K: all of them in one call (one launch for a small batch: `grouped` sized the workspace for it)
        const size_t nver = 2 * t;
        if (nver == 0) return;
        if (together) {  // sender j's row: yv + j nver K, verifier i's columns at i K inside it -- groups that follow each other
            PL(f.gl ? hbmpc_gl_dev_recover_check_degree_strided(ctx, ids.data(), ids.size(), (const uint64_t*)yv, nver * K, K, n, t, nver, K, (uint64_t*)poly,
                                                                status, summ, bad, stream)
                    : hbmpc_dev_recover_check_degree_strided(ctx, ids.data(), ids.size(), (const U256*)yv, nver * K, K, n, t, nver, K, (U256*)poly, status, summ,
                                                             bad, stream));
            return;
        }
        if (grouped) {
            PL(f.gl ? hbmpc_gl_dev_recover_check_degree_strided(ctx, ids.data(), ids.size(), (const uint64_t*)y, K, K, n, t, nver, n * K, (uint64_t*)poly,
                                                                status, summ, bad, stream)
                    : hbmpc_dev_recover_check_degree_strided(ctx, ids.data(), ids.size(), (const U256*)y, K, K, n, t, nver, n * K, (U256*)poly, status, summ,
                                                             bad, stream));
            return;
        }
        for (size_t i = 0; i < nver; ++i)
            PL(f.gl ? hbmpc_gl_dev_recover_check_degree_strided(ctx, ids.data(), ids.size(), (const uint64_t*)(y + i * n * K * f.eb), K, K, n, t, 1, 0,
                                                                (uint64_t*)poly, status, summ, bad, stream)
                    : hbmpc_dev_recover_check_degree_strided(ctx, ids.data(), ids.size(), (const U256*)(y + i * n * K * f.eb), K, K, n, t, 1, 0, (U256*)poly,
                                                             status, summ, bad, stream));
    }
};

// K batch elements per dealer -> (t + 1) K double sharings per party, verified by parties t + 1 .. n - 1, each of which
// interpolates both polynomials through ALL n shares (ran_dou_sha/mod.rs:557-602)
struct RanDouSha : Producer {
    size_t nout;
    unsigned char *coeffs_t, *coeffs_2t, *S_t, *S_2t, *x, *y_t, *y_2t, *poly, *c0_t, *c0_2t, *out_t, *out_2t, *sel_t, *sel_2t, *st_t, *st_2t;
    unsigned char *yv_t = nullptr, *yv_2t = nullptr;
    uint32_t *deg_t, *deg_2t;
    bool grouped, together;
    std::vector<size_t> ids;
    std::vector<Slice> split_t, split_2t;
    static size_t checked_nout(size_t n, size_t t, size_t K) {
        if (n <= 2 * t || K == 0) throw PipeError{InvalidInput};  // the degree-2t sharing needs 2t + 1 <= n points
        return (t + 1) * K;
    }
    RanDouSha(hbmpc_ctx* cx, size_t n_, size_t t_, size_t K_, void* s) : Producer(cx, n_, t_, K_, s), nout(checked_nout(n_, t_, K_)) {
        grouped = (n - t - 1) * K <= grouped_max && !f.gl;  // (Fr only) the verifiers' columns fit one launch of the wave-per-chunk decode: room for all their results
        together = !grouped && 2 * t < n && verifiers_together(n - t - 1);
        const size_t vr = grouped || together ? n - t - 1 : 1;
        const size_t vc = together && f.gl ? vr : 1;  // Goldilocks has no selective decode: the verifiers' full interpolations, all in one call
        arena((n * K * (3 * t + 2) + 5 * n * n * K + (together ? 2 * vr * n * K : 0) + vc * K * n + (2 * vc + 4 * vr) * K + 2 * n * nout) * f.eb +
              (8 * vc + 2 * vr) * K + (1 << 14));
        coeffs_t = take("coeffs_t", n * K * (t + 1)), coeffs_2t = take("coeffs_2t", n * K * (2 * t + 1));
        S_t = take("S_t", n * n * K), S_2t = take("S_2t", n * n * K);
        x = take("x", n * n * K), y_t = take("y_t", n * n * K), y_2t = take("y_2t", n * n * K);
        if (together) yv_t = take("yv_t", vr * n * K), yv_2t = take("yv_2t", vr * n * K);  // [party][verifier][K]
        poly = take("poly", vc * K * n);  // workspace of the verifier interpolations that have no c0-only kernel
        c0_t = take("c0_t", vc * K), c0_2t = take("c0_2t", vc * K);
        deg_t = reinterpret_cast<uint32_t*>(take_bytes("deg_t", 4 * vc * K, K)), deg_2t = reinterpret_cast<uint32_t*>(take_bytes("deg_2t", 4 * vc * K, K));
        sel_t = take("sel_t", 2 * K * vr), sel_2t = take("sel_2t", 2 * K * vr);  // (constant term, top coefficient) of a verifier's two polynomials
        st_t = take_bytes("st_t", K * vr, K), st_2t = take_bytes("st_2t", K * vr, K);
        bad = reinterpret_cast<uint32_t*>(take_bytes("bad", 64, 16));
        out_t = take("out_t", n * nout), out_2t = take("out_2t", n * nout);  // [party][K][t + 1]  (ran_dou_sha/mod.rs:314-331)
        for (size_t i = 0; i < n; ++i) ids.push_back(i);
    }
    void deal() override {  // DouShaNode::init_batch: both sharings of every secret
        deal_one(coeffs_t, t, S_t);
        deal_one(coeffs_2t, 2 * t, S_2t);
    }
    void finish() override {
        // RanDouShaNode::init_batch steps 1, 2, 4-5: rows 0 .. t are the output, per party in the order [k][i]  (ran_dou_sha/mod.rs:314-331)
        mix(S_t, x, y_t, 0, t + 1, split_t.empty() ? std::vector<Slice>{{out_t, nout, 0, K}} : split_t, yv_t);
        mix(S_2t, x, y_2t, 0, t + 1, split_2t.empty() ? std::vector<Slice>{{out_2t, nout, 0, K}} : split_2t, yv_2t);
        clear_bad();
        // step 3: verifiers t + 1 .. n - 1 interpolate both sharings through all n shares and test the degrees and the constant terms
        // (:586-602) -- they keep nothing else of them
        const size_t nver = n - t - 1, v0 = (t + 1) * n * K * f.eb;  // verifier i's sender rows: y + i n K
        if (together && f.gl) {  // the n senders' rows nver K apart: every verifier's interpolation in one call, the verdicts in one
            PL(f.interpolate_c0(ctx, ids.data(), n, yv_t, nver * K, nver * K, n, poly, c0_t, deg_t, stream));
            PL(f.interpolate_c0(ctx, ids.data(), n, yv_2t, nver * K, nver * K, n, poly, c0_2t, deg_2t, stream));
            PL(hbmpc_dev_check_double_share_c0_columns(ctx, c0_t, deg_t, c0_2t, deg_2t, nver * K, K, t, bad, stream));
            return;
        }
        if (together) {  // sender j's row: yv + j nver K, verifier i's columns at i K inside it -- groups that follow each other
            PL(hbmpc_dev_interpolate_degree_check_strided(ctx, ids.data(), n, (const U256*)yv_t, nver * K, K, n, t, nver, K, (U256*)poly, (U256*)sel_t, st_t, stream));
            PL(hbmpc_dev_interpolate_degree_check_strided(ctx, ids.data(), n, (const U256*)yv_2t, nver * K, K, n, 2 * t, nver, K, (U256*)poly, (U256*)sel_2t, st_2t,
                                                          stream));
            PL(hbmpc_dev_check_double_share_sel(ctx, sel_t, st_t, sel_2t, st_2t, nver * K, K, t, bad, stream));
            return;
        }
        if (!f.gl && 2 * t < n && grouped) {  // all verifiers in one call each (hbmpc_dev_interpolate_degree_check_strided)
            PL(hbmpc_dev_interpolate_degree_check_strided(ctx, ids.data(), n, (const U256*)(y_t + v0), K, K, n, t, nver, n * K, (U256*)poly, (U256*)sel_t, st_t,
                                                          stream));
            PL(hbmpc_dev_interpolate_degree_check_strided(ctx, ids.data(), n, (const U256*)(y_2t + v0), K, K, n, 2 * t, nver, n * K, (U256*)poly, (U256*)sel_2t,
                                                          st_2t, stream));
            PL(hbmpc_dev_check_double_share_sel(ctx, sel_t, st_t, sel_2t, st_2t, nver * K, K, t, bad, stream));
            return;
        }
        for (size_t i = t + 1; i < n; ++i) {
            if (!f.gl && 2 * t < n) {  // the two questions answered without the full interpolation
                PL(hbmpc_dev_interpolate_degree_check_strided(ctx, ids.data(), n, (const U256*)(y_t + i * n * K * f.eb), K, K, n, t, 1, 0, (U256*)poly,
                                                              (U256*)sel_t, st_t, stream));
                PL(hbmpc_dev_interpolate_degree_check_strided(ctx, ids.data(), n, (const U256*)(y_2t + i * n * K * f.eb), K, K, n, 2 * t, 1, 0, (U256*)poly,
                                                              (U256*)sel_2t, st_2t, stream));
                PL(hbmpc_dev_check_double_share_sel(ctx, sel_t, st_t, sel_2t, st_2t, K, 0, t, bad, stream));
                continue;
            }
            PL(f.interpolate_c0(ctx, ids.data(), n, y_t + i * n * K * f.eb, K, K, n, poly, c0_t, deg_t, stream));
            PL(f.interpolate_c0(ctx, ids.data(), n, y_2t + i * n * K * f.eb, K, K, n, poly, c0_2t, deg_2t, stream));
            PL(hbmpc_dev_check_double_share_c0(ctx, c0_t, deg_t, c0_2t, deg_2t, K, t, bad, stream));
        }
    }
};

// run_preprocessing's triple part for all n parties, device-resident from the dealers' polynomials to [c]_t: RanSha produces
// 2 N random sharings per party (a = the first N, b = the next N: take_random_shares twice, honeybadger/mod.rs:1307-1316),
// RanDouSha the N double sharings, TripleGen consumes them where they lie.
struct Preprocessing : hbmpc_pipe {
    size_t n, t, N;
    std::unique_ptr<RanSha> rs;
    std::unique_ptr<RanDouSha> rd;
    std::unique_ptr<TripleGen> tg;
    bool in_place;
    Preprocessing(hbmpc_ctx* cx, size_t n_, size_t t_, size_t N_, void* s) : hbmpc_pipe(cx, s), n(n_), t(t_), N(N_) {
        if (n <= 2 * t || N == 0) throw PipeError{InvalidInput};
        rs.reset(new RanSha(cx, n, t, (2 * N + (n - 2 * t) - 1) / (n - 2 * t), 0, s));  // (n - 2t) K >= 2 N
        rd.reset(new RanDouSha(cx, n, t, (N + t) / (t + 1), s));                          // (t + 1) K >= N
        tg.reset(new TripleGen(cx, n, t, N, s));
        // whole batch elements on both sides of every cut: the producers' output slices go straight into TripleGen's
        // [party][N] arrays and nothing is copied
        in_place = N % (n - 2 * t) == 0 && N % (t + 1) == 0;
        if (in_place) {
            const size_t k1 = N / (n - 2 * t), k2 = N / (t + 1);
            rs->split = {Slice{tg->a, N, 0, k1}, Slice{tg->b, N, k1, k1}};
            rd->split_t = {Slice{tg->rt, N, 0, k2}};
            rd->split_2t = {Slice{tg->r2t, N, 0, k2}};
        }
        summ = tg->summ;
    }
    hbmpc_pipe* part(const std::string& name) override {
        return name == "ransha" ? (hbmpc_pipe*)rs.get() : name == "randousha" ? (hbmpc_pipe*)rd.get() : name == "triplegen" ? (hbmpc_pipe*)tg.get() : nullptr;
    }
    void run() override {
        rs->checked = rd->checked = tg->checked = checked;
        rs->run();
        rd->run();
        if (!in_place) {
            // the parties' lists, in the reference's order, become TripleGen's [party][N] inputs: one copy per array for all parties
            const size_t eb = f.eb;
            PL(hbmpc_memcpy_d2d_rows(ctx, tg->a, N * eb, rs->out, rs->nout * eb, N * eb, n, stream));
            PL(hbmpc_memcpy_d2d_rows(ctx, tg->b, N * eb, rs->out + N * eb, rs->nout * eb, N * eb, n, stream));
            PL(hbmpc_memcpy_d2d_rows(ctx, tg->rt, N * eb, rd->out_t, rd->nout * eb, N * eb, n, stream));
            PL(hbmpc_memcpy_d2d_rows(ctx, tg->r2t, N * eb, rd->out_2t, rd->nout * eb, N * eb, n, stream));
        }
        tg->run();
    }
    void verdict(uint32_t out[2]) override {  // both producers: the sum of the failed checks, the first failing element of the first that failed
        uint32_t a[2], b[2];
        rs->verdict(a);
        rd->verdict(b);
        out[0] = a[0] + b[0];
        out[1] = a[0] ? a[1] : b[1];
    }
};

template <class F>
ShareErrorCode guarded(F&& fn) {
    try {
        fn();
        return ShareSuccess;
    } catch (const PipeError& e) {
        return e.rc;
    } catch (const std::bad_alloc&) {
        return InvalidInput;
    }
}
template <class P, class... A>
ShareErrorCode create(hbmpc_ctx* ctx, hbmpc_pipe** out, A... args) {
    if (!ctx || !out) return InvalidInput;
    *out = nullptr;
    return guarded([&] { *out = new P(ctx, args...); });
}

}  // namespace

extern "C" ShareErrorCode hbmpc_pipe_triplegen_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream, hbmpc_pipe** pipe_out) {
    return create<TripleGen>(ctx, pipe_out, n, t, N, stream);
}
extern "C" ShareErrorCode hbmpc_pipe_fpmul_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, size_t k, size_t m, size_t open_senders,
                                                  void* stream, hbmpc_pipe** pipe_out) {
    return create<FpMul>(ctx, pipe_out, n, t, N, k, m, open_senders, stream);
}
extern "C" ShareErrorCode hbmpc_pipe_ransha_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, size_t verify_senders, void* stream,
                                                   hbmpc_pipe** pipe_out) {
    return create<RanSha>(ctx, pipe_out, n, t, K, verify_senders, stream);
}
extern "C" ShareErrorCode hbmpc_pipe_randousha_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, void* stream, hbmpc_pipe** pipe_out) {
    return create<RanDouSha>(ctx, pipe_out, n, t, K, stream);
}
extern "C" ShareErrorCode hbmpc_pipe_preprocessing_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream, hbmpc_pipe** pipe_out) {
    return create<Preprocessing>(ctx, pipe_out, n, t, N, stream);
}
extern "C" void hbmpc_pipe_destroy(hbmpc_pipe* pipe) { delete pipe; }
extern "C" ShareErrorCode hbmpc_pipe_part(hbmpc_pipe* pipe, const char* name, hbmpc_pipe** part_out) {
    if (!pipe || !name || !part_out) return InvalidInput;
    *part_out = pipe->part(name);
    return *part_out ? ShareSuccess : InvalidInput;
}
extern "C" ShareErrorCode hbmpc_pipe_buffer(hbmpc_pipe* pipe, const char* name, void** dev_out, size_t* elements_out) {
    if (!pipe || !name) return InvalidInput;
    const auto it = pipe->buffers.find(name);
    if (it == pipe->buffers.end()) return InvalidInput;
    if (dev_out) *dev_out = it->second.p;
    if (elements_out) *elements_out = it->second.elements;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_pipe_upload(hbmpc_pipe* pipe, const char* name, const void* host, size_t elements) {
    if (!pipe || !name || !host) return InvalidInput;
    const auto it = pipe->buffers.find(name);
    if (it == pipe->buffers.end() || elements > it->second.elements) return InvalidInput;
    return hbmpc_memcpy_h2d(pipe->ctx, it->second.p, host, elements * pipe->f.eb, pipe->stream);
}
extern "C" ShareErrorCode hbmpc_pipe_download(hbmpc_pipe* pipe, const char* name, void* host, size_t elements) {
    if (!pipe || !name || !host) return InvalidInput;
    const auto it = pipe->buffers.find(name);
    if (it == pipe->buffers.end() || elements > it->second.elements) return InvalidInput;
    const ShareErrorCode rc = hbmpc_memcpy_d2h(pipe->ctx, host, it->second.p, elements * pipe->f.eb, pipe->stream);
    return rc != ShareSuccess ? rc : hbmpc_stream_sync(pipe->ctx, pipe->stream);
}
extern "C" ShareErrorCode hbmpc_pipe_set_checked(hbmpc_pipe* pipe, int on) {
    if (!pipe) return InvalidInput;
    pipe->checked = on != 0;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_pipe_run(hbmpc_pipe* pipe) {
    if (!pipe) return InvalidInput;
    return guarded([&] { pipe->run(); });
}
extern "C" ShareErrorCode hbmpc_pipe_deal(hbmpc_pipe* pipe) {
    if (!pipe) return InvalidInput;
    return guarded([&] { pipe->deal(); });
}
extern "C" ShareErrorCode hbmpc_pipe_finish(hbmpc_pipe* pipe) {
    if (!pipe) return InvalidInput;
    return guarded([&] { pipe->finish(); });
}
// Two eager runs, then the same calls are recorded: nothing can be built during capture, and a mid-size decode builds its
// matrix-core table when it first sees a sender set -- the recorded launches are the ones an eager caller gets from then on.
extern "C" ShareErrorCode hbmpc_pipe_capture(hbmpc_pipe* pipe) {
    if (!pipe || !pipe->stream) return InvalidInput;  // capture needs an explicit stream
    return guarded([&] {
        const bool was_checked = pipe->checked;
        pipe->checked = false;  // a captured run only enqueues
        struct Restore {
            hbmpc_pipe* p;
            bool v;
            ~Restore() { p->checked = v; }
        } restore{pipe, was_checked};
        pipe->run();
        pipe->run();
        PL(hbmpc_stream_sync(pipe->ctx, pipe->stream));
        PL(hbmpc_graph_begin_capture(pipe->ctx, pipe->stream));
        try {
            pipe->run();
        } catch (...) {
            hbmpc_graph* g = nullptr;
            (void)hbmpc_graph_end_capture(pipe->ctx, pipe->stream, &g);
            hbmpc_graph_destroy(g);
            throw;
        }
        hbmpc_graph_destroy(pipe->graph);
        pipe->graph = nullptr;
        PL(hbmpc_graph_end_capture(pipe->ctx, pipe->stream, &pipe->graph));
    });
}
extern "C" ShareErrorCode hbmpc_pipe_replay(hbmpc_pipe* pipe) {
    if (!pipe || !pipe->graph) return InvalidInput;
    return hbmpc_graph_launch(pipe->ctx, pipe->graph, pipe->stream);
}
extern "C" ShareErrorCode hbmpc_pipe_sync(hbmpc_pipe* pipe) {
    if (!pipe) return InvalidInput;
    return hbmpc_stream_sync(pipe->ctx, pipe->stream);
}
extern "C" ShareErrorCode hbmpc_pipe_summary(hbmpc_pipe* pipe, hbmpc_recover_summary* summary_out) {
    if (!pipe || !summary_out || !pipe->summ) return InvalidInput;
    const ShareErrorCode rc = hbmpc_memcpy_d2h(pipe->ctx, summary_out, pipe->summ, sizeof *summary_out, pipe->stream);
    return rc != ShareSuccess ? rc : hbmpc_stream_sync(pipe->ctx, pipe->stream);
}
extern "C" ShareErrorCode hbmpc_pipe_verdict(hbmpc_pipe* pipe, uint32_t verdict_out[2]) {
    if (!pipe || !verdict_out) return InvalidInput;
    return guarded([&] { pipe->verdict(verdict_out); });
}
