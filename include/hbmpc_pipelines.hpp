// hbmpc_pipelines.hpp -- device-resident replays of the reference's arithmetic pipelines for ALL n simulated
// parties on one GPU (how every reference test and bench runs: n parties in one process on FakeNetwork), as a
// header-only C++ host over the hbmpc_dev_* entry points of hbmpc_hip.h.  Host-side orchestration only: every
// arithmetic step is a device call, buffers never leave HBM, the parties' all-to-all is a layout (strided sender
// rows).  The same classes exist in Python (mpc-protocols_amd/pipelines.py) for the tests and bench.py.
//
//   TripleGen   TripleGenNode::init_batch + BatchReconNode (degree 2t) + try_finalize_triple_gen
//               triple_gen/triple_generation.rs:304-364,164-232; batch_recon/batch_recon.rs:144-185,332-481
//   FpMul       FPMulNode::init = Multiply (Beaver, RBC path) + TruncPrNode
//               fpmul/fpmul.rs:61-110, mul/multiplication.rs:417-426,57-139, fpmul/truncpr.rs:185-318
//   RanSha      RanShaNode: deal, n x n Vandermonde, verifier reconstruction + degree test, output slice
//               share_gen/share_gen.rs:232-289,401-454,516-530,199-203
//   RanDouSha   DouShaNode deal + RanDouShaNode: both Vandermonde products, verifier interpolations + tests, output slice
//               double_share/double_share_generation.rs:151-215, ran_dou_sha/mod.rs:371-449,569-602,314-331
//   Preprocessing   run_preprocessing's triple part (honeybadger/mod.rs:1239-1393): RanSha -> a, b; RanDouSha -> r; TripleGen
//
// run() only ENQUEUES on the stream; after one eager run the same call sequence can be captured into a HIP graph
// (capture()) and replayed (replay()) -- at the batch sizes the protocols really use that removes the launch
// overhead that dominates (fpmul, 16 parties x 1024 elements, party-batched launches: 0.13 ms eager, 0.08 ms
// replayed; with one launch per party and step it was 1.03 ms / 0.27 ms).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "hbmpc_hip.h"

namespace hbmpc {

inline void pl_check(ShareErrorCode rc, hbmpc_ctx* ctx, const char* what) {
    if (rc != ShareSuccess) throw std::runtime_error(std::string(what) + " -> " + std::to_string((int)rc) + ": " + hbmpc_last_error(ctx));
}

// bump allocator over one hbmpc_dev_alloc block (keeps the pipelines free of per-step allocations)
class DeviceArena {
  public:
    DeviceArena(hbmpc_ctx* ctx, size_t bytes) : ctx_(ctx), size_(bytes) {
        void* p = nullptr;
        pl_check(hbmpc_dev_alloc(ctx, bytes, &p), ctx, "hbmpc_dev_alloc");
        base_ = static_cast<unsigned char*>(p);
    }
    ~DeviceArena() { (void)hbmpc_dev_free(ctx_, base_); }
    DeviceArena(const DeviceArena&) = delete;
    DeviceArena& operator=(const DeviceArena&) = delete;
    U256* take(size_t elements) { return static_cast<U256*>(take_bytes(elements * sizeof(U256))); }
    void* take_bytes(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (off_ + bytes > size_) throw std::runtime_error("DeviceArena exhausted");
        void* p = base_ + off_;
        off_ += bytes;
        return p;
    }

  private:
    hbmpc_ctx* ctx_;
    unsigned char* base_ = nullptr;
    size_t size_, off_ = 0;
};

class CapturablePipeline {
  public:
    virtual ~CapturablePipeline() { hbmpc_graph_destroy(graph_); }
    virtual void run() = 0;  // enqueue only
    // Two eager runs, then the same calls are recorded: nothing can be built during capture, and a mid-size decode builds
    // its matrix-core table the second time it sees a sender set (hbmpc_set_matrix_cores) -- the recorded launches are then
    // the ones an eager caller gets from its second call on.
    void capture() {
        if (!stream_) throw std::runtime_error("capture needs an explicit stream");
        run();
        run();
        pl_check(hbmpc_stream_sync(ctx_, stream_), ctx_, "sync");
        pl_check(hbmpc_graph_begin_capture(ctx_, stream_), ctx_, "begin_capture");
        try {
            run();
        } catch (...) {
            hbmpc_graph* g = nullptr;
            (void)hbmpc_graph_end_capture(ctx_, stream_, &g);
            hbmpc_graph_destroy(g);
            throw;
        }
        hbmpc_graph_destroy(graph_);
        graph_ = nullptr;
        pl_check(hbmpc_graph_end_capture(ctx_, stream_, &graph_), ctx_, "end_capture");
    }
    void replay() { pl_check(hbmpc_graph_launch(ctx_, graph_, stream_), ctx_, "graph_launch"); }
    void sync() { pl_check(hbmpc_stream_sync(ctx_, stream_), ctx_, "sync"); }
    void upload(U256* dst, const U256* src, size_t elements) {
        pl_check(hbmpc_memcpy_h2d(ctx_, dst, src, elements * sizeof(U256), stream_), ctx_, "h2d");
    }
    void download(U256* dst, const U256* src, size_t elements) {
        pl_check(hbmpc_memcpy_d2h(ctx_, dst, src, elements * sizeof(U256), stream_), ctx_, "d2h");
        sync();
    }
    // the summary of the last decode: {n_fallback, n_failed, first_failed, first_error}
    hbmpc_recover_summary last_summary(const hbmpc_recover_summary* dev) {
        hbmpc_recover_summary s;
        pl_check(hbmpc_memcpy_d2h(ctx_, &s, dev, sizeof s, stream_), ctx_, "d2h");
        sync();
        return s;
    }

  protected:
    CapturablePipeline(hbmpc_ctx* ctx, void* stream) : ctx_(ctx), stream_(stream) {}
    hbmpc_ctx* ctx_;
    void* stream_;
    hbmpc_graph* graph_ = nullptr;
};

// n parties, threshold t, N triples (a multiple of 2t+1).  Buffers are [party][N] canonical elements.
class TripleGen : public CapturablePipeline {
  public:
    TripleGen(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream)
        : CapturablePipeline(ctx, stream), n_(n), t_(t), N_(N), m_(2 * t + 1), G_(N / (2 * t + 1)),
          arena_(ctx, (5 * n * N + n * n * (N / (2 * t + 1)) + n * (N / (2 * t + 1)) + N) * 32 + (n + 2) * (N / (2 * t + 1)) + (1 << 14)) {
        if (N % m_) throw std::invalid_argument("N must be a multiple of 2t+1");
        a = arena_.take(n * N), b = arena_.take(n * N), r2t = arena_.take(n * N), rt = arena_.take(n * N), c = arena_.take(n * N);
        Y_ = arena_.take(n * n * G_);
        Z_ = arena_.take(n * G_);
        opened_ = arena_.take(N);
        status_ = static_cast<uint8_t*>(arena_.take_bytes(n * G_));
        summ = static_cast<hbmpc_recover_summary*>(arena_.take_bytes(64));
        for (size_t i = 0; i < n; ++i) ids_.push_back(i);
    }
    void run() override {
        const size_t n = n_, N = N_, G = G_, d = 2 * t_;
        // [ab - r]_2t = a_i b_i - r2t_i (triple_generation.rs:333-340) Vandermonde-encoded in chunks of 2t+1 for every
        // recipient (batch_recon.rs:157-165): a, b, r2t [party][N] -> Y[party][n][G], all parties in ONE launch; the local
        // products stay on chip where the fused kernel covers the shape (c is the workspace of the two-launch path)
        pl_check(hbmpc_dev_triple_encode_parties(ctx_, a, b, r2t, G, n, d, n, c, Y_, stream_), ctx_, "local product + encode");
        // EvalBatch arm for ALL recipients in one call: with Y[p][j][g] the row of sender p for "chunk" j G + g is
        // Y + p (n G) + (j G + g), and the output is already Z[j][g]
        pl_check(hbmpc_dev_batch_recover_strided(ctx_, ids_.data(), n, Y_, n * G, n * G, n, d, t_, 1, Z_, nullptr, status_, summ,
                                                 stream_), ctx_, "decode y_j");
        // RevealBatch arm: everyone interpolates the 2t+1 opened values per chunk from the n broadcast y_j
        pl_check(hbmpc_dev_batch_recover(ctx_, ids_.data(), n, Z_, G, n, d, t_, opened_, nullptr, status_, summ, stream_), ctx_, "open");
        // [c]_t = rt_i + opened  (triple_generation.rs:196-208), all parties in one launch
        pl_check(hbmpc_dev_triple_finalize_parties(ctx_, rt, opened_, N, n, c, stream_), ctx_, "triple_finalize");
    }
    U256 *a, *b, *r2t, *rt, *c;  // [party][N]
    hbmpc_recover_summary* summ;

  private:
    size_t n_, t_, N_, m_, G_;
    DeviceArena arena_;
    U256 *Y_, *Z_, *opened_;
    uint8_t* status_;
    std::vector<size_t> ids_;
};

// Fixed-point multiplication of N element pairs for n parties: Beaver mul (a-x, b-y opened by direct robust
// interpolation, the RBC path FPMulNode always takes) followed by TruncPr with k-bit values and m fractional bits.
// open_senders: how many parties' shares an open interpolates from (0 = the default 2t+1: the reference opens as soon as
// that many have arrived, multiplication.rs:388,617, truncpr.rs:202 -- with d = t a decode with no OEC round, one launch).
class FpMul : public CapturablePipeline {
  public:
    FpMul(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, size_t k, size_t m, void* stream, size_t open_senders = 0)
        : CapturablePipeline(ctx, stream), n_(n), t_(t), N_(N), k_(k), m_(m),
          arena_(ctx, ((12 + m) * n * N + 4 * N) * 32 + 8 * N + (1 << 14)) {
        U256** per_party[] = {&x, &y, &ta, &tb, &tc, &rint, &z, &rdash_, &osh_, &out};
        for (U256** q : per_party) *q = arena_.take(n * N);
        desh_ = arena_.take(2 * n * N);  // [party][2][N]: a party's shares of a - x and of b - y side by side
        rbits = arena_.take(n * m * N);  // [party][bit][N]
        dop_ = arena_.take(2 * N), eop_ = dop_ + N;  // the opened a - x [N], then the opened b - y [N]
        cop_ = arena_.take(N);
        status_ = static_cast<uint8_t*>(arena_.take_bytes(2 * N));
        summ = static_cast<hbmpc_recover_summary*>(arena_.take_bytes(64));
        if (open_senders == 0) open_senders = 2 * t + 1;
        if (open_senders < 2 * t + 1 || open_senders > n) throw std::invalid_argument("FpMul: 2t+1 <= open_senders <= n");
        for (size_t i = 0; i < open_senders; ++i) ids_.push_back(i);
    }
    void run() override {
        const size_t n = n_, N = N_;
        // one launch per step for all parties (the [party][N] arrays are contiguous; opened values are broadcast)
        pl_check(hbmpc_dev_beaver_open_shares_paired(ctx_, ta, tb, x, y, N, n, desh_, stream_), ctx_, "beaver_open_shares");  // multiplication.rs:417-426
        // reconstruct_rbc: per-element recover_secret of a - x and of b - y (:102-139) -- ONE call over the 2 N values of a sender row
        open(desh_, dop_, "open a-x, b-y", 2 * N);
        // finalize_mul (:57-100), r' (truncpr.rs:277-283) and the share TruncPr opens (:294-297): one launch
        pl_check(hbmpc_dev_fpmul_middle(ctx_, tc, x, y, dop_, eop_, rbits, rint, k_, m_, N, n, z, rdash_, osh_, stream_), ctx_, "fpmul_middle");
        open(osh_, cop_, "open b+r");  // truncpr.rs:215
        pl_check(hbmpc_dev_truncpr_finalize_parties(ctx_, z, rdash_, cop_, m_, N, n, out, stream_), ctx_, "truncpr_finalize");  // :216-220
    }
    U256 *x, *y, *ta, *tb, *tc, *rint, *rbits, *z, *out;  // [party][N] (rbits: [party][bit][N])
    hbmpc_recover_summary* summ;

  private:
    void open(const U256* shares, U256* dst, const char* what, size_t values = 0) {
        pl_check(hbmpc_dev_batch_recover_p0(ctx_, ids_.data(), ids_.size(), shares, values ? values : N_, n_, t_, t_, dst, status_, summ, stream_),
                 ctx_, what);
    }
    size_t n_, t_, N_, k_, m_;
    DeviceArena arena_;
    U256 *desh_, *rdash_, *osh_, *dop_, *eop_, *cop_;
    uint8_t* status_;
    std::vector<size_t> ids_;
};

// What RanSha and RanDouSha share.  The dealers' polynomials are the INPUT (coefficient rows [dealer][K][deg + 1], column 0 the
// secret: uploaded by the host or filled on the device by hbmpc_dev_fill_coeffs -- the reference draws them from each party's rng).
// Layouts (all n parties on one device): dealt S[p][j, k] --n x n Vandermonde over the rows p--> y[i][j, k]; what party j
// sends verifier i is y[i][j K .. j K + K): a strided sender row, nothing is copied.
class Producer : public CapturablePipeline {
  protected:
    Producer(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, void* stream) : CapturablePipeline(ctx, stream), n_(n), t_(t), K_(K) {}
    void deal(const U256* coeffs, size_t deg, U256* S) {
        for (size_t p = 0; p < n_; ++p)  // dealer p: compute_shares of its K polynomials
            pl_check(hbmpc_dev_compute_shares(ctx_, coeffs + p * K_ * (deg + 1), K_, n_, deg, S + p * n_ * K_, stream_), ctx_, "deal");
    }
    void mix(const U256* S, U256* x, U256* y) {
        // the share of dealer p for (recipient, element) is row p of S: the n x n map reads the dealers' outputs in place; x is
        // the workspace of the shapes that have to be transposed first
        pl_check(hbmpc_dev_vandermonde_apply_rows(ctx_, S, n_ * K_, n_ * K_, n_, n_ - 1, x, y, stream_), ctx_, "n x n Vandermonde over the dealt shares");
    }
  public:
    // where a slice of every party's output list goes instead of the producer's own buffer: batch elements [k0, k0 + count)
    // of party p to dst + p * stride
    struct Slice {
        U256* dst;
        size_t stride, k0, count;
    };

  protected:
    void clear_bad(uint32_t* bad) {
        static const uint32_t init[2] = {0u, 0xffffffffu};
        pl_check(hbmpc_memcpy_h2d(ctx_, bad, init, sizeof init, stream_), ctx_, "h2d");
    }
    size_t n_, t_, K_;

  public:
    // {number of verifier checks that failed, first failing batch element}: zero means every verifier says OK
    void verdict(const uint32_t* bad_dev, uint32_t out[2]) {
        pl_check(hbmpc_memcpy_d2h(ctx_, out, bad_dev, 8, stream_), ctx_, "d2h");
        sync();
    }
};

// K batch elements per dealer -> (n - 2t) K random degree-t sharings per party, verified by parties 0 .. 2t - 1 from the
// shares of the first verify_senders parties (0 = the default 2t + 1: the handler fires when that many have arrived,
// share_gen.rs:497 -- with degree t a decode with no OEC round)
class RanSha : public Producer {
  public:
    RanSha(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, void* stream, size_t verify_senders = 0)
        : Producer(ctx, n, t, K, stream), nout((n - 2 * t) * K),
          arena_(ctx, (n * K * (t + 1) + 3 * n * n * K + K * (t + 1) + n * (n - 2 * t) * K) * 32 + K + (1 << 14)) {
        if (n <= 2 * t) throw std::invalid_argument("RanSha: n > 2t");
        if (verify_senders == 0) verify_senders = 2 * t + 1;
        if (verify_senders < 2 * t + 1 || verify_senders > n) throw std::invalid_argument("RanSha: 2t+1 <= verify_senders <= n");
        coeffs = arena_.take(n * K * (t + 1));
        S = arena_.take(n * n * K), x_ = arena_.take(n * n * K), y_ = arena_.take(n * n * K);
        poly_ = arena_.take(K * (t + 1));
        status_ = static_cast<uint8_t*>(arena_.take_bytes(K));
        summ_ = static_cast<hbmpc_recover_summary*>(arena_.take_bytes(64));
        bad = static_cast<uint32_t*>(arena_.take_bytes(64));
        out = arena_.take(n * nout);
        for (size_t i = 0; i < verify_senders; ++i) ids_.push_back(i);
    }
    void deal() { Producer::deal(coeffs, t_, S); }
    // everything after the dealers' messages have arrived; with `split`, the output slices go where it says instead of `out`
    void finish(const std::vector<Slice>& split = {}) {
        const size_t n = n_, t = t_, K = K_;
        mix(S, x_, y_);
        clear_bad(bad);
        for (size_t i = 0; i < 2 * t; ++i) {  // verifier i: recover_secret of the K columns + exact-degree test (share_gen.rs:516-530)
            pl_check(hbmpc_dev_batch_recover_strided(ctx_, ids_.data(), ids_.size(), y_ + i * n * K, K, K, n, t, t, 0, poly_, nullptr, status_,
                                                     summ_, stream_), ctx_, "verifier reconstruction");
            pl_check(hbmpc_dev_check_degree(ctx_, poly_, status_, K, t + 1, t, bad, stream_), ctx_, "degree test");
        }
        // rows 2t .. n - 1 of every batch element, per party in the order [k][i - 2t]  (share_gen.rs:199-203)
        const std::vector<Slice> whole = {{out, nout, 0, K}};
        for (const Slice& sl : split.empty() ? whole : split)
            pl_check(hbmpc_dev_transpose(ctx_, y_ + 2 * t * n * K + sl.k0, n - 2 * t, sl.count, n * K, sl.dst, n - 2 * t, n, K, sl.stride, stream_), ctx_,
                     "output shares");
    }
    void run() override {
        deal();
        finish();
    }
    void run(const std::vector<Slice>& split) {
        deal();
        finish(split);
    }
    const size_t nout;  // output shares per party
    U256 *coeffs, *S, *out;  // [dealer][K][t + 1]; [dealer][recipient][K]; [party][K][n - 2t]
    uint32_t* bad;

  private:
    DeviceArena arena_;
    U256 *x_, *y_, *poly_;
    uint8_t* status_;
    hbmpc_recover_summary* summ_;
    std::vector<size_t> ids_;
};

// K batch elements per dealer -> (t + 1) K double sharings per party, verified by parties t + 1 .. n - 1, each of which
// interpolates both polynomials through ALL n shares (ran_dou_sha/mod.rs:557-602)
class RanDouSha : public Producer {
  public:
    RanDouSha(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, void* stream)
        : Producer(ctx, n, t, K, stream), nout((t + 1) * K),
          arena_(ctx, (n * K * (3 * t + 2) + 5 * n * n * K + 2 * K * n + 2 * n * (t + 1) * K) * 32 + 8 * K + (1 << 14)) {
        coeffs_t = arena_.take(n * K * (t + 1)), coeffs_2t = arena_.take(n * K * (2 * t + 1));
        S_t = arena_.take(n * n * K), S_2t = arena_.take(n * n * K);
        x_ = arena_.take(n * n * K), y_t_ = arena_.take(n * n * K), y_2t_ = arena_.take(n * n * K);
        poly_t_ = arena_.take(K * n), poly_2t_ = arena_.take(K * n);
        deg_ = static_cast<uint32_t*>(arena_.take_bytes(4 * K));
        bad = static_cast<uint32_t*>(arena_.take_bytes(64));
        out_t = arena_.take(n * nout), out_2t = arena_.take(n * nout);
        for (size_t i = 0; i < n; ++i) ids_.push_back(i);
    }
    void deal() {  // DouShaNode::init_batch: both sharings of every secret
        Producer::deal(coeffs_t, t_, S_t);
        Producer::deal(coeffs_2t, 2 * t_, S_2t);
    }
    void finish(const std::vector<Slice>& split_t = {}, const std::vector<Slice>& split_2t = {}) {
        const size_t n = n_, t = t_, K = K_;
        mix(S_t, x_, y_t_);    // RanDouShaNode::init_batch step 1
        mix(S_2t, x_, y_2t_);  // step 2
        clear_bad(bad);
        for (size_t i = t + 1; i < n; ++i) {  // step 3: verifier i
            pl_check(hbmpc_dev_batch_interpolate(ctx_, ids_.data(), n, y_t_ + i * n * K, K, K, n, poly_t_, deg_, stream_), ctx_, "interpolate [r]_t");
            pl_check(hbmpc_dev_batch_interpolate(ctx_, ids_.data(), n, y_2t_ + i * n * K, K, K, n, poly_2t_, deg_, stream_), ctx_, "interpolate [r]_2t");
            pl_check(hbmpc_dev_check_double_share(ctx_, poly_t_, poly_2t_, K, n, t, bad, stream_), ctx_, "degree / equal-secret tests");
        }
        // steps 4-5: rows 0 .. t, per party in the order [k][i]  (ran_dou_sha/mod.rs:314-331)
        const std::vector<Slice> whole_t = {{out_t, nout, 0, K}}, whole_2t = {{out_2t, nout, 0, K}};
        for (const Slice& sl : split_t.empty() ? whole_t : split_t)
            pl_check(hbmpc_dev_transpose(ctx_, y_t_ + sl.k0, t + 1, sl.count, n * K, sl.dst, t + 1, n, K, sl.stride, stream_), ctx_, "output [r]_t");
        for (const Slice& sl : split_2t.empty() ? whole_2t : split_2t)
            pl_check(hbmpc_dev_transpose(ctx_, y_2t_ + sl.k0, t + 1, sl.count, n * K, sl.dst, t + 1, n, K, sl.stride, stream_), ctx_, "output [r]_2t");
    }
    void run() override {
        deal();
        finish();
    }
    void run(const std::vector<Slice>& split_t, const std::vector<Slice>& split_2t) {
        deal();
        finish(split_t, split_2t);
    }
    const size_t nout;
    U256 *coeffs_t, *coeffs_2t, *S_t, *S_2t, *out_t, *out_2t;
    uint32_t* bad;

  private:
    DeviceArena arena_;
    U256 *x_, *y_t_, *y_2t_, *poly_t_, *poly_2t_;
    uint32_t* deg_;
    std::vector<size_t> ids_;
};

// run_preprocessing's triple part for all n parties, device-resident from the dealers' polynomials to [c]_t: RanSha produces
// 2 N random sharings per party (a = the first N, b = the next N: take_random_shares twice, honeybadger/mod.rs:1307-1316),
// RanDouSha the N double sharings, TripleGen consumes them where they lie.
class Preprocessing {
  public:
    Preprocessing(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream)
        : rs(ctx, n, t, (2 * N + (n - 2 * t) - 1) / (n - 2 * t), stream), rd(ctx, n, t, (N + t) / (t + 1), stream), tg(ctx, n, t, N, stream),
          ctx_(ctx), stream_(stream), n_(n), t_(t), N_(N) {}
    void run() {
        const size_t t = t_;
        if (N_ % (n_ - 2 * t) == 0 && N_ % (t + 1) == 0) {
            // whole batch elements on both sides of every cut: the producers' output slices go straight into TripleGen's
            // [party][N] arrays and nothing is copied
            const size_t k1 = N_ / (n_ - 2 * t), k2 = N_ / (t + 1);
            rs.run({{tg.a, N_, 0, k1}, {tg.b, N_, k1, k1}});
            rd.run({{tg.rt, N_, 0, k2}}, {{tg.r2t, N_, 0, k2}});
            tg.run();
            return;
        }
        rs.run();
        rd.run();
        for (size_t p = 0; p < n_; ++p) {  // the parties' lists, in the reference's order, become TripleGen's [party][N] inputs
            copy(tg.a + p * N_, rs.out + p * rs.nout);
            copy(tg.b + p * N_, rs.out + p * rs.nout + N_);
            copy(tg.rt + p * N_, rd.out_t + p * rd.nout);
            copy(tg.r2t + p * N_, rd.out_2t + p * rd.nout);
        }
        tg.run();
    }
    RanSha rs;
    RanDouSha rd;
    TripleGen tg;

  private:
    void copy(U256* dst, const U256* src) { pl_check(hbmpc_memcpy_d2d(ctx_, dst, src, N_ * sizeof(U256), stream_), ctx_, "d2d"); }
    hbmpc_ctx* ctx_;
    void* stream_;
    size_t n_, t_, N_;
};

}  // namespace hbmpc
