// hbmpc_pipelines.hpp -- C++ convenience over the hbmpc_pipe_* handles of hbmpc_hip.h (csrc/capi_pipelines.hip): device-resident
// replays of the reference's arithmetic pipelines for ALL n simulated parties on one GPU (how every reference test and bench
// runs: n parties in one process on FakeNetwork).  The call sequencing, the arena layout and the capture rules live in the
// LIBRARY, behind the C ABI; this header only gives the handles RAII, exceptions and typed buffer access.  The same wrappers
// exist in Python (mpc-protocols_amd/pipelines.py) and for Rust (rust/gpu_shares.rs).
//
//   TripleGen      TripleGenNode::init_batch + BatchReconNode (degree 2t) + try_finalize_triple_gen
//                  triple_gen/triple_generation.rs:304-364,164-232; batch_recon/batch_recon.rs:144-185,332-481
//   FpMul          FPMulNode::init = Multiply (Beaver, RBC path) + TruncPrNode
//                  fpmul/fpmul.rs:61-110, mul/multiplication.rs:417-426,57-139, fpmul/truncpr.rs:185-318
//   RanSha         RanShaNode: deal, n x n Vandermonde, verifier reconstruction + degree test, output slice
//                  share_gen/share_gen.rs:232-289,401-454,516-530,199-203
//   RanDouSha      DouShaNode deal + RanDouShaNode: both Vandermonde products, verifier interpolations + tests, output slice
//                  double_share/double_share_generation.rs:151-215, ran_dou_sha/mod.rs:371-449,569-602,314-331
//   Preprocessing  run_preprocessing's triple part (honeybadger/mod.rs:1239-1393): RanSha -> a, b; RanDouSha -> r; TripleGen
//
// run() only ENQUEUES on the stream; capture() records the same call sequence into a HIP graph after two eager runs and
// replay() launches it -- at the batch sizes the protocols really use that removes the launch overhead that dominates.
#pragma once
#include <stdexcept>
#include <string>

#include "hbmpc_hip.h"

namespace hbmpc {

inline void pl_check(ShareErrorCode rc, hbmpc_ctx* ctx, const char* what) {
    if (rc != ShareSuccess) throw std::runtime_error(std::string(what) + " -> " + std::to_string((int)rc) + ": " + hbmpc_last_error(ctx));
}

class Pipeline {
  public:
    Pipeline(const Pipeline&) = delete;
    Pipeline& operator=(const Pipeline&) = delete;
    virtual ~Pipeline() {
        if (owned_) hbmpc_pipe_destroy(h_);
    }
    // a named device buffer of the pipeline (see the list in hbmpc_hip.h), e.g. buffer("a"), buffer("out")
    U256* buffer(const char* name, size_t* elements = nullptr) const {
        void* p = nullptr;
        pl_check(hbmpc_pipe_buffer(h_, name, &p, elements), ctx_, name);
        return static_cast<U256*>(p);
    }
    void upload(const char* name, const U256* src, size_t elements) { pl_check(hbmpc_pipe_upload(h_, name, src, elements), ctx_, name); }
    void download(const char* name, U256* dst, size_t elements) { pl_check(hbmpc_pipe_download(h_, name, dst, elements), ctx_, name); }
    // the same by device pointer (any position inside a buffer), on the pipeline's stream; download synchronises
    void upload(U256* dst_dev, const U256* src, size_t elements) {
        pl_check(hbmpc_memcpy_h2d(ctx_, dst_dev, src, elements * sizeof(U256), stream_), ctx_, "h2d");
    }
    void download(U256* dst, const U256* src_dev, size_t elements) {
        pl_check(hbmpc_memcpy_d2h(ctx_, dst, src_dev, elements * sizeof(U256), stream_), ctx_, "d2h");
        sync();
    }
    void run(bool checked = false) {  // enqueue only; checked: the summary is read back after every decode
        pl_check(hbmpc_pipe_set_checked(h_, checked ? 1 : 0), ctx_, "set_checked");
        pl_check(hbmpc_pipe_run(h_), ctx_, "run");
    }
    void capture() { pl_check(hbmpc_pipe_capture(h_), ctx_, "capture"); }
    void replay() { pl_check(hbmpc_pipe_replay(h_), ctx_, "replay"); }
    void sync() { pl_check(hbmpc_pipe_sync(h_), ctx_, "sync"); }
    hbmpc_recover_summary last_summary() {  // of the last decode: {n_fallback, n_failed, first_failed, first_error}
        hbmpc_recover_summary s;
        pl_check(hbmpc_pipe_summary(h_, &s), ctx_, "summary");
        return s;
    }
    hbmpc_pipe* handle() const { return h_; }

  protected:
    Pipeline(hbmpc_ctx* ctx, hbmpc_pipe* h, void* stream, bool owned = true) : ctx_(ctx), h_(h), stream_(stream), owned_(owned) {}
    hbmpc_ctx* ctx_;
    hbmpc_pipe* h_;
    void* stream_;
    bool owned_;
};

// n parties, threshold t, N triples (a multiple of 2t+1).  Buffers a, b, r2t, rt (inputs), c (output): [party][N].
class TripleGen : public Pipeline {
  public:
    TripleGen(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream) : Pipeline(ctx, create(ctx, n, t, N, stream), stream) { bind(); }
    TripleGen(hbmpc_ctx* ctx, hbmpc_pipe* borrowed, void* stream) : Pipeline(ctx, borrowed, stream, false) { bind(); }
    U256 *a, *b, *r2t, *rt, *c;

  private:
    static hbmpc_pipe* create(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream) {
        hbmpc_pipe* h = nullptr;
        pl_check(hbmpc_pipe_triplegen_create(ctx, n, t, N, stream, &h), ctx, "hbmpc_pipe_triplegen_create");
        return h;
    }
    void bind() { a = buffer("a"), b = buffer("b"), r2t = buffer("r2t"), rt = buffer("rt"), c = buffer("c"); }
};

// Fixed-point multiplication of N element pairs for n parties: Beaver mul + TruncPr with k-bit values and m fractional bits.
// open_senders: how many parties' shares an open interpolates from (0 = the reference's 2t+1).
class FpMul : public Pipeline {
  public:
    FpMul(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, size_t k, size_t m, void* stream, size_t open_senders = 0)
        : Pipeline(ctx, create(ctx, n, t, N, k, m, open_senders, stream), stream) {
        x = buffer("x"), y = buffer("y"), ta = buffer("ta"), tb = buffer("tb"), tc = buffer("tc"), rint = buffer("rint"), rbits = buffer("rbits");
        z = buffer("z"), out = buffer("out");
    }
    U256 *x, *y, *ta, *tb, *tc, *rint, *rbits, *z, *out;  // [party][N] (rbits: [party][bit][N])

  private:
    static hbmpc_pipe* create(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, size_t k, size_t m, size_t open_senders, void* stream) {
        hbmpc_pipe* h = nullptr;
        pl_check(hbmpc_pipe_fpmul_create(ctx, n, t, N, k, m, open_senders, stream, &h), ctx, "hbmpc_pipe_fpmul_create");
        return h;
    }
};

class Producer : public Pipeline {
  public:
    void deal() { pl_check(hbmpc_pipe_deal(h_), ctx_, "deal"); }      // the dealers' compute_shares
    void finish() { pl_check(hbmpc_pipe_finish(h_), ctx_, "finish"); }  // everything after the dealers' messages have arrived
    // {number of verifier checks that failed, first failing batch element}: zero means every verifier says OK (synchronises)
    void verdict(uint32_t out[2]) { pl_check(hbmpc_pipe_verdict(h_, out), ctx_, "verdict"); }

  protected:
    using Pipeline::Pipeline;
};

// K batch elements per dealer -> (n - 2t) K random degree-t sharings per party ("out": [party][K][n - 2t]), verified by parties
// 0 .. 2t - 1 from the shares of the first verify_senders parties (0 = the reference's 2t + 1)
class RanSha : public Producer {
  public:
    RanSha(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, void* stream, size_t verify_senders = 0)
        : Producer(ctx, create(ctx, n, t, K, verify_senders, stream), stream), nout((n - 2 * t) * K) {
        bind();
    }
    RanSha(hbmpc_ctx* ctx, hbmpc_pipe* borrowed, void* stream, size_t n, size_t t, size_t K) : Producer(ctx, borrowed, stream, false), nout((n - 2 * t) * K) { bind(); }
    const size_t nout;       // output shares per party
    U256 *coeffs, *S, *out;  // [dealer][K][t + 1]; [dealer][recipient][K]; [party][K][n - 2t]

  private:
    static hbmpc_pipe* create(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, size_t verify_senders, void* stream) {
        hbmpc_pipe* h = nullptr;
        pl_check(hbmpc_pipe_ransha_create(ctx, n, t, K, verify_senders, stream, &h), ctx, "hbmpc_pipe_ransha_create");
        return h;
    }
    void bind() { coeffs = buffer("coeffs"), S = buffer("S"), out = buffer("out"); }
};

// K batch elements per dealer -> (t + 1) K double sharings per party, verified by parties t + 1 .. n - 1
class RanDouSha : public Producer {
  public:
    RanDouSha(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, void* stream) : Producer(ctx, create(ctx, n, t, K, stream), stream), nout((t + 1) * K) { bind(); }
    RanDouSha(hbmpc_ctx* ctx, hbmpc_pipe* borrowed, void* stream, size_t t, size_t K) : Producer(ctx, borrowed, stream, false), nout((t + 1) * K) { bind(); }
    const size_t nout;
    U256 *coeffs_t, *coeffs_2t, *S_t, *S_2t, *out_t, *out_2t;

  private:
    static hbmpc_pipe* create(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, void* stream) {
        hbmpc_pipe* h = nullptr;
        pl_check(hbmpc_pipe_randousha_create(ctx, n, t, K, stream, &h), ctx, "hbmpc_pipe_randousha_create");
        return h;
    }
    void bind() {
        coeffs_t = buffer("coeffs_t"), coeffs_2t = buffer("coeffs_2t"), S_t = buffer("S_t"), S_2t = buffer("S_2t");
        out_t = buffer("out_t"), out_2t = buffer("out_2t");
    }
};

// run_preprocessing's triple part for all n parties, device-resident from the dealers' polynomials to [c]_t
class Preprocessing : public Pipeline {
  public:
    Preprocessing(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream)
        : Pipeline(ctx, create(ctx, n, t, N, stream), stream), rs(ctx, part("ransha"), stream, n, t, (2 * N + (n - 2 * t) - 1) / (n - 2 * t)),
          rd(ctx, part("randousha"), stream, t, (N + t) / (t + 1)), tg(ctx, part("triplegen"), stream) {}
    RanSha rs;     // borrowed parts: they live as long as this object
    RanDouSha rd;
    TripleGen tg;
    void verdict(uint32_t out[2]) { pl_check(hbmpc_pipe_verdict(h_, out), ctx_, "verdict"); }  // both producers

  private:
    static hbmpc_pipe* create(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream) {
        hbmpc_pipe* h = nullptr;
        pl_check(hbmpc_pipe_preprocessing_create(ctx, n, t, N, stream, &h), ctx, "hbmpc_pipe_preprocessing_create");
        return h;
    }
    hbmpc_pipe* part(const char* name) {
        hbmpc_pipe* p = nullptr;
        pl_check(hbmpc_pipe_part(h_, name, &p), ctx_, name);
        return p;
    }
};

}  // namespace hbmpc
