/*
 * hbmpc_hip.h -- C ABI of the MI355X (gfx950) Shamir share-arithmetic engine for HoneyBadgerMPC.
 *
 * This is the drop-in boundary for ONE path of Stoffel-Labs/mpc-protocols: the field arithmetic
 * behind RobustShare<F> / SecretSharingScheme<F> (F = ark_bls12_381::Fr) that the async protocol
 * code calls inline.  Each entry point names the reference interface it replaces (paths relative
 * to the reference's mpc/src/).  The Rust side of the boundary is delivered as files: rust/hbmpc_sys.rs (generated from
 * this header by tools/gen_rust_sys.py) and rust/gpu_shares.rs (the adaptor); INTEGRATION.md shows where they plug in.
 *
 * Conventions (they follow the reference's own exported C ABI, ffi/c_bindings/mod.rs:17-35 and
 * ffi/c_bindings/share/mod.rs:18-37, so both ABIs look alike):
 *   - U256 = canonical integer < r, four u64 limbs, least-significant limb first
 *     (Fr <-> U256 exactly as ffi/c_bindings/mod.rs:37-49: into_bigint().0 / from_bigint).
 *     Inputs MUST be canonical (< r); the reference panics on a non-canonical U256
 *     (from_bigint(..).unwrap()), this library leaves the result unspecified.
 *   - every call returns a ShareErrorCode; it is the only error channel
 *     (hbmpc_last_error gives a message for the calling thread's last failure on that ctx).
 *   - the CALLER allocates every output buffer (no allocator is shared across the boundary).
 *   - there is no CPU path: every entry point needs a HIP device and fails with
 *     HBMPC_NO_DEVICE when there is none.
 *   - thread-safe and re-entrant: calls on one ctx from several threads serialise on the ctx's
 *     stream; use one ctx per thread (or the hbmpc_dev_* calls with your own streams) to overlap.
 *
 * Array layouts (B secrets / G chunks are the batch dimension):
 *   "chunk-major"  X[G][m]  : the m = degree+1 coefficients/secrets of one chunk are contiguous
 *                             (how the reference holds them: shares.chunks_exact(degree+1),
 *                             batch_recon.rs:160; Vec<Vec<F>> results, robust_interpolate.rs:401)
 *   "party-major"  Y[n][G]  : one party's (sender's / recipient's) values for all chunks are
 *                             contiguous (y_shares_by_recipient[recipient][chunk],
 *                             batch_recon.rs:158-165; evals_by_sender[i].1[c],
 *                             robust_interpolate.rs:285)
 *
 * Two API levels:
 *   hbmpc_*      host pointers, synchronous (H2D, kernels, D2H inside the call)
 *   hbmpc_dev_*  device pointers + a hipStream_t (as void*), asynchronous: for pipelines that keep
 *                shares resident in HBM between protocol steps (and for the benchmark).
 */
#ifndef HBMPC_HIP_H
#define HBMPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ffi/c_bindings/mod.rs:17-21 */
typedef struct {
    uint64_t data[4];
} U256;

/* ffi/c_bindings/share/mod.rs:18-37 (same order, same values); the HBMPC_* values are new and
 * never produced by the reference: they report conditions the Rust code cannot have. */
typedef enum {
    ShareSuccess = 0,
    InsufficientShares = 1,
    DegreeMismatch = 2,
    IdMismatch = 3,
    InvalidInput = 4,
    TypeMismatch = 5,
    NoSuitableDomain = 6,
    PolynomialOperationError = 7,
    DecodingError = 8,
    HBMPC_NO_DEVICE = 100,    /* no HIP device / HIP runtime error (see hbmpc_last_error) */
    HBMPC_OUT_OF_MEMORY = 101 /* device allocation failed */
} ShareErrorCode;

/* ffi/c_bindings/share/mod.rs:50-53 (FieldKind).  Goldilocks64 is this library's extension for the reference's small
 * field (common/math/goldilocks.rs:4-13: p = 2^64 - 2^32 + 1, generator 7), SURVEY.md section 8(f) row 4. */
typedef enum { Bls12_381Fr = 0, Goldilocks64 = 1 } FieldKind;

/* per-chunk status written by the batch-recover calls */
enum {
    HBMPC_CHUNK_OPTIMISTIC = 0, /* all degree+t+1 lowest senders agreed (robust_interpolate.rs:417) */
    HBMPC_CHUNK_FALLBACK = 1    /* recovered by the OEC/Gao path (robust_interpolate.rs:433-438) */
    /* any other value: the ShareErrorCode recover_secret returned for that chunk */
};

typedef struct hbmpc_ctx hbmpc_ctx;

/* summary of one batch-recover call (device API writes it to device memory) */
typedef struct {
    uint32_t n_fallback;   /* chunks that took the OEC/Gao path                         */
    uint32_t n_failed;     /* chunks whose fallback failed                               */
    uint32_t first_failed; /* lowest failing chunk index (valid when n_failed > 0)       */
    uint32_t first_error;  /* its ShareErrorCode: what the reference's `?` would return  */
} hbmpc_recover_summary;

/* ---- Device buffers ----------------------------------------------------------------------
 * Every *_dev pointer must be device memory that is coherent at kernel boundaries of a stream: hipMalloc (or an
 * allocator that sub-allocates hipMalloc blocks, e.g. PyTorch's caching allocator), or hbmpc_dev_alloc.
 * Memory from the stream-ordered pool (hipMallocAsync) is safe only while the pool RETAINS its blocks
 * (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, ...) above the working set).  With the default
 * threshold 0 the pool returns freed blocks to the driver at every synchronisation and re-acquires them; on ROCm 7.2 /
 * gfx950 kernels that use such a block next can read lines of its previous life and lose results to stale dirty lines
 * (tests/cpp/test_pool_buffers.hip: every second decode of a re-allocated buffer set wrong, hipMalloc buffers in the
 * same loop never; tools/repro_stale.hip for the isolated effect).  A multi-kernel call (batch_recover: optimistic
 * kernel, then the fallback kernels through status bytes, lists and counters) cannot be made safe against that from
 * inside; the library keeps its own scratch in hipMalloc memory and reads every atomically written hand-off word with
 * agent-scope atomic loads.  Pointer attributes do not distinguish the two kinds (HIP_POINTER_ATTRIBUTE_MEMPOOL_HANDLE is
 * hipErrorNotSupported on ROCm 7.2), so no buffer is rejected; hbmpc_stream_pool_release_threshold / hbmpc_stream_pool_retain
 * below report and repair the pool's configuration instead. */

/* ---- context --------------------------------------------------------------------------- */
/* device: HIP device ordinal (>= 0).  There is no CPU mode. */
ShareErrorCode hbmpc_create(int device, FieldKind field_kind, hbmpc_ctx** ctx_out);
void hbmpc_destroy(hbmpc_ctx* ctx);
const char* hbmpc_last_error(const hbmpc_ctx* ctx);
const char* hbmpc_version(void);
FieldKind hbmpc_field_of(const hbmpc_ctx* ctx); /* the field the context was created with */

/* ---- device memory / stream helpers (for hosts without their own HIP binding) ------------ */
/* Device buffers passed to hbmpc_dev_* should come from hbmpc_dev_alloc / hipMalloc (what torch's caching
 * allocator uses).  Stream-ordered pool memory (hipMallocAsync) is not recommended for buffers that one
 * kernel writes and the next reads: see DESIGN.md section 4, "Memory the library hands between kernels". */
ShareErrorCode hbmpc_dev_alloc(hbmpc_ctx* ctx, size_t bytes, void** dptr_out);
ShareErrorCode hbmpc_dev_free(hbmpc_ctx* ctx, void* dptr);
/* Hosts that hand the library buffers from the stream-ordered pool (hipMallocAsync; see "Device buffers" above): the release threshold
 * of the CURRENT pool of the context's device -- 0, the platform's default, is the unsafe configuration -- and the one-call remedy, which
 * makes that pool keep its freed blocks (threshold = UINT64_MAX; hipMemPoolTrimTo still returns memory on request).  Call once, before
 * the first hipMallocAsync whose buffer reaches an hbmpc_dev_* call; a host with its own pool (hipMemPoolCreate) sets the attribute on it. */
ShareErrorCode hbmpc_stream_pool_release_threshold(hbmpc_ctx* ctx, uint64_t* threshold_out);
ShareErrorCode hbmpc_stream_pool_retain(hbmpc_ctx* ctx);
ShareErrorCode hbmpc_memcpy_h2d(hbmpc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes, void* stream);
ShareErrorCode hbmpc_memcpy_d2h(hbmpc_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, void* stream);
ShareErrorCode hbmpc_memcpy_d2d(hbmpc_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes, void* stream);
/* `rows` rows of row_bytes bytes each, the rows dst_pitch_bytes / src_pitch_bytes apart (both >= row_bytes): one copy for the same slice of
 * every party's array */
ShareErrorCode hbmpc_memcpy_d2d_rows(hbmpc_ctx* ctx, void* dst_dev, size_t dst_pitch_bytes, const void* src_dev, size_t src_pitch_bytes,
                                     size_t row_bytes, size_t rows, void* stream);
ShareErrorCode hbmpc_stream_sync(hbmpc_ctx* ctx, void* stream); /* stream == NULL: the ctx's own stream */
ShareErrorCode hbmpc_stream_create(hbmpc_ctx* ctx, void** stream_out); /* a non-blocking hipStream_t on the ctx's device */
ShareErrorCode hbmpc_stream_destroy(hbmpc_ctx* ctx, void* stream);

/* ---- multi-GPU: the final gather ---------------------------------------------------------
 * The path shards by batch index (reference: independent preprocessing sessions,
 * honeybadger/mod.rs:1334-1393): a host drives one ctx per device, every ctx computes the party-major outputs
 * [n_rows][counts[r]] of its contiguous slice of the batch, and nothing crosses devices until the consumer wants
 * the rows of the WHOLE batch in one place.  This call is that step for a single-process host (a Rust node owns all
 * its devices; no RCCL bootstrap, no collective: it is a gather of disjoint column ranges, one peer copy per row
 * and shard over xGMI):
 *   shards_dev[r]   on the device of ctxs[r]: [n_rows][strides[r]] elements, the first counts[r] of each row valid
 *   out_dev         on the device of ctxs[root]: [n_rows][out_stride]; shard r lands at column sum(counts[0..r))
 * Elements are those of the contexts' field (all ctxs must share it).  The copies are enqueued on `stream` of the
 * root (NULL = its own stream) after the streams of all source contexts have been drained (sync_sources != 0), or
 * immediately when the caller has ordered them itself (sync_sources = 0).  Peer access root <- source is enabled
 * on first use when the devices differ. */
ShareErrorCode hbmpc_dev_gather_party_major(hbmpc_ctx* const* ctxs, size_t n_shards, size_t root,
                                            const void* const* shards_dev, const size_t* counts, const size_t* strides,
                                            size_t n_rows, void* out_dev, size_t out_stride, int sync_sources,
                                            void* stream);
/* *direct_out = 1 when root's device reads source's memory directly (same device, or peer access over xGMI, which this
 * call enables): the gather is then one strided copy per shard; 0 when the runtime has to stage the rows through the
 * host.  (The reference has no counterpart: its parties exchange bytes through the Network trait.) */
ShareErrorCode hbmpc_dev_peer_access(hbmpc_ctx* root, hbmpc_ctx* source, int* direct_out);
/* on != 0: the root context's gathers take the per-row peer copies (the branch of device pairs without peer access) even where
 * the one 2-D copy per shard applies -- an A/B aid and the way to exercise that branch on a box with one device; same bytes. */
ShareErrorCode hbmpc_set_gather_row_copies(hbmpc_ctx* root, int on);

/* ---- HIP graphs (for hosts without their own HIP binding) ----------------------------------------
 * The reference's regime is many small protocol steps (a few hundred elements per message); a device-resident
 * pipeline of hbmpc_dev_* calls is then bound by kernel-launch overhead.  Every hbmpc_dev_* call is capturable
 * once its tables and scratch exist, i.e. after ONE eager run of the same call sequence with the same shapes on
 * the same stream: begin capture, issue the calls again (they only record), end capture, replay the graph as
 * often as the buffers are refilled.  (fpmul for 16 parties x 1024 elements: 0.13 ms eager, 0.08 ms replayed.)
 * stream must be a real stream (not NULL).  A call that would have to build a table or grow scratch during
 * capture fails with HBMPC_NO_DEVICE and invalidates the capture. */
typedef struct hbmpc_graph hbmpc_graph;
ShareErrorCode hbmpc_graph_begin_capture(hbmpc_ctx* ctx, void* stream);
ShareErrorCode hbmpc_graph_end_capture(hbmpc_ctx* ctx, void* stream, hbmpc_graph** graph_out);
ShareErrorCode hbmpc_graph_launch(hbmpc_ctx* ctx, hbmpc_graph* graph, void* stream);
void hbmpc_graph_destroy(hbmpc_graph* graph);

/* ---- device-resident pipelines (all n simulated parties on one GPU) -------------------------------------------------
 * The reference's arithmetic pipelines as opaque handles, so that the Rust node (or any C caller) gets the call
 * sequencing, the arena layout and the capture rules from the library:
 *   triplegen      TripleGenNode::init_batch + BatchRecon(2t) + finalize   triple_gen/triple_generation.rs:304-364,164-232
 *   fpmul          FPMulNode::init = Multiply (Beaver, RBC path) + TruncPr  fpmul/fpmul.rs:61-110, fpmul/truncpr.rs:185-318 (Fr only)
 *   ransha         RanShaNode: deal, n x n Vandermonde, verifiers, output   share_gen/share_gen.rs:232-289,401-454,516-530,199-203
 *   randousha      DouShaNode deal + RanDouShaNode                           ran_dou_sha/mod.rs:371-449,569-602,314-331
 *   preprocessing  run_preprocessing's triple part: ransha -> a, b; randousha -> r; triplegen   honeybadger/mod.rs:1239-1393
 * Sizes: N triples (a multiple of 2t+1) / N element pairs / K batch elements per dealer.  open_senders, verify_senders: how
 * many parties' shares an open / a RanSha verifier interpolates from; 0 = the reference's 2t + 1 (it acts as soon as that
 * many have arrived: multiplication.rs:388, truncpr.rs:202, share_gen.rs:497).  Every buffer of a pipeline lives in ONE device
 * block owned by the handle and is reached by name (hbmpc_pipe_buffer; elements of the context's field, [party][...] layouts):
 *   triplegen      a, b, r2t, rt (inputs [n][N]); c (output [n][N]); Y, Z, opened, status, summary
 *   fpmul          x, y, ta, tb, tc, rint ([n][N]), rbits ([n][m][N]) (inputs); out ([n][N]); z, rdash, osh, desh, dop, eop, cop
 *   ransha         coeffs ([dealer][K][t+1], column 0 the secret); S ([dealer][recipient][K]); y; out ([party][K][n-2t]); bad
 *   randousha      coeffs_t, coeffs_2t; S_t, S_2t; y_t, y_2t; out_t, out_2t ([party][K][t+1]); bad
 *   preprocessing  its parts by name (hbmpc_pipe_part: "ransha", "randousha", "triplegen"; borrowed handles)
 * run() only ENQUEUES on the handle's stream (checked mode, hbmpc_pipe_set_checked: it reads the summary back after every
 * decode and returns the failing chunk's error where the reference's `?` would).  deal() / finish() are the two halves of a
 * producer's run (tests corrupt the dealt shares in between).  capture() = two eager runs + a recorded one; replay() launches
 * the recording (stream must be a real stream).  summary(): the last decode's hbmpc_recover_summary; verdict(): a producer's
 * {verifier checks that failed, first failing batch element} (preprocessing: both producers) -- both synchronise.
 * Errors: InvalidInput for shapes the reference rejects or unknown names, TypeMismatch for fpmul on a Goldilocks context. */
typedef struct hbmpc_pipe hbmpc_pipe;
ShareErrorCode hbmpc_pipe_triplegen_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream, hbmpc_pipe** pipe_out);
ShareErrorCode hbmpc_pipe_fpmul_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, size_t k, size_t m, size_t open_senders,
                                       void* stream, hbmpc_pipe** pipe_out);
ShareErrorCode hbmpc_pipe_ransha_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, size_t verify_senders, void* stream,
                                        hbmpc_pipe** pipe_out);
ShareErrorCode hbmpc_pipe_randousha_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t K, void* stream, hbmpc_pipe** pipe_out);
ShareErrorCode hbmpc_pipe_preprocessing_create(hbmpc_ctx* ctx, size_t n, size_t t, size_t N, void* stream, hbmpc_pipe** pipe_out);
void hbmpc_pipe_destroy(hbmpc_pipe* pipe);
ShareErrorCode hbmpc_pipe_part(hbmpc_pipe* pipe, const char* name, hbmpc_pipe** part_out);
ShareErrorCode hbmpc_pipe_buffer(hbmpc_pipe* pipe, const char* name, void** dev_out, size_t* elements_out);
ShareErrorCode hbmpc_pipe_upload(hbmpc_pipe* pipe, const char* name, const void* host, size_t elements);
ShareErrorCode hbmpc_pipe_download(hbmpc_pipe* pipe, const char* name, void* host, size_t elements);
ShareErrorCode hbmpc_pipe_set_checked(hbmpc_pipe* pipe, int on);
ShareErrorCode hbmpc_pipe_run(hbmpc_pipe* pipe);
ShareErrorCode hbmpc_pipe_deal(hbmpc_pipe* pipe);
ShareErrorCode hbmpc_pipe_finish(hbmpc_pipe* pipe);
ShareErrorCode hbmpc_pipe_capture(hbmpc_pipe* pipe);
ShareErrorCode hbmpc_pipe_replay(hbmpc_pipe* pipe);
ShareErrorCode hbmpc_pipe_sync(hbmpc_pipe* pipe);
ShareErrorCode hbmpc_pipe_summary(hbmpc_pipe* pipe, hbmpc_recover_summary* summary_out);
ShareErrorCode hbmpc_pipe_verdict(hbmpc_pipe* pipe, uint32_t verdict_out[2]);

/* ==== a3: RobustShare::compute_shares / NonRobustShare::compute_shares ======================
 * replaces honeybadger/robust_interpolate/robust_interpolate.rs:52-82 and
 * common/share/shamir.rs:158-196 for B secrets at once.
 * coeffs[B][d+1] chunk-major: coeffs[b][0] = secret, coeffs[b][1..=d] = the rng draws of
 * DensePolynomial::rand (the host draws them so results stay bit-exact with its rng).
 * shares_out[n][B] party-major: shares_out[j][b] = share with id j of secret b.
 * Errors: InvalidInput if n <= d (:59-64); NoSuitableDomain if next_pow2(n) > 2^32. */
ShareErrorCode hbmpc_compute_shares(hbmpc_ctx* ctx, const U256* coeffs, size_t B, size_t n, size_t d,
                                    U256* shares_out);
ShareErrorCode hbmpc_dev_compute_shares(hbmpc_ctx* ctx, const U256* coeffs_dev, size_t B, size_t n, size_t d,
                                        U256* shares_out_dev, void* stream);

/* ---- seeded variant: the random coefficients are drawn ON THE DEVICE ---------------------------------------
 * Same sharing as above, but coeffs[b][1..=d] come from a keyed stream this library defines, so only the SECRETS
 * cross the bus (1/(d+1) of the upload of hbmpc_compute_shares) and a device-resident dealer needs no host rng at
 * all.  Contract "hbmpc-chacha20-v1" (restated in oracle/spec.py::seeded_coefficient, which the parity tests use):
 *   coefficient k (1 <= k <= d) of polynomial index i = first_index + b is the first candidate < modulus in the
 *   sequence of ChaCha20 blocks (RFC 8439 block function, 20 rounds; key = seed as eight little-endian words;
 *   state words 12..15 = (attempt, k, i mod 2^32, i div 2^32)) for attempt = 0, 1, ...; each 64-byte block yields,
 *   in order, 64/element-bytes little-endian candidates (Fr: two 32-byte integers with bit 255 cleared; Goldilocks:
 *   eight u64).  Uniform over the field by rejection; coefficients at different (seed, i, k) are independent under
 *   the usual ChaCha20 PRF assumption.
 * This is NOT the stream of the reference's rng (DensePolynomial::rand(degree, rng) with the caller's
 * ark_std/rand generator, robust_interpolate.rs:66-67): a deployment that needs bit-equality with a given Rust rng
 * keeps drawing on the host and calls hbmpc_compute_shares.  first_index lets a dealer split one seed over several
 * calls / GPUs without reusing stream positions (rank r of W deals indices [r B, (r+1) B)).
 * SECURITY: the pair (seed, polynomial index) fixes the random coefficients.  Two calls that use the same seed and
 * overlapping index ranges for DIFFERENT secrets produce sharings with identical higher coefficients, so every party
 * learns share_i(s1) - share_i(s2) = s1 - s2.  Nothing in the library can detect that: the caller must never reuse an
 * index under one seed (take first_index from a counter that only grows, as rust/gpu_shares.rs does, or derive a fresh
 * seed per call).  The coefficient workspace and the library's staging buffers hold secret polynomial coefficients
 * and are recycled without being cleared: clear the workspace yourself if the threat model asks for it.
 * secrets (host or device) may be NULL: the secret of every polynomial is then drawn as well, as coefficient k = 0
 * of the same stream -- the dealer loop of RanSha (share_gen/share_gen.rs:249-256: `F::rand(rng)` followed by
 * compute_shares) without a host rng; the secrets are column 0 of the workspace.
 * coeffs_ws_dev[B][d+1]: caller-provided workspace that receives the full coefficient rows (row b = secret b, then
 * the d draws) -- the dealer usually needs them again (e.g. to open or to verify).
 * Errors: as hbmpc_compute_shares. */
ShareErrorCode hbmpc_dev_fill_coeffs(hbmpc_ctx* ctx, const uint8_t seed[32], const U256* secrets_dev, size_t B,
                                     uint64_t first_index, size_t d, U256* coeffs_out_dev, void* stream);
ShareErrorCode hbmpc_dev_compute_shares_seeded(hbmpc_ctx* ctx, const uint8_t seed[32], const U256* secrets_dev, size_t B,
                                               uint64_t first_index, size_t n, size_t d, U256* coeffs_ws_dev,
                                               U256* shares_out_dev, void* stream);
ShareErrorCode hbmpc_compute_shares_seeded(hbmpc_ctx* ctx, const uint8_t seed[32], const U256* secrets, size_t B,
                                           uint64_t first_index, size_t n, size_t d, U256* shares_out);

/* ==== a4+a5: make_vandermonde + apply_vandermonde ===========================================
 * replaces common/share/mod.rs:31-76 as used by BatchReconNode::init_batch_reconstruct[_many]
 * (batch_recon.rs:114-115,157-165), RanSha (share_gen.rs:415-419) and RanDouSha
 * (ran_dou_sha/mod.rs:392-403).
 * x[G][d+1] chunk-major -> y_out[n][G] party-major, y_out[j][g] = sum_k alpha_j^k x[g][k]. */
ShareErrorCode hbmpc_vandermonde_apply(hbmpc_ctx* ctx, const U256* x, size_t G, size_t n, size_t d, U256* y_out);
ShareErrorCode hbmpc_dev_vandermonde_apply(hbmpc_ctx* ctx, const U256* x_dev, size_t G, size_t n, size_t d,
                                           U256* y_out_dev, void* stream);
/* the encodes of several parties in one launch: x[parties][G][d+1] -> y_out[parties][n][G] (parties in 1..65535) */
ShareErrorCode hbmpc_dev_vandermonde_apply_parties(hbmpc_ctx* ctx, const U256* x_dev, size_t G, size_t n, size_t d,
                                                   size_t parties, U256* y_out_dev, void* stream);
/* make_vandermonde alone (common/share/mod.rs:31-45): v_out[n][d+1], host memory. */
ShareErrorCode hbmpc_make_vandermonde(hbmpc_ctx* ctx, size_t n, size_t d, U256* v_out);

/* ==== a7: batch_recover_secret ==============================================================
 * replaces robust_interpolate.rs:284-443.  sender_ids[S] in ARRIVAL order (sorted inside, :313),
 * evals[S][G] party-major in the same order as sender_ids.
 * coeffs_out[G][d+1] chunk-major, zero-padded to d+1 coefficients.
 * ncoeffs_out[G] (nullable): the length of the Vec<F> the reference returns for that chunk:
 *   d+1 on the optimistic path (:419), the trimmed length on the fallback path (:437-438).
 * status_out[G] (nullable): HBMPC_CHUNK_* per chunk.
 * Return value: what the reference returns -- the validation errors of :290-341, or the error of
 * the lowest-index chunk whose fallback fails (:437 `?`); outputs of other chunks are still written.
 * P(0)-only variant (the EvalBatch arm keeps coeffs[0] only, batch_recon.rs:384-391):
 * secrets_out[G]. */
ShareErrorCode hbmpc_batch_recover(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const U256* evals, size_t G,
                                   size_t n, size_t d, size_t t, U256* coeffs_out, uint32_t* ncoeffs_out,
                                   uint8_t* status_out);
ShareErrorCode hbmpc_batch_recover_p0(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const U256* evals,
                                      size_t G, size_t n, size_t d, size_t t, U256* secrets_out,
                                      uint8_t* status_out);
/* device variants: sender_ids is a HOST array (it selects the constant tables); every other
 * pointer is device memory; summary_dev (nullable) receives an hbmpc_recover_summary.  The return
 * value covers validation only; read summary_dev after the stream has drained for chunk errors. */
ShareErrorCode hbmpc_dev_batch_recover(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const U256* evals_dev,
                                       size_t G, size_t n, size_t d, size_t t, U256* coeffs_out_dev,
                                       uint32_t* ncoeffs_out_dev, uint8_t* status_out_dev,
                                       hbmpc_recover_summary* summary_dev, void* stream);
ShareErrorCode hbmpc_dev_batch_recover_p0(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const U256* evals_dev,
                                          size_t G, size_t n, size_t d, size_t t, U256* secrets_out_dev,
                                          uint8_t* status_out_dev, hbmpc_recover_summary* summary_dev,
                                          void* stream);

/* Strided variant: sender i's values start at evals_dev + i * row_stride elements (row_stride >= G).
 * With all n parties on one device the encode output of party p for recipient j is row (p, j) of a
 * Y[n][n][G] array; recipient j decodes with evals_dev = &Y[0][j][0], row_stride = n * G -- the
 * protocol's all-to-all (batch_recon.rs:173-183) becomes a layout choice, no data moves.
 * p0_only != 0: out_dev is secrets[G] (ncoeffs_out_dev ignored), else coeffs[G][d+1]. */
ShareErrorCode hbmpc_dev_batch_recover_strided(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S,
                                               const U256* evals_dev, size_t row_stride, size_t G, size_t n,
                                               size_t d, size_t t, int p0_only, U256* out_dev,
                                               uint32_t* ncoeffs_out_dev, uint8_t* status_out_dev,
                                               hbmpc_recover_summary* summary_dev, void* stream);
/* A P(0)-shaped decode that keeps coefficient k of every chunk's polynomial instead of coefficient 0 (k = 0 is P(0)): out_dev[G].
 * For callers that need one coefficient only -- RanSha's verifiers test the exact degree, i.e. the top coefficient
 * (share_gen.rs:516-530): a sixth of the rows and of the output of the full decode at t = 5.  Only for calls WITHOUT OEC rounds
 * (S == d + t + 1, what the handler passes: it fires when that many have arrived, share_gen.rs:497); InvalidInput otherwise. */
ShareErrorCode hbmpc_dev_batch_recover_coeff_strided(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const U256* evals_dev, size_t row_stride,
                                                     size_t G, size_t n, size_t d, size_t t, size_t k, U256* out_dev, uint8_t* status_out_dev,
                                                     hbmpc_recover_summary* summary_dev, void* stream);

/* ==== a6: RobustShare::recover_secret (one polynomial) ======================================
 * replaces robust_interpolate.rs:94-157 (+ robust_interpolate_fnt :206-266, oec_decode :579-628,
 * gao_rs_decode :456-538).  ids[S]/vals[S]/degrees[S] in any order.
 * coeffs_out has room for d+1 = degrees[0]+1 elements; *ncoeffs_out = trimmed length
 * (DensePolynomial normal form); *secret_out = P(0). */
ShareErrorCode hbmpc_recover_secret(hbmpc_ctx* ctx, const size_t* ids, const size_t* degrees, const U256* vals,
                                    size_t S, size_t n, size_t t, U256* coeffs_out, size_t* ncoeffs_out,
                                    U256* secret_out);
/* gao_rs_decode alone (robust_interpolate.rs:456-538): received[n], erasure positions,
 * k = message length; coeffs_out has room for k elements. */
ShareErrorCode hbmpc_gao_rs_decode(hbmpc_ctx* ctx, const U256* received, size_t k, size_t n,
                                   const size_t* erasure_positions, size_t n_erasures, U256* coeffs_out,
                                   size_t* ncoeffs_out);
/* NonRobustShare::recover_secret (common/share/shamir.rs:199-239): plain Lagrange through ALL
 * supplied shares + degree check (the RanDouSha verifier, ran_dou_sha/mod.rs:569-602). */
ShareErrorCode hbmpc_nonrobust_recover_secret(hbmpc_ctx* ctx, const size_t* ids, const size_t* degrees,
                                              const U256* vals, size_t S, size_t n, U256* coeffs_out,
                                              size_t* ncoeffs_out, U256* secret_out);

/* Batched form of the above for G columns that share the ids (the RanDouSha verifier reconstructs every
 * column of a batch, ran_dou_sha/mod.rs:569-602): plain Lagrange through ALL S
 * points.  coeffs_out[G][S] chunk-major (not trimmed), degree_out[G] = DensePolynomial::degree() of each
 * (0 for the zero polynomial).  The caller applies the reference's checks (degree > share degree =>
 * DegreeMismatch, shamir.rs:235; exact-degree and equal-secret tests of ran_dou_sha/mod.rs:586-589). */
ShareErrorCode hbmpc_batch_interpolate(hbmpc_ctx* ctx, const size_t* ids, size_t S, const U256* evals, size_t G,
                                       size_t n, U256* coeffs_out, uint32_t* degree_out);
ShareErrorCode hbmpc_dev_batch_interpolate(hbmpc_ctx* ctx, const size_t* ids, size_t S, const U256* evals_dev,
                                           size_t row_stride, size_t G, size_t n, U256* coeffs_out_dev,
                                           uint32_t* degree_out_dev, void* stream);
/* What the RanDouSha verifier keeps of an interpolation (ran_dou_sha/mod.rs:586-589: the degree of each of its two polynomials
 * and whether their constant terms agree): c0_out[G] = coefficient 0, degree_out[G] = DensePolynomial::degree().  Through all
 * n shares of an 8- or 16-point domain the kernel writes just these (36 bytes per column instead of 32 S); other shapes
 * interpolate into tmp_coeffs_dev[G][S] (required) and extract them.  hbmpc_dev_check_double_share_c0 is the test on two such
 * pairs: bad[0] += columns with degree_t != t, degree_2t != 2 t or c0_t != c0_2t, bad[1] = min(first such column). */
ShareErrorCode hbmpc_dev_batch_interpolate_c0(hbmpc_ctx* ctx, const size_t* ids, size_t S, const U256* evals_dev, size_t row_stride,
                                              size_t G, size_t n, U256* tmp_coeffs_dev, U256* c0_out_dev, uint32_t* degree_out_dev,
                                              void* stream);
ShareErrorCode hbmpc_dev_check_double_share_c0(hbmpc_ctx* ctx, const void* c0_t_dev, const uint32_t* degree_t_dev, const void* c0_2t_dev,
                                               const uint32_t* degree_2t_dev, size_t G, size_t t, uint32_t* bad_dev, void* stream);
/* The same when the G entries are several verifiers' results over the same `columns` columns, one verifier after the other (G a multiple
 * of columns; 0 = G): bad[1] is the first failing COLUMN (entry index mod columns), bad[0] counts entries as before. */
ShareErrorCode hbmpc_dev_check_double_share_c0_columns(hbmpc_ctx* ctx, const void* c0_t_dev, const uint32_t* degree_t_dev, const void* c0_2t_dev,
                                                       const uint32_t* degree_2t_dev, size_t G, size_t columns, size_t t, uint32_t* bad_dev,
                                                       void* stream);

/* ---- layout and verdict steps of the preprocessing producers (RanSha, DouSha, RanDouSha; either field: the element
 * size follows the context) -----------------------------------------------------------------------------------------
 * The producers of config 4's inputs are Vandermonde products and verifier interpolations the calls above already cover
 * (make_vandermonde(n, n - 1) + apply_vandermonde: share_gen/share_gen.rs:415-419, ran_dou_sha/mod.rs:392-403;
 * recover_secret + degree test: share_gen.rs:516-530; NonRobustShare::recover_secret x 2 + the exact-degree / equal-secret
 * tests: ran_dou_sha/mod.rs:569-602).  What they add is data movement between "who dealt" and "who received" -- in the
 * reference the network, here a transpose -- and the verifiers' verdicts, kept on the device:
 *   transpose       dst[b][c][r] = src[b][r][c] (strides in elements)
 *   check_degree    bad[0] += number of polynomials (coeffs[G][m], not trimmed) whose DensePolynomial::degree() is not
 *                   want_degree or whose status byte (nullable; from hbmpc_dev_batch_recover) is a failure; bad[1] =
 *                   min(bad[1], first such polynomial).  The caller zeroes bad[0] and sets bad[1] = 0xffffffff.
 *   check_double_share   the same for pairs: degree t, degree 2t and equal constant terms */
ShareErrorCode hbmpc_dev_transpose(hbmpc_ctx* ctx, const void* src_dev, size_t rows, size_t cols, size_t src_row_stride, void* dst_dev,
                                   size_t dst_row_stride, size_t batch, size_t src_batch_stride, size_t dst_batch_stride, void* stream);
ShareErrorCode hbmpc_dev_check_degree(hbmpc_ctx* ctx, const void* coeffs_dev, const uint8_t* status_dev, size_t G, size_t m,
                                      size_t want_degree, uint32_t* bad_dev, void* stream);
/* The exact-degree verdict from the top coefficient alone: top_dev[G] = coefficient want_degree of each polynomial (what
 * hbmpc_dev_batch_recover_coeff_strided writes with k = want_degree), status_dev as above.  A polynomial of at most want_degree + 1
 * coefficients has that degree iff the coefficient is not zero (want_degree = 0: only the status counts).  Same bad[] as check_degree. */
ShareErrorCode hbmpc_dev_check_top_coeff(hbmpc_ctx* ctx, const void* top_dev, const uint8_t* status_dev, size_t G, size_t want_degree,
                                         uint32_t* bad_dev, void* stream);
/* RanDouSha's verifier (ran_dou_sha/mod.rs:557-602) interpolates a polynomial through ALL n shares and asks three things of it: is its
 * degree exactly d, and its constant term.  hbmpc_[gl_]dev_interpolate_degree_check_strided answers them without the full interpolation:
 * the first d + 1 points (in id order) interpolate, every other point is a verify row (the points lie on a polynomial of degree <= d iff
 * all of them agree), and only coefficients 0 and d are computed: sel_out_dev[G][2] = (constant term, coefficient d), status_out_dev[G] = 0
 * or DecodingError (the points are on no such polynomial; both coefficients are then zero).  The interpolant of degree < S through the
 * points has degree exactly d iff status is 0 and coefficient d is not zero.  Over Fr on the wave-per-chunk and matrix-core kernels
 * (12 + 7 table rows instead of 2 x 16 at n = 16, t = 5); other shapes run the full interpolation into ws_dev [G][S] and pick the same
 * values from it.  hbmpc_dev_check_double_share_sel: the verdict from two such results (degrees t and 2t, equal constant terms),
 * accumulated in bad_dev as check_double_share does.
 * groups > 1 (interpolate_degree_check) / columns > 0 (check_double_share_sel: G = groups x columns entries): several verifiers in one call,
 * verifier q's sender rows q * group_stride elements after verifier 0's and its results at sel_out_dev + q G 2, status_out_dev + q G; bad[1] is
 * the lowest failing column.  ONE launch while groups * G chunks fit the wave-per-chunk decode, a loop otherwise -- except for groups
 * that follow each other inside the sender rows (group_stride == G, the layout hbmpc_dev_vandermonde_apply_rows_split writes): those are
 * one plain decode of groups * G chunks whatever the size. */
ShareErrorCode hbmpc_dev_interpolate_degree_check_strided(hbmpc_ctx* ctx, const size_t* ids, size_t S, const U256* evals_dev, size_t row_stride,
                                                          size_t G, size_t n, size_t d, size_t groups, size_t group_stride, U256* ws_dev,
                                                          U256* sel_out_dev, uint8_t* status_out_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_interpolate_degree_check_strided(hbmpc_ctx* ctx, const size_t* ids, size_t S, const uint64_t* evals_dev,
                                                             size_t row_stride, size_t G, size_t n, size_t d, size_t groups, size_t group_stride,
                                                             uint64_t* ws_dev, uint64_t* sel_out_dev, uint8_t* status_out_dev, void* stream);
ShareErrorCode hbmpc_dev_check_double_share_sel(hbmpc_ctx* ctx, const void* sel_t_dev, const uint8_t* status_t_dev, const void* sel_2t_dev,
                                                const uint8_t* status_2t_dev, size_t G, size_t columns, size_t t, uint32_t* bad_dev, void* stream);
/* RanSha's verifier in one call (share_gen.rs:516-530): recover_secret of G columns from S senders' shares (rows row_stride elements
 * apart, degree t, threshold t) and the exact-degree test, the verdict accumulated in bad_dev as check_degree does.  ws_dev: G (t + 1)
 * elements of workspace.  With exactly 2t + 1 senders (no OEC round) and hbmpc_set_producer_fusion on, the decode keeps the top
 * coefficient only (hbmpc_dev_batch_recover_coeff_strided + hbmpc_dev_check_top_coeff); otherwise the full decode and
 * hbmpc_dev_check_degree.  Same verdict either way.
 * groups > 1: that many verifiers in one call -- verifier q's sender rows start q * group_stride elements after verifier 0's, bad[1] is the
 * lowest failing column of any of them; ONE launch while groups * G chunks fit the wave-per-chunk decode (no OEC round), a loop otherwise;
 * groups that follow each other inside the sender rows (group_stride == G) are one plain decode of groups * G chunks at any size.
 * ws_dev: max(groups G, G (t + 1)) elements, status_out_dev: groups G bytes. */
ShareErrorCode hbmpc_dev_recover_check_degree_strided(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const U256* evals_dev, size_t row_stride,
                                                      size_t G, size_t n, size_t t, size_t groups, size_t group_stride, U256* ws_dev,
                                                      uint8_t* status_out_dev, hbmpc_recover_summary* summary_dev, uint32_t* bad_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_recover_check_degree_strided(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const uint64_t* evals_dev,
                                                         size_t row_stride, size_t G, size_t n, size_t t, size_t groups, size_t group_stride,
                                                         uint64_t* ws_dev, uint8_t* status_out_dev, hbmpc_recover_summary* summary_dev,
                                                         uint32_t* bad_dev, void* stream);
ShareErrorCode hbmpc_dev_check_double_share(hbmpc_ctx* ctx, const void* coeffs_t_dev, const void* coeffs_2t_dev, size_t G, size_t m,
                                            size_t t, uint32_t* bad_dev, void* stream);

/* ==== a9/a11/a12/a13: element-wise share arithmetic of one party ============================
 * All arrays hold N elements of ONE party (same id, same degree: the id/degree checks of
 * common/mod.rs:167-300 are metadata and stay with the host mirror).  Host-pointer calls;
 * hbmpc_dev_* twins take device pointers + stream. */
/* triple_gen/triple_generation.rs:333-340: out = a*b - r2t */
ShareErrorCode hbmpc_triple_local(hbmpc_ctx* ctx, const U256* a, const U256* b, const U256* r2t, size_t N, U256* out);
/* triple_generation.rs:196-208: c = rt + opened */
ShareErrorCode hbmpc_triple_finalize(hbmpc_ctx* ctx, const U256* rt, const U256* opened, size_t N, U256* c_out);
/* mul/multiplication.rs:417-426: d_sh = a - x, e_sh = b - y */
ShareErrorCode hbmpc_beaver_open_shares(hbmpc_ctx* ctx, const U256* a, const U256* b, const U256* x, const U256* y,
                                        size_t N, U256* d_sh_out, U256* e_sh_out);
/* multiplication.rs:57-100 finalize_mul: z = c - d*e - d*y - e*x  (d, e opened values) */
ShareErrorCode hbmpc_beaver_finalize(hbmpc_ctx* ctx, const U256* c, const U256* x, const U256* y, const U256* d,
                                     const U256* e, size_t N, U256* z_out);
/* fpmul/truncpr.rs:277-283: r_dash[i] = sum_{j<m} 2^j * r_bits[j][i]; r_bits[m][N] bit-major */
ShareErrorCode hbmpc_truncpr_rdash(hbmpc_ctx* ctx, const U256* r_bits, size_t m, size_t N, U256* r_dash_out);
/* truncpr.rs:275,294-297: open = (a + 2^(k-1)) + (2^m * r_int + r_dash) */
ShareErrorCode hbmpc_truncpr_open_share(hbmpc_ctx* ctx, const U256* a, const U256* r_dash, const U256* r_int,
                                        size_t k, size_t m, size_t N, U256* open_out);
/* truncpr.rs:215-220 + fpmul/mod.rs:377-406: d = (a - ((c mod 2^m) - r_dash)) * (2^m)^-1 */
ShareErrorCode hbmpc_truncpr_finalize(hbmpc_ctx* ctx, const U256* a, const U256* r_dash, const U256* c_open,
                                      size_t m, size_t N, U256* d_out);

/* TripleGenNode::init_batch for `parties` simulated parties in one launch: [ab - r]_2t = a_i b_i - r2t_i per element
 * (triple_gen/triple_generation.rs:333-340) followed by the BatchRecon encode of the chunks of d + 1 = 2t + 1 values
 * (batch_recon/batch_recon.rs:157-165): a, b, r2t are [parties][G (d+1)], y_out is [parties][n][G] exactly as
 * hbmpc_dev_vandermonde_apply_parties writes it.  Where a fused kernel exists the local products never touch HBM: over Fr,
 * batches of at least 2^17 chunks over all parties on domains of 16 or 32 points whose pair table fits one CU's LDS
 * (n / 2 rounded up to a power of two, times (d + 1) KB + 256 bytes, <= 160 KB; 2 <= d + 1 <= 15) compute the products
 * inside the matrix-core encode (csrc/kernels_mfma_bfly.hpp, TRIPLE); either field, domains up to 16 points,
 * d + 1 in {3, 5, 7, 9, 11}: the fused FFT kernel; other shapes -- and, over Fr when tmp_dev is given,
 * batches of at most 2 048 chunks over all parties, where two short launches beat the one long one (the small field's
 * fused kernel is light at every size and always runs when it covers the shape) -- run the two launches through
 * tmp_dev (parties G (d+1) elements; may be NULL only when the fused kernel applies -- InvalidInput otherwise).  Results are
 * those of hbmpc_dev_triple_local followed by hbmpc_dev_vandermonde_apply_parties, bit for bit. */
ShareErrorCode hbmpc_dev_triple_encode_parties(hbmpc_ctx* ctx, const U256* a_dev, const U256* b_dev, const U256* r2t_dev,
                                               size_t G, size_t n, size_t d, size_t parties, U256* tmp_dev, U256* y_out_dev,
                                               void* stream);
ShareErrorCode hbmpc_dev_triple_local(hbmpc_ctx* ctx, const U256* a, const U256* b, const U256* r2t, size_t N,
                                      U256* out, void* stream);
ShareErrorCode hbmpc_dev_triple_finalize(hbmpc_ctx* ctx, const U256* rt, const U256* opened, size_t N, U256* c_out,
                                         void* stream);
ShareErrorCode hbmpc_dev_beaver_open_shares(hbmpc_ctx* ctx, const U256* a, const U256* b, const U256* x,
                                            const U256* y, size_t N, U256* d_sh_out, U256* e_sh_out, void* stream);
ShareErrorCode hbmpc_dev_beaver_finalize(hbmpc_ctx* ctx, const U256* c, const U256* x, const U256* y, const U256* d,
                                         const U256* e, size_t N, U256* z_out, void* stream);
/* FPMulNode's local math between its two rounds of opens in one launch, for `parties` parties (c, x, y, r_int, z_out,
 * r_dash_out, open_out: [party][N]; r_bits: [party][m][N]; d, e: the opened values, [N]):
 *   z = c - d e - d y - e x                      (finalize_mul, multiplication.rs:57-100)
 *   r' = sum_{j<m} 2^j r_bits[j]                 (truncpr.rs:277-283)
 *   open = (z + 2^(k-1)) + (2^m r_int + r')      (truncpr.rs:275,294-297)
 * i.e. hbmpc_dev_beaver_finalize_parties, hbmpc_dev_truncpr_rdash_parties and hbmpc_dev_truncpr_open_share back to back
 * with the same bytes in all three outputs; at the batch sizes the protocols use every launch is several microseconds
 * of a ~50 us multiplication. */
ShareErrorCode hbmpc_dev_fpmul_middle(hbmpc_ctx* ctx, const U256* c, const U256* x, const U256* y, const U256* d, const U256* e,
                                      const U256* r_bits, const U256* r_int, size_t k, size_t m, size_t N, size_t parties,
                                      U256* z_out, U256* r_dash_out, U256* open_out, void* stream);
/* beaver_open_shares for `parties` parties at once (a, b, x, y: [party][N]) with a party's two results side by side:
 * de_sh_out[party][0][N] = a - x, [party][1][N] = b - y.  One robust interpolation over 2 N values per sender
 * (hbmpc_dev_batch_recover_p0 with G = 2 N: the sender rows are the parties' rows of de_sh_out) then opens d and e
 * together -- one call instead of two; its output is d[N] followed by e[N]. */
/* TripleGenNode for every party of this device in one call (triple_gen/triple_generation.rs:304-364): [ab - r]_2t = a_i b_i - r2t_i
 * (:333-340) in chunks of 2t + 1, BatchRecon's encode for every recipient (batch_recon.rs:157-165), the recipients' P(0) decodes
 * (:384-391), the coefficient decode of the n revealed values (:457-467) and [c]_t = rt_i + opened (:196-208).  a, b, r2t, rt, c_out:
 * [party][N], N a multiple of 2t + 1 (G = N / (2t + 1) chunks); y_ws [party][recipient][G] and z_ws [recipient][G] are the messages
 * of the two arms (written, as the separate calls write them); opened_out [G][2t + 1]; status_out [n G] as the two decodes leave it
 * ([0, G) the second's, [G, n G) recipients 1 .. n - 1 of the first); the two decodes' summaries (either may be null).  A chunk that
 * fails a decode opens to zero and is counted, and the steps after it run on that zero.
 * With n = 3t + 1 <= 16 (every decode has exactly d + t + 1 senders) and at most hbmpc_set_fused_triplegen chunks (default 1024) the
 * call is ONE launch, a workgroup per chunk (csrc/kernels_triplegen_wg.hpp); otherwise the four launches
 * hbmpc_dev_triple_encode_parties (c_out is its workspace), hbmpc_dev_batch_recover_strided (P(0), n G chunks),
 * hbmpc_dev_batch_recover, hbmpc_dev_triple_finalize_parties.  Same bytes in every buffer either way. */
ShareErrorCode hbmpc_dev_triplegen_parties(hbmpc_ctx* ctx, const U256* a, const U256* b, const U256* r2t, const U256* rt, size_t N, size_t n,
                                           size_t t, U256* y_ws, U256* z_ws, U256* opened_out, U256* c_out, uint8_t* status_out,
                                           hbmpc_recover_summary* summary_first_dev, hbmpc_recover_summary* summary_dev, void* stream);
/* FPMulNode for every party of this device in one call (fpmul/fpmul.rs:61-110): Multiply's opened a - x and b - y
 * (mul/multiplication.rs:417-426, :102-139), finalize_mul (:57-100), TruncPr's r' and opened share (truncpr.rs:277-297, :215) and
 * its last step (:216-220).  Per-party arrays are [party][N] (r_bits [party][m][N]); sender_ids[S] are PARTY ids, degree and
 * threshold t.  Outputs: de_out[2 N] the opened a - x then b - y; z, r', the share TruncPr opens; c_open_out[N] that share opened;
 * d_out the parties' shares of the truncated product; status_out[2 N] as the two decodes leave it ([0, N) the second open's,
 * [N, 2 N) the b - y half of the first); the two opens' summaries (either may be null).  A chunk that fails its verification
 * opens to zero and is counted, and the steps after it run on that zero -- as a caller of the separate functions who does not
 * look at the summary in between gets them.
 * With exactly 2t + 1 senders (what the reference opens from: multiplication.rs:388, truncpr.rs:202) and at most
 * hbmpc_set_fused_fpmul elements (default 2048) the call is ONE launch, a wave per element (csrc/kernels_fpmul_wave.hpp: at these
 * sizes a multiplication is bound by launches and by a lone wave's chain of multiplications per step, not by bytes); otherwise it
 * is the five launches hbmpc_dev_beaver_open_shares_paired, hbmpc_dev_batch_recover_p0 (2 N values per sender),
 * hbmpc_dev_fpmul_middle, hbmpc_dev_batch_recover_p0, hbmpc_dev_truncpr_finalize_parties -- the first two as one from
 * hbmpc_set_fpmul_pair_decode elements on (default 8192: the decode forms a - x and b - y of its 2t + 1 senders as it loads them).
 * Every output buffer holds the same bytes in every form; de_sh_ws [party][2][N] is the workspace of the five-launch form (contents
 * unspecified afterwards). */
ShareErrorCode hbmpc_dev_fpmul_parties(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const U256* a, const U256* b, const U256* c,
                                       const U256* x, const U256* y, const U256* r_bits, const U256* r_int, size_t k, size_t m, size_t N,
                                       size_t n, size_t t, U256* de_sh_ws, U256* de_out, U256* z_out, U256* r_dash_out, U256* open_sh_out,
                                       U256* c_open_out, U256* d_out, uint8_t* status_out, hbmpc_recover_summary* summary_first_dev,
                                       hbmpc_recover_summary* summary_dev, void* stream);
ShareErrorCode hbmpc_dev_beaver_open_shares_paired(hbmpc_ctx* ctx, const U256* a, const U256* b, const U256* x, const U256* y,
                                                   size_t N, size_t parties, U256* de_sh_out, void* stream);
ShareErrorCode hbmpc_dev_truncpr_rdash(hbmpc_ctx* ctx, const U256* r_bits, size_t m, size_t N, U256* r_dash_out,
                                       void* stream);
ShareErrorCode hbmpc_dev_truncpr_open_share(hbmpc_ctx* ctx, const U256* a, const U256* r_dash, const U256* r_int,
                                            size_t k, size_t m, size_t N, U256* open_out, void* stream);
ShareErrorCode hbmpc_dev_truncpr_finalize(hbmpc_ctx* ctx, const U256* a, const U256* r_dash, const U256* c_open,
                                          size_t m, size_t N, U256* d_out, void* stream);

/* Party-batched forms for hosts that hold the [party][N] arrays of several parties contiguously (every reference test
 * and bench simulates all n parties in one process): ONE launch for all parties.  Per-party arrays are
 * [parties][N]; the PUBLIC operands -- the opened values every party shares (opened, d, e, c_open) -- are [N].
 * triple_local, beaver_open_shares and truncpr_open_share have no public operand: pass N * parties to the plain
 * entry points.  r_bits is [parties][m][N].  parties in 1..65535. */
ShareErrorCode hbmpc_dev_triple_finalize_parties(hbmpc_ctx* ctx, const U256* rt, const U256* opened, size_t N, size_t parties,
                                                 U256* c_out, void* stream);
ShareErrorCode hbmpc_dev_beaver_finalize_parties(hbmpc_ctx* ctx, const U256* c, const U256* x, const U256* y, const U256* d,
                                                 const U256* e, size_t N, size_t parties, U256* z_out, void* stream);
ShareErrorCode hbmpc_dev_truncpr_rdash_parties(hbmpc_ctx* ctx, const U256* r_bits, size_t m, size_t N, size_t parties,
                                               U256* r_dash_out, void* stream);
ShareErrorCode hbmpc_dev_truncpr_finalize_parties(hbmpc_ctx* ctx, const U256* a, const U256* r_dash, const U256* c_open,
                                                  size_t m, size_t N, size_t parties, U256* d_out, void* stream);

/* ==== a9: share (+, -, *) share of one party (common/mod.rs:167-300: Add, Sub, share_mul) ====
 * op: 0 = a + b, 1 = a - b, 2 = a * b (element-wise, N elements). */
ShareErrorCode hbmpc_fr_op(hbmpc_ctx* ctx, int op, const U256* a, const U256* b, size_t N, U256* out);
ShareErrorCode hbmpc_dev_fr_op(hbmpc_ctx* ctx, int op, const U256* a, const U256* b, size_t N, U256* out, void* stream);
/* share (+, -, *) ONE field element and element - share: Add<F>, Sub<F>, Mul<F>, from_scalar_sub of common/mod.rs:205-280.
 * op: 0 = a + s, 1 = a - s, 2 = a * s, 3 = s - a.  `scalar` points to one canonical element in HOST memory in both
 * variants (it travels in the kernel arguments); a non-canonical scalar is InvalidInput. */
ShareErrorCode hbmpc_fr_op_scalar(hbmpc_ctx* ctx, int op, const U256* a, const U256* scalar, size_t N, U256* out);
ShareErrorCode hbmpc_dev_fr_op_scalar(hbmpc_ctx* ctx, int op, const U256* a_dev, const U256* scalar_host, size_t N,
                                      U256* out_dev, void* stream);

/* ==== wire codec: ark-serialize "compressed" payloads of the path ====
 * Vec<F>  = u64-LE length, then 32-byte LE canonical elements (EvalBatch / RevealBatch payloads,
 *           batch_recon.rs:173-176, 396-398; read back by common/utils.rs:3-21 deser_bounded_vec).
 * Vec<RobustShare<F>> = u64-LE length, then 48-byte records value | id u64 | degree u64
 *           (common/mod.rs:92-99; share_gen.rs:262-266).
 * pack_fvec: rows_dev[n_rows] (row r at rows_dev + r*row_stride elements, G elements each) ->
 *   payload r at payloads_dev + r*payload_stride_bytes (8-byte aligned, >= 8 + 32 G bytes).
 *   After hbmpc_dev_vandermonde_apply, row j IS the EvalBatch payload body for recipient j.
 * unpack_fvec: the inverse; status_dev[r] = 0, or InvalidInput (4) when the length prefix is not G or an
 *   element is not canonical (ark returns SerializationError::InvalidData for both).
 * pack_shares / unpack_shares: N values of one party (id, degree) <-> one Vec<RobustShare> payload;
 *   status_dev[0] = 0 / 4 (length, non-canonical) / 3 (a record's id differs) / 2 (degree differs).
 * validate_canonical: status_dev[0] = 4 if some element >= r (what Fr::from_bigint(..).unwrap() checks,
 *   ffi/c_bindings/mod.rs:37-41). */
ShareErrorCode hbmpc_dev_pack_fvec(hbmpc_ctx* ctx, const U256* rows_dev, size_t row_stride, size_t G, size_t n_rows,
                                   void* payloads_dev, size_t payload_stride_bytes, void* stream);
ShareErrorCode hbmpc_dev_unpack_fvec(hbmpc_ctx* ctx, const void* payloads_dev, size_t payload_stride_bytes,
                                     size_t payload_bytes, size_t G, size_t n_rows, U256* rows_dev, size_t row_stride,
                                     uint32_t* status_dev, void* stream);
/* apply_vandermonde with the input given as d + 1 ROWS: row i = coefficient i of all G chunks, x_row_stride elements apart
 * (x_row_stride >= G).  This is the shape of the preprocessing producers' mixing step (share_gen.rs:150-175,
 * ran_dou_sha/mod.rs:270-312: every recipient multiplies the n shares it was dealt by make_vandermonde(n, n - 1)): with all
 * parties' buffers on one device the share of dealer p for (recipient, element) is row p of the dealt array, so the n x n
 * map reads the dealers' outputs where they lie -- y[i][g] = sum_p alpha_i^p x[p][g] -- and the [recipient][element][dealer]
 * copy is never made.  Large Fr batches with 2 <= d + 1 <= 16 on domains of 8 .. 256 points run on the matrix cores straight
 * from the rows (csrc/kernels_mfma_bfly.hpp); every other shape transposes into tmp_dev (G * (d + 1) elements; may be null
 * when the direct kernel covers the shape -- a call that needs it and has none fails with InvalidInput) and takes the
 * chunk-major encode.  Output as hbmpc_dev_vandermonde_apply: y[n][G]. */
ShareErrorCode hbmpc_dev_vandermonde_apply_rows(hbmpc_ctx* ctx, const U256* x_rows_dev, size_t x_row_stride, size_t G, size_t n,
                                                size_t d, U256* tmp_dev, U256* y_out_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_vandermonde_apply_rows(hbmpc_ctx* ctx, const uint64_t* x_rows_dev, size_t x_row_stride, size_t G, size_t n,
                                                   size_t d, uint64_t* tmp_dev, uint64_t* y_out_dev, void* stream);
/* The same mixing step with the parties' OUTPUT rows written as the lists the reference returns (share_gen.rs:199-203: rows
 * 2t .. n-1, ran_dou_sha/mod.rs:314-331: rows 0 .. t -- per party in the order [batch element k][row]), so that no pass copies
 * them out of y afterwards.  The G = parties * K chunks are (party j, batch element k) = g / K, g % K.  Output rows
 * [list_row0, list_row0 + list_rows) of chunk (j, k) go to
 *     slices[s].dst_dev + (j * party_stride + (k - k0) * list_rows + (row - list_row0)) elements
 * for the slice s whose [k0, k0 + count) holds k (at most two slices, ascending and disjoint: a consumer that wants the first
 * N of a party's list in one array and the next N in another -- TripleGen's a and b -- names both; batch elements no slice
 * holds are not written).  The other rows go to y_out[row][G] as above; the list rows of y_out are unspecified.  Large Fr batches
 * on domains of 8 and 16 points (and the shapes of the lane kernel named below) write the lists from the kernel that computes them; every other shape (and
 * hbmpc_set_producer_fusion(ctx, 0)) computes all of y_out and copies the slices out with hbmpc_dev_transpose -- same bytes. */
typedef struct {
    void* dst_dev;
    size_t party_stride; /* elements between the lists of consecutive parties */
    size_t k0, count;    /* batch elements [k0, k0 + count) of every party */
} hbmpc_list_slice;
ShareErrorCode hbmpc_dev_vandermonde_apply_rows_lists(hbmpc_ctx* ctx, const U256* x_rows_dev, size_t x_row_stride, size_t G, size_t n,
                                                      size_t d, U256* tmp_dev, U256* y_out_dev, size_t list_row0, size_t list_rows, size_t K,
                                                      const hbmpc_list_slice* slices, size_t n_slices, void* stream);
ShareErrorCode hbmpc_gl_dev_vandermonde_apply_rows_lists(hbmpc_ctx* ctx, const uint64_t* x_rows_dev, size_t x_row_stride, size_t G, size_t n,
                                                         size_t d, uint64_t* tmp_dev, uint64_t* y_out_dev, size_t list_row0, size_t list_rows,
                                                         size_t K, const hbmpc_list_slice* slices, size_t n_slices, void* stream);
/* The same with the OTHER rows -- everything outside [list_row0, list_row0 + list_rows): what the parties send the verifiers --
 * written party-major to others_out_dev instead of y_out[row][G]: row number r' among them (r' = row below the lists, row - list_rows
 * above) of chunk (party j, batch element k) goes to
 *     others_out_dev + (j * (n - list_rows) + r') * K + k  elements.
 * A sender's shares for ALL verifiers are then one contiguous row of (n - list_rows) K elements, and the verifiers' decodes are ONE
 * call over (verifier, k) chunks with sender rows (n - list_rows) K apart instead of one call per verifier -- at the reference's own
 * batch sizes (K of 7 000 .. 15 000 columns, n = 16) thirty launches of 10 - 13 us each become three (profiles/r04_protocol_batch_sizes.txt).
 * others_out_dev holds parties * (n - list_rows) * K elements (below 4 GiB); y_out_dev (n * G elements) is workspace here: its contents
 * are unspecified.  Shapes a list kernel covers (hbmpc_dev_apply_rows_lists_in_kernel says which, for d = n - 1: Fr with 5 .. 16 parties at
 * large batches on the matrix cores, Fr with 3 .. 8 parties and Goldilocks with 3 .. 16 at any batch size on the single-pass lane kernel,
 * csrc/kernels_eval.hpp: k_eval_fft1_mix) write both kinds of rows from the kernel that computes them, in one launch; every other shape
 * computes y_out and copies -- same bytes either way. */
ShareErrorCode hbmpc_dev_vandermonde_apply_rows_split(hbmpc_ctx* ctx, const U256* x_rows_dev, size_t x_row_stride, size_t G, size_t n, size_t d,
                                                      U256* tmp_dev, U256* y_out_dev, size_t list_row0, size_t list_rows, size_t K,
                                                      const hbmpc_list_slice* slices, size_t n_slices, U256* others_out_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_vandermonde_apply_rows_split(hbmpc_ctx* ctx, const uint64_t* x_rows_dev, size_t x_row_stride, size_t G, size_t n,
                                                         size_t d, uint64_t* tmp_dev, uint64_t* y_out_dev, size_t list_row0, size_t list_rows,
                                                         size_t K, const hbmpc_list_slice* slices, size_t n_slices, uint64_t* others_out_dev,
                                                         void* stream);
ShareErrorCode hbmpc_dev_apply_rows_lists_in_kernel(hbmpc_ctx* ctx, size_t G, size_t n, size_t d, int* yes_out);
/* In-place wire path -- no pack / unpack pass.  A payload that starts 8 bytes before a 32-byte boundary has its
 * elements 32-byte aligned, so the encode kernel writes the payload bodies itself and the decode reads them where
 * they arrived:
 *   send:    hbmpc_dev_encode_fvec = vandermonde_apply with payload r's body as output row r + the n length prefixes
 *            (payloads_dev + 8 divisible by 32, payload_stride_bytes a multiple of 32 and >= 8 + 32 G) -- the same
 *            bytes as vandermonde_apply followed by pack_fvec;
 *   receive: hbmpc_dev_validate_fvec = unpack_fvec's checks (length prefix, canonical elements; status per payload)
 *            without the copy, then hbmpc_dev_batch_recover_slots with evals_dev = payloads_dev + 8 bytes,
 *            row_stride = payload_stride_bytes / 32 and row_slots[i] = the slot (payload index, < 256) of the i-th
 *            listed sender decodes out of the payloads of the senders that passed (a sender whose payload failed
 *            is simply not listed; hbmpc_dev_batch_recover_strided is the case row_slots[i] = i).
 * hbmpc_[gl_]dev_vandermonde_apply_strided is the underlying call: output row j at y_out_dev + j * y_row_stride
 * elements (y_row_stride >= G). */
ShareErrorCode hbmpc_dev_vandermonde_apply_strided(hbmpc_ctx* ctx, const U256* x_dev, size_t G, size_t n, size_t d,
                                                   U256* y_out_dev, size_t y_row_stride, void* stream);
ShareErrorCode hbmpc_dev_encode_fvec(hbmpc_ctx* ctx, const U256* x_dev, size_t G, size_t n, size_t d, void* payloads_dev,
                                     size_t payload_stride_bytes, void* stream);
ShareErrorCode hbmpc_dev_validate_fvec(hbmpc_ctx* ctx, const void* payloads_dev, size_t payload_stride_bytes,
                                       size_t payload_bytes, size_t G, size_t n_rows, uint32_t* status_dev, void* stream);
ShareErrorCode hbmpc_dev_batch_recover_slots(hbmpc_ctx* ctx, const size_t* sender_ids, const size_t* row_slots, size_t S,
                                             const U256* evals_dev, size_t row_stride, size_t G, size_t n, size_t d,
                                             size_t t, int p0_only, U256* out_dev, uint32_t* ncoeffs_out_dev,
                                             uint8_t* status_out_dev, hbmpc_recover_summary* summary_dev, void* stream);
ShareErrorCode hbmpc_dev_pack_shares(hbmpc_ctx* ctx, const U256* values_dev, size_t N, size_t id, size_t degree,
                                     void* payload_dev, void* stream);
ShareErrorCode hbmpc_dev_unpack_shares(hbmpc_ctx* ctx, const void* payload_dev, size_t payload_bytes, size_t N, size_t id,
                                       size_t degree, U256* values_dev, uint32_t* status_dev, void* stream);
ShareErrorCode hbmpc_dev_validate_canonical(hbmpc_ctx* ctx, const U256* a_dev, size_t N, uint32_t* status_dev,
                                            void* stream);

/* ==== Goldilocks variants ==========================================
 * The reference instantiates the same generic code for GoldilocksField = Fp64<MontBackend<..>>
 * (common/math/goldilocks.rs:4-13; RanShaNode / RanDouShaNode / TripleGenNode / RandBit over it,
 * honeybadger/mod.rs:316-324).  These entry points are the hbmpc_* calls above with 8-byte canonical
 * little-endian elements (what ark serialises for Fp64) instead of U256; same argument meaning, layouts,
 * validation order and error codes.  They need a context created with FieldKind Goldilocks64; a context
 * serves one field only (the other family returns TypeMismatch).  Non-canonical inputs (>= p) are the
 * caller's error, as for Fr.  TruncPr is big-field-only in the reference and has no hbmpc_gl_ form; of the wire
 * codec the in-place Vec<F> path exists for both fields (below). */
ShareErrorCode hbmpc_gl_compute_shares(hbmpc_ctx* ctx, const uint64_t* coeffs, size_t B, size_t n, size_t d,
                                       uint64_t* shares_out);
ShareErrorCode hbmpc_gl_dev_compute_shares(hbmpc_ctx* ctx, const uint64_t* coeffs_dev, size_t B, size_t n, size_t d,
                                           uint64_t* shares_out_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_fill_coeffs(hbmpc_ctx* ctx, const uint8_t seed[32], const uint64_t* secrets_dev, size_t B,
                                        uint64_t first_index, size_t d, uint64_t* coeffs_out_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_compute_shares_seeded(hbmpc_ctx* ctx, const uint8_t seed[32], const uint64_t* secrets_dev,
                                                  size_t B, uint64_t first_index, size_t n, size_t d,
                                                  uint64_t* coeffs_ws_dev, uint64_t* shares_out_dev, void* stream);
ShareErrorCode hbmpc_gl_compute_shares_seeded(hbmpc_ctx* ctx, const uint8_t seed[32], const uint64_t* secrets, size_t B,
                                              uint64_t first_index, size_t n, size_t d, uint64_t* shares_out);
ShareErrorCode hbmpc_gl_vandermonde_apply(hbmpc_ctx* ctx, const uint64_t* x, size_t G, size_t n, size_t d, uint64_t* y_out);
ShareErrorCode hbmpc_gl_dev_vandermonde_apply(hbmpc_ctx* ctx, const uint64_t* x_dev, size_t G, size_t n, size_t d,
                                              uint64_t* y_out_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_vandermonde_apply_strided(hbmpc_ctx* ctx, const uint64_t* x_dev, size_t G, size_t n, size_t d,
                                                      uint64_t* y_out_dev, size_t y_row_stride, void* stream);
/* Goldilocks wire payloads (ark Fp64: u64-LE length + 8-byte LE canonical elements = the in-memory form): any 8-byte-
 * aligned payload has aligned elements, so the in-place path needs no alignment trick: stride a multiple of 8. */
ShareErrorCode hbmpc_gl_dev_encode_fvec(hbmpc_ctx* ctx, const uint64_t* x_dev, size_t G, size_t n, size_t d,
                                        void* payloads_dev, size_t payload_stride_bytes, void* stream);
ShareErrorCode hbmpc_gl_dev_validate_fvec(hbmpc_ctx* ctx, const void* payloads_dev, size_t payload_stride_bytes,
                                          size_t payload_bytes, size_t G, size_t n_rows, uint32_t* status_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_batch_recover_slots(hbmpc_ctx* ctx, const size_t* sender_ids, const size_t* row_slots, size_t S,
                                                const uint64_t* evals_dev, size_t row_stride, size_t G, size_t n, size_t d,
                                                size_t t, int p0_only, uint64_t* out_dev, uint32_t* ncoeffs_out_dev,
                                                uint8_t* status_out_dev, hbmpc_recover_summary* summary_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_vandermonde_apply_parties(hbmpc_ctx* ctx, const uint64_t* x_dev, size_t G, size_t n, size_t d,
                                                      size_t parties, uint64_t* y_out_dev, void* stream);
ShareErrorCode hbmpc_gl_make_vandermonde(hbmpc_ctx* ctx, size_t n, size_t d, uint64_t* v_out);
ShareErrorCode hbmpc_gl_batch_recover(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const uint64_t* evals, size_t G,
                                      size_t n, size_t d, size_t t, uint64_t* coeffs_out, uint32_t* ncoeffs_out,
                                      uint8_t* status_out);
ShareErrorCode hbmpc_gl_batch_recover_p0(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const uint64_t* evals,
                                         size_t G, size_t n, size_t d, size_t t, uint64_t* secrets_out, uint8_t* status_out);
ShareErrorCode hbmpc_gl_dev_batch_recover(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const uint64_t* evals_dev,
                                          size_t G, size_t n, size_t d, size_t t, uint64_t* coeffs_out_dev,
                                          uint32_t* ncoeffs_out_dev, uint8_t* status_out_dev,
                                          hbmpc_recover_summary* summary_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_batch_recover_p0(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const uint64_t* evals_dev,
                                             size_t G, size_t n, size_t d, size_t t, uint64_t* secrets_out_dev,
                                             uint8_t* status_out_dev, hbmpc_recover_summary* summary_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_batch_recover_strided(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S,
                                                  const uint64_t* evals_dev, size_t row_stride, size_t G, size_t n, size_t d,
                                                  size_t t, int p0_only, uint64_t* out_dev, uint32_t* ncoeffs_out_dev,
                                                  uint8_t* status_out_dev, hbmpc_recover_summary* summary_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_batch_recover_coeff_strided(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const uint64_t* evals_dev,
                                                        size_t row_stride, size_t G, size_t n, size_t d, size_t t, size_t k, uint64_t* out_dev,
                                                        uint8_t* status_out_dev, hbmpc_recover_summary* summary_dev, void* stream);
ShareErrorCode hbmpc_gl_batch_interpolate(hbmpc_ctx* ctx, const size_t* ids, size_t S, const uint64_t* evals, size_t G,
                                          size_t n, uint64_t* coeffs_out, uint32_t* degree_out);
ShareErrorCode hbmpc_gl_dev_batch_interpolate_c0(hbmpc_ctx* ctx, const size_t* ids, size_t S, const uint64_t* evals_dev, size_t row_stride,
                                                 size_t G, size_t n, uint64_t* tmp_coeffs_dev, uint64_t* c0_out_dev,
                                                 uint32_t* degree_out_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_batch_interpolate(hbmpc_ctx* ctx, const size_t* ids, size_t S, const uint64_t* evals_dev,
                                              size_t row_stride, size_t G, size_t n, uint64_t* coeffs_out_dev,
                                              uint32_t* degree_out_dev, void* stream);
ShareErrorCode hbmpc_gl_recover_secret(hbmpc_ctx* ctx, const size_t* ids, const size_t* degrees, const uint64_t* vals,
                                       size_t S, size_t n, size_t t, uint64_t* coeffs_out, size_t* ncoeffs_out,
                                       uint64_t* secret_out);
ShareErrorCode hbmpc_gl_gao_rs_decode(hbmpc_ctx* ctx, const uint64_t* received, size_t k, size_t n,
                                      const size_t* erasure_positions, size_t n_erasures, uint64_t* coeffs_out,
                                      size_t* ncoeffs_out);
ShareErrorCode hbmpc_gl_nonrobust_recover_secret(hbmpc_ctx* ctx, const size_t* ids, const size_t* degrees,
                                                 const uint64_t* vals, size_t S, size_t n, uint64_t* coeffs_out,
                                                 size_t* ncoeffs_out, uint64_t* secret_out);
ShareErrorCode hbmpc_gl_fr_op(hbmpc_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t N, uint64_t* out);
ShareErrorCode hbmpc_gl_fr_op_scalar(hbmpc_ctx* ctx, int op, const uint64_t* a, const uint64_t* scalar, size_t N, uint64_t* out);
ShareErrorCode hbmpc_gl_dev_fr_op_scalar(hbmpc_ctx* ctx, int op, const uint64_t* a_dev, const uint64_t* scalar_host, size_t N,
                                         uint64_t* out_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_fr_op(hbmpc_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t N, uint64_t* out,
                                  void* stream);
ShareErrorCode hbmpc_gl_triple_local(hbmpc_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* r2t, size_t N,
                                     uint64_t* out);
ShareErrorCode hbmpc_gl_dev_triple_encode_parties(hbmpc_ctx* ctx, const uint64_t* a_dev, const uint64_t* b_dev,
                                                  const uint64_t* r2t_dev, size_t G, size_t n, size_t d, size_t parties,
                                                  uint64_t* tmp_dev, uint64_t* y_out_dev, void* stream);
ShareErrorCode hbmpc_gl_dev_triple_local(hbmpc_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* r2t, size_t N,
                                         uint64_t* out, void* stream);
ShareErrorCode hbmpc_gl_triple_finalize(hbmpc_ctx* ctx, const uint64_t* rt, const uint64_t* opened, size_t N, uint64_t* c_out);
ShareErrorCode hbmpc_gl_dev_triple_finalize(hbmpc_ctx* ctx, const uint64_t* rt, const uint64_t* opened, size_t N,
                                            uint64_t* c_out, void* stream);
ShareErrorCode hbmpc_gl_beaver_open_shares(hbmpc_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* x,
                                           const uint64_t* y, size_t N, uint64_t* d_sh_out, uint64_t* e_sh_out);
ShareErrorCode hbmpc_gl_dev_beaver_open_shares(hbmpc_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* x,
                                               const uint64_t* y, size_t N, uint64_t* d_sh_out, uint64_t* e_sh_out,
                                               void* stream);
ShareErrorCode hbmpc_gl_dev_triple_finalize_parties(hbmpc_ctx* ctx, const uint64_t* rt, const uint64_t* opened, size_t N,
                                                    size_t parties, uint64_t* c_out, void* stream);
/* hbmpc_dev_triplegen_parties over the small field (PreprocNodesSmallField, honeybadger/mod.rs:316-324): same arguments, 8-byte
 * elements; one launch under the same conditions up to half as many chunks (0.011 ms for 1 100 triples per party, four launches 0.029). */
ShareErrorCode hbmpc_gl_dev_triplegen_parties(hbmpc_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* r2t, const uint64_t* rt, size_t N,
                                              size_t n, size_t t, uint64_t* y_ws, uint64_t* z_ws, uint64_t* opened_out, uint64_t* c_out,
                                              uint8_t* status_out, hbmpc_recover_summary* summary_first_dev, hbmpc_recover_summary* summary_dev,
                                              void* stream);
ShareErrorCode hbmpc_gl_dev_beaver_finalize_parties(hbmpc_ctx* ctx, const uint64_t* c, const uint64_t* x, const uint64_t* y,
                                                    const uint64_t* d, const uint64_t* e, size_t N, size_t parties,
                                                    uint64_t* z_out, void* stream);
ShareErrorCode hbmpc_gl_beaver_finalize(hbmpc_ctx* ctx, const uint64_t* c, const uint64_t* x, const uint64_t* y,
                                        const uint64_t* d, const uint64_t* e, size_t N, uint64_t* z_out);
ShareErrorCode hbmpc_gl_dev_beaver_finalize(hbmpc_ctx* ctx, const uint64_t* c, const uint64_t* x, const uint64_t* y,
                                            const uint64_t* d, const uint64_t* e, size_t N, uint64_t* z_out, void* stream);
ShareErrorCode hbmpc_gl_dev_beaver_open_shares_paired(hbmpc_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* x,
                                                      const uint64_t* y, size_t N, size_t parties, uint64_t* de_sh_out,
                                                      void* stream);

/* ---- A/B aid: 0 = unsaturated 9x29-bit limbs (default, fast), 1 = saturated 8x32-bit limbs
 * (the straightforward formulation; same results, kept as a cross-check). */
ShareErrorCode hbmpc_set_field_impl(hbmpc_ctx* ctx, int impl);
/* Large bls12-381 Fr decodes (batch_recover* of at least min_chunks chunks, 2 <= d + 1 <= 15, d + t + 1 - (d + 1)
 * verify rows that fit one CU's LDS) run the verify and coefficient rows on the matrix cores: every row is a constant
 * vector (a function of n, d, t and the sender ids) times batch data, i.e. an int8 GEMM over the bytes of the canonical
 * elements followed by one carry pass and one small-quotient reduction (csrc/kernels_mfma.hpp).  Results are
 * bit-identical to the lane-per-chunk kernels; on = 0 switches back to them (A/B aid, parity suites run both); on = 2
 * keeps the matrix cores but not the workgroup-per-tile kernel that batches with fewer 32-chunk tiles than waves take
 * (up to 16 384 chunks on a 256-CU chip: a tile's rows are shared by the waves of a workgroup instead of walked by one wave);
 * on = 3 keeps the matrix cores but gives every evaluation point its own table row, where the default takes the points of
 * a large encode in pairs (k, k + size/2) -- alpha_{k + size/2} = -alpha_k on a domain of roots of unity, so the even and
 * the odd coefficients' digit sums are computed once and added / subtracted: half the matrix-core work and LDS operand
 * traffic per output (csrc/kernels_mfma_bfly.hpp).
 * A Goldilocks context has the same switch: for 2 <= d + 1 <= 16 its encodes on domains beyond 16 points run on the
 * matrix cores from 4 096 chunks, its decodes from 2 048 chunks when they are a single launch (exactly d + t + 1 senders)
 * and beyond the small-batch range (8 192 chunks) otherwise (csrc/kernels_mfma_gl.hpp: the table is a few KB, fits the
 * LDS whole and costs microseconds to build, so there is no sender-set rule).
 * min_chunks = 0 keeps the current thresholds.  Defaults: a decode takes the path from 4 096 chunks -- from 2 048 when it
 * is given exactly d + t + 1 senders, i.e. is a single launch -- whether or not its sender set has been seen before: the
 * set's byte-digit table (239 KB for n = 31) is expanded ON THE DEVICE from the rows x (d + 1) coefficients the host
 * computes (csrc/kernels_tables.hpp; a first call costs 1.1 - 1.4x a repeat at 8 192 chunks); evaluations (one table per
 * (n, d), never rebuilt) take it from 2 049 chunks, i.e. right above the wave-per-chunk kernel's range -- on domains
 * beyond 16 points at every size; on domains of 8 and 16 points at every size as well (the workgroup-per-tile kernel while
 * a workgroup has at most two tiles -- 16 384 chunks on 256 CUs -- the point pairs beyond); 4-point domains keep the
 * single-pass FFT beyond the workgroup-per-tile range.  A nonzero min_chunks caps these thresholds at it.  Nothing can be built during a
 * graph capture: a call whose table is missing then records the lane kernels, so run the sequence once eagerly first. */
ShareErrorCode hbmpc_set_matrix_cores(hbmpc_ctx* ctx, int on, size_t min_chunks);
/* A decode of a sender set the context has not seen before, with OEC rounds available (more than d + t + 1 senders):
 * the tables only chunks that FAIL the verification read -- the Gao rounds' Lagrange bases and the second-chance windows
 * (robust_interpolate.rs:589-593; 0.4 - 1.3 ms of host arithmetic for n = 31 .. 64) -- are built after the first kernel,
 * and only if it flagged a chunk: the call looks at the flagged counter once (one synchronisation of its stream, in a
 * call that would otherwise have spent far longer building).  With honest senders such a first call costs 1.6 - 2.5x a
 * repeat at 64 chunks (4.6 - 36x with the tables built up front).
 *   on = 1 (default): the host-pointer entry points (hbmpc_batch_recover*, which synchronise anyway);
 *   on = 2: the hbmpc_dev_* entry points as well -- their first call with a new sender set then waits for its first kernel,
 *           and a later graph capture of that call needs the tables, i.e. a warm-up run made with on = 0 (or bad data);
 *   on = 0: everything up front, as rounds 1 and 2 did (A/B aid).
 * Inside a graph capture nothing can be looked at and the tables must exist.  Results are identical in every mode. */
ShareErrorCode hbmpc_set_lazy_fallback_tables(hbmpc_ctx* ctx, int on);
/* The producers' fused steps (the mixing step writes the parties' lists itself; the RanDouSha verifier's interpolation keeps
 * only c0 and the degree): on (default) / off = every row into y and separate copy / test passes (A/B aid; same bytes). */
ShareErrorCode hbmpc_set_producer_fusion(hbmpc_ctx* ctx, int on);
/* hbmpc_dev_fpmul_parties runs as one launch up to max_elements batch elements (default 2048; 0: always the five separate
 * launches).  Same bytes either way (A/B aid). */
ShareErrorCode hbmpc_set_fused_fpmul(hbmpc_ctx* ctx, size_t max_elements);
/* hbmpc_dev_triplegen_parties runs as one launch up to max_chunks chunks of 2t + 1 triples (default 1024; 0: always the four
 * separate launches).  Same bytes either way (A/B aid). */
ShareErrorCode hbmpc_set_fused_triplegen(hbmpc_ctx* ctx, size_t max_chunks);
/* hbmpc_dev_fpmul_parties' five-launch form from min_elements batch elements on (default 8192; (size_t)-1: never): the first open
 * forms the senders' a - x and b - y as the decode loads them instead of reading them back from a launch that wrote every
 * party's (four launches; de_sh_ws is then left untouched).  Same bytes in every output (A/B aid). */
ShareErrorCode hbmpc_set_fpmul_pair_decode(hbmpc_ctx* ctx, size_t min_elements);
/* A decode that is given exactly d + t + 1 senders -- what BatchRecon passes: it decodes as soon as that many have
 * arrived (batch_recon.rs:371-389) -- has no OEC round: a chunk that fails the verification can only fail
 * (DecodingError, robust_interpolate.rs:625).  Such a call is ONE kernel launch: the decode kernel writes the failure
 * (status, zero coefficients, length 0) and its last workgroup the summary.  on = 0 sends those chunks through the
 * OEC/Gao kernel as every other call does (A/B aid: same bytes either way; default on). */
ShareErrorCode hbmpc_set_single_launch_decode(hbmpc_ctx* ctx, int on);
/* test aid: workgroups of a matrix-core launch (0 = one per CU, the default); a small number makes a small batch walk
 * the multi-tile loop of every wave */
ShareErrorCode hbmpc_set_matrix_core_workgroups(hbmpc_ctx* ctx, int workgroups);
/* test aid: the matrix-core byte-digit table of a sender set (ids ascending) exactly as the library builds it: expanded on
 * the device from the host-computed coefficients (on_device != 0: what a decode uses) or by the host reference the CPU
 * tests pin against the oracle.  *bytes_out = its size; out (nullable) receives it when cap suffices. */
ShareErrorCode hbmpc_debug_mfma_table(hbmpc_ctx* ctx, const size_t* sorted_ids, size_t S, size_t n, size_t d, size_t t, int on_device,
                                      uint8_t* out, size_t cap, size_t* bytes_out);
/* test aid: 1 = route every shape through the generic (runtime-shaped) kernels */
ShareErrorCode hbmpc_set_force_generic(hbmpc_ctx* ctx, int on);
/* Secret hygiene.  The host-pointer calls stage their arguments through per-context pools (device buffers, pinned host
 * blocks) that are recycled between calls, so copies of polynomial coefficients (secrets) and shares stay there until
 * a later call overwrites them.  hbmpc_scrub_staging zeroes everything the pools hold at that moment (it drains the
 * context's stream first; a block checked out by a host-pointer call running on another thread, and the per-stream
 * kernel scratch, which only ever holds flagged-chunk indices and counters, are not in the pools); hbmpc_destroy does it
 * before freeing.  Buffers the CALLER allocated (hbmpc_dev_alloc, its own host arrays) are
 * the caller's to clear. */
ShareErrorCode hbmpc_scrub_staging(hbmpc_ctx* ctx);
/* The device-table cache of a context (twiddles, Vandermonde rows, one Lagrange/verify table and one OEC/Gao table
 * per sender set) is bounded: at 512 tables everything not referenced by a captured graph is evicted (unlinked now,
 * freed at the next eviction, so a concurrent call that already looked a table up never loses it).
 * stats_out = {tables cached, of which pinned by graphs, evicted-but-not-yet-freed, evictions so far}. */
ShareErrorCode hbmpc_cache_stats(hbmpc_ctx* ctx, size_t stats_out[4]);
/* Host-pointer calls whose inputs + outputs fit 2 MiB stage through pinned host memory mapped into the device
 * (the kernels read and write it over PCIe: no DMA commands; a one-polynomial hbmpc_recover_secret takes ~45 us
 * instead of ~110 us).  zero_copy = 0 routes them through device buffers + copies like large calls (A/B aid;
 * results are identical). */
ShareErrorCode hbmpc_set_small_call_staging(hbmpc_ctx* ctx, int zero_copy);
/* batch_recover calls of at most max_chunks chunks (default 8192), and compute_shares / vandermonde_apply calls of
 * at most max_chunks / 4, run wave-per-chunk kernels (one table row / one evaluation point per lane: the latency of
 * m products instead of (t + out_width) m, of d multiplications instead of a whole FFT; the chip is not full at
 * these sizes anyway); larger calls run one lane per chunk.  0 = always one lane per chunk.  Same results either
 * way (A/B aid). */
ShareErrorCode hbmpc_set_small_batch_chunks(hbmpc_ctx* ctx, size_t max_chunks);
/* Chunks that fail the optimistic verification first try up to four cheap candidates -- the polynomials through the
 * lowest degree+1 senders, the next degree+1, the last degree+1 of the OEC prefix and a window straddling the first
 * two -- and accept one that disagrees with at most
 * min(t, S - (degree+t+1)) of the shares the reference's OEC rounds would look at: that IS the polynomial oec_decode
 * returns (robust_interpolate.rs:579-628; it is unique).  Only what no candidate resolves runs the OEC/Gao kernel.  A
 * single Byzantine sender is always resolved here.  on = 0 sends every flagged chunk to OEC/Gao (A/B aid; same
 * results, statuses and error codes either way). */
ShareErrorCode hbmpc_set_second_chance(hbmpc_ctx* ctx, int on);

/* ---- measurement aid: register-resident Montgomery-multiply loop (integer-ALU ceiling) ------
 * Runs `iters` dependent modmuls in each of `threads` lanes, writes one U256 per lane to out_dev
 * (so nothing is optimised away) and returns nothing else; time it with events on `stream`. */
ShareErrorCode hbmpc_dev_modmul_ubench(hbmpc_ctx* ctx, U256* out_dev, size_t threads, uint32_t iters, void* stream);
/* measurement aid: the memory traffic of an encode with no arithmetic -- reads x[G][m] chunk-major, writes y[n][G] party-major in
 * the access shapes of the matrix-core encode (every loaded word reaches every stored value) -- so that a bench line can state
 * what the memory system delivers for exactly its kernel's loads and stores, measured in the same run. */
ShareErrorCode hbmpc_dev_traffic_ubench(hbmpc_ctx* ctx, const U256* x_dev, size_t G, size_t m, U256* y_dev, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HBMPC_HIP_H */
