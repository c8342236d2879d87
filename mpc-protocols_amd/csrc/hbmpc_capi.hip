// hbmpc_capi.hip -- the C ABI of include/hbmpc_hip.h: context, table cache, kernel dispatch.
// There is no CPU data path here: host code only validates arguments, builds the small constant
// tables (tables.hpp) and launches kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hbmpc_hip.h"
#include "fr_gold.hpp"
#include "kernels_elem.hpp"
#include "kernels_fpmul_wave.hpp"
#include "kernels_triplegen_wg.hpp"
#include "launchers.hpp"
#include "tables.hpp"
#include "tables_mfma.hpp"
#include "tables_mfma_gl.hpp"
#include "kernels_mfma.hpp"
#include "kernels_mfma_gl.hpp"

using namespace hbmpc;

constexpr size_t STAGE_PIN_BLOCK = (size_t)2048 << 10;  // pinned host block of the small-call staging path (Stage, hbmpc_scrub_staging)
struct hbmpc_ctx {
    int device = 0;
    int impl = IMPL_U29;
    bool force_generic = false;                    // tests: route every shape through the generic kernels
    hipStream_t stream = nullptr;
    std::mutex mu;                                 // guards the table cache, the scratch map and the staging pool
    std::mutex enqueue_mu;                         // keeps multi-launch sequences that share scratch contiguous on a stream
    struct Tab {
        uint32_t* p = nullptr;
        bool pinned = false;  // referenced by a captured HIP graph: never evicted
        std::array<size_t, 5> aux = {0, 0, 0, 0, 0};  // offsets inside the buffer (OEC/Gao tables): one unit with the table
    };
    struct Scratch {
        void* p = nullptr;
        size_t cap = 0;
        bool pinned = false;  // referenced by a captured HIP graph: never freed before hbmpc_destroy
        bool dirty = true;    // the batch_recover counters at its start are not known to be zero (fresh, or a failed call)
    };
    std::map<std::string, Tab> tables;             // device-resident constant tables
    std::vector<uint32_t*> retired_tables;         // evicted one flush ago; freed at the next flush (see get_table)
    std::vector<void*> retired_scratch;            // outgrown scratch that a graph may still reference
    size_t evictions = 0;                          // flushes of the table cache so far
    std::vector<void*> pin_free;                   // pinned, device-mapped staging blocks of small host-pointer calls (Stage)
    size_t wide_max_chunks = 8192;                 // batch_recover calls up to this many chunks (evaluations: a quarter of it) use the wave-per-chunk kernels
    bool second_chance = true;                     // flagged chunks try two cheap interpolation candidates before OEC/Gao
    bool direct_fail = true;                       // a decode with no OEC round (S == d + t + 1) is ONE launch: failures are written by the first kernel
    bool zero_copy = true;                         // small host-pointer calls stage through mapped host memory
    bool matrix_cores = true;                      // large Fr decodes run the int8 MFMA formulation (kernels_mfma.hpp)
    size_t mfma_min_chunks = 65536;                // A/B aid since the tables are expanded on the device: the thresholds below decide
    size_t mfma_min_cached = 4096;                 // ... from this many chunks on (the crossover with the wave-per-chunk kernels)
    size_t mfma_min_direct = 2048;                 // ... and from this many when the call has no OEC round (one launch)
    size_t mfma_min_encode = 2049;                 // encodes (one table per (n, d), never rebuilt): right above the wave-per-chunk range
    int lazy_fallback_tables = 1;                  // a new sender set's OEC / Gao and second-chance tables are built when a chunk needs them: 1 = host-pointer calls, 2 = all
    bool device_tables = true;                     // the matrix-core table of a new sender set is expanded on the device (kernels_tables.hpp)
    size_t pair_decode_min = 8192;                // hbmpc_dev_fpmul_parties: from this many elements the first open forms its shares at load time
    size_t fused_triplegen_max = 1024;             // hbmpc_dev_triplegen_parties: one launch (a workgroup per chunk of 2t + 1 triples) up to this many chunks (0: never)
    size_t fused_fpmul_max = 2048;                 // hbmpc_dev_fpmul_parties: one launch (a wave per element) up to this many elements (0: never)
    bool gather_row_copies = false;                // hbmpc_dev_gather_party_major: take the per-row peer copies even where the 2-D copy applies (A/B aid)
    bool list_rows_in_kernel = true;               // the producers' mixing step writes the parties' lists itself (k_mfma_bfly<.., LISTS>)
    bool mfma_bfly = true;                         // large encodes take the domain points in pairs (kernels_mfma_bfly.hpp)
    bool mfma_team = true;                         // batches with fewer tiles than waves: a workgroup per tile (kernels_mfma_team.hpp)
    size_t mfma_min_gold = 4096;                   // Goldilocks encodes (tiny tables, one workgroup kind): from this many chunks
    size_t mfma_min_gold_direct = 2048;            // Goldilocks decodes without OEC rounds (one launch): flat ~7 us against a wave-per-chunk kernel that grows
    size_t mfma_min_gold_oec = 8193;               // Goldilocks decodes with OEC rounds (four launches against the small-batch path's two): beyond its range
    std::map<size_t, HFr> inv_pow2;                // (2^m)^-1 per m: a field inversion is ~20 us of host time, more than a small launch
    std::map<size_t, std::shared_ptr<DomainInv<HFr>>> dom_fr;  // per n: domain elements + inverse differences (tables.hpp), built once
    std::map<size_t, std::shared_ptr<DomainInv<HGl>>> dom_gl;
    int n_cus = 256;
    int mfma_wgs = 0;                              // test aid: workgroups of a matrix-core launch (0 = one per CU)
    std::map<hipStream_t, Scratch> scratch;        // per-stream scratch (calls on one stream are ordered)
    std::vector<std::pair<void*, size_t>> stage_free;  // device staging buffers of the host-pointer API, kept between calls
    size_t stage_bytes = 0;
};

// the calling thread's last failure message (hbmpc_last_error): thread-local, so concurrent callers of one
// context never write the same string
static thread_local std::string g_err;
// > 0 while the calling thread is between hbmpc_graph_begin_capture and _end_capture (capture mode is thread-local):
// tables and scratch it looks up are then pinned for the graph's lifetime, and nothing may be allocated
static thread_local int g_capturing = 0;
static const ShareErrorCode HBMPC_NOT_FUSED = (ShareErrorCode)9999;  // internal: a fused form does not cover the call (nothing was enqueued)
// internal: batch_recover_dev was asked to form the senders' values at load time and the call cannot take that form

#define HIP_TRY(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            (void)(ctx);                                                                                \
            g_err = std::string(#call) + ": " + hipGetErrorString(e__);                            \
            return e__ == hipErrorOutOfMemory ? HBMPC_OUT_OF_MEMORY : HBMPC_NO_DEVICE;                  \
        }                                                                                               \
    } while (0)

static ShareErrorCode fail(hbmpc_ctx* ctx, ShareErrorCode rc, const char* msg) {
    (void)ctx;
    g_err = msg;
    return rc;
}

static bool is_gold(const hbmpc_ctx* ctx) { return ctx->impl == IMPL_GOLD; }
static size_t ebytes(const hbmpc_ctx* ctx) { return impl_ebytes(ctx->impl); }
// entry points are typed per field (U256* vs uint64_t*): a context only serves the field it was created for
#define REQ_FR(ctx) do { if ((ctx) && is_gold(ctx)) return fail((ctx), TypeMismatch, "this context was created for Goldilocks: use the hbmpc_gl_* entry points"); } while (0)
#define REQ_GL(ctx) do { if ((ctx) && !is_gold(ctx)) return fail((ctx), TypeMismatch, "this context was created for bls12-381 Fr: hbmpc_gl_* needs a Goldilocks64 context"); } while (0)

// ---- table cache -------------------------------------------------------------------------------
// Looks the table up or builds it, all under ctx->mu.  `aux` (optional): in = what the builder computed alongside the
// words (read after build() returns), out = what is stored with the table -- a table and its offsets are published
// together, so a second thread that finds the table cached also finds its layout.
// frees a fresh device block unless it was published (the HIP_TRY early returns between hipMalloc and the cache entry)
struct DevBlockGuard {
    uint32_t* p = nullptr;
    ~DevBlockGuard() {
        if (p) (void)hipFree(p);
    }
    uint32_t* release() {
        uint32_t* q = p;
        p = nullptr;
        return q;
    }
};
template <class Build>
static ShareErrorCode get_table(hbmpc_ctx* ctx, const std::string& key, Build build, const uint32_t** out,
                                std::array<size_t, 5>* aux = nullptr) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        if (g_capturing) it->second.pinned = true;
        *out = it->second.p;
        if (aux) *aux = it->second.aux;
        return ShareSuccess;
    }
    if (g_capturing) return fail(ctx, HBMPC_NO_DEVICE, "a table would have to be built during graph capture: run the call sequence once eagerly first");
    if (ctx->tables.size() >= 512) {
        // Bound the cache in a long-running node (every new sender set is a new table).  Eviction is two-phase so
        // that a concurrent call which has looked a table up but not launched yet never sees it freed: tables
        // evicted now are only unlinked, and freed at the NEXT flush (>= 512 table builds later), after a device
        // synchronise for kernels still running on caller streams.  Tables a captured graph references stay.
        ++ctx->evictions;
        (void)hipDeviceSynchronize();
        for (uint32_t* q : ctx->retired_tables) (void)hipFree(q);
        ctx->retired_tables.clear();
        for (auto t = ctx->tables.begin(); t != ctx->tables.end();) {
            if (t->second.pinned) {
                ++t;
                continue;
            }
            ctx->retired_tables.push_back(t->second.p);
            t = ctx->tables.erase(t);
        }
    }
    std::vector<uint32_t> host = build();
    if (host.empty()) host.push_back(0);
    DevBlockGuard blk;
    HIP_TRY(ctx, hipMalloc(&blk.p, host.size() * 4));
    HIP_TRY(ctx, hipMemcpy(blk.p, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    hbmpc_ctx::Tab& tab = ctx->tables[key];
    tab.p = blk.release();
    if (aux) tab.aux = *aux;
    *out = tab.p;
    return ShareSuccess;
}
// A table whose bulk is EXPANDED ON THE DEVICE from a small host-built seed (the matrix-core byte-digit table: a few KB of
// coefficients -> hundreds of KB): the entry is dev_words of table followed by the seed; `expand` enqueues the kernels on
// the context's own stream, which is drained before the entry is published -- so whichever stream or thread uses the
// entry next is ordered behind the build by the host, and the caller's stream is never waited for.
template <class Build, class Expand>
static ShareErrorCode get_table_expanded(hbmpc_ctx* ctx, const std::string& key, size_t dev_words, Build build_seed, Expand expand,
                                         const uint32_t** out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        if (g_capturing) it->second.pinned = true;
        *out = it->second.p;
        return ShareSuccess;
    }
    if (g_capturing) return fail(ctx, HBMPC_NO_DEVICE, "a table would have to be built during graph capture: run the call sequence once eagerly first");
    if (ctx->tables.size() >= 512) {  // same two-phase flush as get_table
        ++ctx->evictions;
        (void)hipDeviceSynchronize();
        for (uint32_t* q : ctx->retired_tables) (void)hipFree(q);
        ctx->retired_tables.clear();
        for (auto t = ctx->tables.begin(); t != ctx->tables.end();) {
            if (t->second.pinned) {
                ++t;
                continue;
            }
            ctx->retired_tables.push_back(t->second.p);
            t = ctx->tables.erase(t);
        }
    }
    // (ctx->mu is held across the build and the drain of the context's own stream: a second thread that wants the SAME new table
    // must find it finished, and builds are rare -- one per new sender set -- and short, two launches of a few microseconds)
    std::vector<uint32_t> seed = build_seed();
    DevBlockGuard blk;
    HIP_TRY(ctx, hipMalloc(&blk.p, (dev_words + seed.size() + 16) * 4));
    HIP_TRY(ctx, hipMemcpyAsync(blk.p + dev_words, seed.data(), seed.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    expand(blk.p, blk.p + dev_words, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // also keeps `seed` (pageable host memory) alive until the copy has run
    hbmpc_ctx::Tab& tab = ctx->tables[key];
    tab.p = blk.release();
    *out = tab.p;
    return ShareSuccess;
}
static bool table_cached(hbmpc_ctx* ctx, const std::string& key) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ctx->tables.find(key) != ctx->tables.end();
}
// Per-stream scratch from plain hipMalloc.  NOT hipMallocAsync: data written to stream-ordered-pool memory by
// one kernel was observed stale for workgroups of the NEXT kernel that run on other XCDs (their L2 kept the
// value an earlier memset/kernel had left there), which silently dropped flagged chunks; ordinary hipMalloc
// memory is coherent at kernel boundaries.  Calls on one stream are ordered, so they can share a buffer; it
// only grows (after draining the stream).
static ShareErrorCode get_scratch(hbmpc_ctx* ctx, hipStream_t s, size_t bytes, void** out, bool* dirty = nullptr) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto& slot = ctx->scratch[s];
    if (slot.cap < bytes) {
        if (g_capturing) return fail(ctx, HBMPC_NO_DEVICE, "scratch would have to grow during graph capture: run the call sequence once eagerly first");
        if (slot.p) {
            HIP_TRY(ctx, hipStreamSynchronize(s));
            if (slot.pinned) ctx->retired_scratch.push_back(slot.p);  // a graph replays kernels that point into it
            else HIP_TRY(ctx, hipFree(slot.p));
            slot = hbmpc_ctx::Scratch();
        }
        const size_t want = bytes < (1u << 16) ? (1u << 16) : bytes + bytes / 2;
        HIP_TRY(ctx, hipMalloc(&slot.p, want));
        slot.cap = want;
    }
    if (g_capturing) slot.pinned = true;
    if (dirty) *dirty = slot.dirty;
    *out = slot.p;
    return ShareSuccess;
}
static void set_scratch_dirty(hbmpc_ctx* ctx, hipStream_t s, bool dirty) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->scratch.find(s);
    if (it != ctx->scratch.end()) it->second.dirty = dirty;
}
static std::string key(const char* kind, std::initializer_list<size_t> v, int impl) {
    std::string k = kind;
    for (size_t x : v) k += ":" + std::to_string(x);
    return k + "#" + std::to_string(impl);
}

static HFr rdev_value(int impl) {  // Rdev mod r as a field value: 2^256 (sat32) or 2^261 (u29)
    HFr two = HFr::from_u64(2), p = HFr::one();
    const int bits = impl == IMPL_U29 ? 261 : 256;
    for (int i = 0; i < bits; ++i) p = p * two;
    return p;
}
// r2: mont(x, r2) = x * Rdev (canonical -> Montgomery).  c0: a constant in device-constant form.
// c1_plain: a canonical value in plain limb form (for load_const).
static ElemConsts elem_consts(int impl, const HFr* c0 = nullptr, const HFr* c1_plain = nullptr) {
    ElemConsts cs = {};
    std::vector<uint32_t> v;
    if (impl == IMPL_GOLD) {  // no Montgomery form: mont(x, 1) = x
        cs.r2[0] = 1;
        return cs;
    }
    put_const(v, rdev_value(impl), impl);
    for (int i = 0; i < impl_nl(impl); ++i) cs.r2[i] = v[i];
    if (c0) {
        v.clear();
        put_const(v, *c0, impl);
        for (int i = 0; i < impl_nl(impl); ++i) cs.c0[i] = v[i];
    }
    if (c1_plain) {
        v.clear();
        put_plain(v, *c1_plain, impl);
        for (int i = 0; i < impl_nl(impl); ++i) cs.c1[i] = v[i];
    }
    return cs;
}

// (2^m)^-1, computed once per m and context (TruncPr's last step, truncpr.rs:216-220)
static HFr inv_pow2(hbmpc_ctx* ctx, size_t m) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->inv_pow2.find(m);
    if (it != ctx->inv_pow2.end()) return it->second;
    if (ctx->inv_pow2.size() > 4096) ctx->inv_pow2.clear();
    const HFr v = HFr::from_u64(2).pow_u64(m).inv();
    ctx->inv_pow2.emplace(m, v);
    return v;
}
// ---- context -----------------------------------------------------------------------------------
extern "C" const char* hbmpc_version(void) { return "hbmpc-hip 0.1 (gfx950)"; }

extern "C" ShareErrorCode hbmpc_create(int device, FieldKind field_kind, hbmpc_ctx** ctx_out) {
    if (!ctx_out) return InvalidInput;
    *ctx_out = nullptr;
    if (field_kind != Bls12_381Fr && field_kind != Goldilocks64) return TypeMismatch;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        g_err = "hbmpc_create: no HIP device " + std::to_string(device) + " (this library has no CPU path)";
        return HBMPC_NO_DEVICE;
    }
    hbmpc_ctx* ctx = new hbmpc_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        g_err = "hbmpc_create: hipSetDevice/hipStreamCreate failed";
        delete ctx;
        return HBMPC_NO_DEVICE;
    }
    const char* env = getenv("HBMPC_FIELD_IMPL");
    if (env && std::string(env) == "sat32") ctx->impl = IMPL_SAT32;
    if (field_kind == Goldilocks64) ctx->impl = IMPL_GOLD;
    env = getenv("HBMPC_MATRIX_CORES");
    if (env && std::string(env) == "0") ctx->matrix_cores = false;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cus = prop.multiProcessorCount;
    *ctx_out = ctx;
    return ShareSuccess;
}
// Staging memory of the host-pointer API holds copies of what callers passed in (polynomial coefficients, i.e. secrets,
// shares) until a later call overwrites it: zero what the pools hold.
extern "C" ShareErrorCode hbmpc_scrub_staging(hbmpc_ctx* ctx) {
    // (blocks checked out by a host-pointer call running on another thread, and the per-stream scratch, are not in the
    // pools and are not touched)
    if (!ctx) return InvalidInput;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (auto& b : ctx->stage_free) HIP_TRY(ctx, hipMemsetAsync(b.first, 0, b.second, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (void* q : ctx->pin_free) memset(q, 0, STAGE_PIN_BLOCK);
    return ShareSuccess;
}
extern "C" void hbmpc_destroy(hbmpc_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hbmpc_scrub_staging(ctx);
    for (auto& kv : ctx->tables) (void)hipFree(kv.second.p);
    for (uint32_t* q : ctx->retired_tables) (void)hipFree(q);
    for (auto& kv : ctx->scratch) (void)hipFree(kv.second.p);
    for (void* q : ctx->retired_scratch) (void)hipFree(q);
    for (void* q : ctx->pin_free) (void)hipHostFree(q);
    for (auto& b : ctx->stage_free) (void)hipFree(b.first);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
extern "C" const char* hbmpc_last_error(const hbmpc_ctx*) { return g_err.c_str(); }
extern "C" FieldKind hbmpc_field_of(const hbmpc_ctx* ctx) { return ctx && is_gold(ctx) ? Goldilocks64 : Bls12_381Fr; }
extern "C" ShareErrorCode hbmpc_set_field_impl(hbmpc_ctx* ctx, int impl) {
    if (!ctx || (impl != IMPL_U29 && impl != IMPL_SAT32)) return InvalidInput;
    REQ_FR(ctx);
    ctx->impl = impl;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_matrix_cores(hbmpc_ctx* ctx, int on, size_t min_chunks) {  // either field
    if (!ctx) return InvalidInput;
    ctx->matrix_cores = on != 0;
    ctx->mfma_team = on != 2;  // 2: without the workgroup-per-tile kernel of small batches (A/B aid)
    ctx->mfma_bfly = on != 3;  // 3: large encodes with one table row per point instead of per point pair (A/B aid)
    if (min_chunks) {
        ctx->mfma_min_gold = std::min<size_t>(min_chunks, 4096);
        ctx->mfma_min_gold_direct = std::min<size_t>(min_chunks, 2048);
        ctx->mfma_min_gold_oec = std::min<size_t>(min_chunks, 8193);
        ctx->mfma_min_chunks = min_chunks;
        ctx->mfma_min_cached = std::min<size_t>(min_chunks, 4096);
        ctx->mfma_min_direct = std::min<size_t>(min_chunks, 2048);
        ctx->mfma_min_encode = std::min<size_t>(min_chunks, 2049);
    }
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_fused_fpmul(hbmpc_ctx* ctx, size_t max_elements) {
    if (!ctx) return InvalidInput;
    ctx->fused_fpmul_max = max_elements;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_fused_triplegen(hbmpc_ctx* ctx, size_t max_chunks) {
    if (!ctx) return InvalidInput;
    ctx->fused_triplegen_max = max_chunks;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_fpmul_pair_decode(hbmpc_ctx* ctx, size_t min_elements) {
    if (!ctx) return InvalidInput;
    ctx->pair_decode_min = min_elements;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_gather_row_copies(hbmpc_ctx* ctx, int on) {  // either field
    if (!ctx) return InvalidInput;
    ctx->gather_row_copies = on != 0;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_producer_fusion(hbmpc_ctx* ctx, int on) {  // either field
    if (!ctx) return InvalidInput;
    ctx->list_rows_in_kernel = on != 0;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_lazy_fallback_tables(hbmpc_ctx* ctx, int on) {  // either field
    if (!ctx) return InvalidInput;
    if (on < 0 || on > 2) return InvalidInput;
    ctx->lazy_fallback_tables = on;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_single_launch_decode(hbmpc_ctx* ctx, int on) {  // either field
    if (!ctx) return InvalidInput;
    ctx->direct_fail = on != 0;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_matrix_core_workgroups(hbmpc_ctx* ctx, int workgroups) {
    if (!ctx || workgroups < 0) return InvalidInput;
    ctx->mfma_wgs = workgroups;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_small_batch_chunks(hbmpc_ctx* ctx, size_t max_chunks) {
    if (!ctx) return InvalidInput;
    ctx->wide_max_chunks = max_chunks;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_second_chance(hbmpc_ctx* ctx, int on) {
    if (!ctx) return InvalidInput;
    ctx->second_chance = on != 0;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_cache_stats(hbmpc_ctx* ctx, size_t stats_out[4]) {
    if (!ctx || !stats_out) return InvalidInput;
    std::lock_guard<std::mutex> lk(ctx->mu);
    size_t pinned = 0;
    for (auto& kv : ctx->tables) pinned += kv.second.pinned;
    stats_out[0] = ctx->tables.size(), stats_out[1] = pinned, stats_out[2] = ctx->retired_tables.size(), stats_out[3] = ctx->evictions;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_set_force_generic(hbmpc_ctx* ctx, int on) {  // either field
    if (!ctx) return InvalidInput;
    ctx->force_generic = on != 0;
    return ShareSuccess;
}

extern "C" ShareErrorCode hbmpc_dev_alloc(hbmpc_ctx* ctx, size_t bytes, void** dptr_out) {
    if (!ctx || !dptr_out) return InvalidInput;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(dptr_out, bytes ? bytes : 1));
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_free(hbmpc_ctx* ctx, void* dptr) {
    if (!ctx) return InvalidInput;
    HIP_TRY(ctx, hipFree(dptr));
    return ShareSuccess;
}
// The stream-ordered pool of the context's device (hipMallocAsync): caller buffers from it are safe only while the pool retains its
// freed blocks (include/hbmpc_hip.h, "Device buffers").  Pointer attributes do not say which allocator a buffer came from
// (HIP_POINTER_ATTRIBUTE_MEMPOOL_HANDLE: hipErrorNotSupported on ROCm 7.2, tools/probe_pool_attr.hip), so the check is on the pool.
extern "C" ShareErrorCode hbmpc_stream_pool_release_threshold(hbmpc_ctx* ctx, uint64_t* threshold_out) {
    if (!ctx || !threshold_out) return InvalidInput;
    hipMemPool_t pool = nullptr;
    HIP_TRY(ctx, hipDeviceGetMemPool(&pool, ctx->device));
    HIP_TRY(ctx, hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, threshold_out));
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_stream_pool_retain(hbmpc_ctx* ctx) {
    if (!ctx) return InvalidInput;
    hipMemPool_t pool = nullptr;
    uint64_t keep = ~0ull;
    HIP_TRY(ctx, hipDeviceGetMemPool(&pool, ctx->device));
    HIP_TRY(ctx, hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    return ShareSuccess;
}
static hipStream_t pick(hbmpc_ctx* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }
extern "C" ShareErrorCode hbmpc_memcpy_h2d(hbmpc_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
    if (!ctx) return InvalidInput;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, pick(ctx, stream)));
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_memcpy_d2h(hbmpc_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
    if (!ctx) return InvalidInput;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, pick(ctx, stream)));
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_memcpy_d2d_rows(hbmpc_ctx* ctx, void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes,
                                                size_t row_bytes, size_t rows, void* stream) {
    if (!ctx) return InvalidInput;
    if ((rows && row_bytes && (!dst || !src)) || dst_pitch_bytes < row_bytes || src_pitch_bytes < row_bytes) return fail(ctx, InvalidInput, "null buffer or pitch below the row");
    if (rows == 0 || row_bytes == 0) return ShareSuccess;
    HIP_TRY(ctx, hipMemcpy2DAsync(dst, dst_pitch_bytes, src, src_pitch_bytes, row_bytes, rows, hipMemcpyDeviceToDevice, pick(ctx, stream)));
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_memcpy_d2d(hbmpc_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
    if (!ctx) return InvalidInput;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, pick(ctx, stream)));
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_stream_sync(hbmpc_ctx* ctx, void* stream) {
    if (!ctx) return InvalidInput;
    HIP_TRY(ctx, hipStreamSynchronize(pick(ctx, stream)));
    return ShareSuccess;
}

extern "C" ShareErrorCode hbmpc_stream_create(hbmpc_ctx* ctx, void** stream_out) {
    if (!ctx || !stream_out) return InvalidInput;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = s;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_gather_party_major(hbmpc_ctx* const* ctxs, size_t n_shards, size_t root,
                                                       const void* const* shards_dev, const size_t* counts,
                                                       const size_t* strides, size_t n_rows, void* out_dev, size_t out_stride,
                                                       int sync_sources, void* stream) {
    if (!ctxs || n_shards == 0 || root >= n_shards || !ctxs[root]) return InvalidInput;
    hbmpc_ctx* rc_ctx = ctxs[root];
    if (!shards_dev || !counts || !strides || !out_dev) return fail(rc_ctx, InvalidInput, "null argument");
    size_t total = 0;
    for (size_t r = 0; r < n_shards; ++r) {
        if (!ctxs[r]) return fail(rc_ctx, InvalidInput, "null context in the shard list");
        if (is_gold(ctxs[r]) != is_gold(rc_ctx)) return fail(rc_ctx, TypeMismatch, "contexts of different fields");
        if (strides[r] < counts[r]) return fail(rc_ctx, InvalidInput, "shard row stride below its column count");
        if (counts[r] && !shards_dev[r]) return fail(rc_ctx, InvalidInput, "null shard");
        total += counts[r];
    }
    if (out_stride < total) return fail(rc_ctx, InvalidInput, "output row stride below the total column count");
    const size_t eb = ebytes(rc_ctx);
    if (sync_sources) {
        for (size_t r = 0; r < n_shards; ++r) {
            HIP_TRY(rc_ctx, hipSetDevice(ctxs[r]->device));
            HIP_TRY(rc_ctx, hipStreamSynchronize(ctxs[r]->stream));
        }
    }
    HIP_TRY(rc_ctx, hipSetDevice(rc_ctx->device));
    hipStream_t s = pick(rc_ctx, stream);
    size_t col = 0;
    for (size_t r = 0; r < n_shards; ++r) {
        const int src_dev = ctxs[r]->device;
        bool direct = src_dev == rc_ctx->device;
        if (rc_ctx->gather_row_copies) direct = false;  // A/B aid: the branch a pair of devices without peer access takes
        else if (!direct) {
            int can = 0;
            HIP_TRY(rc_ctx, hipDeviceCanAccessPeer(&can, rc_ctx->device, src_dev));
            if (can) {
                const hipError_t e = hipDeviceEnablePeerAccess(src_dev, 0);  // idempotent across calls
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(rc_ctx, e);
                (void)hipGetLastError();
                direct = true;
            }
        }
        if (counts[r] && n_rows) {
            char* dst = (char*)out_dev + col * eb;
            const char* src = (const char*)shards_dev[r];
            if (direct) {
                // ONE strided copy per shard: n_rows row segments of counts[r] elements (the destination reads the source
                // over xGMI when the devices differ)
                HIP_TRY(rc_ctx, hipMemcpy2DAsync(dst, out_stride * eb, src, strides[r] * eb, counts[r] * eb, n_rows, hipMemcpyDeviceToDevice, s));
            } else {
                // no peer access between the two devices: the runtime stages every row through the host
                for (size_t j = 0; j < n_rows; ++j)
                    HIP_TRY(rc_ctx, hipMemcpyPeerAsync(dst + j * out_stride * eb, rc_ctx->device, src + j * strides[r] * eb, src_dev, counts[r] * eb, s));
            }
        }
        col += counts[r];
    }
    return ShareSuccess;
}
// 1: the root's device reads the source's memory directly (same device, or peer access over xGMI, enabled here);
// 0: hbmpc_dev_gather_party_major falls back to copies the runtime stages through the host
extern "C" ShareErrorCode hbmpc_dev_peer_access(hbmpc_ctx* root, hbmpc_ctx* source, int* direct_out) {
    if (!root || !source || !direct_out) return InvalidInput;
    *direct_out = 0;
    if (root->device == source->device) {
        *direct_out = 1;
        return ShareSuccess;
    }
    int can = 0;
    HIP_TRY(root, hipSetDevice(root->device));
    HIP_TRY(root, hipDeviceCanAccessPeer(&can, root->device, source->device));
    if (can) {
        const hipError_t e = hipDeviceEnablePeerAccess(source->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(root, e);
        (void)hipGetLastError();
        *direct_out = 1;
    }
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_stream_destroy(hbmpc_ctx* ctx, void* stream) {
    if (!ctx || !stream) return InvalidInput;
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    {
        std::lock_guard<std::mutex> lk(ctx->mu);  // the stream's scratch goes with it
        auto it = ctx->scratch.find((hipStream_t)stream);
        if (it != ctx->scratch.end()) {
            if (it->second.pinned) ctx->retired_scratch.push_back(it->second.p);  // a graph may be replayed elsewhere
            else (void)hipFree(it->second.p);
            ctx->scratch.erase(it);
        }
    }
    HIP_TRY(ctx, hipStreamDestroy((hipStream_t)stream));
    return ShareSuccess;
}

// ---- HIP graphs ----------------------------------------------------------------------------------
struct hbmpc_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t captured_on = nullptr;  // the stream whose scratch (decode counters) the recorded kernels point into
    void* scratch = nullptr;
};
extern "C" ShareErrorCode hbmpc_graph_begin_capture(hbmpc_ctx* ctx, void* stream) {
    if (!ctx) return InvalidInput;
    if (!stream) return fail(ctx, InvalidInput, "graph capture needs an explicit stream");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    ++g_capturing;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_graph_end_capture(hbmpc_ctx* ctx, void* stream, hbmpc_graph** graph_out) {
    if (!ctx || !graph_out) return InvalidInput;
    *graph_out = nullptr;
    if (!stream) return fail(ctx, InvalidInput, "graph capture needs an explicit stream");
    if (g_capturing > 0) --g_capturing;
    hbmpc_graph* g = new hbmpc_graph();
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        g_err = std::string("graph capture failed: ") + hipGetErrorString(e);
        hbmpc_graph_destroy(g);
        return HBMPC_NO_DEVICE;
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->scratch.find((hipStream_t)stream);
        g->captured_on = (hipStream_t)stream;
        g->scratch = it != ctx->scratch.end() ? it->second.p : nullptr;
    }
    *graph_out = g;
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_graph_launch(hbmpc_ctx* ctx, hbmpc_graph* graph, void* stream) {
    if (!ctx || !graph || !graph->exec) return InvalidInput;
    hipStream_t s = pick(ctx, stream);
    {
        // the recorded decodes expect their counters at zero and carry no clear of their own (batch_recover_dev): a call that
        // failed between its launches since the capture is the one case in which they are not
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->scratch.find(graph->captured_on);
        if (graph->scratch && it != ctx->scratch.end() && it->second.p == graph->scratch && it->second.dirty) {
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            HIP_TRY(ctx, hipMemsetAsync(graph->scratch, 0, 128, s));
            it->second.dirty = false;
        }
    }
    HIP_TRY(ctx, hipGraphLaunch(graph->exec, s));
    return ShareSuccess;
}
extern "C" void hbmpc_graph_destroy(hbmpc_graph* graph) {
    if (!graph) return;
    if (graph->exec) (void)hipGraphExecDestroy(graph->exec);
    if (graph->graph) (void)hipGraphDestroy(graph->graph);
    delete graph;
}

// ---- a3 / a5: evaluation on the domain ---------------------------------------------------------
// apply_vandermonde / compute_shares as int8 MFMA tiles (kernels_mfma.hpp): row j of the table is (alpha_j^k)_k
// x_row_stride != 0: x is given as d + 1 ROWS of G elements, x_row_stride elements apart (the point-pair kernel only)
// lists (the producers' mixing step, MfmaRowsArgs::list): only the point-pair kernel writes them; with lists set the call
// returns false unless that kernel ran
struct ListSpec {
    size_t row0, rows, K;
    const hbmpc_list_slice* slices;
    size_t n_slices;
    void* others = nullptr;  // non-null: the rows outside the lists go here, party-major: others[(j (n - rows) + r') K + k] (hbmpc_dev_vandermonde_apply_rows_split)
};
// x[P][G][M] -> y[P][n][G] as ONE launch of the point-pair kernel over the P G chunks (k_mfma_bfly<.., LISTS> with every row party-major,
// tu_mfma_bfly.inc: launch_lists): domains of 8 and 16 points, M <= 11, dense output rows, enough tiles over all parties to fill the chip
static bool party_batched_one_launch(const hbmpc_ctx* ctx, size_t G, size_t n, size_t dp1, const EvalOut& y) {
    const size_t size = domain_size(n);
    return y.parties > 1 && y.ys == 0 && ctx->impl == IMPL_U29 && ctx->matrix_cores && ctx->mfma_bfly && !ctx->force_generic && size >= 8 && size <= 16 &&
           n > size / 2 && dp1 >= 2 && dp1 <= 11 && ((size_t)y.parties * G + 31) / 32 > (size_t)(ctx->mfma_wgs ? ctx->mfma_wgs : ctx->n_cus) * 2 &&
           (size_t)y.parties * n * G * 32 < ((size_t)1 << 32) && (size_t)y.parties * G * dp1 * 32 < ((size_t)1 << 32);
}
static bool try_mfma_eval(hbmpc_ctx* ctx, const uint32_t* x, size_t G, size_t n, size_t dp1, EvalOut y, hipStream_t s,
                          ShareErrorCode* rc_out, size_t x_row_stride = 0, const ListSpec* lists = nullptr) {
    *rc_out = ShareSuccess;
    const size_t rowb = mf_row_bytes(dp1);
    mf::MfmaRowsArgs a;
    memset(&a, 0, sizeof a);
    // up to two tiles per workgroup: the workgroup-per-tile kernel (kernels_mfma_team.hpp; no verify rows here, so no barriers)
    const int nwg = ctx->mfma_wgs ? ctx->mfma_wgs : ctx->n_cus;
    // (measured, 4 096 .. 16 384 chunks: n = 20, d = 6: 6.7 .. 10.0 us against 19 .. 20; n = 31, d = 10 -- three roles --
    // 9.8 and 15.1 us against 17.6 and 18.2 at 4 096 and 8 192 chunks, behind at 16 384)
    // several parties' encodes in one launch (below): the tiles of all of them count
    const bool parties_one = party_batched_one_launch(ctx, G, n, dp1, y) && !lists && !x_row_stride;
    bool team = ctx->mfma_team && x_row_stride == 0 && dp1 <= MF_MAX_M && (G + 31) / 32 <= (size_t)nwg * 2 && !parties_one;
    bool plain_ok = x_row_stride == 0 && dp1 <= MF_MAX_M && mf::mf_plan_roles((int)n, 0, (int)((160 * 1024 - (team ? 128 : 0)) / rowb), nwg, &a);
    if (plain_ok && team && a.nroles > 1 && (G + 31) / 32 > (size_t)nwg) {
        team = false;
        plain_ok = mf::mf_plan_roles((int)n, 0, (int)((160 * 1024) / rowb), nwg, &a);
    }
    if (!plain_ok) team = false;  // the point pairs below may still fit (half the rows)
    auto vandermonde = [&] {
        std::vector<HFr> el = domain_elements<HFr>(n, n);
        std::vector<std::vector<HFr>> V(n, std::vector<HFr>(dp1));
        for (size_t j = 0; j < n; ++j) {
            HFr p = HFr::one();
            for (size_t k = 0; k < dp1; ++k) {
                V[j][k] = p;
                p = p * el[j];
            }
        }
        return V;
    };
    a.G = G;
    a.in_chunk_major = 1;
    a.nv = 0;
    a.out_party_major = 1;
    a.out_stride = y.ys ? y.ys : G;
    const int mi = (int)dp1;
    // Large batches take the points in pairs (k, k + size / 2): alpha_{k + size/2} = -alpha_k, so both outputs come from
    // the same M MFMAs (kernels_mfma_bfly.hpp) -- config 2: 0.16 ms against 0.18, config 3's encode 0.32 against 0.50
    // (profiles/r03_mfma_bfly_ubench.txt).  The workgroup-per-tile kernel keeps the plain rows.
    const size_t half = domain_size(n) / 2;
    if (!team && ctx->mfma_bfly && half >= 2 && n > half) {
        mf::MfmaRowsArgs b = a;
        if (mf::mf_plan_pairs((int)half, (int)((160 * 1024) / mf_bfly_row_bytes(dp1)), nwg, &b)) {
            const uint32_t* tab;
            std::array<size_t, 5> aux = {0, 0, 0, 0, 0};  // aux[0]: the table's words (0: its digit-sum bound does not hold, tables_mfma.hpp)
            *rc_out = get_table(ctx, key("mfbfly", {n, dp1}, ctx->impl), [&] {
                std::vector<uint32_t> tbl = build_mfma_bfly_table(vandermonde(), dp1, half);
                aux[0] = tbl.size();
                return tbl;
            }, &tab, &aux);
            if (*rc_out != ShareSuccess) return true;
            bool ok = aux[0] != 0;
            b.table = (const uint8_t*)tab;
            b.half = (int)half, b.nout = (int)n;
            if (x_row_stride) {
                b.in_chunk_major = 0, b.row_stride = x_row_stride;
                for (size_t i = 0; i < dp1; ++i) b.rows.set(i, (unsigned)i);
            }
            if (lists) {
                b.list_row0 = (int)lists->row0, b.list_rows = (int)lists->rows, b.list_K = (uint32_t)lists->K;
                if (lists->others) b.other_stride = (uint32_t)((n - lists->rows) * lists->K);
                for (size_t k = 0; k < 2; ++k) {
                    const bool have = k < lists->n_slices;
                    b.list[k].dst = have ? (uint8_t*)lists->slices[k].dst_dev : nullptr;
                    b.list[k].stride = have ? lists->slices[k].party_stride : 0;
                    b.list[k].k0 = have ? (uint32_t)lists->slices[k].k0 : 0u;
                    b.list[k].count = have ? (uint32_t)lists->slices[k].count : 0u;
                }
            }
            // party-batched calls x[P][G][M] -> y[P][n][G]: ONE launch over the P G chunks where the instance with party-major "other" rows
            // covers the shape (MfmaRowsArgs::other_stride with no list rows: chunk (p, g) of row k goes to ((p n + k) G + g)) -- the
            // dealers' encodes of the producers; otherwise a launch per party
            if (parties_one && b.nroles == 1 && (b.role[0].nrows == 4 || b.role[0].nrows == 8)) {
                mf::MfmaRowsArgs pb = b;  // the plan does not depend on the number of chunks (workgroups stride over the tiles)
                pb.G = (size_t)y.parties * G, pb.list_K = (uint32_t)G, pb.other_stride = (uint32_t)(n * G), pb.list_row0 = 0, pb.list_rows = 0;
                pb.in = (const uint8_t*)x, pb.out = (uint8_t*)y.y;
                if (launch_mfma_bfly_a(mi, pb, ctx->device, s) || launch_mfma_bfly_b(mi, pb, ctx->device, s) || launch_mfma_bfly_c(mi, pb, ctx->device, s) ||
                    launch_mfma_bfly_d(mi, pb, ctx->device, s))
                    return true;
            }
            for (unsigned p = 0; p < y.parties && ok; ++p) {  // party-batched calls: one launch per party
                b.in = (const uint8_t*)x + (size_t)p * G * dp1 * 32;
                b.out = lists && lists->others ? (uint8_t*)lists->others : (uint8_t*)y.y + (size_t)p * n * b.out_stride * 32;
                ok = launch_mfma_bfly_a(mi, b, ctx->device, s) || launch_mfma_bfly_b(mi, b, ctx->device, s) ||
                     launch_mfma_bfly_c(mi, b, ctx->device, s) || launch_mfma_bfly_d(mi, b, ctx->device, s);
                if (!ok && p > 0) return false;  // cannot happen: the first party's launch decides
            }
            if (ok) return true;
        }
    }
    if (!plain_ok || lists) return false;
    const uint32_t* tab;
    *rc_out = get_table(ctx, key("mfvand", {n, dp1}, ctx->impl), [&] { return build_mfma_table(vandermonde(), dp1); }, &tab);
    if (*rc_out != ShareSuccess) return true;
    a.table = (const uint8_t*)tab;
    // party-batched calls (x[P][G][d+1] -> y[P][n][G]): one launch per party (each is >= tens of microseconds)
    for (unsigned p = 0; p < y.parties; ++p) {
        a.in = (const uint8_t*)x + (size_t)p * G * dp1 * 32;
        a.out = (uint8_t*)y.y + (size_t)p * n * a.out_stride * 32;
        if (!(launch_mfma_rows_a(mi, a, ctx->device, s, team) || launch_mfma_rows_b(mi, a, ctx->device, s, team) ||
              launch_mfma_rows_c(mi, a, ctx->device, s, team) || launch_mfma_rows_d(mi, a, ctx->device, s, team)))
            return false;
    }
    return true;
}

// the same over Goldilocks (kernels_mfma_gl.hpp): all n rows in every workgroup's LDS
static bool try_mfma_eval_gl(hbmpc_ctx* ctx, const uint32_t* x, size_t G, size_t n, size_t dp1, EvalOut y, hipStream_t s,
                             ShareErrorCode* rc_out) {
    *rc_out = ShareSuccess;
    if (mfgl_table_bytes(n, dp1) + 2048 > 160 * 1024) return false;
    const uint32_t* tab;
    *rc_out = get_table(ctx, key("mfvandgl", {n, dp1}, ctx->impl), [&] {
        std::vector<HGl> el = domain_elements<HGl>(n, n);
        std::vector<std::vector<HGl>> V(n, std::vector<HGl>(dp1));
        for (size_t j = 0; j < n; ++j) {
            HGl p = HGl::one();
            for (size_t k = 0; k < dp1; ++k) {
                V[j][k] = p;
                p = p * el[j];
            }
        }
        return build_mfma_table_gl(V, dp1);
    }, &tab);
    if (*rc_out != ShareSuccess) return true;
    mf::MfmaGlArgs a;
    memset(&a, 0, sizeof a);
    a.G = G;
    a.in_chunk_major = 1;
    a.table = (const uint8_t*)tab;
    a.m = (int)dp1, a.nrows = (int)n, a.nv = 0;
    a.out_party_major = 1;
    a.out_stride = y.ys ? y.ys : G;
    const size_t ntiles = (G + 31) / 32;
    const unsigned grid = (unsigned)std::min<size_t>((ntiles + 3) / 4, (size_t)(ctx->mfma_wgs ? ctx->mfma_wgs : ctx->n_cus * 4));
    for (unsigned p = 0; p < y.parties; ++p) {  // party-batched calls: one launch per party
        a.in = (const uint8_t*)x + (size_t)p * G * dp1 * 8;
        a.out = (uint8_t*)y.y + (size_t)p * n * a.out_stride * 8;
        if (!launch_mfma_rows_gl(a, grid, ctx->device, s)) return false;
    }
    return true;
}

// alpha_j^k, j < n, k <= d, as constants [n][d + 1] (k_eval_wide_dot, k_triplegen_wg)
template <class H = HFr>
static ShareErrorCode vmat_table(hbmpc_ctx* ctx, size_t n, size_t d, const uint32_t** out) {
    const int impl = ctx->impl;
    return get_table(ctx, key("vmat", {n, d}, impl), [&] {
        std::vector<uint32_t> w;
        for (const H& al : domain_elements<H>(n, n)) {
            H p = H::one();
            for (size_t k = 0; k <= d; ++k) {
                put_const(w, p, impl);
                p = p * al;
            }
        }
        return w;
    }, out);
}
static ShareErrorCode eval_impl(hbmpc_ctx* ctx, const uint32_t* x, size_t G, size_t n, size_t d, EvalOut y,
                                hipStream_t s) {
    const size_t size = domain_size(n), dp1 = d + 1;
    ShareErrorCode rc_mf = ShareSuccess;
    const int impl = ctx->impl;
    const bool gold = impl == IMPL_GOLD;
    if (G * y.parties <= ctx->wide_max_chunks / 4 && !ctx->force_generic) {  // small batch: wave per chunk
        if (impl == IMPL_U29 && n * dp1 * 36 <= 48 * 1024) {  // as a table product, the lanes sharing a point's terms (k_eval_wide_dot)
            const uint32_t* vmat;
            ShareErrorCode rc = vmat_table(ctx, n, d, &vmat);
            if (rc != ShareSuccess) return rc;
            launch_eval_wide_dot(x, G, (int)n, (int)dp1, vmat, y, s);
            return ShareSuccess;
        }
        const uint32_t* alpha;
        ShareErrorCode rc = get_table(ctx, key("alpha", {n}, impl), [&] {
            return gold ? build_alpha<HGl>(n, impl) : build_alpha<HFr>(n, impl);
        }, &alpha);
        if (rc != ShareSuccess) return rc;
        launch_eval_wide(impl, x, G, (int)n, (int)dp1, alpha, y, s);
        return ShareSuccess;
    }
    // mid-size batches on small domains as well: up to two tiles per workgroup the workgroup-per-tile matrix-core kernel
    // beats the single-pass FFT on latency -- n = 16, d = 5: 5.8 us against 12.3 us at 2 100 .. 4 096 chunks, 8.4 against
    // 13.3 at 16 384 (profiles/r02_team_kernel_encode.txt); at 2^20 the two tie
    if (impl == IMPL_U29 && size <= 16 && ctx->matrix_cores && ctx->mfma_team && !ctx->force_generic && y.parties == 1 && dp1 >= 2 &&
        dp1 <= MF_MAX_M && G >= ctx->mfma_min_encode && (G + 31) / 32 <= (size_t)(ctx->mfma_wgs ? ctx->mfma_wgs : ctx->n_cus) * 2 &&
        try_mfma_eval(ctx, x, G, n, dp1, y, s, &rc_mf))
        return rc_mf;
    // large batches on small domains: with the points taken in pairs the matrix-core encode is ahead of the single-pass FFT
    if (impl == IMPL_U29 && size <= 16 && size >= 8 && ctx->matrix_cores && ctx->mfma_bfly && !ctx->force_generic && y.parties <= 64 &&
        dp1 >= 2 && dp1 <= MF_BFLY_MAX_M && (G + 31) / 32 > (size_t)(ctx->mfma_wgs ? ctx->mfma_wgs : ctx->n_cus) * 2 &&
        G * std::max(dp1, (size_t)1) * 32 < ((size_t)1 << 32) && try_mfma_eval(ctx, x, G, n, dp1, y, s, &rc_mf))
        return rc_mf;
    // several parties' mid-size batches: one launch over all of them (the dealers' encodes of the producers)
    if (party_batched_one_launch(ctx, G, n, dp1, y) && try_mfma_eval(ctx, x, G, n, dp1, y, s, &rc_mf)) return rc_mf;
    if ((impl == IMPL_U29 || gold) && size <= 16 && !ctx->force_generic) {
        const uint32_t* tw;
        ShareErrorCode rc = get_table(ctx, key("tw", {size}, impl), [&] {
            return gold ? build_twiddles<HGl>(size, impl) : build_twiddles<HFr>(size, impl);
        }, &tw);
        if (rc != ShareSuccess) return rc;
        const int lg = ilog2(size), c = (int)dp1, nn = (int)n;
        if (gold ? launch_gold_fft1(lg, c, x, G, nn, tw, y, s)
                 : (lg < 4 ? launch_fft1_lo(lg, c, x, G, nn, tw, y, s)
                           : (launch_fft1_16a(c, x, G, nn, tw, y, s) || launch_fft1_16b(c, x, G, nn, tw, y, s) ||
                              launch_fft1_16c(c, x, G, nn, tw, y, s) || launch_fft1_16d(c, x, G, nn, tw, y, s))))
            return ShareSuccess;
    } else if (impl == IMPL_U29 && ctx->matrix_cores && !ctx->force_generic && y.parties <= 64 && dp1 >= 2 &&
               dp1 <= (ctx->mfma_bfly ? MF_BFLY_MAX_M : MF_MAX_M) && G >= ctx->mfma_min_encode && G * dp1 * 32 < ((size_t)1 << 32) && n <= 255 &&
               try_mfma_eval(ctx, x, G, n, dp1, y, s, &rc_mf)) {
        // domains beyond 16 points: the dense n x (d + 1) map on the matrix cores beats the multi-pass FFT (config 3's
        // encode: 0.45 ms against 0.62 ms); up to 16 points the single-pass FFT stays (config 2: a tie at 0.187 ms)
        return rc_mf;
    } else if (gold && ctx->matrix_cores && !ctx->force_generic && y.parties <= 64 && dp1 >= 2 && dp1 <= MFGL_MAX_M &&
               G >= ctx->mfma_min_gold && n <= 255 && try_mfma_eval_gl(ctx, x, G, n, dp1, y, s, &rc_mf)) {
        return rc_mf;
    } else if ((impl == IMPL_U29 || gold) && size <= 256 && dp1 <= 32 && !ctx->force_generic) {
        const size_t P = size / 16;
        const uint32_t *tw16, *twist;
        ShareErrorCode rc = get_table(ctx, key("tw", {16}, impl), [&] {
            return gold ? build_twiddles<HGl>(16, impl) : build_twiddles<HFr>(16, impl);
        }, &tw16);
        if (rc != ShareSuccess) return rc;
        rc = get_table(ctx, key("twist", {size, dp1}, impl), [&] {
            return gold ? build_twist<HGl>(size, P, dp1, impl) : build_twist<HFr>(size, P, dp1, impl);
        }, &twist);
        if (rc != ShareSuccess) return rc;
        const int c = (int)dp1, nn = (int)n, pp = (int)P;
        if (gold ? launch_gold_fftP(c, x, G, nn, pp, tw16, twist, y, s)
                 : (launch_fftP_a(c, x, G, nn, pp, tw16, twist, y, s) || launch_fftP_b(c, x, G, nn, pp, tw16, twist, y, s) ||
                    launch_fftP_c(c, x, G, nn, pp, tw16, twist, y, s) || launch_fftP_d(c, x, G, nn, pp, tw16, twist, y, s) ||
                    launch_fftP_fold(c, x, G, nn, pp, tw16, twist, y, s)))
            return ShareSuccess;
    }
    const uint32_t* alpha;
    ShareErrorCode rc = get_table(ctx, key("alpha", {n}, impl), [&] {
        return impl == IMPL_GOLD ? build_alpha<HGl>(n, impl) : build_alpha<HFr>(n, impl);
    }, &alpha);
    if (rc != ShareSuccess) return rc;
    launch_eval_generic(impl, x, G, (int)n, (int)dp1, alpha, y, s);
    return ShareSuccess;
}

static ShareErrorCode eval_dev(hbmpc_ctx* ctx, const void* x, size_t G, size_t n, size_t d, void* y, void* stream,
                               size_t parties = 1, size_t ystride = 0) {
    if (!ctx) return InvalidInput;
    if (parties == 0 || parties > 65535) return fail(ctx, InvalidInput, "parties must be in 1..65535");
    if (n <= d) return fail(ctx, InvalidInput, "number of shares must be greater than the degree");  // :59-64
    if (n == 0 || n > ((size_t)1 << 32)) return fail(ctx, NoSuitableDomain, "no radix-2 domain of that size");
    if (n > (1u << 20) || d > (1u << 20)) return fail(ctx, InvalidInput, "n, d beyond the supported range");
    if (ystride != 0 && ystride < G) return fail(ctx, InvalidInput, "output row stride must be >= G");
    if (G == 0) return ShareSuccess;
    if (!x || !y) return fail(ctx, InvalidInput, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    ShareErrorCode rc = eval_impl(ctx, (const uint32_t*)x, G, n, d, EvalOut{(uint32_t*)y, ystride, (unsigned)parties}, s);
    if (rc != ShareSuccess) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}

extern "C" ShareErrorCode hbmpc_dev_compute_shares(hbmpc_ctx* ctx, const U256* coeffs, size_t B, size_t n, size_t d,
                                                   U256* shares_out, void* stream) {
    REQ_FR(ctx);
    return eval_dev(ctx, coeffs, B, n, d, shares_out, stream);
}
extern "C" ShareErrorCode hbmpc_dev_vandermonde_apply(hbmpc_ctx* ctx, const U256* x, size_t G, size_t n, size_t d,
                                                      U256* y_out, void* stream) {
    REQ_FR(ctx);
    return eval_dev(ctx, x, G, n, d, y_out, stream);
}
// x[parties][G][d+1] -> y[parties][n][G]: the encodes of several parties in one launch
extern "C" ShareErrorCode hbmpc_dev_vandermonde_apply_parties(hbmpc_ctx* ctx, const U256* x, size_t G, size_t n, size_t d,
                                                              size_t parties, U256* y_out, void* stream) {
    REQ_FR(ctx);
    return eval_dev(ctx, x, G, n, d, y_out, stream, parties);
}
extern "C" ShareErrorCode hbmpc_gl_dev_vandermonde_apply_parties(hbmpc_ctx* ctx, const uint64_t* x, size_t G, size_t n,
                                                                 size_t d, size_t parties, uint64_t* y_out, void* stream) {
    REQ_GL(ctx);
    return eval_dev(ctx, x, G, n, d, y_out, stream, parties);
}
extern "C" ShareErrorCode hbmpc_gl_dev_compute_shares(hbmpc_ctx* ctx, const uint64_t* coeffs, size_t B, size_t n, size_t d,
                                                      uint64_t* shares_out, void* stream) {
    REQ_GL(ctx);
    return eval_dev(ctx, coeffs, B, n, d, shares_out, stream);
}
extern "C" ShareErrorCode hbmpc_gl_dev_vandermonde_apply(hbmpc_ctx* ctx, const uint64_t* x, size_t G, size_t n, size_t d,
                                                         uint64_t* y_out, void* stream) {
    REQ_GL(ctx);
    return eval_dev(ctx, x, G, n, d, y_out, stream);
}

// host wrapper helper: RAII device staging buffers for the host-pointer API.  They are recycled through a small
// per-context pool: hipMalloc + hipFree (which synchronises the device) cost ~0.45 ms per call, half the time of
// a 2^14-secret hbmpc_compute_shares.  Every user of a buffer runs on the context's own stream, so a recycled
// buffer is ordered behind its previous use.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hbmpc_ctx* owner = nullptr;
    ~DevBuf() {
        if (!p) return;
        std::lock_guard<std::mutex> lk(owner->mu);
        if (owner->stage_free.size() < 16 && owner->stage_bytes + cap <= ((size_t)2 << 30)) {
            owner->stage_free.emplace_back(p, cap);
            owner->stage_bytes += cap;
        } else {
            (void)hipFree(p);
        }
    }
    hipError_t alloc(hbmpc_ctx* ctx, size_t bytes) {
        owner = ctx;
        if (bytes == 0) bytes = 1;
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            int best = -1;
            for (int i = 0; i < (int)ctx->stage_free.size(); ++i) {
                const size_t c = ctx->stage_free[i].second;
                if (c >= bytes && c <= 4 * bytes + (1u << 20) && (best < 0 || c < ctx->stage_free[best].second)) best = i;
            }
            if (best >= 0) {
                p = ctx->stage_free[best].first;
                cap = ctx->stage_free[best].second;
                ctx->stage_bytes -= cap;
                ctx->stage_free.erase(ctx->stage_free.begin() + best);
                return hipSuccess;
            }
        }
        cap = bytes < (1u << 16) ? (1u << 16) : bytes + bytes / 4;
        return hipMalloc(&p, cap);
    }
};


// ---- staging of one host-pointer call -------------------------------------------------------------------------
// Large calls: device buffers from the DevBuf pool + async copies on the context stream.  Small calls (everything
// fits one 2 MiB block) stage through pinned host memory that is mapped into the device: inputs are memcpy'd in,
// the kernels read and write that block over PCIe, outputs are memcpy'd out after the stream sync -- no DMA commands
// at all.  On this box one 512-byte hipMemcpy + sync costs 16-27 us and a kernel launch + sync 25 us, so a
// one-polynomial recover_secret (1 upload, 3 launches, 4 downloads) drops from 111 us to about 45 us
// (tools/time_small_calls.py).  Only plain loads/stores touch the mapped block (the kernels' atomics -- flag lists,
// counters -- live in device scratch; the host path derives its summary from the status bytes instead).
struct Stage {
    static constexpr size_t BLOCK = STAGE_PIN_BLOCK;
    hbmpc_ctx* ctx;
    bool mapped = false, finished = false;
    char* hblk = nullptr;  // host view of the block
    char* dblk = nullptr;  // device view
    size_t off = 0;
    std::deque<DevBuf> bufs;
    struct Out {
        void* dst;
        const void* src;
        size_t bytes;
    };
    std::vector<Out> outs;

    Stage(hbmpc_ctx* c, size_t total_bytes, size_t n_bufs) : ctx(c) {
        if (!c->zero_copy || total_bytes + 256 * n_bufs > BLOCK) return;
        {
            std::lock_guard<std::mutex> lk(c->mu);
            if (!c->pin_free.empty()) {
                hblk = (char*)c->pin_free.back();
                c->pin_free.pop_back();
            }
        }
        if (!hblk && hipHostMalloc((void**)&hblk, BLOCK, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            hblk = nullptr;
            return;  // no pinned memory to be had: the device-buffer path serves the call
        }
        if (hipHostGetDevicePointer((void**)&dblk, hblk, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(hblk);
            hblk = nullptr;
            return;
        }
        mapped = true;
    }
    ~Stage() {
        if (!hblk) return;
        if (!finished) (void)hipStreamSynchronize(ctx->stream);  // an error path: kernels may still be writing the block
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (ctx->pin_free.size() < 16) ctx->pin_free.push_back(hblk);
        else (void)hipHostFree(hblk);
    }
    Stage(const Stage&) = delete;
    Stage& operator=(const Stage&) = delete;

    hipError_t alloc(size_t bytes, void** p) {
        if (mapped) {
            *p = dblk + off;
            off += (bytes + 255) & ~(size_t)255;
            return off <= BLOCK ? hipSuccess : hipErrorOutOfMemory;  // cannot happen: sized in the constructor
        }
        bufs.emplace_back();
        hipError_t e = bufs.back().alloc(ctx, bytes);
        *p = bufs.back().p;
        return e;
    }
    hipError_t in(const void* src, size_t bytes, void** p) {
        hipError_t e = alloc(bytes, p);
        if (e != hipSuccess || bytes == 0) return e;
        if (mapped) {
            memcpy(hblk + ((char*)*p - dblk), src, bytes);
            return hipSuccess;
        }
        return hipMemcpyAsync(*p, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    }
    hipError_t out(void* dst, const void* p, size_t bytes) {  // p: a pointer returned by alloc()/in()
        if (!dst || bytes == 0) return hipSuccess;
        if (mapped) {
            outs.push_back({dst, hblk + ((const char*)p - dblk), bytes});
            return hipSuccess;
        }
        return hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t finish() {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        finished = true;
        if (e != hipSuccess) return e;
        for (const Out& o : outs) memcpy(o.dst, o.src, o.bytes);
        return hipSuccess;
    }
};
extern "C" ShareErrorCode hbmpc_set_small_call_staging(hbmpc_ctx* ctx, int zero_copy) {
    if (!ctx) return InvalidInput;
    ctx->zero_copy = zero_copy != 0;
    return ShareSuccess;
}

static ShareErrorCode eval_host(hbmpc_ctx* ctx, const void* x, size_t G, size_t n, size_t d, void* y) {
    if (!ctx) return InvalidInput;
    if (n <= d) return fail(ctx, InvalidInput, "number of shares must be greater than the degree");
    // the range checks of eval_dev, before any size arithmetic with n and d
    if (n == 0 || n > ((size_t)1 << 32)) return fail(ctx, NoSuitableDomain, "no radix-2 domain of that size");
    if (n > (1u << 20) || d > (1u << 20)) return fail(ctx, InvalidInput, "n, d beyond the supported range");
    if (G == 0) return ShareSuccess;
    if (!x || !y) return fail(ctx, InvalidInput, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t eb = ebytes(ctx);
    Stage st(ctx, G * (d + 1 + n) * eb, 2);
    void *dx, *dy;
    HIP_TRY(ctx, st.in(x, G * (d + 1) * eb, &dx));
    HIP_TRY(ctx, st.alloc(G * n * eb, &dy));
    ShareErrorCode rc = eval_dev(ctx, dx, G, n, d, dy, nullptr);
    if (rc != ShareSuccess) return rc;
    HIP_TRY(ctx, st.out(y, dy, G * n * eb));
    HIP_TRY(ctx, st.finish());
    return ShareSuccess;
}
// ---- seeded compute_shares: the random coefficients are generated on the device ("hbmpc-chacha20-v1") ---------
static ShareErrorCode fill_coeffs_dev(hbmpc_ctx* ctx, const uint8_t seed[32], const void* secrets, size_t B,
                                      uint64_t first_index, size_t d, void* coeffs_out, void* stream) {
    if (!ctx) return InvalidInput;
    if (!seed) return fail(ctx, InvalidInput, "null seed");
    if (d > (1u << 20)) return fail(ctx, InvalidInput, "degree beyond the supported range");
    if (B == 0) return ShareSuccess;
    if (!coeffs_out) return fail(ctx, InvalidInput, "null buffer");  // secrets may be null: they are drawn too
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t k[8];
    memcpy(k, seed, 32);  // little-endian words
    launch_fill_coeffs((int)(ebytes(ctx) / 4), k, (const uint32_t*)secrets, B, first_index, (int)(d + 1), (uint32_t*)coeffs_out,
                       pick(ctx, stream));
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
static ShareErrorCode compute_shares_seeded_dev(hbmpc_ctx* ctx, const uint8_t seed[32], const void* secrets, size_t B,
                                                uint64_t first_index, size_t n, size_t d, void* coeffs_ws, void* shares_out,
                                                void* stream) {
    if (ctx && n <= d) return fail(ctx, InvalidInput, "number of shares must be greater than the degree");
    ShareErrorCode rc = fill_coeffs_dev(ctx, seed, secrets, B, first_index, d, coeffs_ws, stream);
    if (rc != ShareSuccess) return rc;
    return eval_dev(ctx, coeffs_ws, B, n, d, shares_out, stream);
}
static ShareErrorCode compute_shares_seeded_host(hbmpc_ctx* ctx, const uint8_t seed[32], const void* secrets, size_t B,
                                                 uint64_t first_index, size_t n, size_t d, void* shares_out) {
    if (!ctx) return InvalidInput;
    if (n <= d) return fail(ctx, InvalidInput, "number of shares must be greater than the degree");
    if (B == 0) return ShareSuccess;
    if (!shares_out || !seed) return fail(ctx, InvalidInput, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t eb = ebytes(ctx);
    Stage st(ctx, B * (1 + d + 1 + n) * eb, 3);  // only the SECRETS cross the bus on the way in: 1/(d+1) of what hbmpc_compute_shares uploads
    void *ds = nullptr, *dc, *dy;
    if (secrets) HIP_TRY(ctx, st.in(secrets, B * eb, &ds));
    HIP_TRY(ctx, st.alloc(B * (d + 1) * eb, &dc));
    HIP_TRY(ctx, st.alloc(B * n * eb, &dy));
    ShareErrorCode rc = compute_shares_seeded_dev(ctx, seed, ds, B, first_index, n, d, dc, dy, nullptr);
    if (rc != ShareSuccess) return rc;
    HIP_TRY(ctx, st.out(shares_out, dy, B * n * eb));
    HIP_TRY(ctx, st.finish());
    return ShareSuccess;
}
#define TYPED_SEEDED(T, REQ, PFX)                                                                                          \
    extern "C" ShareErrorCode PFX##dev_fill_coeffs(hbmpc_ctx* ctx, const uint8_t seed[32], const T* secrets_dev, size_t B,  \
                                                   uint64_t first_index, size_t d, T* coeffs_out_dev, void* stream) {       \
        REQ(ctx);                                                                                                          \
        return fill_coeffs_dev(ctx, seed, secrets_dev, B, first_index, d, coeffs_out_dev, stream);                         \
    }                                                                                                                      \
    extern "C" ShareErrorCode PFX##dev_compute_shares_seeded(hbmpc_ctx* ctx, const uint8_t seed[32], const T* secrets_dev, \
                                                             size_t B, uint64_t first_index, size_t n, size_t d,           \
                                                             T* coeffs_ws_dev, T* shares_out_dev, void* stream) {          \
        REQ(ctx);                                                                                                          \
        return compute_shares_seeded_dev(ctx, seed, secrets_dev, B, first_index, n, d, coeffs_ws_dev, shares_out_dev, stream); \
    }                                                                                                                      \
    extern "C" ShareErrorCode PFX##compute_shares_seeded(hbmpc_ctx* ctx, const uint8_t seed[32], const T* secrets, size_t B, \
                                                         uint64_t first_index, size_t n, size_t d, T* shares_out) {        \
        REQ(ctx);                                                                                                          \
        return compute_shares_seeded_host(ctx, seed, secrets, B, first_index, n, d, shares_out);                           \
    }
TYPED_SEEDED(U256, REQ_FR, hbmpc_)
TYPED_SEEDED(uint64_t, REQ_GL, hbmpc_gl_)

extern "C" ShareErrorCode hbmpc_compute_shares(hbmpc_ctx* ctx, const U256* coeffs, size_t B, size_t n, size_t d,
                                               U256* shares_out) {
    REQ_FR(ctx);
    return eval_host(ctx, coeffs, B, n, d, shares_out);
}
extern "C" ShareErrorCode hbmpc_vandermonde_apply(hbmpc_ctx* ctx, const U256* x, size_t G, size_t n, size_t d,
                                                  U256* y_out) {
    REQ_FR(ctx);
    return eval_host(ctx, x, G, n, d, y_out);
}
extern "C" ShareErrorCode hbmpc_gl_compute_shares(hbmpc_ctx* ctx, const uint64_t* coeffs, size_t B, size_t n, size_t d,
                                                  uint64_t* shares_out) {
    REQ_GL(ctx);
    return eval_host(ctx, coeffs, B, n, d, shares_out);
}
extern "C" ShareErrorCode hbmpc_gl_vandermonde_apply(hbmpc_ctx* ctx, const uint64_t* x, size_t G, size_t n, size_t d,
                                                     uint64_t* y_out) {
    REQ_GL(ctx);
    return eval_host(ctx, x, G, n, d, y_out);
}
// common/share/mod.rs:31-45.  A constant table (rows [1, alpha_j, ..., alpha_j^d]), built where every other
// table of this library is built: on the host.
template <class H>
static ShareErrorCode make_vandermonde_t(hbmpc_ctx* ctx, size_t n, size_t d, void* v_out) {
    if (!ctx) return InvalidInput;
    if (n == 0) return ShareSuccess;
    if (!v_out) return fail(ctx, InvalidInput, "null buffer");
    if (domain_size(n) > ((size_t)1 << 32)) return fail(ctx, NoSuitableDomain, "no radix-2 domain of that size");
    const std::vector<H> el = domain_elements<H>(n, n);
    for (size_t j = 0; j < n; ++j) {
        H p = H::one();
        for (size_t k = 0; k <= d; ++k) {
            uint64_t c[4];
            p.to_canon(c);
            memcpy((uint8_t*)v_out + (j * (d + 1) + k) * H::EBYTES, c, H::EBYTES);
            p = p * el[j];
        }
    }
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_make_vandermonde(hbmpc_ctx* ctx, size_t n, size_t d, U256* v_out) {
    REQ_FR(ctx);
    return make_vandermonde_t<HFr>(ctx, n, d, v_out);
}
extern "C" ShareErrorCode hbmpc_gl_make_vandermonde(hbmpc_ctx* ctx, size_t n, size_t d, uint64_t* v_out) {
    REQ_GL(ctx);
    return make_vandermonde_t<HGl>(ctx, n, d, v_out);
}

// ---- element-wise ------------------------------------------------------------------------------
#define ELEM_PROLOGUE                                                           \
    if (!ctx) return InvalidInput;                                              \
    if (N == 0) return ShareSuccess;                                            \
    HIP_TRY(ctx, hipSetDevice(ctx->device));                                    \
    hipStream_t s = pick(ctx, stream);                                          \
    const unsigned grid = (unsigned)((N + 255) / 256);                          \
    (void)grid;

#define BY_IMPL(KERNEL, ...)                                                                         \
    do {                                                                                             \
        if (ctx->impl == IMPL_U29)                                                                   \
            hipLaunchKernelGGL((KERNEL<U29>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);             \
        else                                                                                         \
            hipLaunchKernelGGL((KERNEL<Sat32>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);           \
        HIP_TRY(ctx, hipGetLastError());                                                             \
    } while (0)

// the same with dim3(grid, P): party-batched kernels (blockIdx.y = party)
#define BY_FIELD_P(P, KERNEL, ...)                                                                   \
    do {                                                                                             \
        const dim3 gp(grid, (unsigned)(P));                                                          \
        if (ctx->impl == IMPL_U29)                                                                   \
            hipLaunchKernelGGL((KERNEL<U29>), gp, dim3(256), 0, s, __VA_ARGS__);                     \
        else if (ctx->impl == IMPL_SAT32)                                                            \
            hipLaunchKernelGGL((KERNEL<Sat32>), gp, dim3(256), 0, s, __VA_ARGS__);                   \
        else                                                                                         \
            hipLaunchKernelGGL((KERNEL<Gold>), gp, dim3(256), 0, s, __VA_ARGS__);                    \
        HIP_TRY(ctx, hipGetLastError());                                                             \
    } while (0)
#define BY_IMPL_P(P, KERNEL, ...)                                                                    \
    do {                                                                                             \
        const dim3 gp(grid, (unsigned)(P));                                                          \
        if (ctx->impl == IMPL_U29)                                                                   \
            hipLaunchKernelGGL((KERNEL<U29>), gp, dim3(256), 0, s, __VA_ARGS__);                     \
        else                                                                                         \
            hipLaunchKernelGGL((KERNEL<Sat32>), gp, dim3(256), 0, s, __VA_ARGS__);                   \
        HIP_TRY(ctx, hipGetLastError());                                                             \
    } while (0)
#define CHECK_PARTIES(P) do { if ((P) == 0 || (P) > 65535) return fail(ctx, InvalidInput, "parties must be in 1..65535"); } while (0)

#define BY_FIELD(KERNEL, ...)                                                                        \
    do {                                                                                             \
        if (ctx->impl == IMPL_U29)                                                                   \
            hipLaunchKernelGGL((KERNEL<U29>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);             \
        else if (ctx->impl == IMPL_SAT32)                                                            \
            hipLaunchKernelGGL((KERNEL<Sat32>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);           \
        else                                                                                         \
            hipLaunchKernelGGL((KERNEL<Gold>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);            \
        HIP_TRY(ctx, hipGetLastError());                                                             \
    } while (0)

#define W(p) ((const uint32_t*)(p))
#define WO(p) ((uint32_t*)(p))

static ShareErrorCode fr_op_any(hbmpc_ctx* ctx, int op, const void* a, const void* b, size_t N, void* out, void* stream) {
    ELEM_PROLOGUE
    if (op < 0 || op > 2) return fail(ctx, InvalidInput, "op must be 0 (add), 1 (sub) or 2 (mul)");
    const ElemConsts cs = elem_consts(ctx->impl);
    if (ctx->impl == IMPL_GOLD) {
        if (op == 0) hipLaunchKernelGGL((k_binop<Gold, OP_ADD>), dim3(grid), dim3(256), 0, s, W(a), W(b), N, cs, WO(out));
        if (op == 1) hipLaunchKernelGGL((k_binop<Gold, OP_SUB>), dim3(grid), dim3(256), 0, s, W(a), W(b), N, cs, WO(out));
        if (op == 2) hipLaunchKernelGGL((k_binop<Gold, OP_MUL>), dim3(grid), dim3(256), 0, s, W(a), W(b), N, cs, WO(out));
    } else if (ctx->impl == IMPL_U29) {
        if (op == 0) hipLaunchKernelGGL((k_binop<U29, OP_ADD>), dim3(grid), dim3(256), 0, s, W(a), W(b), N, cs, WO(out));
        if (op == 1) hipLaunchKernelGGL((k_binop<U29, OP_SUB>), dim3(grid), dim3(256), 0, s, W(a), W(b), N, cs, WO(out));
        if (op == 2) hipLaunchKernelGGL((k_binop<U29, OP_MUL>), dim3(grid), dim3(256), 0, s, W(a), W(b), N, cs, WO(out));
    } else {
        if (op == 0) hipLaunchKernelGGL((k_binop<Sat32, OP_ADD>), dim3(grid), dim3(256), 0, s, W(a), W(b), N, cs, WO(out));
        if (op == 1) hipLaunchKernelGGL((k_binop<Sat32, OP_SUB>), dim3(grid), dim3(256), 0, s, W(a), W(b), N, cs, WO(out));
        if (op == 2) hipLaunchKernelGGL((k_binop<Sat32, OP_MUL>), dim3(grid), dim3(256), 0, s, W(a), W(b), N, cs, WO(out));
    }
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
// scalar: ONE canonical element in HOST memory
static ShareErrorCode fr_op_scalar_any(hbmpc_ctx* ctx, int op, const void* a, const void* scalar, size_t N, void* out, void* stream) {
    if (ctx && !scalar) return fail(ctx, InvalidInput, "null scalar");
    ELEM_PROLOGUE
    if (op < 0 || op > 3) return fail(ctx, InvalidInput, "op must be 0 (a + s), 1 (a - s), 2 (a * s) or 3 (s - a)");
    ScalarArg sc;
    memset(&sc, 0, sizeof sc);
    memcpy(sc.w, scalar, ebytes(ctx));
    if (ctx->impl == IMPL_GOLD) {
        uint64_t v;
        memcpy(&v, scalar, 8);
        if (v >= HGl::P) return fail(ctx, InvalidInput, "scalar is not a canonical field element");
    } else {
        uint64_t v[4];
        memcpy(v, scalar, 32);
        if (HFr::geq(v)) return fail(ctx, InvalidInput, "scalar is not a canonical field element");
    }
    const ElemConsts cs = elem_consts(ctx->impl);
#define SCALAR_OPS(F)                                                                                                       \
    do {                                                                                                                    \
        if (op == 0) hipLaunchKernelGGL((k_scalarop<F, OP_ADD>), dim3(grid), dim3(256), 0, s, W(a), sc, N, cs, WO(out));    \
        if (op == 1) hipLaunchKernelGGL((k_scalarop<F, OP_SUB>), dim3(grid), dim3(256), 0, s, W(a), sc, N, cs, WO(out));    \
        if (op == 2) hipLaunchKernelGGL((k_scalarop<F, OP_MUL>), dim3(grid), dim3(256), 0, s, W(a), sc, N, cs, WO(out));    \
        if (op == 3) hipLaunchKernelGGL((k_scalarop<F, OP_RSUB>), dim3(grid), dim3(256), 0, s, W(a), sc, N, cs, WO(out));   \
    } while (0)
    if (ctx->impl == IMPL_GOLD) SCALAR_OPS(Gold);
    else if (ctx->impl == IMPL_U29) SCALAR_OPS(U29);
    else SCALAR_OPS(Sat32);
#undef SCALAR_OPS
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
static ShareErrorCode triple_local_any(hbmpc_ctx* ctx, const void* a, const void* b, const void* r2t, size_t N, void* out,
                                       void* stream) {
    ELEM_PROLOGUE
    const ElemConsts cs = elem_consts(ctx->impl);
    BY_FIELD(k_triple_local, W(a), W(b), W(r2t), N, cs, WO(out));
    return ShareSuccess;
}
// TripleGenNode::init_batch for `parties` parties at once: x = a b - r2t per element, then the Vandermonde encode of the
// chunks of d + 1 -- one launch where k_eval_fft1_triple covers the shape, otherwise the two separate ones through `tmp`
static ShareErrorCode triple_encode_any(hbmpc_ctx* ctx, const void* a, const void* b, const void* r2t, size_t G, size_t n, size_t d,
                                        size_t parties, void* tmp, void* y, void* stream) {
    if (!ctx) return InvalidInput;
    if (parties == 0 || parties > 65535) return fail(ctx, InvalidInput, "parties must be in 1..65535");
    if (n <= d) return fail(ctx, InvalidInput, "number of shares must be greater than the degree");
    if (n == 0 || n > ((size_t)1 << 32)) return fail(ctx, NoSuitableDomain, "no radix-2 domain of that size");
    if (n > (1u << 20) || d > (1u << 20)) return fail(ctx, InvalidInput, "n, d beyond the supported range");
    if (G == 0) return ShareSuccess;
    if (!a || !b || !r2t || !y) return fail(ctx, InvalidInput, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    const size_t size = domain_size(n), dp1 = d + 1;
    // small batches with a workspace: the product kernel + the wave-per-chunk evaluation are two short launches, the fused
    // kernel one long one (a lane walks ~12 k instructions for its chunk): 1100 triples x 16 parties, the whole triple
    // generation 0.043 ms against 0.057 ms
    const bool small_two = tmp && G * parties <= ctx->wide_max_chunks / 4;
    // Large batches on 9 .. 32 points: the products are computed inside the matrix-core encode with the points in pairs
    // (k_mfma_bfly<.., TRIPLE>, kernels_mfma_bfly.hpp) -- config 4's 16 parties x 381 300 chunks: 2.22 - 2.28 ms against
    // 2.53 - 2.55 of the fused FFT kernel on the same box (profiles/r03_mfma_bfly_triple.txt).  From 2^14 chunks over all parties: at the
    // reference node's batch of 4 096 chunks x 16 parties the whole TripleGen step takes 0.073 ms against 0.093 with the lane kernel, a tie
    // at 2^14 (profiles/r04_protocol_batch_sizes.txt)
    if (ctx->impl == IMPL_U29 && ctx->matrix_cores && ctx->mfma_bfly && !ctx->force_generic && dp1 >= 2 && dp1 <= MF_MAX_M && size >= 16 &&
        size <= 32 && n > size / 2 && G * parties >= ((size_t)1 << 14) && G * dp1 * 32 < ((size_t)1 << 32) && parties <= 65535) {
        const size_t half = size / 2;
        mf::MfmaRowsArgs ma;
        memset(&ma, 0, sizeof ma);
        const int nwg = ctx->mfma_wgs ? ctx->mfma_wgs : ctx->n_cus;
        if (half * mf_bfly_row_bytes(dp1) <= 160 * 1024 && mf::mf_plan_pairs((int)half, (int)half, nwg, &ma)) {
            const uint32_t* tab;
            std::array<size_t, 5> aux = {0, 0, 0, 0, 0};  // aux[0]: the table's words (0: the digit-sum bound could not be proved)
            // rows alpha_j^i R, R = 2^261: the kernel hands over (a b - r2t) / R (one Montgomery reduction, no conversions)
            ShareErrorCode rc = get_table(ctx, key("mfbflyR", {n, dp1}, ctx->impl), [&] {
                std::vector<HFr> el = domain_elements<HFr>(n, n);
                HFr R = HFr::one();
                const HFr two = HFr::from_u64(2);
                for (int i = 0; i < 261; ++i) R = R * two;
                std::vector<std::vector<HFr>> V(n, std::vector<HFr>(dp1));
                for (size_t j = 0; j < n; ++j) {
                    HFr p = R;
                    for (size_t k = 0; k < dp1; ++k) V[j][k] = p, p = p * el[j];
                }
                std::vector<uint32_t> tbl = build_mfma_bfly_table(V, dp1, half);
                aux[0] = tbl.size();
                return tbl;
            }, &tab, &aux);
            if (rc != ShareSuccess) return rc;
            ma.in = (const uint8_t*)a, ma.in_b = (const uint8_t*)b, ma.in_r = (const uint8_t*)r2t;
            ma.parties = (int)parties, ma.G = G, ma.in_chunk_major = 1, ma.nv = 0;
            ma.table = (const uint8_t*)tab, ma.half = (int)half, ma.nout = (int)n;
            ma.out = (uint8_t*)y, ma.out_party_major = 1, ma.out_stride = G;
            const int mi = (int)dp1;
            if (aux[0] != 0 && (launch_mfma_bfly_a(mi, ma, ctx->device, s) || launch_mfma_bfly_b(mi, ma, ctx->device, s) ||
                                launch_mfma_bfly_c(mi, ma, ctx->device, s) || launch_mfma_bfly_d(mi, ma, ctx->device, s))) {
                HIP_TRY(ctx, hipGetLastError());
                return ShareSuccess;
            }
        }
    }
    if (!small_two && ctx->impl == IMPL_U29 && size <= 16 && !ctx->force_generic) {
        const uint32_t* tw;
        ShareErrorCode rc = get_table(ctx, key("tw", {size}, ctx->impl), [&] { return build_twiddles<HFr>(size, ctx->impl); }, &tw);
        if (rc != ShareSuccess) return rc;
        const ElemConsts cs = elem_consts(ctx->impl);
        if (launch_fft1_triple(ilog2(size), (int)dp1, W(a), W(b), W(r2t), G, (int)n, tw, EvalOut{WO(y), 0, (unsigned)parties}, cs.r2, s)) {
            HIP_TRY(ctx, hipGetLastError());
            return ShareSuccess;
        }
    }
    if (ctx->impl == IMPL_GOLD && size <= 16 && !ctx->force_generic) {
        const uint32_t* tw;
        ShareErrorCode rc = get_table(ctx, key("tw", {size}, ctx->impl), [&] { return build_twiddles<HGl>(size, ctx->impl); }, &tw);
        if (rc != ShareSuccess) return rc;
        if (launch_fft1_triple_gold(ilog2(size), (int)dp1, W(a), W(b), W(r2t), G, (int)n, tw, EvalOut{WO(y), 0, (unsigned)parties}, s)) {
            HIP_TRY(ctx, hipGetLastError());
            return ShareSuccess;
        }
    }
    if (!tmp) return fail(ctx, InvalidInput, "no fused kernel for this shape: pass a workspace of parties * G * (d + 1) elements");
    ShareErrorCode rc = triple_local_any(ctx, a, b, r2t, parties * G * dp1, tmp, stream);
    if (rc != ShareSuccess) return rc;
    return eval_dev(ctx, tmp, G, n, d, y, stream, parties);
}
static ShareErrorCode triple_finalize_any(hbmpc_ctx* ctx, const void* rt, const void* opened, size_t N, void* c_out,
                                          void* stream, size_t parties = 1) {
    ELEM_PROLOGUE
    CHECK_PARTIES(parties);
    BY_FIELD_P(parties, k_triple_finalize, W(rt), W(opened), N, WO(c_out));
    return ShareSuccess;
}
static ShareErrorCode beaver_open_any(hbmpc_ctx* ctx, const void* a, const void* b, const void* x, const void* y, size_t N,
                                      void* d_sh, void* e_sh, void* stream) {
    ELEM_PROLOGUE
    BY_FIELD(k_beaver_open, W(a), W(b), W(x), W(y), N, WO(d_sh), WO(e_sh));
    return ShareSuccess;
}
static ShareErrorCode beaver_open_pair_any(hbmpc_ctx* ctx, const void* a, const void* b, const void* x, const void* y, size_t N,
                                           size_t parties, void* de, void* stream) {
    ELEM_PROLOGUE
    CHECK_PARTIES(parties);
    BY_FIELD_P(parties, k_beaver_open_pair, W(a), W(b), W(x), W(y), N, WO(de));
    return ShareSuccess;
}
static ShareErrorCode beaver_finalize_any(hbmpc_ctx* ctx, const void* c, const void* x, const void* y, const void* d,
                                          const void* e, size_t N, void* z, void* stream, size_t parties = 1) {
    ELEM_PROLOGUE
    CHECK_PARTIES(parties);
    const ElemConsts cs = elem_consts(ctx->impl);
    // the kernel loops over the parties (large N: the public operands are converted once per element) or spreads them
    // over gridDim.y (small N: latency)
    BY_FIELD_P(N >= ((size_t)1 << 16) ? 1 : parties, k_beaver_finalize, W(c), W(x), W(y), W(d), W(e), N, cs, WO(z), (unsigned)parties);
    return ShareSuccess;
}
#define TYPED_PAIR(T, REQ, PFX)                                                                                          \
    extern "C" ShareErrorCode PFX##dev_fr_op(hbmpc_ctx* ctx, int op, const T* a, const T* b, size_t N, T* out,           \
                                             void* stream) {                                                             \
        REQ(ctx);                                                                                                        \
        return fr_op_any(ctx, op, a, b, N, out, stream);                                                                 \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##dev_fr_op_scalar(hbmpc_ctx* ctx, int op, const T* a, const T* scalar_host, size_t N,  \
                                                    T* out, void* stream) {                                              \
        REQ(ctx);                                                                                                        \
        return fr_op_scalar_any(ctx, op, a, scalar_host, N, out, stream);                                                \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##dev_triple_encode_parties(hbmpc_ctx* ctx, const T* a, const T* b, const T* r2t,       \
                                                             size_t G, size_t n, size_t d, size_t parties, T* tmp,        \
                                                             T* y_out, void* stream) {                                   \
        REQ(ctx);                                                                                                        \
        return triple_encode_any(ctx, a, b, r2t, G, n, d, parties, tmp, y_out, stream);                                  \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##dev_triple_local(hbmpc_ctx* ctx, const T* a, const T* b, const T* r2t, size_t N,      \
                                                    T* out, void* stream) {                                              \
        REQ(ctx);                                                                                                        \
        return triple_local_any(ctx, a, b, r2t, N, out, stream);                                                         \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##dev_triple_finalize(hbmpc_ctx* ctx, const T* rt, const T* opened, size_t N, T* c_out, \
                                                       void* stream) {                                                   \
        REQ(ctx);                                                                                                        \
        return triple_finalize_any(ctx, rt, opened, N, c_out, stream);                                                   \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##dev_beaver_open_shares(hbmpc_ctx* ctx, const T* a, const T* b, const T* x, const T* y, \
                                                          size_t N, T* d_sh, T* e_sh, void* stream) {                    \
        REQ(ctx);                                                                                                        \
        return beaver_open_any(ctx, a, b, x, y, N, d_sh, e_sh, stream);                                                  \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##dev_beaver_finalize(hbmpc_ctx* ctx, const T* c, const T* x, const T* y, const T* d,   \
                                                       const T* e, size_t N, T* z, void* stream) {                       \
        REQ(ctx);                                                                                                        \
        return beaver_finalize_any(ctx, c, x, y, d, e, N, z, stream);                                                    \
    }
TYPED_PAIR(U256, REQ_FR, hbmpc_)
TYPED_PAIR(uint64_t, REQ_GL, hbmpc_gl_)
#define TYPED_PARTIES(T, REQ, PFX)                                                                                       \
    extern "C" ShareErrorCode PFX##dev_beaver_open_shares_paired(hbmpc_ctx* ctx, const T* a, const T* b, const T* x,     \
                                                                 const T* y, size_t N, size_t parties, T* de_sh_out,     \
                                                                 void* stream) {                                         \
        REQ(ctx);                                                                                                        \
        return beaver_open_pair_any(ctx, a, b, x, y, N, parties, de_sh_out, stream);                                     \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##dev_triple_finalize_parties(hbmpc_ctx* ctx, const T* rt, const T* opened, size_t N,   \
                                                               size_t parties, T* c_out, void* stream) {                 \
        REQ(ctx);                                                                                                        \
        return triple_finalize_any(ctx, rt, opened, N, c_out, stream, parties);                                          \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##dev_beaver_finalize_parties(hbmpc_ctx* ctx, const T* c, const T* x, const T* y,       \
                                                               const T* d, const T* e, size_t N, size_t parties, T* z,   \
                                                               void* stream) {                                           \
        REQ(ctx);                                                                                                        \
        return beaver_finalize_any(ctx, c, x, y, d, e, N, z, stream, parties);                                           \
    }
TYPED_PARTIES(U256, REQ_FR, hbmpc_)
TYPED_PARTIES(uint64_t, REQ_GL, hbmpc_gl_)
static ShareErrorCode truncpr_rdash_impl(hbmpc_ctx* ctx, const U256* r_bits, size_t m, size_t N, size_t parties, U256* r_dash,
                                         void* stream) {
    REQ_FR(ctx);
    ELEM_PROLOGUE
    CHECK_PARTIES(parties);
    if (m > 4096) return fail(ctx, InvalidInput, "m beyond the supported range");
    const uint32_t* pow2;
    const int impl = ctx->impl;
    ShareErrorCode rc = get_table(ctx, key("pow2", {m}, impl), [&] { return build_pow2(m, impl); }, &pow2);
    if (rc != ShareSuccess) return rc;
    BY_IMPL_P(parties, k_truncpr_rdash, W(r_bits), (int)m, N, pow2, WO(r_dash));
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_truncpr_rdash(hbmpc_ctx* ctx, const U256* r_bits, size_t m, size_t N,
                                                  U256* r_dash, void* stream) {
    return truncpr_rdash_impl(ctx, r_bits, m, N, 1, r_dash, stream);
}
extern "C" ShareErrorCode hbmpc_dev_truncpr_rdash_parties(hbmpc_ctx* ctx, const U256* r_bits, size_t m, size_t N,
                                                          size_t parties, U256* r_dash, void* stream) {
    return truncpr_rdash_impl(ctx, r_bits, m, N, parties, r_dash, stream);
}
extern "C" ShareErrorCode hbmpc_dev_truncpr_open_share(hbmpc_ctx* ctx, const U256* a, const U256* r_dash,
                                                       const U256* r_int, size_t k, size_t m, size_t N, U256* open_out,
                                                       void* stream) {
    REQ_FR(ctx);
    if (ctx && k == 0) return fail(ctx, InvalidInput, "k must be >= 1 (2^(k-1))");
    ELEM_PROLOGUE
    const HFr two = HFr::from_u64(2);
    const HFr p2m = two.pow_u64(m), p2k = two.pow_u64(k - 1);
    const ElemConsts cs = elem_consts(ctx->impl, &p2m, &p2k);
    BY_IMPL(k_truncpr_open, W(a), W(r_dash), W(r_int), N, cs, WO(open_out));
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_fpmul_middle(hbmpc_ctx* ctx, const U256* c, const U256* x, const U256* y, const U256* d,
                                                 const U256* e, const U256* r_bits, const U256* r_int, size_t k, size_t m, size_t N,
                                                 size_t parties, U256* z_out, U256* r_dash_out, U256* open_out, void* stream) {
    REQ_FR(ctx);
    if (ctx && k == 0) return fail(ctx, InvalidInput, "k must be >= 1 (2^(k-1))");
    ELEM_PROLOGUE
    CHECK_PARTIES(parties);
    if (m > 4096) return fail(ctx, InvalidInput, "m beyond the supported range");
    if (!c || !x || !y || !d || !e || !r_int || !z_out || !r_dash_out || !open_out || (m && !r_bits)) return fail(ctx, InvalidInput, "null buffer");
    const uint32_t* pow2;
    const int impl = ctx->impl;
    ShareErrorCode rc = get_table(ctx, key("pow2", {m}, impl), [&] { return build_pow2(m, impl); }, &pow2);
    if (rc != ShareSuccess) return rc;
    const HFr two = HFr::from_u64(2);
    const HFr p2m = two.pow_u64(m), p2k = two.pow_u64(k - 1);
    const ElemConsts cs = elem_consts(impl, &p2m, &p2k);
    BY_IMPL_P(N >= ((size_t)1 << 16) ? 1 : parties, k_fpmul_middle, W(c), W(x), W(y), W(d), W(e), W(r_bits), W(r_int), (int)m, N, cs, pow2,
              WO(z_out), WO(r_dash_out), WO(open_out), (unsigned)parties);
    return ShareSuccess;
}
static ShareErrorCode truncpr_finalize_impl(hbmpc_ctx* ctx, const U256* a, const U256* r_dash, const U256* c_open, size_t m,
                                            size_t N, size_t parties, U256* d_out, void* stream) {
    REQ_FR(ctx);
    // fpmul/mod.rs:381-406 indexes bytes[m/8] when m % 8 != 0: out of bounds (a panic) from m = 257 on
    if (ctx && m % 8 != 0 && m / 8 >= 32) return fail(ctx, InvalidInput, "m: bytes[m/8] out of bounds in the reference");
    ELEM_PROLOGUE
    CHECK_PARTIES(parties);
    const HFr inv = inv_pow2(ctx, m);
    const ElemConsts cs = elem_consts(ctx->impl, &inv);
    BY_IMPL_P(parties, k_truncpr_finalize, W(a), W(r_dash), W(c_open), (int)(m > 256 ? 256 : m), N, cs, WO(d_out));
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_truncpr_finalize(hbmpc_ctx* ctx, const U256* a, const U256* r_dash,
                                                     const U256* c_open, size_t m, size_t N, U256* d_out,
                                                     void* stream) {
    return truncpr_finalize_impl(ctx, a, r_dash, c_open, m, N, 1, d_out, stream);
}
extern "C" ShareErrorCode hbmpc_dev_truncpr_finalize_parties(hbmpc_ctx* ctx, const U256* a, const U256* r_dash,
                                                             const U256* c_open, size_t m, size_t N, size_t parties,
                                                             U256* d_out, void* stream) {
    return truncpr_finalize_impl(ctx, a, r_dash, c_open, m, N, parties, d_out, stream);
}
extern "C" ShareErrorCode hbmpc_dev_modmul_ubench(hbmpc_ctx* ctx, U256* out_dev, size_t threads, uint32_t iters,
                                                  void* stream) {
    REQ_FR(ctx);
    size_t N = threads;
    ELEM_PROLOGUE
    if (threads % 256) return fail(ctx, InvalidInput, "threads must be a multiple of 256");
    const ElemConsts cs = elem_consts(ctx->impl);
    BY_IMPL(k_modmul_ubench, WO(out_dev), iters, cs);
    return ShareSuccess;
}

extern "C" ShareErrorCode hbmpc_dev_traffic_ubench(hbmpc_ctx* ctx, const U256* x_dev, size_t G, size_t m, U256* y_dev, size_t n, void* stream) {
    REQ_FR(ctx);
    if (!x_dev || !y_dev || G == 0 || m == 0 || n == 0 || m > 64 || n > 256) return fail(ctx, InvalidInput, "null buffer or shape out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_traffic_ubench, dim3((unsigned)ctx->n_cus), dim3(768), 0, pick(ctx, stream), (const uint4*)x_dev, G, (int)m, (uint4*)y_dev, (int)n);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}

// host wrappers for the element-wise calls: n_in inputs of N elements (first input may be m*N), n_out outputs
template <class Fn>
static ShareErrorCode elem_host(hbmpc_ctx* ctx, std::initializer_list<std::pair<const void*, size_t>> ins,
                                std::initializer_list<std::pair<void*, size_t>> outs, Fn fn) {
    if (!ctx) return InvalidInput;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t eb = ebytes(ctx);
    size_t total = 0;
    for (auto& in : ins) total += in.second * eb;
    for (auto& o : outs) total += o.second * eb;
    Stage st(ctx, total, ins.size() + outs.size());
    std::vector<const void*> pi;
    std::vector<void*> po;
    for (auto& in : ins) {
        if (in.second && !in.first) return fail(ctx, InvalidInput, "null buffer");
        void* p = nullptr;
        HIP_TRY(ctx, st.in(in.first, in.second * eb, &p));
        pi.push_back(p);
    }
    for (auto& o : outs) {
        if (o.second && !o.first) return fail(ctx, InvalidInput, "null buffer");
        void* p = nullptr;
        HIP_TRY(ctx, st.alloc(o.second * eb, &p));
        po.push_back(p);
    }
    ShareErrorCode rc = fn(pi, po);
    if (rc != ShareSuccess) return rc;
    size_t k = 0;
    for (auto& o : outs) HIP_TRY(ctx, st.out(o.first, po[k++], o.second * eb));
    HIP_TRY(ctx, st.finish());
    return ShareSuccess;
}
typedef std::vector<const void*> VI;
typedef std::vector<void*> VO;
#define CU(p) ((const U256*)(p))
#define MU(p) ((U256*)(p))

#define TYPED_HOST(T, REQ, PFX)                                                                                          \
    extern "C" ShareErrorCode PFX##fr_op(hbmpc_ctx* ctx, int op, const T* a, const T* b, size_t N, T* out) {             \
        REQ(ctx);                                                                                                        \
        return elem_host(ctx, {{a, N}, {b, N}}, {{out, N}},                                                              \
                         [&](VI& i, VO& o) { return fr_op_any(ctx, op, i[0], i[1], N, o[0], nullptr); });                \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##fr_op_scalar(hbmpc_ctx* ctx, int op, const T* a, const T* scalar, size_t N, T* out) {  \
        REQ(ctx);                                                                                                        \
        return elem_host(ctx, {{a, N}}, {{out, N}},                                                                      \
                         [&](VI& i, VO& o) { return fr_op_scalar_any(ctx, op, i[0], scalar, N, o[0], nullptr); });       \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##triple_local(hbmpc_ctx* ctx, const T* a, const T* b, const T* r2t, size_t N, T* out) { \
        REQ(ctx);                                                                                                        \
        return elem_host(ctx, {{a, N}, {b, N}, {r2t, N}}, {{out, N}},                                                    \
                         [&](VI& i, VO& o) { return triple_local_any(ctx, i[0], i[1], i[2], N, o[0], nullptr); });       \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##triple_finalize(hbmpc_ctx* ctx, const T* rt, const T* opened, size_t N, T* c_out) {   \
        REQ(ctx);                                                                                                        \
        return elem_host(ctx, {{rt, N}, {opened, N}}, {{c_out, N}},                                                      \
                         [&](VI& i, VO& o) { return triple_finalize_any(ctx, i[0], i[1], N, o[0], nullptr); });          \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##beaver_open_shares(hbmpc_ctx* ctx, const T* a, const T* b, const T* x, const T* y,    \
                                                      size_t N, T* d_sh, T* e_sh) {                                      \
        REQ(ctx);                                                                                                        \
        return elem_host(ctx, {{a, N}, {b, N}, {x, N}, {y, N}}, {{d_sh, N}, {e_sh, N}}, [&](VI& i, VO& o) {              \
            return beaver_open_any(ctx, i[0], i[1], i[2], i[3], N, o[0], o[1], nullptr);                                 \
        });                                                                                                              \
    }                                                                                                                    \
    extern "C" ShareErrorCode PFX##beaver_finalize(hbmpc_ctx* ctx, const T* c, const T* x, const T* y, const T* d,       \
                                                   const T* e, size_t N, T* z) {                                         \
        REQ(ctx);                                                                                                        \
        return elem_host(ctx, {{c, N}, {x, N}, {y, N}, {d, N}, {e, N}}, {{z, N}}, [&](VI& i, VO& o) {                    \
            return beaver_finalize_any(ctx, i[0], i[1], i[2], i[3], i[4], N, o[0], nullptr);                             \
        });                                                                                                              \
    }
TYPED_HOST(U256, REQ_FR, hbmpc_)
TYPED_HOST(uint64_t, REQ_GL, hbmpc_gl_)
extern "C" ShareErrorCode hbmpc_truncpr_rdash(hbmpc_ctx* ctx, const U256* r_bits, size_t m, size_t N, U256* r_dash) {
    REQ_FR(ctx);
    return elem_host(ctx, {{r_bits, m * N}}, {{r_dash, N}},
                     [&](VI& i, VO& o) { return hbmpc_dev_truncpr_rdash(ctx, CU(i[0]), m, N, MU(o[0]), nullptr); });
}
extern "C" ShareErrorCode hbmpc_truncpr_open_share(hbmpc_ctx* ctx, const U256* a, const U256* r_dash,
                                                   const U256* r_int, size_t k, size_t m, size_t N, U256* open_out) {
    REQ_FR(ctx);
    return elem_host(ctx, {{a, N}, {r_dash, N}, {r_int, N}}, {{open_out, N}}, [&](VI& i, VO& o) {
        return hbmpc_dev_truncpr_open_share(ctx, CU(i[0]), CU(i[1]), CU(i[2]), k, m, N, MU(o[0]), nullptr);
    });
}
extern "C" ShareErrorCode hbmpc_truncpr_finalize(hbmpc_ctx* ctx, const U256* a, const U256* r_dash,
                                                 const U256* c_open, size_t m, size_t N, U256* d_out) {
    REQ_FR(ctx);
    return elem_host(ctx, {{a, N}, {r_dash, N}, {c_open, N}}, {{d_out, N}}, [&](VI& i, VO& o) {
        return hbmpc_dev_truncpr_finalize(ctx, CU(i[0]), CU(i[1]), CU(i[2]), m, N, MU(o[0]), nullptr);
    });
}

// ---- wire codec ---------------------------------------------------
extern "C" ShareErrorCode hbmpc_dev_pack_fvec(hbmpc_ctx* ctx, const U256* rows_dev, size_t row_stride, size_t G,
                                              size_t n_rows, void* payloads_dev, size_t payload_stride_bytes,
                                              void* stream) {
    REQ_FR(ctx);
    if (!ctx) return InvalidInput;
    if (n_rows == 0) return ShareSuccess;
    if (!rows_dev || !payloads_dev) return fail(ctx, InvalidInput, "null buffer");
    if (payload_stride_bytes % 8 || payload_stride_bytes < 8 + 32 * G || ((uintptr_t)payloads_dev & 7) || row_stride < G)
        return fail(ctx, InvalidInput, "payload stride must be 8-byte aligned and >= 8 + 32 G; row_stride >= G");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    launch_pack_fvec((const uint64_t*)rows_dev, row_stride, G, n_rows, (uint64_t*)payloads_dev, payload_stride_bytes / 8,
                     pick(ctx, stream));
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_unpack_fvec(hbmpc_ctx* ctx, const void* payloads_dev, size_t payload_stride_bytes,
                                                size_t payload_bytes, size_t G, size_t n_rows, U256* rows_dev,
                                                size_t row_stride, uint32_t* status_dev, void* stream) {
    REQ_FR(ctx);
    if (!ctx) return InvalidInput;
    if (n_rows == 0) return ShareSuccess;
    if (!rows_dev || !payloads_dev || !status_dev) return fail(ctx, InvalidInput, "null buffer");
    if (payload_bytes < 8 || payload_bytes < 8 + 32 * G)  // common/utils.rs:7-9; a short read fails every element
        return fail(ctx, InvalidInput, "payload shorter than its length prefix requires");
    if (payload_stride_bytes % 8 || payload_stride_bytes < payload_bytes || ((uintptr_t)payloads_dev & 7) || row_stride < G)
        return fail(ctx, InvalidInput, "payload stride must be 8-byte aligned and >= payload_bytes; row_stride >= G");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    HIP_TRY(ctx, hipMemsetAsync(status_dev, 0, n_rows * 4, s));
    launch_unpack_fvec((const uint64_t*)payloads_dev, payload_stride_bytes / 8, G, n_rows, (uint64_t*)rows_dev, row_stride,
                       status_dev, s);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
// In-place wire path (no pack / unpack pass): see include/hbmpc_hip.h
extern "C" ShareErrorCode hbmpc_dev_vandermonde_apply_strided(hbmpc_ctx* ctx, const U256* x_dev, size_t G, size_t n, size_t d,
                                                              U256* y_out_dev, size_t y_row_stride, void* stream) {
    REQ_FR(ctx);
    return eval_dev(ctx, x_dev, G, n, d, y_out_dev, stream, 1, y_row_stride);
}
extern "C" ShareErrorCode hbmpc_gl_dev_vandermonde_apply_strided(hbmpc_ctx* ctx, const uint64_t* x_dev, size_t G, size_t n,
                                                                 size_t d, uint64_t* y_out_dev, size_t y_row_stride,
                                                                 void* stream) {
    REQ_GL(ctx);
    return eval_dev(ctx, x_dev, G, n, d, y_out_dev, stream, 1, y_row_stride);
}
// x given as d + 1 rows (see include/hbmpc_hip.h): the point-pair matrix-core kernel reads them in place; every other
// shape goes through the workspace (transpose, then the chunk-major encode)
static ShareErrorCode eval_rows_any(hbmpc_ctx* ctx, const void* x_rows, size_t x_row_stride, size_t G, size_t n, size_t d, void* tmp,
                                    void* y, void* stream, const ListSpec* lists = nullptr) {
    if (!ctx) return InvalidInput;
    if (n <= d) return fail(ctx, InvalidInput, "number of shares must be greater than the degree");
    if (n == 0 || n > ((size_t)1 << 32)) return fail(ctx, NoSuitableDomain, "no radix-2 domain of that size");
    if (n > (1u << 20) || d > (1u << 20)) return fail(ctx, InvalidInput, "n, d beyond the supported range");
    if (x_row_stride < G) return fail(ctx, InvalidInput, "input row stride must be >= G");
    if (lists) {
        if (lists->rows == 0 || lists->row0 + lists->rows > n || lists->K == 0 || G % lists->K != 0 || lists->n_slices == 0 || lists->n_slices > 2 ||
            !lists->slices)
            return fail(ctx, InvalidInput, "list rows / slices out of range");
        for (size_t k = 0; k < lists->n_slices; ++k) {
            const hbmpc_list_slice& sl = lists->slices[k];
            if (!sl.dst_dev || sl.count == 0 || sl.k0 + sl.count > lists->K || sl.party_stride < sl.count * lists->rows)
                return fail(ctx, InvalidInput, "list slice out of range");
        }
        if (lists->n_slices == 2 && lists->slices[0].k0 + lists->slices[0].count > lists->slices[1].k0)
            return fail(ctx, InvalidInput, "list slices must be in ascending, disjoint ranges of batch elements");
    }
    if (G == 0) return ShareSuccess;
    if (!x_rows || !y) return fail(ctx, InvalidInput, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    const size_t size = domain_size(n), dp1 = d + 1;
    ShareErrorCode rc_mf = ShareSuccess;
    const bool mf_shape = ctx->impl == IMPL_U29 && ctx->matrix_cores && ctx->mfma_bfly && !ctx->force_generic && dp1 >= 2 && dp1 <= MF_BFLY_MAX_M &&
                          size >= 8 && n <= 255 && (G + 31) / 32 > (size_t)(ctx->mfma_wgs ? ctx->mfma_wgs : ctx->n_cus) * 2 && G * 32 < ((size_t)1 << 32);
    if (lists && lists->others && (lists->rows >= n || (G / lists->K) * (n - lists->rows) * lists->K * 32 >= ((size_t)1 << 32)))
        return fail(ctx, InvalidInput, "party-major other rows: needs a row outside the lists and less than 4 GiB of them");
    // the list rows straight from the kernel that computes them (k_mfma_bfly<.., LISTS>): one role, G < 2^32 chunks
    if (lists && mf_shape && ctx->list_rows_in_kernel && try_mfma_eval(ctx, (const uint32_t*)x_rows, G, n, dp1, EvalOut{(uint32_t*)y, 0, 1}, s, &rc_mf, x_row_stride, lists)) {
        if (rc_mf != ShareSuccess) return rc_mf;
        HIP_TRY(ctx, hipGetLastError());
        return ShareSuccess;
    }
    // Goldilocks, n inputs as rows, a domain of 4 .. 16 points: the single-pass lane kernel reads the rows in place and writes the lists and the
    // party-major rows itself (k_eval_fft1_mix) -- one launch instead of a transpose either side of the encode
    // (over Fr the same kernel on domains of 4 and 8 points -- 3 .. 8 parties -- where the matrix-core list kernel above did not take the call)
    const bool mix_gl = is_gold(ctx) && size <= 16, mix_fr = ctx->impl == IMPL_U29 && size <= 8;
    if (lists && (mix_gl || mix_fr) && ctx->list_rows_in_kernel && !ctx->force_generic && dp1 == n && n >= 3) {
        const uint32_t* tw;
        const ShareErrorCode rc = get_table(ctx, key("tw", {size}, ctx->impl), [&] {
            return mix_gl ? build_twiddles<HGl>(size, ctx->impl) : build_twiddles<HFr>(size, ctx->impl);
        }, &tw);
        if (rc != ShareSuccess) return rc;
        MixOut o;
        memset(&o, 0, sizeof o);
        o.y = (uint32_t*)y, o.others = (uint32_t*)lists->others, o.K = lists->K, o.row0 = (int)lists->row0, o.rows = (int)lists->rows;
        for (size_t k = 0; k < lists->n_slices; ++k)
            o.list[k] = MixOut::Slice{(uint32_t*)lists->slices[k].dst_dev, lists->slices[k].party_stride, lists->slices[k].k0, lists->slices[k].count};
        if (mix_gl ? launch_gold_fft1_mix(ilog2(size), (int)dp1, (const uint32_t*)x_rows, x_row_stride, G, (int)n, tw, o, s)
                   : launch_fft1_mix_lo(ilog2(size), (int)dp1, (const uint32_t*)x_rows, x_row_stride, G, (int)n, tw, o, s)) {
            HIP_TRY(ctx, hipGetLastError());
            return ShareSuccess;
        }
    }
    auto copy_lists = [&]() -> ShareErrorCode {  // every row is in y[row][G]: the list rows are copied out per slice
        if (!lists) return ShareSuccess;
        const size_t parties = G / lists->K;
        if (lists->others) {  // and the other rows into their party-major place
            launch_rows_party_major(is_gold(ctx) ? 1 : 4, (const uint64_t*)y, G, lists->K, (int)lists->row0, (int)lists->rows, (int)(n - lists->rows),
                                    (uint64_t*)lists->others, s);
            HIP_TRY(ctx, hipGetLastError());
        }
        for (size_t k = 0; k < lists->n_slices; ++k) {
            const hbmpc_list_slice& sl = lists->slices[k];
            const ShareErrorCode rc = hbmpc_dev_transpose(ctx, (const uint8_t*)y + (lists->row0 * G + sl.k0) * ebytes(ctx), lists->rows, sl.count, G,
                                                          sl.dst_dev, lists->rows, parties, lists->K, sl.party_stride, stream);
            if (rc != ShareSuccess) return rc;
        }
        return ShareSuccess;
    };
    if (mf_shape && try_mfma_eval(ctx, (const uint32_t*)x_rows, G, n, dp1, EvalOut{(uint32_t*)y, 0, 1}, s, &rc_mf, x_row_stride)) {
        if (rc_mf != ShareSuccess) return rc_mf;
        HIP_TRY(ctx, hipGetLastError());
        return copy_lists();
    }
    if (!tmp) return fail(ctx, InvalidInput, "no kernel reads this shape from rows: pass a workspace of G * (d + 1) elements");
    if ((dp1 + 15) / 16 > 65535) return fail(ctx, InvalidInput, "d beyond the launch grid");
    launch_transpose(is_gold(ctx) ? 1 : 4, (const uint64_t*)x_rows, dp1, G, x_row_stride, (uint64_t*)tmp, dp1, 1, 0, 0, s);
    HIP_TRY(ctx, hipGetLastError());
    const ShareErrorCode rc = eval_dev(ctx, tmp, G, n, d, y, stream);
    return rc != ShareSuccess ? rc : copy_lists();
}
extern "C" ShareErrorCode hbmpc_dev_vandermonde_apply_rows(hbmpc_ctx* ctx, const U256* x_rows_dev, size_t x_row_stride, size_t G, size_t n,
                                                           size_t d, U256* tmp_dev, U256* y_out_dev, void* stream) {
    REQ_FR(ctx);
    return eval_rows_any(ctx, x_rows_dev, x_row_stride, G, n, d, tmp_dev, y_out_dev, stream);
}
extern "C" ShareErrorCode hbmpc_gl_dev_vandermonde_apply_rows(hbmpc_ctx* ctx, const uint64_t* x_rows_dev, size_t x_row_stride, size_t G,
                                                              size_t n, size_t d, uint64_t* tmp_dev, uint64_t* y_out_dev, void* stream) {
    REQ_GL(ctx);
    return eval_rows_any(ctx, x_rows_dev, x_row_stride, G, n, d, tmp_dev, y_out_dev, stream);
}
static ShareErrorCode eval_rows_lists_any(hbmpc_ctx* ctx, const void* x_rows, size_t x_row_stride, size_t G, size_t n, size_t d, void* tmp, void* y,
                                          size_t list_row0, size_t list_rows, size_t K, const hbmpc_list_slice* slices, size_t n_slices, void* stream,
                                          void* others = nullptr, bool want_others = false) {
    if (ctx && want_others && !others) return fail(ctx, InvalidInput, "null buffer");
    const ListSpec ls{list_row0, list_rows, K, slices, n_slices, others};
    return eval_rows_any(ctx, x_rows, x_row_stride, G, n, d, tmp, y, stream, &ls);
}
// would the mixing step write its lists (and party-major other rows) from the kernel that computes them?  (otherwise: all rows to y, then copies)
extern "C" ShareErrorCode hbmpc_dev_apply_rows_lists_in_kernel(hbmpc_ctx* ctx, size_t G, size_t n, size_t d, int* yes_out) {
    if (!ctx || !yes_out) return InvalidInput;
    const size_t size = domain_size(n), dp1 = d + 1;
    if (is_gold(ctx)) {  // k_eval_fft1_mix: the n x n mixing step on domains of up to 16 points, any batch size
        *yes_out = ctx->list_rows_in_kernel && !ctx->force_generic && dp1 == n && size <= 16 && n >= 3;
        return ShareSuccess;
    }
    if (ctx->impl == IMPL_U29 && ctx->list_rows_in_kernel && !ctx->force_generic && dp1 == n && size <= 8 && n >= 3) {  // the same kernel over Fr, 3 .. 8 parties
        *yes_out = 1;
        return ShareSuccess;
    }
    *yes_out = ctx->impl == IMPL_U29 && ctx->matrix_cores && ctx->mfma_bfly && !ctx->force_generic && ctx->list_rows_in_kernel && dp1 >= 2 &&
               dp1 >= 5 && dp1 <= MF_BFLY_MAX_M && size >= 8 && size <= 16 && n > size / 2 &&  // tu_mfma_bfly.inc: launch_lists
               (G + 31) / 32 > (size_t)(ctx->mfma_wgs ? ctx->mfma_wgs : ctx->n_cus) * 2 && G * 32 < ((size_t)1 << 32);
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_vandermonde_apply_rows_split(hbmpc_ctx* ctx, const U256* x_rows_dev, size_t x_row_stride, size_t G, size_t n, size_t d,
                                                                 U256* tmp_dev, U256* y_out_dev, size_t list_row0, size_t list_rows, size_t K,
                                                                 const hbmpc_list_slice* slices, size_t n_slices, U256* others_out_dev, void* stream) {
    REQ_FR(ctx);
    return eval_rows_lists_any(ctx, x_rows_dev, x_row_stride, G, n, d, tmp_dev, y_out_dev, list_row0, list_rows, K, slices, n_slices, stream, others_out_dev, true);
}
extern "C" ShareErrorCode hbmpc_gl_dev_vandermonde_apply_rows_split(hbmpc_ctx* ctx, const uint64_t* x_rows_dev, size_t x_row_stride, size_t G, size_t n,
                                                                    size_t d, uint64_t* tmp_dev, uint64_t* y_out_dev, size_t list_row0, size_t list_rows,
                                                                    size_t K, const hbmpc_list_slice* slices, size_t n_slices, uint64_t* others_out_dev,
                                                                    void* stream) {
    REQ_GL(ctx);
    return eval_rows_lists_any(ctx, x_rows_dev, x_row_stride, G, n, d, tmp_dev, y_out_dev, list_row0, list_rows, K, slices, n_slices, stream, others_out_dev, true);
}
extern "C" ShareErrorCode hbmpc_dev_vandermonde_apply_rows_lists(hbmpc_ctx* ctx, const U256* x_rows_dev, size_t x_row_stride, size_t G, size_t n,
                                                                 size_t d, U256* tmp_dev, U256* y_out_dev, size_t list_row0, size_t list_rows,
                                                                 size_t K, const hbmpc_list_slice* slices, size_t n_slices, void* stream) {
    REQ_FR(ctx);
    return eval_rows_lists_any(ctx, x_rows_dev, x_row_stride, G, n, d, tmp_dev, y_out_dev, list_row0, list_rows, K, slices, n_slices, stream);
}
extern "C" ShareErrorCode hbmpc_gl_dev_vandermonde_apply_rows_lists(hbmpc_ctx* ctx, const uint64_t* x_rows_dev, size_t x_row_stride, size_t G,
                                                                    size_t n, size_t d, uint64_t* tmp_dev, uint64_t* y_out_dev, size_t list_row0,
                                                                    size_t list_rows, size_t K, const hbmpc_list_slice* slices, size_t n_slices,
                                                                    void* stream) {
    REQ_GL(ctx);
    return eval_rows_lists_any(ctx, x_rows_dev, x_row_stride, G, n, d, tmp_dev, y_out_dev, list_row0, list_rows, K, slices, n_slices, stream);
}
static ShareErrorCode encode_fvec_any(hbmpc_ctx* ctx, const void* x_dev, size_t G, size_t n, size_t d, void* payloads_dev,
                                      size_t payload_stride_bytes, void* stream) {
    if (!ctx) return InvalidInput;
    if (!payloads_dev) return fail(ctx, InvalidInput, "null buffer");
    const size_t eb = ebytes(ctx);  // Fr: bodies must be 32-byte aligned; Goldilocks: 8-byte elements, any 8-byte-aligned payload
    if (((uintptr_t)payloads_dev + 8) % eb || payload_stride_bytes % eb || payload_stride_bytes < 8 + eb * G)
        return fail(ctx, InvalidInput, eb == 32 ? "payloads must start 8 bytes before a 32-byte boundary, at a stride that is a multiple of 32 and >= 8 + 32 G"
                                                : "payloads must be 8-byte aligned, at a stride that is a multiple of 8 and >= 8 + 8 G");
    ShareErrorCode rc = eval_dev(ctx, x_dev, G, n, d, (char*)payloads_dev + 8, stream, 1, payload_stride_bytes / eb);
    if (rc != ShareSuccess || G == 0) return rc;
    launch_fvec_prefix((uint64_t*)payloads_dev, payload_stride_bytes / 8, G, n, pick(ctx, stream));
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
static ShareErrorCode validate_fvec_any(hbmpc_ctx* ctx, const void* payloads_dev, size_t payload_stride_bytes,
                                        size_t payload_bytes, size_t G, size_t n_rows, uint32_t* status_dev, void* stream) {
    if (!ctx) return InvalidInput;
    if (n_rows == 0) return ShareSuccess;
    if (!payloads_dev || !status_dev) return fail(ctx, InvalidInput, "null buffer");
    const size_t eb = ebytes(ctx);
    if (payload_bytes < 8 || payload_bytes < 8 + eb * G)
        return fail(ctx, InvalidInput, "payload shorter than its length prefix requires");
    if (payload_stride_bytes % 8 || payload_stride_bytes < payload_bytes || ((uintptr_t)payloads_dev & 7))
        return fail(ctx, InvalidInput, "payload stride must be 8-byte aligned and >= payload_bytes");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    HIP_TRY(ctx, hipMemsetAsync(status_dev, 0, n_rows * 4, s));
    launch_validate_fvec((const uint64_t*)payloads_dev, payload_stride_bytes / 8, G, n_rows, status_dev, s, is_gold(ctx));
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
#define TYPED_WIRE(T, REQ, PFX)                                                                                            \
    extern "C" ShareErrorCode PFX##dev_encode_fvec(hbmpc_ctx* ctx, const T* x_dev, size_t G, size_t n, size_t d,           \
                                                   void* payloads_dev, size_t payload_stride_bytes, void* stream) {         \
        REQ(ctx);                                                                                                          \
        return encode_fvec_any(ctx, x_dev, G, n, d, payloads_dev, payload_stride_bytes, stream);                           \
    }                                                                                                                      \
    extern "C" ShareErrorCode PFX##dev_validate_fvec(hbmpc_ctx* ctx, const void* payloads_dev, size_t payload_stride_bytes, \
                                                     size_t payload_bytes, size_t G, size_t n_rows, uint32_t* status_dev,  \
                                                     void* stream) {                                                       \
        REQ(ctx);                                                                                                          \
        return validate_fvec_any(ctx, payloads_dev, payload_stride_bytes, payload_bytes, G, n_rows, status_dev, stream);   \
    }
TYPED_WIRE(U256, REQ_FR, hbmpc_)
TYPED_WIRE(uint64_t, REQ_GL, hbmpc_gl_)
extern "C" ShareErrorCode hbmpc_dev_pack_shares(hbmpc_ctx* ctx, const U256* values_dev, size_t N, size_t id,
                                                size_t degree, void* payload_dev, void* stream) {
    REQ_FR(ctx);
    if (!ctx) return InvalidInput;
    if (!payload_dev || (N && !values_dev) || ((uintptr_t)payload_dev & 7)) return fail(ctx, InvalidInput, "bad buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    launch_pack_shares((const uint64_t*)values_dev, N, id, degree, (uint64_t*)payload_dev, pick(ctx, stream));
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_unpack_shares(hbmpc_ctx* ctx, const void* payload_dev, size_t payload_bytes, size_t N,
                                                  size_t id, size_t degree, U256* values_dev, uint32_t* status_dev,
                                                  void* stream) {
    REQ_FR(ctx);
    if (!ctx) return InvalidInput;
    if (!payload_dev || !status_dev || (N && !values_dev) || ((uintptr_t)payload_dev & 7)) return fail(ctx, InvalidInput, "bad buffer");
    if (payload_bytes < 8 + 48 * N) return fail(ctx, InvalidInput, "payload shorter than its length prefix requires");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    HIP_TRY(ctx, hipMemsetAsync(status_dev, 0, 4, s));
    launch_unpack_shares((const uint64_t*)payload_dev, N, id, degree, (uint64_t*)values_dev, status_dev, s);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_validate_canonical(hbmpc_ctx* ctx, const U256* a_dev, size_t N, uint32_t* status_dev,
                                                       void* stream) {
    REQ_FR(ctx);
    if (!ctx) return InvalidInput;
    if (!status_dev || (N && !a_dev)) return fail(ctx, InvalidInput, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    HIP_TRY(ctx, hipMemsetAsync(status_dev, 0, 4, s));
    if (N) launch_validate_canonical((const uint64_t*)a_dev, N, status_dev, s);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}

// ---- layout / verdict steps of the preprocessing producers (either field: the element size follows the context) ----
extern "C" ShareErrorCode hbmpc_dev_transpose(hbmpc_ctx* ctx, const void* src_dev, size_t rows, size_t cols, size_t src_row_stride,
                                              void* dst_dev, size_t dst_row_stride, size_t batch, size_t src_batch_stride,
                                              size_t dst_batch_stride, void* stream) {
    if (!ctx) return InvalidInput;
    if (rows == 0 || cols == 0 || batch == 0) return ShareSuccess;
    if (!src_dev || !dst_dev) return fail(ctx, InvalidInput, "null buffer");
    if (src_row_stride < cols || dst_row_stride < rows) return fail(ctx, InvalidInput, "row stride below the row length");
    if (batch > 65535 || (rows + 15) / 16 > 65535) return fail(ctx, InvalidInput, "batch / rows beyond the launch grid");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    launch_transpose(is_gold(ctx) ? 1 : 4, (const uint64_t*)src_dev, rows, cols, src_row_stride, (uint64_t*)dst_dev, dst_row_stride, batch,
                     src_batch_stride, dst_batch_stride, s);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_check_degree(hbmpc_ctx* ctx, const void* coeffs_dev, const uint8_t* status_dev, size_t G, size_t m,
                                                 size_t want_degree, uint32_t* bad_dev, void* stream) {
    if (!ctx) return InvalidInput;
    if (!bad_dev || (G && !coeffs_dev) || m == 0 || m > 65535) return fail(ctx, InvalidInput, "null buffer or bad length");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    if (G) launch_check_degree(is_gold(ctx) ? 1 : 4, (const uint64_t*)coeffs_dev, status_dev, G, (int)m, (int)want_degree, bad_dev, s);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
// columns > 0: the G entries are several verifiers' columns one after the other; bad[1] is the lowest failing COLUMN (g mod columns)
static ShareErrorCode check_top_coeff_any(hbmpc_ctx* ctx, const void* top_dev, const uint8_t* status_dev, size_t G, size_t want_degree, uint32_t* bad_dev,
                                          void* stream, size_t columns) {
    if (!ctx) return InvalidInput;
    if (!bad_dev || (G && !top_dev) || want_degree > 65535) return fail(ctx, InvalidInput, "null buffer or bad degree");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    if (G) launch_check_top_coeff(is_gold(ctx) ? 1 : 4, (const uint64_t*)top_dev, status_dev, G, (int)want_degree, bad_dev, s, columns);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_check_top_coeff(hbmpc_ctx* ctx, const void* top_dev, const uint8_t* status_dev, size_t G, size_t want_degree,
                                                    uint32_t* bad_dev, void* stream) {
    return check_top_coeff_any(ctx, top_dev, status_dev, G, want_degree, bad_dev, stream, 0);
}
extern "C" ShareErrorCode hbmpc_dev_check_double_share_sel(hbmpc_ctx* ctx, const void* sel_t_dev, const uint8_t* status_t_dev, const void* sel_2t_dev,
                                                           const uint8_t* status_2t_dev, size_t G, size_t columns, size_t t, uint32_t* bad_dev, void* stream) {
    if (!ctx) return InvalidInput;
    if (!bad_dev || (G && (!sel_t_dev || !sel_2t_dev || !status_t_dev || !status_2t_dev))) return fail(ctx, InvalidInput, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    if (columns && G % columns != 0) return fail(ctx, InvalidInput, "G must be a multiple of the columns");
    if (G) launch_check_double_sel(is_gold(ctx) ? 1 : 4, (const uint64_t*)sel_t_dev, status_t_dev, (const uint64_t*)sel_2t_dev, status_2t_dev, G, (int)t, bad_dev, s, columns);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_check_double_share(hbmpc_ctx* ctx, const void* coeffs_t_dev, const void* coeffs_2t_dev, size_t G, size_t m,
                                                       size_t t, uint32_t* bad_dev, void* stream) {
    if (!ctx) return InvalidInput;
    if (!bad_dev || (G && (!coeffs_t_dev || !coeffs_2t_dev)) || m == 0 || m > 65535) return fail(ctx, InvalidInput, "null buffer or bad length");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    if (G) launch_check_double(is_gold(ctx) ? 1 : 4, (const uint64_t*)coeffs_t_dev, (const uint64_t*)coeffs_2t_dev, G, (int)m, (int)t, bad_dev, s);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
// columns: 0, or the G entries are several verifiers' results for the same `columns` columns, one verifier after the other (G a multiple of it):
// bad[1] is then the first failing COLUMN
extern "C" ShareErrorCode hbmpc_dev_check_double_share_c0_columns(hbmpc_ctx* ctx, const void* c0_t_dev, const uint32_t* degree_t_dev, const void* c0_2t_dev,
                                                                  const uint32_t* degree_2t_dev, size_t G, size_t columns, size_t t, uint32_t* bad_dev,
                                                                  void* stream) {
    if (!ctx) return InvalidInput;
    if (!bad_dev || (G && (!c0_t_dev || !c0_2t_dev || !degree_t_dev || !degree_2t_dev))) return fail(ctx, InvalidInput, "null buffer");
    if (columns && G % columns != 0) return fail(ctx, InvalidInput, "G must be a multiple of the number of columns");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    if (G) launch_check_double_c0(is_gold(ctx) ? 1 : 4, (const uint64_t*)c0_t_dev, degree_t_dev, (const uint64_t*)c0_2t_dev, degree_2t_dev, G, (int)t, bad_dev, s, columns);
    HIP_TRY(ctx, hipGetLastError());
    return ShareSuccess;
}
extern "C" ShareErrorCode hbmpc_dev_check_double_share_c0(hbmpc_ctx* ctx, const void* c0_t_dev, const uint32_t* degree_t_dev, const void* c0_2t_dev,
                                                          const uint32_t* degree_2t_dev, size_t G, size_t t, uint32_t* bad_dev, void* stream) {
    return hbmpc_dev_check_double_share_c0_columns(ctx, c0_t_dev, degree_t_dev, c0_2t_dev, degree_2t_dev, G, 0, t, bad_dev, stream);
}

#include "capi_recover.inc"

// ---- TripleGenNode for all parties on this device (triple_gen/triple_generation.rs:304-364) --------------------------------------
// T: U256 (H = HFr) or uint64_t (H = HGl)
template <class H, class T>
static ShareErrorCode triplegen_parties_any(hbmpc_ctx* ctx, const T* a, const T* b, const T* r2t, const T* rt, size_t N, size_t n, size_t t, T* y_ws,
                                            T* z_ws, T* opened_out, T* c_out, uint8_t* status_out, hbmpc_recover_summary* summary_first_dev,
                                            hbmpc_recover_summary* summary_dev, void* stream) {
    constexpr bool gold = std::is_same<T, uint64_t>::value;
    if (!a || !b || !r2t || !rt || !y_ws || !z_ws || !opened_out || !c_out) return fail(ctx, InvalidInput, "null buffer");
    const size_t M = 2 * t + 1, d = 2 * t;
    if (N == 0 || n == 0 || n > 255 || N % M != 0) return fail(ctx, InvalidInput, "N must be a positive multiple of 2t + 1; n in 1 .. 255");
    if (n < 3 * t + 1) return fail(ctx, InvalidInput, "n must be >= 3t + 1 for Byzantine fault tolerance");
    const size_t G = N / M;
    std::vector<size_t> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i] = i;
    // Small batches: one launch, a workgroup per chunk (kernels_triplegen_wg.hpp).  n = 3t + 1 <= 16: every recipient decodes from
    // exactly d + t + 1 senders (no OEC round), and the (party, recipient) pairs fit the workgroup.
    // (over Goldilocks the four launches are flat at ~28 us and overtake at ~600 chunks: half the threshold)
    if (G <= (gold ? ctx->fused_triplegen_max / 2 : ctx->fused_triplegen_max) && n == 3 * t + 1 && n <= 16 && (gold || ctx->impl == IMPL_U29) &&
        !ctx->force_generic && ctx->direct_fail) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipStream_t s = pick(ctx, stream);
        const int impl = ctx->impl;
        TripleGenWgArgs ta;
        memset(&ta, 0, sizeof ta);
        const std::shared_ptr<const DomainInv<H>> dom = domain_inv<H>(ctx, n);
        ShareErrorCode rc = get_table(ctx, ids_key("rec", ids, n, n, d, t, impl), [&] {
            RecoverTables T2 = build_recover_tables<H>(*dom, ids, d, t, impl);
            std::vector<uint32_t> both = T2.vm;
            both.insert(both.end(), T2.bc.begin(), T2.bc.end());
            return both;
        }, &ta.tab);
        if (rc != ShareSuccess) return rc;
        rc = vmat_table<H>(ctx, n, d, &ta.vmat);
        if (rc != ShareSuccess) return rc;
        const ElemConsts cs = elem_consts(impl);
        memcpy(ta.r2, cs.r2, sizeof ta.r2);
        ta.a = (const uint32_t*)a, ta.b = (const uint32_t*)b, ta.r2t = (const uint32_t*)r2t, ta.rt = (const uint32_t*)rt;
        ta.Y = (uint32_t*)y_ws, ta.Z = (uint32_t*)z_ws, ta.opened = (uint32_t*)opened_out, ta.c = (uint32_t*)c_out, ta.status = status_out;
        ta.G = G, ta.N = N, ta.n = (int)n, ta.t = (int)t;
        std::lock_guard<std::mutex> enqueue_lock(ctx->enqueue_mu);
        void* scratch;
        bool dirty = false;
        rc = get_scratch(ctx, s, 2048, &scratch, &dirty);
        if (rc != ShareSuccess) return rc;
        ta.counters = (uint32_t*)scratch;
        ta.summary_first = summary_first_dev ? (uint32_t*)summary_first_dev : ta.counters + 4;
        ta.summary = summary_dev ? (uint32_t*)summary_dev : ta.counters + 4;
        if (dirty) HIP_TRY(ctx, hipMemsetAsync(ta.counters, 0, 128, s));
        set_scratch_dirty(ctx, s, true);
        launch_triplegen_wg(impl, ta, s);
        HIP_TRY(ctx, hipGetLastError());
        set_scratch_dirty(ctx, s, false);  // the kernel's last workgroup leaves the counters at zero
        return ShareSuccess;
    }
    // [ab - r]_2t Vandermonde-encoded in chunks of 2t + 1 for every recipient (batch_recon.rs:157-165), all parties in one launch;
    // EvalBatch arm for ALL recipients in one call: the row of sender p for "chunk" j G + g is y_ws + p (n G) + (j G + g);
    // RevealBatch arm: everyone interpolates the 2t + 1 opened values per chunk from the n broadcast values;
    // [c]_t = rt_i + opened (triple_generation.rs:196-208)
    ShareErrorCode rc;
    if constexpr (gold) {
        rc = hbmpc_gl_dev_triple_encode_parties(ctx, a, b, r2t, G, n, d, n, c_out, y_ws, stream);
        if (rc != ShareSuccess) return rc;
        rc = hbmpc_gl_dev_batch_recover_strided(ctx, ids.data(), n, y_ws, n * G, n * G, n, d, t, 1, z_ws, nullptr, status_out, summary_first_dev, stream);
        if (rc != ShareSuccess) return rc;
        rc = hbmpc_gl_dev_batch_recover(ctx, ids.data(), n, z_ws, G, n, d, t, opened_out, nullptr, status_out, summary_dev, stream);
        if (rc != ShareSuccess) return rc;
        return hbmpc_gl_dev_triple_finalize_parties(ctx, rt, opened_out, N, n, c_out, stream);
    } else {
        rc = hbmpc_dev_triple_encode_parties(ctx, a, b, r2t, G, n, d, n, c_out, y_ws, stream);
        if (rc != ShareSuccess) return rc;
        rc = hbmpc_dev_batch_recover_strided(ctx, ids.data(), n, y_ws, n * G, n * G, n, d, t, 1, z_ws, nullptr, status_out, summary_first_dev, stream);
        if (rc != ShareSuccess) return rc;
        rc = hbmpc_dev_batch_recover(ctx, ids.data(), n, z_ws, G, n, d, t, opened_out, nullptr, status_out, summary_dev, stream);
        if (rc != ShareSuccess) return rc;
        return hbmpc_dev_triple_finalize_parties(ctx, rt, opened_out, N, n, c_out, stream);
    }
}
extern "C" ShareErrorCode hbmpc_dev_triplegen_parties(hbmpc_ctx* ctx, const U256* a, const U256* b, const U256* r2t, const U256* rt, size_t N,
                                                      size_t n, size_t t, U256* y_ws, U256* z_ws, U256* opened_out, U256* c_out, uint8_t* status_out,
                                                      hbmpc_recover_summary* summary_first_dev, hbmpc_recover_summary* summary_dev, void* stream) {
    REQ_FR(ctx);
    return triplegen_parties_any<HFr>(ctx, a, b, r2t, rt, N, n, t, y_ws, z_ws, opened_out, c_out, status_out, summary_first_dev, summary_dev, stream);
}
extern "C" ShareErrorCode hbmpc_gl_dev_triplegen_parties(hbmpc_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* r2t, const uint64_t* rt,
                                                         size_t N, size_t n, size_t t, uint64_t* y_ws, uint64_t* z_ws, uint64_t* opened_out,
                                                         uint64_t* c_out, uint8_t* status_out, hbmpc_recover_summary* summary_first_dev,
                                                         hbmpc_recover_summary* summary_dev, void* stream) {
    REQ_GL(ctx);
    return triplegen_parties_any<HGl>(ctx, a, b, r2t, rt, N, n, t, y_ws, z_ws, opened_out, c_out, status_out, summary_first_dev, summary_dev, stream);
}

// ---- FPMulNode for all parties on this device (fpmul/fpmul.rs:61-110) ----------------------------------------------------------
// The table of k_fpmul_wave: the verify rows and the P(0) row of the decode's table, and the P(0) row scaled by R (its
// dot product is the opened value in Montgomery form: what finalize_mul multiplies the shares by)
static ShareErrorCode fpmul_wave_table(hbmpc_ctx* ctx, const SortedSenders& ss, size_t n, size_t t, const uint32_t** out) {
    const int impl = ctx->impl;
    const std::shared_ptr<const DomainInv<HFr>> dom = domain_inv<HFr>(ctx, n);
    return get_table(ctx, ids_key("fpw", ss.ids, 2 * t + 1, n, t, t, impl), [&] {
        const auto rows = recover_coeff_rows<HFr>(*dom, ss.ids, t, t);  // t verify rows, then the coefficient rows
        const HFr R = rdev_value(impl);
        std::vector<uint32_t> w;
        for (size_t r = 0; r <= t; ++r)
            for (size_t i = 0; i <= t; ++i) put_const(w, rows[r][i], impl);
        for (size_t i = 0; i <= t; ++i) put_const(w, rows[t][i] * R, impl);
        return w;
    }, out);
}
extern "C" ShareErrorCode hbmpc_dev_fpmul_parties(hbmpc_ctx* ctx, const size_t* sender_ids, size_t S, const U256* a, const U256* b, const U256* c,
                                                  const U256* x, const U256* y, const U256* r_bits, const U256* r_int, size_t k, size_t m, size_t N,
                                                  size_t n, size_t t, U256* de_sh_ws, U256* de_out, U256* z_out, U256* r_dash_out,
                                                  U256* open_sh_out, U256* c_open_out, U256* d_out, uint8_t* status_out,
                                                  hbmpc_recover_summary* summary_first_dev, hbmpc_recover_summary* summary_dev, void* stream) {
    REQ_FR(ctx);
    if (k == 0) return fail(ctx, InvalidInput, "k must be >= 1 (2^(k-1))");
    if (m > 4096) return fail(ctx, InvalidInput, "m beyond the supported range");
    if (m % 8 != 0 && m / 8 >= 32) return fail(ctx, InvalidInput, "m: bytes[m/8] out of bounds in the reference");
    if (!a || !b || !c || !x || !y || !r_int || (m && !r_bits) || !de_sh_ws || !de_out || !z_out || !r_dash_out || !open_sh_out || !c_open_out || !d_out)
        return fail(ctx, InvalidInput, "null buffer");
    if (N == 0 || n == 0 || n > 255) return fail(ctx, InvalidInput, "N, n out of range");
    // Small batches: the whole multiplication is one launch, a wave per element (kernels_fpmul_wave.hpp).  Larger ones, calls
    // with OEC rounds available (S > 2t + 1) and the other field implementations run the five separate launches below -- the
    // same bytes in every output buffer.
    if (N <= ctx->fused_fpmul_max && S == 2 * t + 1 && ctx->impl == IMPL_U29 && !ctx->force_generic && ctx->direct_fail && n <= 64 && t <= 30 &&
        (4 + m) * n <= 4096) {
        SortedSenders ss;
        ShareErrorCode rc = validate_senders(ctx, sender_ids, S, N, n, t, t, &ss);
        if (rc != ShareSuccess) return rc;
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipStream_t s = pick(ctx, stream);
        const int impl = ctx->impl;
        FpmulWaveArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.N = N, fa.parties = (int)n, fa.m = (int)m, fa.needed = (int)(2 * t + 1), fa.M = (int)(t + 1), fa.mask_bits = (int)(m > 256 ? 256 : m);
        // the products of a table row are shared by up to four adjacent lanes (a DPP quad) while the rows still fit the wave
        while (fa.lk1 < 2 && ((t + 2) << (fa.lk1 + 1)) <= 32 && ((size_t)2 << fa.lk1) <= t + 1) ++fa.lk1;
        while (fa.lk3 < 2 && ((t + 1) << (fa.lk3 + 1)) <= 64 && ((size_t)2 << fa.lk3) <= t + 1) ++fa.lk3;
        if (launch_fpmul_wave(fa, ctx->device, s, true)) {
            rc = fpmul_wave_table(ctx, ss, n, t, &fa.tab);
            if (rc != ShareSuccess) return rc;
            rc = get_table(ctx, key("pow2", {m}, impl), [&] { return build_pow2(m, impl); }, &fa.pow2);
            if (rc != ShareSuccess) return rc;
            const HFr two = HFr::from_u64(2);
            const HFr p2m = two.pow_u64(m), p2k = two.pow_u64(k - 1), inv = inv_pow2(ctx, m);
            const ElemConsts cs = elem_consts(impl, &p2m, &p2k), ci = elem_consts(impl, &inv);
            memcpy(fa.c0, cs.c0, sizeof fa.c0), memcpy(fa.c1, cs.c1, sizeof fa.c1), memcpy(fa.cinv, ci.c0, sizeof fa.cinv);
            fa.ta = (const uint32_t*)a, fa.tb = (const uint32_t*)b, fa.tc = (const uint32_t*)c, fa.x = (const uint32_t*)x, fa.y = (const uint32_t*)y;
            fa.r_bits = (const uint32_t*)r_bits, fa.r_int = (const uint32_t*)r_int;
            fa.de_out = (uint32_t*)de_out, fa.z = (uint32_t*)z_out, fa.r_dash = (uint32_t*)r_dash_out, fa.open_sh = (uint32_t*)open_sh_out;
            fa.out = (uint32_t*)d_out, fa.c_open = (uint32_t*)c_open_out, fa.status = status_out;
            for (size_t i = 0; i < S; ++i) fa.rows.set(i, (unsigned)ss.ids[i]);  // the per-party arrays are indexed by party id
            // the decodes' counters (see batch_recover_dev): zero between calls, cleared here only when that is not known
            std::lock_guard<std::mutex> enqueue_lock(ctx->enqueue_mu);
            void* scratch;
            bool dirty = false;
            rc = get_scratch(ctx, s, 2048, &scratch, &dirty);
            if (rc != ShareSuccess) return rc;
            fa.counters = (uint32_t*)scratch;
            fa.summary_first = summary_first_dev ? (uint32_t*)summary_first_dev : fa.counters + 4;  // the scratch's local summary slot
            fa.summary = summary_dev ? (uint32_t*)summary_dev : fa.counters + 4;
            if (dirty) HIP_TRY(ctx, hipMemsetAsync(fa.counters, 0, 128, s));
            set_scratch_dirty(ctx, s, true);
            launch_fpmul_wave(fa, ctx->device, s, false);
            HIP_TRY(ctx, hipGetLastError());
            set_scratch_dirty(ctx, s, false);  // the kernel's last workgroup leaves the counters at zero
            return ShareSuccess;
        }
    }
    // the shares Multiply opens (multiplication.rs:417-426) and reconstruct_rbc's recover_secret of a - x and of b - y (:102-139):
    // ONE interpolation over the 2 N values of a sender row
    // Large batches: the shares are formed as the matrix-core decode loads them (kernels_mfma.hpp, k_mfma_rows<.., SUB>) -- 4 (2t + 1)
    // loads per element instead of a launch that writes all n parties' two shares and a decode that reads 2t + 1 of them back
    // (config 5: 0.14 + 0.065 ms -> 0.11; ahead from ~8 000 elements, tools/sweep_fused_fpmul.py).
    ShareErrorCode rc = HBMPC_NOT_FUSED;
    if (N >= ctx->pair_decode_min) {
        PairInput pi = {(const uint32_t*)a, (const uint32_t*)b, (const uint32_t*)x, (const uint32_t*)y, N};
        rc = batch_recover_dev(ctx, sender_ids, S, nullptr, 2 * N, n, t, t, de_out, nullptr, status_out, summary_first_dev, true, stream, 0, nullptr, false, &pi);
    }
    if (rc == HBMPC_NOT_FUSED) {
        rc = hbmpc_dev_beaver_open_shares_paired(ctx, a, b, x, y, N, n, de_sh_ws, stream);
        if (rc != ShareSuccess) return rc;
        rc = hbmpc_dev_batch_recover_p0(ctx, sender_ids, S, de_sh_ws, 2 * N, n, t, t, de_out, status_out, summary_first_dev, stream);
    }
    if (rc != ShareSuccess) return rc;
    // finalize_mul (:57-100), r' (truncpr.rs:277-283), the share TruncPr opens (:294-297), its open (truncpr.rs:215), the last step (:216-220)
    rc = hbmpc_dev_fpmul_middle(ctx, c, x, y, de_out, de_out + N, r_bits, r_int, k, m, N, n, z_out, r_dash_out, open_sh_out, stream);
    if (rc != ShareSuccess) return rc;
    rc = hbmpc_dev_batch_recover_p0(ctx, sender_ids, S, open_sh_out, N, n, t, t, c_open_out, status_out, summary_dev, stream);
    if (rc != ShareSuccess) return rc;
    return hbmpc_dev_truncpr_finalize_parties(ctx, z_out, r_dash_out, c_open_out, m, N, n, d_out, stream);
}
