// launchers.hpp -- the kernel instantiations live in several translation units (tu_*.hip) so that
// they compile in parallel; these are their entry points.  Each returns false when the requested
// shape is not one it instantiates.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "eval_out.hpp"
#include "kernels_gao.hpp"
#include "kernels_recover.hpp"

#include <atomic>

namespace hbmpc {

// More than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device).  Returns
// false when the launch must not be attempted (the caller falls back to the lane kernels): a device ordinal beyond the
// table, or the attribute call failed.  Threads that drive different contexts on one device may race here: the flag is
// atomic and setting the attribute twice is harmless.
constexpr int HBMPC_MAX_DEVICES = 64;
inline bool ensure_dynamic_lds(const void* kernel, std::atomic<bool>* flags, int device, size_t lds_bytes) {
    if (lds_bytes <= 64 * 1024) return true;
    if (device < 0 || device >= HBMPC_MAX_DEVICES) return false;
    if (flags[device].load(std::memory_order_acquire)) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    flags[device].store(true, std::memory_order_release);
    return true;
}

// single-pass pruned FFT, U29, size = 2^log <= 16, cnt = d+1 coefficients
bool launch_fft1_lo(int log, int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s);
bool launch_fft1_16a(int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s);
bool launch_fft1_16b(int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s);
bool launch_fft1_16c(int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s);
bool launch_fft1_16d(int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s);
// triple_gen's local product fused into the encode (k_eval_fft1_triple); false when the shape is not instantiated
bool launch_fft1_triple(int lg, int cnt, const uint32_t* a, const uint32_t* b, const uint32_t* r2t, size_t G, int n,
                        const uint32_t* tw, EvalOut y, const uint32_t r2[9], hipStream_t s);
bool launch_fft1_triple_gold(int lg, int cnt, const uint32_t* a, const uint32_t* b, const uint32_t* r2t, size_t G, int n,
                             const uint32_t* tw, EvalOut y, hipStream_t s);
// multi-pass (size = 16 P), U29, dp1 <= 32
bool launch_fftP_a(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16, const uint32_t* twist,
                   EvalOut y, hipStream_t s);
bool launch_fftP_b(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16, const uint32_t* twist,
                   EvalOut y, hipStream_t s);
bool launch_fftP_c(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16, const uint32_t* twist,
                   EvalOut y, hipStream_t s);
bool launch_fftP_d(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16, const uint32_t* twist,
                   EvalOut y, hipStream_t s);
bool launch_fftP_fold(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16, const uint32_t* twist,
                      EvalOut y, hipStream_t s);
// Goldilocks instantiations of the same templates
bool launch_gold_fft1(int log, int cnt, const uint32_t* x, size_t G, int n, const uint32_t* tw, EvalOut y, hipStream_t s);
bool launch_gold_fftP(int dp1, const uint32_t* x, size_t G, int n, int P, const uint32_t* tw16, const uint32_t* twist,
                      EvalOut y, hipStream_t s);
bool launch_gold_recover(int m, bool p0, const RecoverArgs& ra, unsigned grid, hipStream_t s);
// generic Horner evaluation (impl: 0 = U29, 1 = Sat32, 2 = Goldilocks)
void launch_eval_generic(int impl, const uint32_t* x, size_t G, int n, int dp1, const uint32_t* alpha, EvalOut y,
                         hipStream_t s);
// batch recover, U29, register-resident m <= 16
bool launch_recover_a(int m, bool p0, const RecoverArgs& ra, unsigned grid, hipStream_t s);
bool launch_recover_b(int m, bool p0, const RecoverArgs& ra, unsigned grid, hipStream_t s);
bool launch_recover_c(int m, bool p0, const RecoverArgs& ra, unsigned grid, hipStream_t s);
bool launch_recover_d(int m, bool p0, const RecoverArgs& ra, unsigned grid, hipStream_t s);

void launch_recover_generic(int impl, bool p0, const RecoverArgs& ra, unsigned grid, hipStream_t s);
// matrix-core form of the constant-matrix maps (kernels_mfma.hpp), m = 2 .. 15; false when m is not instantiated there.
// team: the workgroup-per-tile form for batches with fewer tiles than waves (kernels_mfma_team.hpp)
namespace mf { struct MfmaRowsArgs; struct MfmaGlArgs; }
bool launch_mfma_rows_gl(const mf::MfmaGlArgs& a, unsigned grid, int device, hipStream_t s);  // Goldilocks (kernels_mfma_gl.hpp)
// the decode whose sender values are differences formed after loading (k_mfma_rows<.., SUB>), m = 2 .. 11
bool mfma_sub_covers(int m);
bool launch_mfma_rows_sub(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s);
bool launch_mfma_rows_a(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s, bool team = false);
bool launch_mfma_rows_b(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s, bool team = false);
bool launch_mfma_rows_c(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s, bool team = false);
bool launch_mfma_rows_d(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s, bool team = false);
// the encode on a domain of roots of unity with the points taken in pairs (k, k + size / 2): half the MFMAs (kernels_mfma_bfly.hpp);
// with a.in_b set: the fused local product + encode of triple generation (one role of 8 or 16 pairs; false otherwise)
bool launch_mfma_bfly_a(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s);
bool launch_mfma_bfly_b(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s);
bool launch_mfma_bfly_c(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s);
bool launch_mfma_bfly_d(int m, const mf::MfmaRowsArgs& a, int device, hipStream_t s);
// small batches: one wave per chunk, one evaluation point / one table row per lane (k_eval_wide, k_batch_recover_wide)
void launch_eval_wide(int impl, const uint32_t* x, size_t G, int n, int dp1, const uint32_t* alpha, EvalOut y, hipStream_t s);
void launch_eval_wide_dot(const uint32_t* x, size_t G, int n, int dp1, const uint32_t* vmat, EvalOut y, hipStream_t s);  // U29, vmat [n][dp1] <= 48 KB
void launch_recover_wide(int impl, bool p0, const RecoverArgs& ra, const SecondArgs* fused_second, hipStream_t s, int ow_sel = 0);  // ow_sel: output rows when the table holds selected ones
// FPMulNode for all parties of a small batch in one launch (kernels_fpmul_wave.hpp); false: the shape does not fit a workgroup's LDS
struct FpmulWaveArgs;
struct TripleGenWgArgs;
void launch_triplegen_wg(int impl, const TripleGenWgArgs& a, hipStream_t s);  // TripleGenNode, a workgroup per chunk (kernels_triplegen_wg.hpp): U29 or Goldilocks
bool launch_fpmul_wave(const FpmulWaveArgs& a, int device, hipStream_t s, bool dry_run);
// flagged chunks: two cheap interpolation candidates before the OEC/Gao kernel (k_second_chance)
void launch_second_chance(int impl, const SecondArgs& a, unsigned grid, hipStream_t s);
bool launch_second_chance_m(int m, const SecondArgs& a, unsigned grid, hipStream_t s);  // U29, m = 2 .. 16 at compile time; false otherwise
// OEC / Gao, matvec
void launch_gao_u29(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s, bool inline_unscale);
void launch_gao_sat(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s, bool inline_unscale);
void launch_gao_gold(const GaoArgs& ga, size_t n, unsigned grid, hipStream_t s, bool inline_unscale);
void launch_matvec(int impl, const uint32_t* lb, const uint32_t* y, int S, uint32_t* out, hipStream_t s);

// seeded coefficient generation (kernels_rng.hpp); ew = u32 words per element
void launch_fill_coeffs(int ew, const uint32_t seed[8], const uint32_t* secrets, size_t B, uint64_t first_index, int dp1,
                        uint32_t* coeffs, hipStream_t s);
// wire codec (kernels_codec.hpp)
void launch_pack_fvec(const uint64_t* rows, size_t row_stride, size_t G, size_t n_rows, uint64_t* payloads,
                      size_t payload_stride_words, hipStream_t s);
void launch_fvec_prefix(uint64_t* payloads, size_t payload_stride_words, size_t G, size_t n_rows, hipStream_t s);
void launch_validate_fvec(const uint64_t* payloads, size_t payload_stride_words, size_t G, size_t n_rows, uint32_t* status,
                          hipStream_t s, bool gold = false);
void launch_unpack_fvec(const uint64_t* payloads, size_t payload_stride_words, size_t G, size_t n_rows, uint64_t* rows,
                        size_t row_stride, uint32_t* status, hipStream_t s);
void launch_pack_shares(const uint64_t* values, size_t N, uint64_t id, uint64_t degree, uint64_t* payload, hipStream_t s);
void launch_unpack_shares(const uint64_t* payload, size_t N, uint64_t id, uint64_t degree, uint64_t* values,
                          uint32_t* status, hipStream_t s);
void launch_validate_canonical(const uint64_t* a, size_t N, uint32_t* status, hipStream_t s);
void launch_poly_degree(const uint64_t* coeffs, size_t G, int m, int ew64, uint32_t* degree_out, hipStream_t s, uint64_t* c0_out = nullptr);  // c0_out: the constant terms too
void launch_mfma_table(const uint64_t* coeff, int m, int rows, const uint64_t e[4], uint32_t bmag, uint8_t* table, uint64_t* partial,
                       hipStream_t s);  // kernels_tables.hpp
// layout + verdict kernels of the preprocessing producers (kernels_codec.hpp); ew64 = 64-bit words per element
void launch_transpose(int ew64, const uint64_t* src, size_t rows, size_t cols, size_t src_row_stride, uint64_t* dst, size_t dst_row_stride,
                      size_t batch, size_t src_batch_stride, size_t dst_batch_stride, hipStream_t s);
void launch_check_degree(int ew64, const uint64_t* coeffs, const uint8_t* status, size_t G, int m, int want, uint32_t* bad, hipStream_t s);
void launch_check_double_sel(int ew64, const uint64_t* sel_t, const uint8_t* st_t, const uint64_t* sel_2t, const uint8_t* st_2t, size_t G, int t, uint32_t* bad,
                             hipStream_t s, size_t columns = 0);
void launch_pick_two(int ew64, const uint64_t* coeffs, size_t G, int m, int d, uint64_t* sel, uint8_t* status, hipStream_t s);
void launch_check_top_coeff(int ew64, const uint64_t* top, const uint8_t* status, size_t G, int want, uint32_t* bad, hipStream_t s, size_t columns = 0);
void launch_check_double(int ew64, const uint64_t* ct, const uint64_t* c2t, size_t G, int m, int t, uint32_t* bad, hipStream_t s);
void launch_check_double_c0(int ew64, const uint64_t* c0t, const uint32_t* degt, const uint64_t* c02t, const uint32_t* deg2t, size_t G, int t,
                            uint32_t* bad, hipStream_t s, size_t columns = 0);
bool launch_fft1_mix_lo(int log, int cnt, const uint32_t* x, size_t xs, size_t G, int n, const uint32_t* tw, const MixOut& o, hipStream_t s);
bool launch_gold_fft1_mix(int log, int cnt, const uint32_t* x, size_t xs, size_t G, int n, const uint32_t* tw, const MixOut& o, hipStream_t s);
void launch_rows_party_major(int ew64, const uint64_t* src, size_t G, size_t K, int row0, int rows, int nother, uint64_t* dst, hipStream_t s);
void launch_take_c0(int ew64, const uint64_t* coeffs, size_t G, int m, uint64_t* c0, hipStream_t s);

}  // namespace hbmpc
