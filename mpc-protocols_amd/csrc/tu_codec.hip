#include "kernels_codec.hpp"
#include "kernels_tables.hpp"
#include "launchers.hpp"
namespace hbmpc {
void launch_pack_fvec(const uint64_t* rows, size_t row_stride, size_t G, size_t n_rows, uint64_t* payloads,
                      size_t payload_stride_words, hipStream_t s) {
    dim3 grid((unsigned)((G + 255) / 256 ? (G + 255) / 256 : 1), (unsigned)n_rows);
    hipLaunchKernelGGL(k_pack_fvec, grid, dim3(256), 0, s, rows, row_stride, G, payloads, payload_stride_words);
}
void launch_unpack_fvec(const uint64_t* payloads, size_t payload_stride_words, size_t G, size_t n_rows, uint64_t* rows,
                        size_t row_stride, uint32_t* status, hipStream_t s) {
    dim3 grid((unsigned)((G + 255) / 256 ? (G + 255) / 256 : 1), (unsigned)n_rows);
    hipLaunchKernelGGL(k_unpack_fvec, grid, dim3(256), 0, s, payloads, payload_stride_words, G, rows, row_stride, status);
}
void launch_pack_shares(const uint64_t* values, size_t N, uint64_t id, uint64_t degree, uint64_t* payload, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_shares, dim3((unsigned)((N + 255) / 256 ? (N + 255) / 256 : 1)), dim3(256), 0, s, values, N, id,
                       degree, payload);
}
void launch_unpack_shares(const uint64_t* payload, size_t N, uint64_t id, uint64_t degree, uint64_t* values,
                          uint32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(k_unpack_shares, dim3((unsigned)((N + 255) / 256 ? (N + 255) / 256 : 1)), dim3(256), 0, s, payload, N,
                       id, degree, values, status);
}
void launch_validate_canonical(const uint64_t* a, size_t N, uint32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(k_validate_canonical, dim3((unsigned)((N + 255) / 256 ? (N + 255) / 256 : 1)), dim3(256), 0, s, a, N,
                       status);
}
void launch_poly_degree(const uint64_t* coeffs, size_t G, int m, int ew64, uint32_t* degree_out, hipStream_t s, uint64_t* c0_out) {
    hipLaunchKernelGGL(k_poly_degree, dim3((unsigned)((G + 255) / 256 ? (G + 255) / 256 : 1)), dim3(256), 0, s, coeffs, G, m,
                       ew64, degree_out, c0_out);
}
void launch_fvec_prefix(uint64_t* payloads, size_t payload_stride_words, size_t G, size_t n_rows, hipStream_t s) {
    hipLaunchKernelGGL(k_fvec_prefix, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, payloads, payload_stride_words, G, n_rows);
}
void launch_validate_fvec(const uint64_t* payloads, size_t payload_stride_words, size_t G, size_t n_rows, uint32_t* status,
                          hipStream_t s, bool gold) {
    dim3 grid((unsigned)((G + 255) / 256 ? (G + 255) / 256 : 1), (unsigned)n_rows);
    if (gold) hipLaunchKernelGGL(k_validate_fvec_gl, grid, dim3(256), 0, s, payloads, payload_stride_words, G, status);
    else hipLaunchKernelGGL(k_validate_fvec, grid, dim3(256), 0, s, payloads, payload_stride_words, G, status);
}
void launch_transpose(int ew64, const uint64_t* src, size_t rows, size_t cols, size_t src_row_stride, uint64_t* dst, size_t dst_row_stride,
                      size_t batch, size_t src_batch_stride, size_t dst_batch_stride, hipStream_t s) {
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 15) / 16), (unsigned)batch);
    if (ew64 == 4) hipLaunchKernelGGL(k_transpose<4>, grid, dim3(256), 0, s, src, rows, cols, src_row_stride, dst, dst_row_stride, src_batch_stride, dst_batch_stride);
    else hipLaunchKernelGGL(k_transpose<1>, grid, dim3(256), 0, s, src, rows, cols, src_row_stride, dst, dst_row_stride, src_batch_stride, dst_batch_stride);
}
void launch_check_degree(int ew64, const uint64_t* coeffs, const uint8_t* status, size_t G, int m, int want, uint32_t* bad, hipStream_t s) {
    const dim3 grid((unsigned)((G + 255) / 256));
    if (ew64 == 4) hipLaunchKernelGGL(k_check_degree<4>, grid, dim3(256), 0, s, coeffs, status, G, m, want, bad);
    else hipLaunchKernelGGL(k_check_degree<1>, grid, dim3(256), 0, s, coeffs, status, G, m, want, bad);
}
void launch_check_top_coeff(int ew64, const uint64_t* top, const uint8_t* status, size_t G, int want, uint32_t* bad, hipStream_t s, size_t columns) {
    const dim3 grid((unsigned)((G + 255) / 256));
    if (ew64 == 4) hipLaunchKernelGGL(k_check_top_coeff<4>, grid, dim3(256), 0, s, top, status, G, want, bad, columns);
    else hipLaunchKernelGGL(k_check_top_coeff<1>, grid, dim3(256), 0, s, top, status, G, want, bad, columns);
}
void launch_check_double_sel(int ew64, const uint64_t* sel_t, const uint8_t* st_t, const uint64_t* sel_2t, const uint8_t* st_2t, size_t G, int t, uint32_t* bad,
                             hipStream_t s, size_t columns) {
    const dim3 grid((unsigned)((G + 255) / 256));
    if (ew64 == 4) hipLaunchKernelGGL(k_check_double_sel<4>, grid, dim3(256), 0, s, sel_t, st_t, sel_2t, st_2t, G, t, bad, columns);
    else hipLaunchKernelGGL(k_check_double_sel<1>, grid, dim3(256), 0, s, sel_t, st_t, sel_2t, st_2t, G, t, bad, columns);
}
void launch_pick_two(int ew64, const uint64_t* coeffs, size_t G, int m, int d, uint64_t* sel, uint8_t* status, hipStream_t s) {
    const dim3 grid((unsigned)((G + 255) / 256));
    if (ew64 == 4) hipLaunchKernelGGL(k_pick_two<4>, grid, dim3(256), 0, s, coeffs, G, m, d, sel, status);
    else hipLaunchKernelGGL(k_pick_two<1>, grid, dim3(256), 0, s, coeffs, G, m, d, sel, status);
}
void launch_check_double(int ew64, const uint64_t* ct, const uint64_t* c2t, size_t G, int m, int t, uint32_t* bad, hipStream_t s) {
    const dim3 grid((unsigned)((G + 255) / 256));
    if (ew64 == 4) hipLaunchKernelGGL(k_check_double<4>, grid, dim3(256), 0, s, ct, c2t, G, m, t, bad);
    else hipLaunchKernelGGL(k_check_double<1>, grid, dim3(256), 0, s, ct, c2t, G, m, t, bad);
}
void launch_check_double_c0(int ew64, const uint64_t* c0t, const uint32_t* degt, const uint64_t* c02t, const uint32_t* deg2t, size_t G, int t,
                            uint32_t* bad, hipStream_t s, size_t columns) {
    const dim3 grid((unsigned)((G + 255) / 256));
    if (columns == 0) columns = G;
    if (ew64 == 4) hipLaunchKernelGGL(k_check_double_c0<4>, grid, dim3(256), 0, s, c0t, degt, c02t, deg2t, G, t, bad, columns);
    else hipLaunchKernelGGL(k_check_double_c0<1>, grid, dim3(256), 0, s, c0t, degt, c02t, deg2t, G, t, bad, columns);
}
void launch_rows_party_major(int ew64, const uint64_t* src, size_t G, size_t K, int row0, int rows, int nother, uint64_t* dst, hipStream_t s) {
    const dim3 grid((unsigned)((G + 255) / 256), (unsigned)nother);
    if (ew64 == 4) hipLaunchKernelGGL(k_rows_party_major<4>, grid, dim3(256), 0, s, src, G, K, row0, rows, nother, dst);
    else hipLaunchKernelGGL(k_rows_party_major<1>, grid, dim3(256), 0, s, src, G, K, row0, rows, nother, dst);
}
void launch_take_c0(int ew64, const uint64_t* coeffs, size_t G, int m, uint64_t* c0, hipStream_t s) {
    const dim3 grid((unsigned)((G + 255) / 256));
    if (ew64 == 4) hipLaunchKernelGGL(k_take_c0<4>, grid, dim3(256), 0, s, coeffs, G, m, c0);
    else hipLaunchKernelGGL(k_take_c0<1>, grid, dim3(256), 0, s, coeffs, G, m, c0);
}
// the matrix-core byte-digit table expanded on the device from rows x m canonical coefficients (kernels_tables.hpp);
// partial: rows * m scratch elements
void launch_mfma_table(const uint64_t* coeff, int m, int rows, const uint64_t e[4], uint32_t bmag, uint8_t* table, uint64_t* partial,
                       hipStream_t s) {
    hipLaunchKernelGGL(tb::k_mfma_table_slabs, dim3((unsigned)((rows * m + 63) / 64)), dim3(64), 0, s, coeff, m, rows, table, partial);
    tb::TableE E;
    for (int k = 0; k < 4; ++k) E.w[k] = e[k];
    hipLaunchKernelGGL(tb::k_mfma_table_bias, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, partial, m, rows, E, bmag, table);
}
}
