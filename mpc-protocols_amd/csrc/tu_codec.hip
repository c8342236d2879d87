#include "kernels_codec.hpp"
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
void launch_poly_degree(const uint64_t* coeffs, size_t G, int m, int ew64, uint32_t* degree_out, hipStream_t s) {
    hipLaunchKernelGGL(k_poly_degree, dim3((unsigned)((G + 255) / 256 ? (G + 255) / 256 : 1)), dim3(256), 0, s, coeffs, G, m,
                       ew64, degree_out);
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
}
