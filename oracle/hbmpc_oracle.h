/* hbmpc_oracle.h -- CPU restatement of the reference path (TEST INFRASTRUCTURE ONLY; see the
 * header of hbmpc_oracle.c: PARITY UNPINNED at the stored-bytes level).
 * Signatures mirror include/hbmpc_hip.h without the ctx (same layouts, same error codes). */
#ifndef HBMPC_ORACLE_H
#define HBMPC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { uint64_t data[4]; } U256;
enum { ShareSuccess = 0, InsufficientShares, DegreeMismatch, IdMismatch, InvalidInput, TypeMismatch,
       NoSuitableDomain, PolynomialOperationError, DecodingError };

int oracle_compute_shares(const U256* coeffs, size_t B, size_t n, size_t d, U256* shares_out);
int oracle_make_vandermonde(size_t n, size_t d, U256* v_out);
int oracle_vandermonde_apply(const U256* x, size_t G, size_t n, size_t d, U256* y_out);
int oracle_batch_recover(const size_t* sender_ids, size_t S, const U256* evals, size_t G, size_t n, size_t d,
                         size_t t, U256* coeffs_out, uint32_t* ncoeffs_out, uint8_t* status_out);
int oracle_batch_recover_p0(const size_t* sender_ids, size_t S, const U256* evals, size_t G, size_t n, size_t d,
                            size_t t, U256* secrets_out, uint8_t* status_out);
int oracle_recover_secret(const size_t* ids, const size_t* degrees, const U256* vals, size_t S, size_t n, size_t t,
                          U256* coeffs_out, size_t* ncoeffs_out, U256* secret_out);
int oracle_gao_rs_decode(const U256* received, size_t k, size_t n, const size_t* erasure_positions,
                         size_t n_erasures, U256* coeffs_out, size_t* ncoeffs_out);
int oracle_nonrobust_recover_secret(const size_t* ids, const size_t* degrees, const U256* vals, size_t S, size_t n,
                                    U256* coeffs_out, size_t* ncoeffs_out, U256* secret_out);
int oracle_triple_local(const U256* a, const U256* b, const U256* r2t, size_t N, U256* out);
int oracle_triple_finalize(const U256* rt, const U256* opened, size_t N, U256* c_out);
int oracle_beaver_open_shares(const U256* a, const U256* b, const U256* x, const U256* y, size_t N, U256* d_sh,
                              U256* e_sh);
int oracle_beaver_finalize(const U256* c, const U256* x, const U256* y, const U256* d, const U256* e, size_t N,
                           U256* z_out);
int oracle_truncpr_rdash(const U256* r_bits, size_t m, size_t N, U256* r_dash_out);
int oracle_truncpr_open_share(const U256* a, const U256* r_dash, const U256* r_int, size_t k, size_t m, size_t N,
                              U256* open_out);
int oracle_truncpr_finalize(const U256* a, const U256* r_dash, const U256* c_open, size_t m, size_t N, U256* d_out);
void oracle_fr_mul(const U256* a, const U256* b, size_t N, U256* out);
void oracle_fr_add(const U256* a, const U256* b, size_t N, U256* out);
void oracle_fr_sub(const U256* a, const U256* b, size_t N, U256* out);
void oracle_fr_inv(const U256* a, size_t N, U256* out);
void oracle_domain_elements(size_t n, size_t count, U256* out);
void oracle_fill_random(uint64_t seed, size_t N, U256* out);
#ifdef __cplusplus
}
#endif
#endif
