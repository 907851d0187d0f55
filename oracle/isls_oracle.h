/* isls_oracle.h -- declarations of the CPU oracle (test infrastructure; see isls_oracle_impl.h). */
#ifndef ISLS_ORACLE_H
#define ISLS_ORACLE_H
#include "../include/isls_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#define ORACLE_DECL(sfx)                                                                              \
    int oracle_riccati_gain_##sfx(const isls_gain_args *a);                                           \
    int oracle_riccati_ff_##sfx(const isls_ff_args *a);                                               \
    int oracle_rollout_ls_##sfx(const isls_rollout_args *a);                                          \
    int oracle_admm_update_##sfx(const isls_admm_args *a);                                            \
    int oracle_expand_quadratic_##sfx(const isls_expand_args *a);                                     \
    int oracle_linearize_##sfx(const isls_linearize_args *a);                                         \
    int oracle_accept_step_##sfx(const isls_accept_args *a);                                          \
    int oracle_outer_advance_##sfx(const isls_advance_args *a);                                       \
    int oracle_reduce_convergence_##sfx(int32_t B, const void *cost, const void *res,                 \
                                        const int32_t *active, const int32_t *status, void *out5);    \
    int oracle_project_rows_##sfx(const isls_project_args *a);                                        \
    int oracle_sls_admm_##sfx(const isls_sls_admm_args *a);                                           \
    int oracle_sls_closed_loop_##sfx(int32_t M, int32_t N, int32_t n, int32_t m, const void *A,       \
                                     const void *B, const void *K, const void *k, const void *x0,     \
                                     void *x_log, void *u_log);                                       \
    int oracle_dense_closed_loop_##sfx(const isls_dense_loop_args *a);                                \
    int oracle_ilqr_admm_outer_##sfx(const isls_outer_args *a);
ORACLE_DECL(f64)
ORACLE_DECL(f32)
int oracle_set_threads(int n); /* sets the OpenMP team size (n>0) and returns the current maximum */
#ifdef __cplusplus
}
#endif
#endif
