// rollout.hip -- dispatcher of the line-search rollout (kernel template: rollout_kernel.hpp; one translation unit per
// (n, m, model) family: rollout_family.hip compiled with -DISLS_FAM_*), plus the Monte-Carlo closed loop of a dense controller.
#include "rollout_kernel.hpp"

namespace isls {

ISLS_ROLLOUT_FAMILY_DECL(4, 2, ISLS_MODEL_LTI)
ISLS_ROLLOUT_FAMILY_DECL(4, 2, ISLS_MODEL_CAR)
ISLS_ROLLOUT_FAMILY_DECL(4, 2, ISLS_MODEL_DI)
ISLS_ROLLOUT_FAMILY_DECL(4, 2, ISLS_MODEL_TASSA)
ISLS_ROLLOUT_FAMILY_DECL(9, 3, ISLS_MODEL_LTI)
ISLS_ROLLOUT_FAMILY_DECL(9, 3, ISLS_MODEL_ARM3R)
ISLS_ROLLOUT_FAMILY_DECL(6, 3, ISLS_MODEL_LTI)
ISLS_ROLLOUT_FAMILY_DECL(6, 3, ISLS_MODEL_DI)
ISLS_ROLLOUT_FAMILY_DECL(2, 1, ISLS_MODEL_LTI)
ISLS_ROLLOUT_FAMILY_DECL(2, 1, ISLS_MODEL_DI)
ISLS_ROLLOUT_FAMILY_DECL(3, 1, ISLS_MODEL_LTI)
ISLS_ROLLOUT_FAMILY_DECL(6, 2, ISLS_MODEL_LTI)
ISLS_ROLLOUT_FAMILY_DECL(2, 2, ISLS_MODEL_LTI)
ISLS_ROLLOUT_FAMILY_DECL(3, 3, ISLS_MODEL_LTI)

template <typename T>
int launch_rollout(const isls_rollout_args &a, hipStream_t s, const isls_admm_args *fused, bool *did_fuse, bool last)
{
    if (did_fuse) *did_fuse = false;
    if (a.B < 0 || a.N < 1 || a.L < 1 || a.L > 64) return ISLS_ERR_ARG;
    if (!a.model_par || !a.K || !a.k || !a.alphas || !a.x_out || !a.u_out) return ISLS_ERR_ARG;
    if (a.cost_model == ISLS_COST_VIA && (!a.Qtab || !a.ztab || !a.seq)) return ISLS_ERR_ARG;
    if (!(a.flags & ISLS_RO_ABSOLUTE) && (!a.xhat || !a.uhat)) return ISLS_ERR_ARG;
    if ((a.flags & ISLS_RO_ABSOLUTE) && !a.x0) return ISLS_ERR_ARG;
    if ((a.flags & ISLS_RO_ACCEPT_TEST) && !a.cost_cur) return ISLS_ERR_ARG;
    if (a.wq.p && (!a.zx || !a.lx)) return ISLS_ERR_ARG;
    if (a.wr.p && (!a.zu || !a.lu)) return ISLS_ERR_ARG;
    if (a.B == 0) return ISLS_OK;
    if (!dims_supported(a.n, a.m)) return launch_rollout_generic<T>(a, s);     // generic.hip (no fused ADMM update: *did_fuse stays false)
    RoP<T> p;
    p.B = a.B; p.N = a.N; p.L = a.L; p.flags = a.flags;
    p.par = (const T *)a.model_par; p.par_sb = a.model_par_sb;
    p.K = (const T *)a.K; p.k = (const T *)a.k; p.xhat = (const T *)a.xhat; p.uhat = (const T *)a.uhat;
    p.x0 = (const T *)a.x0; p.alphas = (const T *)a.alphas;
    p.Qtab = (const T *)a.Qtab; p.ztab = (const T *)a.ztab; p.Qtab_sb = a.Qtab_sb; p.ztab_sb = a.ztab_sb;
    p.seq = a.seq; p.qnz = a.q_nonzero;
    p.u_std = (T)a.u_std;
    p.wq = View<T>(a.wq); p.wr = View<T>(a.wr);
    p.zx = (const T *)a.zx; p.lx = (const T *)a.lx; p.zu = (const T *)a.zu; p.lu = (const T *)a.lu;
    p.cost_cur = (const T *)a.cost_cur;
    p.cost_all = (T *)a.cost_all; p.cost_new = (T *)a.cost_new; p.x_out = (T *)a.x_out; p.u_out = (T *)a.u_out;
    p.best = a.best; p.status = a.status; p.active = a.active;
    p.cost_model = a.cost_model; p.cpar = (const T *)a.cost_par;
    if (a.cost_model != ISLS_COST_VIA && (a.cost_model != ISLS_COST_PHUBER || a.model != ISLS_MODEL_TASSA || !a.cost_par))
        return ISLS_ERR_UNSUPPORTED;
    p.nseg = 1; p.seg_len = a.N; p.stage_on = 0;             // set by the family launcher
    p.fa_on = 0;
    p.fa_last = last ? 1 : 0;
    p.fa_zx = p.fa_lx = p.fa_zu = p.fa_lu = p.fa_res = p.fa_res_prev = nullptr;
    p.fa_active = p.fa_iters = nullptr;
    p.fa_proj_x = p.fa_proj_u = 0;
    p.fa_relax = p.fa_tol_abs = p.fa_tol_rel = T(0);
    if (fused) {                                               // validated by the caller (rollout_can_fuse_admm)
        const isls_admm_args &f = *fused;
        p.fa_on = 1; p.fa_proj_x = f.proj_x; p.fa_proj_u = f.proj_u;
        p.fa_relax = (T)f.relax; p.fa_tol_abs = (T)f.tol_abs; p.fa_tol_rel = (T)f.tol_rel;
        p.fa_zx = (T *)f.zx; p.fa_lx = (T *)f.lx; p.fa_zu = (T *)f.zu; p.fa_lu = (T *)f.lu;
        p.fa_xlo = View<T>(f.x_lo); p.fa_xhi = View<T>(f.x_hi); p.fa_ulo = View<T>(f.u_lo); p.fa_uhi = View<T>(f.u_hi);
        p.fa_res = (T *)f.res; p.fa_res_prev = (T *)f.res_prev; p.fa_active = f.active; p.fa_iters = f.iters;
    }
    int rc = ISLS_ERR_UNSUPPORTED;
#define FAMILY(NX_, NU_, MODEL_) \
    if (a.n == NX_ && a.m == NU_ && a.model == MODEL_) rc = launch_rollout_family<T, NX_, NU_, MODEL_>(p, a, s, fused != nullptr, nullptr);
    FAMILY(4, 2, ISLS_MODEL_LTI)
    FAMILY(4, 2, ISLS_MODEL_CAR)
    FAMILY(4, 2, ISLS_MODEL_DI)
    FAMILY(4, 2, ISLS_MODEL_TASSA)
    FAMILY(9, 3, ISLS_MODEL_LTI)
    FAMILY(9, 3, ISLS_MODEL_ARM3R)
    FAMILY(6, 3, ISLS_MODEL_LTI)
    FAMILY(6, 3, ISLS_MODEL_DI)
    FAMILY(2, 1, ISLS_MODEL_LTI)
    FAMILY(2, 1, ISLS_MODEL_DI)
    FAMILY(3, 1, ISLS_MODEL_LTI)
    FAMILY(6, 2, ISLS_MODEL_LTI)
    FAMILY(2, 2, ISLS_MODEL_LTI)
    FAMILY(3, 3, ISLS_MODEL_LTI)
#undef FAMILY
    if (did_fuse) *did_fuse = rc == ISLS_OK && p.fa_on != 0;   // the family launcher drops the fused update when the stage does not fit
    return rc;
}
template int launch_rollout<double>(const isls_rollout_args &, hipStream_t, const isls_admm_args *, bool *, bool);
template int launch_rollout<float>(const isls_rollout_args &, hipStream_t, const isls_admm_args *, bool *, bool);

// ---------------------------------------------------------------------------------------------------------------------
// Monte-Carlo closed loop of a dense causal controller about a nominal (iSLSBase.get_trajectory_sls,
// isls/isls_base.py:28-42): one thread per initial state, the state history is the thread's own x_log row.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
struct DenseLoopP {
    int M, N;
    const T *par, *K, *k, *xhat, *uhat, *x0;
    T *x_log, *u_log;
};

template <typename T, int NX, int NU, int MODEL>
__global__ __launch_bounds__(64) void dense_closed_loop_kernel(DenseLoopP<T> p)
{
    extern __shared__ __align__(16) unsigned char dl_smem[];
    Model<T, NX, NU, MODEL> mdl;
    mdl.load(p.par, reinterpret_cast<T *>(dl_smem), threadIdx.x, 64);
    __syncthreads();
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= p.M) return;
    const int N = p.N;
    T *xs = p.x_log + (int64_t)s * N * NX, *us = p.u_log + (int64_t)s * N * NU;
    T x[NX], u[NU], xn[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = p.x0[(int64_t)s * NX + j];
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < NX; ++j) xs[i * NX + j] = x[j];
#pragma unroll
        for (int r = 0; r < NU; ++r) {
            const T *Kr = p.K + (int64_t)(i * NU + r) * N * NX;
            T acc = T(0);
            for (int j = 0; j < (i + 1) * NX; ++j) acc += (xs[j] - (p.xhat ? p.xhat[j] : T(0))) * Kr[j];
            u[r] = (acc + p.k[i * NU + r]) + (p.uhat ? p.uhat[i * NU + r] : T(0));
            us[i * NU + r] = u[r];
        }
        mdl.step(x, u, xn);
#pragma unroll
        for (int j = 0; j < NX; ++j) x[j] = xn[j];
    }
}

template <typename T>
int launch_dense_closed_loop(const isls_dense_loop_args &a, hipStream_t s)
{
    if (a.M < 0 || a.N < 1 || !a.model_par || !a.K || !a.k || !a.x0 || !a.x_log || !a.u_log) return ISLS_ERR_ARG;
    if (a.M == 0) return ISLS_OK;
    DenseLoopP<T> p;
    p.M = a.M; p.N = a.N;
    p.par = (const T *)a.model_par; p.K = (const T *)a.K; p.k = (const T *)a.k;
    p.xhat = (const T *)a.xhat; p.uhat = (const T *)a.uhat; p.x0 = (const T *)a.x0;
    p.x_log = (T *)a.x_log; p.u_log = (T *)a.u_log;
    const int grid = (a.M + 63) / 64;
#define LAUNCH(NX_, NU_, MODEL_)                                                                                          \
    hipLaunchKernelGGL((dense_closed_loop_kernel<T, NX_, NU_, MODEL_>), dim3(grid), dim3(64),                              \
                       sizeof(T) * (Model<T, NX_, NU_, MODEL_>::LDS_WORDS + 1), s, p)
    if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_LTI) LAUNCH(4, 2, ISLS_MODEL_LTI);
    else if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_CAR) LAUNCH(4, 2, ISLS_MODEL_CAR);
    else if (a.n == 9 && a.m == 3 && a.model == ISLS_MODEL_LTI) LAUNCH(9, 3, ISLS_MODEL_LTI);
    else if (a.n == 9 && a.m == 3 && a.model == ISLS_MODEL_ARM3R) LAUNCH(9, 3, ISLS_MODEL_ARM3R);
    else if (a.n == 6 && a.m == 3 && a.model == ISLS_MODEL_LTI) LAUNCH(6, 3, ISLS_MODEL_LTI);
    else if (a.n == 2 && a.m == 1 && a.model == ISLS_MODEL_LTI) LAUNCH(2, 1, ISLS_MODEL_LTI);
    else if (a.n == 6 && a.m == 3 && a.model == ISLS_MODEL_DI) LAUNCH(6, 3, ISLS_MODEL_DI);
    else if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_DI) LAUNCH(4, 2, ISLS_MODEL_DI);
    else if (a.n == 2 && a.m == 1 && a.model == ISLS_MODEL_DI) LAUNCH(2, 1, ISLS_MODEL_DI);
    else if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_TASSA) LAUNCH(4, 2, ISLS_MODEL_TASSA);
    else if (a.n == 3 && a.m == 1 && a.model == ISLS_MODEL_LTI) LAUNCH(3, 1, ISLS_MODEL_LTI);
    else if (a.n == 6 && a.m == 2 && a.model == ISLS_MODEL_LTI) LAUNCH(6, 2, ISLS_MODEL_LTI);
    else if (a.n == 2 && a.m == 2 && a.model == ISLS_MODEL_LTI) LAUNCH(2, 2, ISLS_MODEL_LTI);
    else if (a.n == 3 && a.m == 3 && a.model == ISLS_MODEL_LTI) LAUNCH(3, 3, ISLS_MODEL_LTI);
    else return ISLS_ERR_UNSUPPORTED;
#undef LAUNCH
    return check_launch();
}
template int launch_dense_closed_loop<double>(const isls_dense_loop_args &, hipStream_t);
template int launch_dense_closed_loop<float>(const isls_dense_loop_args &, hipStream_t);

// The ADMM update can ride on the winner replay when it is the plain element-wise form (no set projections), works
// on the arrays this rollout writes, shares its active mask and no acceptance test can keep the old nominal.
bool rollout_can_fuse_admm(const isls_rollout_args &r, const isls_admm_args &a)
{
    if (r.flags & (ISLS_RO_ACCEPT_TEST | ISLS_RO_ABSOLUTE)) return false;
    if (a.B != r.B || a.N != r.N || a.n != r.n || a.m != r.m || !a.res) return false;
    if (a.xx != r.x_out || a.xu != r.u_out || a.active != r.active) return false;
    if ((a.zx && a.proj_x == ISLS_PROJ_SETS) || (a.zu && a.proj_u == ISLS_PROJ_SETS)) return false;
    if (a.zx && (!a.lx || (a.proj_x == ISLS_PROJ_BOX && (!a.x_lo.p || !a.x_hi.p)))) return false;
    if (a.zu && (!a.lu || (a.proj_u == ISLS_PROJ_BOX && (!a.u_lo.p || !a.u_hi.p)))) return false;
    return true;
}

}  // namespace isls
