// project.hip -- isls_project_rows_*: row-wise projections / project_set_convex for P problems x R rows
// (isls/projections.py; device code in projections.hpp).  One workgroup per problem, one thread per row; the
// reference stops all rows of a call together, so every inner iteration ends with a workgroup-wide max.
#include "projections.hpp"

namespace isls {

template <typename T>
struct ProjP {
    int P, R, nsets, max_iter, algorithm;
    T rho, threshold;
    int kind[kMaxSets], dim[kMaxSets];
    const T *A[kMaxSets], *b[kMaxSets], *par[kMaxSets];
    int64_t A_sp[kMaxSets], b_sp[kMaxSets], par_sp[kMaxSets];
    const T *y_in;
    T *y_out;
    int64_t in_sp, in_sr, out_sp, out_sr;
    int32_t *iters;
    const int32_t *active;
    const int32_t *row_mask;
};

// BLOCK = largest workgroup the instance is launched with: up to 256 threads (R <= 256, the usual case: rows = time
// steps or N*m) a wave may use the whole register file; 1024-thread workgroups are capped at 128 VGPRs.
template <typename T, int D, int BLOCK>
__global__ __launch_bounds__(BLOCK, (BLOCK <= 256 ? 2 : 1)) void project_rows_kernel(ProjP<T> p)
{
    __shared__ T red[2][2][16];                                // [parity][a / b][wavefront]
    const int pb = blockIdx.x, r = threadIdx.x;
    if (p.active != nullptr && p.active[pb] == 0) return;      // uniform per workgroup
    const bool inr = r < p.R;
    const bool row = inr && (p.row_mask == nullptr || p.row_mask[inr ? r : 0] != 0);   // rows outside the mask pass through
    T x0[D], x[D];
    const T *src = p.y_in + (int64_t)pb * p.in_sp + (int64_t)(inr ? r : 0) * p.in_sr;
#pragma unroll
    for (int j = 0; j < D; ++j) x0[j] = src[j];
    CSet<T> sets[kMaxSets];
#pragma unroll
    for (int s = 0; s < kMaxSets; ++s) {
        sets[s].kind = p.kind[s];
        sets[s].dim = p.dim[s];
        sets[s].A = p.A[s] ? p.A[s] + (int64_t)pb * p.A_sp[s] : nullptr;
        sets[s].b = p.b[s] ? p.b[s] + (int64_t)pb * p.b_sp[s] : nullptr;
        sets[s].par = p.par[s] ? p.par[s] + (int64_t)pb * p.par_sp[s] : nullptr;
    }
#ifdef ISLS_PROJECT_SET_STAGE                                      // opt-in here: measured 7 % slower for config 4's rows (151 vs 141 us:
                                                                   // the fp64 kernel spills more); sls_admm.hip has it on (+13-19 %)
    __shared__ T set_lds[kMaxSets * kSetLdsWords];
    CSetLds<T> lsets[kMaxSets];
    stage_sets_lds<T>(sets, lsets, p.nsets, D, set_lds);
#else
    CSet<T> (&lsets)[kMaxSets] = sets;
#endif
    int it = 0;
    if (p.algorithm == ISLS_PROJ_ALG_ADMM && p.nsets == 1 && sets[0].A == nullptr) {   // direct primitive
        T v[kMaxSetDim];
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) v[i] = i < D ? x0[i < D ? i : 0] : T(0);
        project_primitive<T>(sets[0].kind, D, sets[0].par, v);
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = v[j];
    } else {
        const int nw = (blockDim.x + 63) >> 6, wid = r >> 6;
        // Workgroup-wide maxima once per inner iteration: two barriers per call (-DISLS_ONE_BARRIER: alternating buffers, one
        // barrier, none for a single wavefront -- measured slower, see sls_admm.hip).
        int par = 0;
        auto block_max = [&](T &a, T &b) {
            if (!row) { a = T(0); b = T(0); }
            a = wave_max(a);
            b = wave_max(b);
#ifdef ISLS_ONE_BARRIER
            if (nw == 1) return;
#endif
            if ((r & 63) == 0) { red[par][0][wid] = a; red[par][1][wid] = b; }
            __syncthreads();
            T ma = red[par][0][0], mb = red[par][1][0];
            for (int w = 1; w < nw; ++w) { ma = red[par][0][w] > ma ? red[par][0][w] : ma; mb = red[par][1][w] > mb ? red[par][1][w] : mb; }
#ifndef ISLS_ONE_BARRIER                                           // default: second barrier (see the comment at `par`)
            __syncthreads();
#else
            par ^= 1;
#endif
            a = ma;
            b = mb;
        };
        if (p.algorithm == ISLS_PROJ_ALG_DYKSTRA)
            it = dykstra_row<T, D>(x0, p.nsets, lsets, p.max_iter, p.threshold, x, block_max);
        else if (p.algorithm == ISLS_PROJ_ALG_SOC)
            it = project_soc_row<T, D>(x0, lsets[0], p.rho, p.max_iter, p.threshold, x, block_max);
        else
            it = project_set_convex_row<T, D>(x0, p.nsets, lsets, p.rho, p.max_iter, p.threshold, x, block_max);
    }
    if (inr) {
        T *dst = p.y_out + (int64_t)pb * p.out_sp + (int64_t)r * p.out_sr;
#pragma unroll
        for (int j = 0; j < D; ++j) dst[j] = row ? x[j] : x0[j];
    }
    if (r == 0 && p.iters) p.iters[pb] = it;
}

template <typename T>
int launch_project(const isls_project_args &a, hipStream_t s)
{
    if (a.P < 0 || a.R < 1 || a.R > 1024 || a.d < 1 || a.d > kMaxRowDim || a.nsets < 1 || a.nsets > kMaxSets) return ISLS_ERR_ARG;
    if (!a.y_in || !a.y_out) return ISLS_ERR_ARG;
    if (a.algorithm < ISLS_PROJ_ALG_ADMM || a.algorithm > ISLS_PROJ_ALG_SOC) return ISLS_ERR_UNSUPPORTED;
    const bool dyk = a.algorithm == ISLS_PROJ_ALG_DYKSTRA;
    const bool direct = a.algorithm == ISLS_PROJ_ALG_ADMM && a.nsets == 1 && a.sets[0].A == nullptr;
    for (int i = 0; i < a.nsets; ++i) {
        const isls_cset &c = a.sets[i];
        if (c.kind < ISLS_SET_BOX || c.kind > ISLS_SET_MULTILINEAR) return ISLS_ERR_UNSUPPORTED;
        if (c.dim < 1 || c.dim > kMaxSetDim) return ISLS_ERR_ARG;
        if (!direct && !dyk && (!c.A || !c.b)) return ISLS_ERR_ARG;
        if (c.kind != ISLS_SET_SOC_UNIT && !c.par) return ISLS_ERR_ARG;
        if ((direct || dyk) && c.dim != a.d) return ISLS_ERR_ARG;      // the set acts on the row itself
    }
    if (a.algorithm == ISLS_PROJ_ALG_SOC && (a.nsets != 1 || a.sets[0].kind != ISLS_SET_SOC_UNIT)) return ISLS_ERR_ARG;
    if (!direct && a.max_iter < 1) return ISLS_ERR_ARG;
    if (!direct && !dyk && !(a.rho > 0)) return ISLS_ERR_ARG;
    if (a.P == 0) return ISLS_OK;
    ProjP<T> p = {};
    p.P = a.P; p.R = a.R; p.nsets = a.nsets; p.max_iter = a.max_iter; p.algorithm = a.algorithm;
    p.rho = (T)a.rho; p.threshold = (T)a.threshold;
    for (int i = 0; i < kMaxSets; ++i) {
        const bool on = i < a.nsets;
        p.kind[i] = on ? a.sets[i].kind : 0;
        p.dim[i] = on ? a.sets[i].dim : 0;
        p.A[i] = on ? (const T *)a.sets[i].A : nullptr;
        p.b[i] = on ? (const T *)a.sets[i].b : nullptr;
        p.par[i] = on ? (const T *)a.sets[i].par : nullptr;
        p.A_sp[i] = on ? a.sets[i].A_sp : 0; p.b_sp[i] = on ? a.sets[i].b_sp : 0; p.par_sp[i] = on ? a.sets[i].par_sp : 0;
    }
    p.y_in = (const T *)a.y_in; p.y_out = (T *)a.y_out;
    p.in_sp = a.in_sp; p.in_sr = a.in_sr; p.out_sp = a.out_sp; p.out_sr = a.out_sr;
    p.iters = a.iters; p.active = a.active; p.row_mask = a.row_mask;
    const int threads = ((a.R + 63) / 64) * 64;
#define CALL(D_)                                                                                              \
    if (threads <= 256) hipLaunchKernelGGL((project_rows_kernel<T, D_, 256>), dim3(a.P), dim3(threads), 0, s, p); \
    else if (threads <= 512) hipLaunchKernelGGL((project_rows_kernel<T, D_, 512>), dim3(a.P), dim3(threads), 0, s, p); /* 256 VGPRs: no spills at fp64 */ \
    else hipLaunchKernelGGL((project_rows_kernel<T, D_, 1024>), dim3(a.P), dim3(threads), 0, s, p)
    switch (a.d) {
        case 1: CALL(1); break;
        case 2: CALL(2); break;
        case 3: CALL(3); break;
        default: CALL(4); break;
    }
#undef CALL
    int rc = check_launch();
    if (rc == ISLS_OK && a.next) {                             // next stage, in place on this stage's output
        isls_project_args nx = *a.next;
        nx.P = a.P; nx.R = a.R; nx.d = a.d;
        nx.y_in = a.y_out; nx.y_out = a.y_out;
        nx.in_sp = nx.out_sp = a.out_sp; nx.in_sr = nx.out_sr = a.out_sr;
        nx.iters = nullptr; nx.active = a.active;
        rc = launch_project<T>(nx, s);
    }
    return rc;
}
template int launch_project<double>(const isls_project_args &, hipStream_t);
template int launch_project<float>(const isls_project_args &, hipStream_t);

}  // namespace isls
