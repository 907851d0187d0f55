// sls_admm.hip -- isls_sls_admm_*: the whole ADMM loop of SLS.ADMM_SLS (isls/sls.py:319-454, control chance
// constraints) for P problems in one launch.
//
// One workgroup per problem, one thread per row y = [d_u, phi_u] of the (N*m) x (1+p) variable.  Nothing but the
// shared (N*m)^2 inverse (L2-resident: 20-180 KB) and the problem's right-hand side is read from HBM; z, lambda and
// the iterates live in registers for all iterations, the right-hand side of the x-step goes through LDS.  The
// x-step is a (N*m)^2 x (1+p) product per problem and iteration (10-90 kflop): far below anything worth an MFMA
// tile -- the time goes into the project_set_convex inner iterations (projections.hpp), which are per-row work.
#include "projections.hpp"

namespace isls {

template <typename T>
struct SlsP {
    int P, R, max_iter, nsets, inner_max_iter;
    T alpha, tol, rel_tol, rho, threshold;
    const T *Linv, *r_side, *rr;
    int kind[kMaxSets], dim[kMaxSets];
    const T *A[kMaxSets], *b[kMaxSets], *par[kMaxSets];
    int64_t A_sp[kMaxSets], b_sp[kMaxSets], par_sp[kMaxSets];
    T *x_u, *z, *lmb, *logs;
    int32_t *iters;
};

// Occupancy is the lever of this kernel (one workgroup per problem, the time in the per-row projection iterations): the
// 256-thread form is held to 256 registers at fp64 (two workgroups per CU: measured 153 -> 268 it/s at DI-3D when the
// projection code had grown past that line) and, for rows of three or four entries, to 128 at fp32 (four wavefronts per SIMD).
template <typename T, int D, int BLOCK>
__global__ __launch_bounds__(BLOCK, (BLOCK <= 256 ? (sizeof(T) == 4 ? (D >= 3 ? 4 : 1) : 2) : 1)) void sls_admm_kernel(SlsP<T> p)
{
    extern __shared__ __align__(16) unsigned char sls_smem[];
    T *rhs = reinterpret_cast<T *>(sls_smem);                  // [R][D]
    __shared__ T red[2][2][16];                                // [parity][a / b][wavefront]
    const int pb = blockIdx.x, r = threadIdx.x, R = p.R;
    const bool row = r < R;
    const int rr_i = row ? r : 0;
    const int nw = (blockDim.x + 63) >> 6, wid = r >> 6;
    CSet<T> sets[kMaxSets];
#pragma unroll
    for (int s = 0; s < kMaxSets; ++s) {
        sets[s].kind = p.kind[s];
        sets[s].dim = p.dim[s];
        sets[s].A = p.A[s] ? p.A[s] + (int64_t)pb * p.A_sp[s] : nullptr;
        sets[s].b = p.b[s] ? p.b[s] + (int64_t)pb * p.b_sp[s] : nullptr;
        sets[s].par = p.par[s] ? p.par[s] + (int64_t)pb * p.par_sp[s] : nullptr;
    }
#ifndef ISLS_NO_SET_STAGE                                          // set operands in LDS behind typed pointers (stage_sets_lds)
    __shared__ T set_lds[kMaxSets * kSetLdsWords];
    CSetLds<T> lsets[kMaxSets];
    stage_sets_lds<T>(sets, lsets, p.nsets, D, set_lds);
#else
    CSet<T> (&lsets)[kMaxSets] = sets;
#endif
    // workgroup-wide max (project_set_convex: once per inner iteration) and sum (residual norms); idle threads contribute
    // zeros.  Two barriers per call (partials visible; buffer free again).  -DISLS_ONE_BARRIER builds the form with alternating
    // buffers (one barrier per call, none for a one-wavefront problem): measured SLOWER on MI355X (config 5, B = 8192: DI-1D
    // fp32 7505 vs 7794 it/s, DI-3D 3162 vs 3391; isls_admm's row projection 156 vs 150 us) -- the second barrier keeps the
    // wavefronts of a workgroup in step through the set projections, whose loads then hit the same lines together.
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
#ifndef ISLS_ONE_BARRIER
        __syncthreads();
#else
        par ^= 1;
#endif
        a = ma;
        b = mb;
    };
    auto block_sum = [&](T &a, T &b) {
        if (!row) { a = T(0); b = T(0); }
        a = wave_sum(a);
        b = wave_sum(b);
#ifdef ISLS_ONE_BARRIER
        if (nw == 1) return;
#endif
        if ((r & 63) == 0) { red[par][0][wid] = a; red[par][1][wid] = b; }
        __syncthreads();
        T sa = T(0), sb = T(0);
        for (int w = 0; w < nw; ++w) { sa += red[par][0][w]; sb += red[par][1][w]; }
#ifndef ISLS_ONE_BARRIER
        __syncthreads();
#else
        par ^= 1;
#endif
        a = sa;
        b = sb;
    };

    const T *rs = p.r_side + ((int64_t)pb * R + rr_i) * D;
    const T wgt = p.rr[rr_i];
    T rside[D], x[D], z[D], lmb[D];
#pragma unroll
    for (int c = 0; c < D; ++c) { rside[c] = rs[c]; x[c] = T(0); z[c] = T(0); lmb[c] = T(0); }
    T prim = T(1e6), dual = T(1e6);
    int it = 0;
#ifdef ISLS_DIAG
    unsigned long long cyc_x = 0, cyc_p = 0, cyc_r = 0, t0_ = __builtin_readcyclecounter();
#define SLS_STAMP(acc) { const unsigned long long t1_ = __builtin_readcyclecounter(); acc += t1_ - t0_; t0_ = t1_; }
#else
#define SLS_STAMP(acc)
#endif
    for (int j = 0; j < p.max_iter; ++j) {
        ++it;
        // x-step: x_u = Linv (r_side + Rr (z - lmb))                                   (sls.py:375-384)
        if (row) {
#pragma unroll
            for (int c = 0; c < D; ++c) rhs[r * D + c] = rside[c] + wgt * (z[c] - lmb[c]);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < D; ++c) x[c] = T(0);
        // (the loads of a chunk are issued together: a wavefront is parked on these L2 round trips for most of its life --
        // rocprofv3 SQ_WAIT_ANY 60 % of the wave cycles, profiles/r03_config5_pmc.json -- and one wait per element cost R of them)
        constexpr int XC = 8;
        for (int k0 = 0; k0 < R; k0 += XC) {
            T l[XC];
#pragma unroll
            for (int q = 0; q < XC; ++q) {
                const int k = k0 + q < R ? k0 + q : R - 1;     // clamped (surplus products are multiplied by zero)
                l[q] = p.Linv[(int64_t)k * R + rr_i];          // symmetric: column rr_i of row k, coalesced over the rows
            }
#pragma unroll
            for (int q = 0; q < XC; ++q) {
                const int k = k0 + q < R ? k0 + q : R - 1;
                const T lq = k0 + q < R ? l[q] : T(0);
#pragma unroll
                for (int c = 0; c < D; ++c) x[c] += lq * rhs[k * D + c];
            }
        }
        __syncthreads();
        SLS_STAMP(cyc_x)
        // z-step: z = project_u(alpha x + (1-alpha) z + lmb), all rows of the problem in one project_set_convex call
        T v[D], zn[D];
#pragma unroll
        for (int c = 0; c < D; ++c) v[c] = (p.alpha * x[c] + (T(1) - p.alpha) * z[c]) + lmb[c];
        project_set_convex_row<T, D>(v, p.nsets, lsets, p.rho, p.inner_max_iter, p.threshold, zn, block_max);
        SLS_STAMP(cyc_p)
        const T prev_prim = prim, prev_dual = dual;
        T p2 = T(0), d2 = T(0);
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const T pr = x[c] - zn[c];
            const T dz = zn[c] - z[c];
            lmb[c] += pr;
            z[c] = zn[c];
            p2 += (wgt * pr) * (wgt * pr);
            d2 += (wgt * dz) * (wgt * dz);
        }
        block_sum(p2, d2);
        prim = sqrt(p2);
        dual = sqrt(d2);
        if (r == 0 && p.logs) {
            T *lg = p.logs + ((int64_t)pb * p.max_iter + j) * 2;
            lg[0] = prim;
            lg[1] = dual;
        }
        if (prim < p.tol && dual < p.tol) break;                                         // sls.py:417
        const T pc = fabs(prev_prim - prim) / (prev_prim + T(1e-30));
        const T dc = fabs(prev_dual - dual) / (prev_dual + T(1e-30));
        if (pc < p.rel_tol && dc < p.rel_tol) break;                                     // sls.py:424-430
        SLS_STAMP(cyc_r)
    }
#ifdef ISLS_DIAG
    if (r == 0 && (pb == 0 || pb == 4000))
        printf("sls_admm diag problem %d: %d iterations, cycles x-step %llu projection %llu residual/stop %llu (R=%d D=%d)\n", pb, it,
               cyc_x, cyc_p, cyc_r, R, D);
#endif
    if (row) {
        const int64_t o = ((int64_t)pb * R + r) * D;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            p.x_u[o + c] = x[c];
            if (p.z) p.z[o + c] = z[c];
            if (p.lmb) p.lmb[o + c] = lmb[c];
        }
    }
    if (r == 0 && p.iters) p.iters[pb] = it;
}

// Closed loop of the dense causal controller, one thread per initial state; the state history is the x_log row itself.
template <typename T>
__global__ __launch_bounds__(64) void sls_closed_loop_kernel(int M, int N, int n, int m, const T *__restrict__ A,
                                                             const T *__restrict__ B, const T *__restrict__ K,
                                                             const T *__restrict__ k, const T *__restrict__ x0,
                                                             T *x_log, T *u_log)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= M) return;
    T *xs = x_log + (int64_t)s * N * n, *us = u_log + (int64_t)s * N * m;
    for (int j = 0; j < n; ++j) xs[j] = x0[(int64_t)s * n + j];
    for (int i = 0; i < N; ++i) {
        for (int r = 0; r < m; ++r) {
            const T *Kr = K + (int64_t)(i * m + r) * N * n;
            T acc = T(0);
            for (int j = 0; j < (i + 1) * n; ++j) acc += xs[j] * Kr[j];
            us[i * m + r] = acc + k[i * m + r];
        }
        if (i + 1 < N) {
            for (int a = 0; a < n; ++a) {
                T acc = T(0);
                for (int j = 0; j < n; ++j) acc += A[a * n + j] * xs[i * n + j];
                for (int r = 0; r < m; ++r) acc += B[a * m + r] * us[i * m + r];
                xs[(i + 1) * n + a] = acc;
            }
        }
    }
}

template <typename T>
int launch_sls_closed_loop(int M, int N, int n, int m, const void *A, const void *B, const void *K, const void *k,
                           const void *x0, void *x_log, void *u_log, hipStream_t s)
{
    if (M < 0 || N < 1 || n < 1 || m < 1 || !A || !B || !K || !k || !x0 || !x_log || !u_log) return ISLS_ERR_ARG;
    if (M == 0) return ISLS_OK;
    hipLaunchKernelGGL((sls_closed_loop_kernel<T>), dim3((M + 63) / 64), dim3(64), 0, s, M, N, n, m, (const T *)A, (const T *)B,
                       (const T *)K, (const T *)k, (const T *)x0, (T *)x_log, (T *)u_log);
    return check_launch();
}
template int launch_sls_closed_loop<double>(int, int, int, int, const void *, const void *, const void *, const void *,
                                            const void *, void *, void *, hipStream_t);
template int launch_sls_closed_loop<float>(int, int, int, int, const void *, const void *, const void *, const void *,
                                           const void *, void *, void *, hipStream_t);

template <typename T>
int launch_sls_admm(const isls_sls_admm_args &a, hipStream_t s)
{
    const isls_project_args &pj = a.proj;
    if (a.P < 0 || a.R < 1 || a.R > 1024 || a.D < 1 || a.D > kMaxRowDim || a.max_iter < 1) return ISLS_ERR_ARG;
    if (!a.Linv || !a.r_side || !a.rr || !a.x_u) return ISLS_ERR_ARG;
    if (pj.nsets < 1 || pj.nsets > kMaxSets || pj.max_iter < 1 || !(pj.rho > 0)) return ISLS_ERR_ARG;
    for (int i = 0; i < pj.nsets; ++i) {
        const isls_cset &c = pj.sets[i];
        if (c.kind < ISLS_SET_BOX || c.kind > ISLS_SET_MULTILINEAR) return ISLS_ERR_UNSUPPORTED;
        if (c.dim < 1 || c.dim > kMaxSetDim || !c.A || !c.b) return ISLS_ERR_ARG;
        if (c.kind != ISLS_SET_SOC_UNIT && !c.par) return ISLS_ERR_ARG;
    }
    if (a.P == 0) return ISLS_OK;
    SlsP<T> p = {};
    p.P = a.P; p.R = a.R; p.max_iter = a.max_iter; p.nsets = pj.nsets; p.inner_max_iter = pj.max_iter;
    p.alpha = (T)a.alpha; p.tol = (T)a.tol; p.rel_tol = (T)a.rel_tol; p.rho = (T)pj.rho; p.threshold = (T)pj.threshold;
    p.Linv = (const T *)a.Linv; p.r_side = (const T *)a.r_side; p.rr = (const T *)a.rr;
    for (int i = 0; i < kMaxSets; ++i) {
        const bool on = i < pj.nsets;
        p.kind[i] = on ? pj.sets[i].kind : 0;
        p.dim[i] = on ? pj.sets[i].dim : 0;
        p.A[i] = on ? (const T *)pj.sets[i].A : nullptr;
        p.b[i] = on ? (const T *)pj.sets[i].b : nullptr;
        p.par[i] = on ? (const T *)pj.sets[i].par : nullptr;
        p.A_sp[i] = on ? pj.sets[i].A_sp : 0; p.b_sp[i] = on ? pj.sets[i].b_sp : 0; p.par_sp[i] = on ? pj.sets[i].par_sp : 0;
    }
    p.x_u = (T *)a.x_u; p.z = (T *)a.z; p.lmb = (T *)a.lmb; p.logs = (T *)a.logs; p.iters = a.iters;
    const int threads = ((a.R + 63) / 64) * 64;
    const size_t smem = (size_t)a.R * a.D * sizeof(T);
#define CALL(D_)                                                                                                  \
    if (threads <= 256) hipLaunchKernelGGL((sls_admm_kernel<T, D_, 256>), dim3(a.P), dim3(threads), smem, s, p);  \
    else if (threads <= 512) hipLaunchKernelGGL((sls_admm_kernel<T, D_, 512>), dim3(a.P), dim3(threads), smem, s, p); \
    else hipLaunchKernelGGL((sls_admm_kernel<T, D_, 1024>), dim3(a.P), dim3(threads), smem, s, p)
    switch (a.D) {
        case 1: CALL(1); break;
        case 2: CALL(2); break;
        case 3: CALL(3); break;
        default: CALL(4); break;
    }
#undef CALL
    return check_launch();
}
template int launch_sls_admm<double>(const isls_sls_admm_args &, hipStream_t);
template int launch_sls_admm<float>(const isls_sls_admm_args &, hipStream_t);

}  // namespace isls
