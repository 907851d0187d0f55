// admm.hip -- ADMM consensus / dual update with projection, residual norms and stop rules (gfx950).
//
// Reference semantics: the body of ADMM() after the x-step, isls/admm.py:43-85, with project_x /
// project_u given as descriptors (box = np.clip, isls/projections.py:7-11).
//
// Mapping: pure streaming work (HBM-bound): one 64-lane wavefront per trajectory sweeps the flat
// [N*n] and [N*m] vectors with coalesced 512-byte accesses, keeps the four partial sums of squares in
// registers, reduces them with a butterfly and lets lane 0 evaluate the stop rules of that trajectory.
#include "isls_common.hpp"

namespace isls {

template <typename T>
struct AdmmP {
    int B, N, n, m, proj_x, proj_u;
    T relax, tol_abs, tol_rel;
    const T *xx, *xu;
    T *zx, *lx, *zu, *lu;
    View<T> x_lo, x_hi, u_lo, u_hi;
    T *res, *res_prev;
    int32_t *active, *iters;
    const T *x_work, *u_work;      // ISLS_PROJ_SETS: the projected argument, produced by admm_argument_kernel + project_rows
};

// relax*x + (1-relax)*z + lmb of one block, written out for the set projection (ISLS_PROJ_SETS)
template <typename T>
__global__ __launch_bounds__(64) void admm_argument_kernel(int N, int d, T relax, const T *x, const T *z, const T *l, T *work,
                                                           const int32_t *active)
{
    const int b = blockIdx.x;
    if (active && active[b] == 0) return;
    const int64_t o = (int64_t)b * N * d;
    for (int e = threadIdx.x; e < N * d; e += kWave) work[o + e] = (relax * x[o + e] + (T(1) - relax) * z[o + e]) + l[o + e];
}

template <typename T>
__device__ __forceinline__ void admm_block(int N, int d, int proj, T relax, const T *x, T *z, T *l,
                                           const View<T> &lo, const View<T> &hi, const T *work, int b, T &p2, T &d2)
{
    p2 = T(0); d2 = T(0);
    const int cnt = N * d;
    for (int e = threadIdx.x; e < cnt; e += kWave) {
        const T zp = z[e], xv = x[e], lv = l[e];
        const T arg = (relax * xv + (T(1) - relax) * zp) + lv;      // admm.py:48-49
        T zn = arg;
        if (proj == ISLS_PROJ_BOX) {
            const int t = e / d, i = e - t * d;
            const T lo_v = lo.at(b, t)[i], hi_v = hi.at(b, t)[i];
            zn = arg < lo_v ? lo_v : arg;                            // np.clip
            zn = zn > hi_v ? hi_v : zn;
        } else if (proj == ISLS_PROJ_SETS) {
            zn = work[e];
        }
        const T r = xv - zn;                                         // admm.py:51
        l[e] = lv + r;                                               // admm.py:52
        z[e] = zn;
        p2 += r * r;
        d2 += (zn - zp) * (zn - zp);
    }
    p2 = wave_sum(p2);
    d2 = wave_sum(d2);
}

template <typename T>
__global__ __launch_bounds__(64) void admm_update_kernel(AdmmP<T> p)
{
    const int b = blockIdx.x;
    if (p.active && p.active[b] == 0) return;                        // uniform per workgroup
    T prim = T(0), dual = T(0), p2, d2;
    if (p.zx) {
        const int64_t o = (int64_t)b * p.N * p.n;
        admm_block<T>(p.N, p.n, p.proj_x, p.relax, p.xx + o, p.zx + o, p.lx + o, p.x_lo, p.x_hi, p.x_work ? p.x_work + o : nullptr, b, p2, d2);
        prim += sqrt(p2); dual += sqrt(d2);
    }
    if (p.zu) {
        const int64_t o = (int64_t)b * p.N * p.m;
        admm_block<T>(p.N, p.m, p.proj_u, p.relax, p.xu + o, p.zu + o, p.lu + o, p.u_lo, p.u_hi, p.u_work ? p.u_work + o : nullptr, b, p2, d2);
        prim += sqrt(p2); dual += sqrt(d2);
    }
    if (threadIdx.x == 0) {
        T *res = p.res + (int64_t)b * 2;
        T *prev = p.res_prev ? p.res_prev + (int64_t)b * 2 : nullptr;
        if (p.active && prev) {
            bool stop = false;
            if (prim < p.tol_abs && dual < p.tol_abs) stop = true;                         // admm.py:72
            else {
                const T pc = fabs(prev[0] - prim) / (prev[0] + T(1e-30));                  // admm.py:78-80
                const T dc = fabs(prev[1] - dual) / (prev[1] + T(1e-30));
                stop = pc < p.tol_rel && dc < p.tol_rel;
            }
            if (stop) p.active[b] = 0;
        }
        res[0] = prim; res[1] = dual;
        if (prev) { prev[0] = prim; prev[1] = dual; }
        if (p.iters) p.iters[b] += 1;
    }
}

template <typename T>
int launch_admm(const isls_admm_args &a, hipStream_t s)
{
    if (a.B < 0 || a.N < 1 || a.n < 1 || a.m < 1 || !a.res) return ISLS_ERR_ARG;
    if (a.zx && (!a.lx || !a.xx)) return ISLS_ERR_ARG;
    if (a.zu && (!a.lu || !a.xu)) return ISLS_ERR_ARG;
    if (a.zx && a.proj_x == ISLS_PROJ_BOX && (!a.x_lo.p || !a.x_hi.p)) return ISLS_ERR_ARG;
    if (a.zu && a.proj_u == ISLS_PROJ_BOX && (!a.u_lo.p || !a.u_hi.p)) return ISLS_ERR_ARG;
    if (a.proj_x < ISLS_PROJ_NONE || a.proj_x > ISLS_PROJ_SETS || a.proj_u < ISLS_PROJ_NONE || a.proj_u > ISLS_PROJ_SETS)
        return ISLS_ERR_UNSUPPORTED;
    if (a.zx && a.proj_x == ISLS_PROJ_SETS && (!a.x_sets || !a.x_work || a.x_col0 < 0 || a.x_col0 + a.x_sets->d > a.n)) return ISLS_ERR_ARG;
    if (a.zu && a.proj_u == ISLS_PROJ_SETS && (!a.u_sets || !a.u_work || a.u_col0 < 0 || a.u_col0 + a.u_sets->d > a.m)) return ISLS_ERR_ARG;
    if (a.B == 0) return ISLS_OK;
    // set projections: argument -> scratch, project_set_convex over the time steps of the coordinate block in place
    for (int blk = 0; blk < 2; ++blk) {
        const bool isx = blk == 0;
        if (!(isx ? a.zx : a.zu) || (isx ? a.proj_x : a.proj_u) != ISLS_PROJ_SETS) continue;
        const int d = isx ? a.n : a.m, col0 = isx ? a.x_col0 : a.u_col0;
        T *work = (T *)(isx ? a.x_work : a.u_work);
        hipLaunchKernelGGL((admm_argument_kernel<T>), dim3(a.B), dim3(64), 0, s, a.N, d, (T)a.relax,
                           (const T *)(isx ? a.xx : a.xu), (const T *)(isx ? a.zx : a.zu), (const T *)(isx ? a.lx : a.lu), work,
                           (const int32_t *)a.active);
        isls_project_args pr = *(isx ? a.x_sets : a.u_sets);
        pr.P = a.B; pr.R = a.N;
        pr.y_in = work + col0; pr.y_out = work + col0;
        pr.in_sp = pr.out_sp = (int64_t)a.N * d;
        pr.in_sr = pr.out_sr = d;
        pr.iters = nullptr; pr.active = a.active;
        const int rc = launch_project<T>(pr, s);
        if (rc != ISLS_OK) return rc;
    }
    AdmmP<T> p;
    p.B = a.B; p.N = a.N; p.n = a.n; p.m = a.m; p.proj_x = a.proj_x; p.proj_u = a.proj_u;
    p.relax = (T)a.relax; p.tol_abs = (T)a.tol_abs; p.tol_rel = (T)a.tol_rel;
    p.xx = (const T *)a.xx; p.xu = (const T *)a.xu;
    p.zx = (T *)a.zx; p.lx = (T *)a.lx; p.zu = (T *)a.zu; p.lu = (T *)a.lu;
    p.x_lo = View<T>(a.x_lo); p.x_hi = View<T>(a.x_hi); p.u_lo = View<T>(a.u_lo); p.u_hi = View<T>(a.u_hi);
    p.res = (T *)a.res; p.res_prev = (T *)a.res_prev; p.active = a.active; p.iters = a.iters;
    p.x_work = a.proj_x == ISLS_PROJ_SETS ? (const T *)a.x_work : nullptr;
    p.u_work = a.proj_u == ISLS_PROJ_SETS ? (const T *)a.u_work : nullptr;
    hipLaunchKernelGGL((admm_update_kernel<T>), dim3(a.B), dim3(64), 0, s, p);
    return check_launch();
}
template int launch_admm<double>(const isls_admm_args &, hipStream_t);
template int launch_admm<float>(const isls_admm_args &, hipStream_t);

}  // namespace isls
