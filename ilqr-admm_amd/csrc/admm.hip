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
};

template <typename T>
__device__ __forceinline__ void admm_block(int N, int d, int proj, T relax, const T *x, T *z, T *l,
                                           const View<T> &lo, const View<T> &hi, int b, T &p2, T &d2)
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
        admm_block<T>(p.N, p.n, p.proj_x, p.relax, p.xx + o, p.zx + o, p.lx + o, p.x_lo, p.x_hi, b, p2, d2);
        prim += sqrt(p2); dual += sqrt(d2);
    }
    if (p.zu) {
        const int64_t o = (int64_t)b * p.N * p.m;
        admm_block<T>(p.N, p.m, p.proj_u, p.relax, p.xu + o, p.zu + o, p.lu + o, p.u_lo, p.u_hi, b, p2, d2);
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
    if ((a.proj_x != ISLS_PROJ_NONE && a.proj_x != ISLS_PROJ_BOX) || (a.proj_u != ISLS_PROJ_NONE && a.proj_u != ISLS_PROJ_BOX))
        return ISLS_ERR_UNSUPPORTED;
    if (a.B == 0) return ISLS_OK;
    AdmmP<T> p;
    p.B = a.B; p.N = a.N; p.n = a.n; p.m = a.m; p.proj_x = a.proj_x; p.proj_u = a.proj_u;
    p.relax = (T)a.relax; p.tol_abs = (T)a.tol_abs; p.tol_rel = (T)a.tol_rel;
    p.xx = (const T *)a.xx; p.xu = (const T *)a.xu;
    p.zx = (T *)a.zx; p.lx = (T *)a.lx; p.zu = (T *)a.zu; p.lu = (T *)a.lu;
    p.x_lo = View<T>(a.x_lo); p.x_hi = View<T>(a.x_hi); p.u_lo = View<T>(a.u_lo); p.u_hi = View<T>(a.u_hi);
    p.res = (T *)a.res; p.res_prev = (T *)a.res_prev; p.active = a.active; p.iters = a.iters;
    hipLaunchKernelGGL((admm_update_kernel<T>), dim3(a.B), dim3(64), 0, s, p);
    return check_launch();
}
template int launch_admm<double>(const isls_admm_args &, hipStream_t);
template int launch_admm<float>(const isls_admm_args &, hipStream_t);

}  // namespace isls
