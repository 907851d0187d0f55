// misc.hip -- the streaming kernels either side of the recursions (gfx950):
//   expand_quadratic : via-point quadratic cost expansion about the nominal + nominal cost
//                      (isls/isls.py:263-271, isls/sls.py:132-137, isls/sls_base.py:25-44)
//   linearize        : A_t, B_t of the built-in models along the nominal (notebooks' get_AB callbacks)
//   reduce_convergence / outer_begin : bookkeeping of the outer loop (isls/isls.py:414-421, admm.py:25-26)
// All are HBM-bound: one wavefront per trajectory, flat coalesced sweeps over the trajectory's arrays.
#include "isls_common.hpp"

namespace isls {

// ------------------------------------------------------------------------------------------------
template <typename T>
struct ExpP {
    int B, N, n, m;
    const T *Qtab, *ztab;
    int64_t Qtab_sb, ztab_sb;
    const int32_t *seq;
    T u_std;
    View<T> Qr, Rr;
    const T *xhat, *uhat;
    T *Cxx, *Cuu, *c0x, *c0u, *cost;
    const int32_t *active;
    int cost_model;
    const T *cpar;
    const int32_t *qnz;
};

template <typename T>
__device__ __forceinline__ void expand_body(const ExpP<T> &p, int b)
{
    const int N = p.N, n = p.n, m = p.m, lane = threadIdx.x;
    const T *Qtab = p.Qtab + (int64_t)b * p.Qtab_sb, *ztab = p.ztab + (int64_t)b * p.ztab_sb;
    const int64_t bN = (int64_t)b * N;
    T cx_sum = T(0), cu_sum = T(0);
    if (p.cost_model == ISLS_COST_PHUBER) {
        // gradient and (diagonal) Hessian of sum_t [cu.u^2 + cx.ph(x,px)] + cf.ph(x_{N-1},pf), ph = sqrt(x^2+p^2)-p:
        // what Tutorial.ipynb cell 16 obtains from autograd (ph' = x/s, ph'' = p^2/s^3 with s = sqrt(x^2+p^2))
        const T *cu = p.cpar, *cx = cu + m, *px = cx + n, *cf = px + n, *pf = cf + n;
        for (int e = lane; e < N * n; e += kWave) {
            const int t = e / n, i = e - t * n;
            const T x = p.xhat ? p.xhat[bN * n + e] : T(0);
            const T s1 = sqrt(x * x + px[i] * px[i]);
            T g = cx[i] * (x / s1), h = cx[i] * ((px[i] * px[i]) / (s1 * s1 * s1)), c = cx[i] * (s1 - px[i]);
            if (t == N - 1) {
                const T s2 = sqrt(x * x + pf[i] * pf[i]);
                g += cf[i] * (x / s2);
                h += cf[i] * ((pf[i] * pf[i]) / (s2 * s2 * s2));
                c += cf[i] * (s2 - pf[i]);
            }
            p.c0x[bN * n + e] = g;
            cx_sum += c;
            if (p.Cxx) {
                T *row = p.Cxx + (bN + t) * n * n + i * n;
                for (int j = 0; j < n; ++j) row[j] = (j == i ? h : T(0)) + (p.Qr.p ? T(2) * p.Qr.at(b, t)[i * n + j] : T(0));
            }
        }
        for (int e = lane; e < N * m; e += kWave) {
            const int t = e / m, i = e - t * m;
            const T uu = p.uhat ? p.uhat[bN * m + e] : T(0);
            p.c0u[bN * m + e] = T(2) * (cu[i] * uu);
            cu_sum += cu[i] * (uu * uu);
            if (p.Cuu) {
                T *row = p.Cuu + (bN + t) * m * m + i * m;
                for (int j = 0; j < m; ++j) row[j] = (j == i ? T(2) * cu[i] : T(0)) + (p.Rr.p ? T(2) * p.Rr.at(b, t)[i * m + j] : T(0));
            }
        }
        if (p.cost) {
            cx_sum = wave_sum(cx_sum);
            cu_sum = wave_sum(cu_sum);
            if (lane == 0) p.cost[b] = cx_sum + cu_sum;
        }
        return;
    }
    // c0x[t,i] = 2 * sum_j Q_t[i,j] (xhat[t,j] - z_t[j]);   cost_x = sum d_i (Q d)_i
    for (int e = lane; e < N * n; e += kWave) {
        const int t = e / n, i = e - t * n;
        if (p.qnz && p.qnz[t] == 0) {                          // Q_t == 0: gradient and cost term vanish, nothing to read
            p.c0x[bN * n + e] = T(0);
            continue;
        }
        const T *Q = Qtab + (int64_t)p.seq[t] * n * n + i * n, *z = ztab + (int64_t)p.seq[t] * n;
        const T *xh = p.xhat ? p.xhat + (bN + t) * n : nullptr;
        T sacc = T(0);
        for (int j = 0; j < n; ++j) sacc += Q[j] * ((xh ? xh[j] : T(0)) - z[j]);
        p.c0x[bN * n + e] = T(2) * sacc;
        cx_sum += ((xh ? xh[i] : T(0)) - z[i]) * sacc;
    }
    for (int e = lane; e < N * m; e += kWave) {
        const T uu = p.uhat ? p.uhat[bN * m + e] : T(0);
        p.c0u[bN * m + e] = T(2) * (p.u_std * uu);
        cu_sum += uu * (p.u_std * uu);
    }
    if (p.Cxx) {
        for (int e = lane; e < N * n * n; e += kWave) {
            const int t = e / (n * n), r = e - t * n * n;
            T v = T(2) * Qtab[(int64_t)p.seq[t] * n * n + r];
            if (p.Qr.p) v += T(2) * p.Qr.at(b, t)[r];
            p.Cxx[bN * n * n + e] = v;
        }
    }
    if (p.Cuu) {
        for (int e = lane; e < N * m * m; e += kWave) {
            const int t = e / (m * m), r = e - t * m * m;
            T v = ((r / m) == (r % m)) ? T(2) * p.u_std : T(0);
            if (p.Rr.p) v += T(2) * p.Rr.at(b, t)[r];
            p.Cuu[bN * m * m + e] = v;
        }
    }
    if (p.cost) {
        cx_sum = wave_sum(cx_sum);
        cu_sum = wave_sum(cu_sum);
        if (lane == 0) p.cost[b] = cx_sum + cu_sum;
    }
}

template <typename T>
__global__ __launch_bounds__(64) void expand_kernel(ExpP<T> p)
{
    const int b = blockIdx.x;
    if (p.active && p.active[b] == 0) return;
    expand_body(p, b);
}

template <typename T>
static int expand_params(const isls_expand_args &a, ExpP<T> &p)
{
    if (a.B < 0 || a.N < 1 || a.n < 1 || a.m < 1 || !a.c0x || !a.c0u) return ISLS_ERR_ARG;
    if (a.cost_model == ISLS_COST_VIA && (!a.Qtab || !a.ztab || !a.seq)) return ISLS_ERR_ARG;
    if (a.cost_model != ISLS_COST_VIA && (a.cost_model != ISLS_COST_PHUBER || !a.cost_par)) return ISLS_ERR_UNSUPPORTED;
    p.B = a.B; p.N = a.N; p.n = a.n; p.m = a.m;
    p.Qtab = (const T *)a.Qtab; p.ztab = (const T *)a.ztab; p.Qtab_sb = a.Qtab_sb; p.ztab_sb = a.ztab_sb;
    p.seq = a.seq; p.u_std = (T)a.u_std;
    p.Qr = View<T>(a.Qr); p.Rr = View<T>(a.Rr);
    p.xhat = (const T *)a.xhat; p.uhat = (const T *)a.uhat;
    p.Cxx = (T *)a.Cxx; p.Cuu = (T *)a.Cuu; p.c0x = (T *)a.c0x; p.c0u = (T *)a.c0u; p.cost = (T *)a.cost;
    p.active = a.active;
    p.cost_model = a.cost_model; p.cpar = (const T *)a.cost_par; p.qnz = a.q_nonzero;
    return ISLS_OK;
}

template <typename T>
int launch_expand(const isls_expand_args &a, hipStream_t s)
{
    ExpP<T> p;
    const int rc = expand_params<T>(a, p);
    if (rc != ISLS_OK) return rc;
    if (a.B == 0) return ISLS_OK;
    hipLaunchKernelGGL((expand_kernel<T>), dim3(a.B), dim3(64), 0, s, p);
    return check_launch();
}
template int launch_expand<double>(const isls_expand_args &, hipStream_t);
template int launch_expand<float>(const isls_expand_args &, hipStream_t);

// ------------------------------------------------------------------------------------------------
template <typename T>
struct LinP {
    int B, N, n, m, model;
    const T *par;
    int64_t par_sb;
    const T *xhat, *uhat;
    T *A, *Bm;
    const int32_t *active;
};

template <typename T>
__device__ __forceinline__ void linearize_body(const LinP<T> &p, int b, T *tab)   // tab: [N][8] per-step trig terms (LDS)
{
    const int N = p.N, n = p.n, m = p.m, lane = threadIdx.x;
    const T *par = p.par + (int64_t)b * p.par_sb;
    const int64_t bN = (int64_t)b * N;
    T *A = p.A + bN * n * n, *Bm = p.Bm + bN * n * m;
    if (p.model == ISLS_MODEL_LTI || p.model == ISLS_MODEL_DI) {
        // state-independent Jacobians: one step's [A | B] pattern is built in LDS (two periods of each, so that a pair of
        // adjacent words never wraps) and streamed out N times as 16-byte stores -- running pattern indices, no division per
        // element (the element-wise form spent ~100 integer instructions on e % (n n), / n, % n per 8-byte store)
        const int nn = n * n, nm = n * m, d = n / 2;
        T *patA = tab, *patB = tab + 2 * nn;
        for (int e = lane; e < 2 * nn; e += kWave) {
            const int w = e < nn ? e : e - nn, r = w / n, c = w - r * n;
            patA[e] = p.model == ISLS_MODEL_LTI ? par[w] : ((r == c) ? T(1) : ((r < d && c == r + d) ? par[0] : T(0)));
        }
        for (int e = lane; e < 2 * nm; e += kWave) {
            const int w = e < nm ? e : e - nm, r = w / m, c = w - r * m;
            patB[e] = p.model == ISLS_MODEL_LTI ? par[nn + w] : ((r == c) ? par[1] : ((r == c + d) ? par[2] : T(0)));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        typedef T V2 __attribute__((ext_vector_type(2)));
        typedef V2 V2u __attribute__((aligned(sizeof(T))));    // the block of an odd trajectory of an odd-sized matrix is 8-byte aligned
        auto stream = [&](T *dst, const T *pat, int per, int total) {
            const int pairs = total / 2;
            int idx = (2 * lane) % per;
            const int step = (2 * kWave) % per;
            for (int q = lane; q < pairs; q += kWave) {
                V2 v;
                v.x = pat[idx];
                v.y = pat[idx + 1];
                *reinterpret_cast<V2u *>(dst + 2 * q) = v;
                idx += step;
                idx = idx >= per ? idx - per : idx;
            }
            if ((total & 1) && lane == 0) dst[total - 1] = pat[(total - 1) % per];
        };
        stream(A, patA, nn, N * nn);
        stream(Bm, patB, nm, N * nm);
        return;
    }
    const T dt = par[0];
    if (p.model == ISLS_MODEL_ARM3R) {
        // phase 1: J(q + qd dt + u dt^2/2) per step: tab[t] = {J00,J01,J02,J10,J11,J12}
        for (int t = lane; t < N; t += kWave) {
            const T *x = p.xhat + (bN + t) * 9, *u = p.uhat + (bN + t) * 3;
            T c = T(0), sn[3], cs[3];
            for (int j = 0; j < 3; ++j) {
                c += x[j] + x[3 + j] * dt + T(0.5) * u[j] * (dt * dt);
                sin_cos(c, sn[j], cs[j]);
            }
            for (int j = 0; j < 3; ++j) {
                T j0 = T(0), j1 = T(0);
                for (int i = j; i < 3; ++i) { j0 += sn[i]; j1 += cs[i]; }
                tab[t * 8 + j] = -j0;
                tab[t * 8 + 3 + j] = j1;
            }
        }
        __syncthreads();
        // phase 2: flat coalesced fill (3DoF notebooks cell 10)
        for (int e = lane; e < N * 81; e += kWave) {
            const int t = e / 81, r = (e - t * 81) / 9, c = e % 9;
            T v = T(0);
            if (r < 6) v = (r == c) ? T(1) : ((r < 3 && c == r + 3) ? dt : T(0));
            else if (r < 8 && c < 3) v = tab[t * 8 + (r - 6) * 3 + c];
            else if (r < 8 && c < 6) v = tab[t * 8 + (r - 6) * 3 + (c - 3)] * dt;
            A[e] = v;
        }
        for (int e = lane; e < N * 27; e += kWave) {
            const int t = e / 27, r = (e - t * 27) / 3, c = e % 3;
            T v = T(0);
            if (r < 3) v = (r == c) ? T(0.5) * (dt * dt) : T(0);
            else if (r < 6) v = (r - 3 == c) ? dt : T(0);
            else if (r < 8) v = (T(0.5) * tab[t * 8 + (r - 6) * 3 + c]) * (dt * dt);
            Bm[e] = v;
        }
        return;
    }
    if (p.model == ISLS_MODEL_TASSA) {
        // Tutorial.ipynb cell 8 differentiated by hand (the notebook uses autograd):
        //   f = dt v, sw = sin w, cw = cos w, r = sqrt(d^2 - (sw f)^2), b = f cw + d - r
        //   db/df = cw + sw^2 f / r, db/dw = -f sw + sw cw f^2 / r, d(asin(sw f / d))/dw = cw f / r, /df = sw / r
        // tab[t] = {b, db/df, db/dw, sin th, cos th, cw f / r, sw / r}
        const T d = par[1];
        for (int t = lane; t < N; t += kWave) {
            const T *x = p.xhat + (bN + t) * 4, *u = p.uhat + (bN + t) * 2;
            T sw, cw;
            sin_cos(u[0], sw, cw);
            const T f = dt * x[3];
            const T r = sqrt(d * d - (sw * f) * (sw * f));
            tab[t * 8 + 0] = (f * cw + d) - r;
            tab[t * 8 + 1] = cw + (sw * sw * f) / r;
            tab[t * 8 + 2] = -f * sw + (sw * cw * f * f) / r;
            sin_cos(x[2], tab[t * 8 + 3], tab[t * 8 + 4]);
            tab[t * 8 + 5] = (cw * f) / r;
            tab[t * 8 + 6] = sw / r;
        }
        __syncthreads();
        for (int e = lane; e < N * 16; e += kWave) {
            const int t = e / 16, r = (e % 16) / 4, c = e % 4;
            const T *tb = tab + t * 8;
            T v = (r == c) ? T(1) : T(0);
            if (r == 0 && c == 2) v = -tb[0] * tb[3];
            else if (r == 1 && c == 2) v = tb[0] * tb[4];
            else if (r == 0 && c == 3) v = (tb[1] * dt) * tb[4];
            else if (r == 1 && c == 3) v = (tb[1] * dt) * tb[3];
            else if (r == 2 && c == 3) v = tb[6] * dt;
            A[e] = v;
        }
        for (int e = lane; e < N * 8; e += kWave) {
            const int t = e / 8, r = (e % 8) / 2, c = e % 2;
            const T *tb = tab + t * 8;
            T v = T(0);
            if (r == 0 && c == 0) v = tb[2] * tb[4];
            else if (r == 1 && c == 0) v = tb[2] * tb[3];
            else if (r == 2 && c == 0) v = tb[5];
            else if (r == 3 && c == 1) v = dt;
            Bm[e] = v;
        }
        return;
    }
    // car-simple (Car notebooks cell 6): tab[t] = {sin th, cos th, v, u0}
    for (int t = lane; t < N; t += kWave) {
        const T *x = p.xhat + (bN + t) * 4, *u = p.uhat + (bN + t) * 2;
        sin_cos(x[2], tab[t * 8 + 0], tab[t * 8 + 1]); tab[t * 8 + 2] = x[3]; tab[t * 8 + 3] = u[0];
    }
    __syncthreads();
    for (int e = lane; e < N * 16; e += kWave) {
        const int t = e / 16, r = (e % 16) / 4, c = e % 4;
        const T sn = tab[t * 8], cs = tab[t * 8 + 1], v_ = tab[t * 8 + 2], u0 = tab[t * 8 + 3];
        T v = (r == c) ? T(1) : T(0);
        if (r == 0 && c == 2) v = -dt * v_ * sn;
        else if (r == 1 && c == 2) v = dt * v_ * cs;
        else if (r == 0 && c == 3) v = dt * cs;
        else if (r == 1 && c == 3) v = dt * sn;
        else if (r == 2 && c == 3) v = dt * u0;
        A[e] = v;
    }
    for (int e = lane; e < N * 8; e += kWave) {
        const int t = e / 8, r = (e % 8) / 2, c = e % 2;
        T v = T(0);
        if (r == 2 && c == 0) v = dt * tab[t * 8 + 2];
        else if (r == 3 && c == 1) v = dt;
        Bm[e] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(64) void linearize_kernel(LinP<T> p)
{
    extern __shared__ __align__(16) unsigned char lin_smem[];
    const int b = blockIdx.x;
    if (p.active && p.active[b] == 0) return;
    linearize_body(p, b, reinterpret_cast<T *>(lin_smem));
}

template <typename T>
static int linearize_params(const isls_linearize_args &a, LinP<T> &p, size_t *tab_words)
{
    if (a.B < 0 || a.N < 1 || !a.model_par || !a.A || !a.Bm) return ISLS_ERR_ARG;
    if (a.model != ISLS_MODEL_LTI && a.model != ISLS_MODEL_DI && (!a.xhat || !a.uhat)) return ISLS_ERR_ARG;
    if (a.model == ISLS_MODEL_ARM3R && !(a.n == 9 && a.m == 3)) return ISLS_ERR_UNSUPPORTED;
    if (a.model == ISLS_MODEL_CAR && !(a.n == 4 && a.m == 2)) return ISLS_ERR_UNSUPPORTED;
    if (a.model == ISLS_MODEL_DI && !(a.n == 2 * a.m)) return ISLS_ERR_UNSUPPORTED;
    if (a.model == ISLS_MODEL_TASSA && !(a.n == 4 && a.m == 2)) return ISLS_ERR_UNSUPPORTED;
    if (a.model < ISLS_MODEL_LTI || a.model > ISLS_MODEL_TASSA) return ISLS_ERR_UNSUPPORTED;
    if ((size_t)a.N * 8 * sizeof(T) > 60000) return ISLS_ERR_UNSUPPORTED;
    p.B = a.B; p.N = a.N; p.n = a.n; p.m = a.m; p.model = a.model;
    p.par = (const T *)a.model_par; p.par_sb = a.model_par_sb;
    p.xhat = (const T *)a.xhat; p.uhat = (const T *)a.uhat; p.A = (T *)a.A; p.Bm = (T *)a.Bm; p.active = a.active;
    *tab_words = (size_t)a.N * 8 > (size_t)2 * a.n * (a.n + a.m) ? (size_t)a.N * 8 : (size_t)2 * a.n * (a.n + a.m);
    return ISLS_OK;
}

template <typename T>
int launch_linearize(const isls_linearize_args &a, hipStream_t s)
{
    LinP<T> p;
    size_t tab_words = 0;
    const int rc = linearize_params<T>(a, p, &tab_words);
    if (rc != ISLS_OK) return rc;
    if (a.B == 0) return ISLS_OK;
    hipLaunchKernelGGL((linearize_kernel<T>), dim3(a.B), dim3(64), tab_words * sizeof(T), s, p);
    return check_launch();
}
template int launch_linearize<double>(const isls_linearize_args &, hipStream_t);
template int launch_linearize<float>(const isls_linearize_args &, hipStream_t);

// ------------------------------------------------------------------------------------------------
// end of an outer iteration: nominal <- x-step, cost log tail, outer stop rules (isls/isls.py:488-499)
// returns (lane 0 only) whether a stop rule fired
template <typename T>
__device__ __forceinline__ bool accept_body(int b, int N, int n, int m, const T *xx, const T *xu, const T *cost_new,
                                            T *xhat, T *uhat, T *cost, T *hist, int32_t *hist_len,
                                            T tol_cost, T tol_osc, int32_t *outer_active)
{
    const int64_t ox = (int64_t)b * N * n, ou = (int64_t)b * N * m;
    for (int e = threadIdx.x; e < N * n; e += kWave) xhat[ox + e] = xx[ox + e];
    for (int e = threadIdx.x; e < N * m; e += kWave) uhat[ou + e] = xu[ou + e];
    if (threadIdx.x != 0) return false;
    const T prev = cost[b], cur = cost_new[b];
    cost[b] = cur;
    bool stop = false;
    if (tol_cost >= T(0) && fabs(cur - prev) < tol_cost) stop = true;             // isls.py:493
    if (hist) {
        T *h = hist + (int64_t)b * 8;
        int len = hist_len[b];
        if (len < 8) h[len++] = cur;
        else {
            for (int i = 0; i < 7; ++i) h[i] = h[i + 1];
            h[7] = cur;
        }
        hist_len[b] = len;
        if (!stop && tol_osc >= T(0) && len >= 5) {                               // isls.py:497
            T a4 = T(0), b4 = T(0);
            for (int i = len - 4; i < len; ++i) a4 += h[i];
            for (int i = 0; i < len - 4; ++i) b4 += h[i];
            if (fabs(a4 / T(4) - b4 / T(len - 4)) < tol_osc) stop = true;
        }
    }
    if (stop && outer_active) outer_active[b] = 0;
    return stop && outer_active != nullptr;
}

template <typename T>
__global__ __launch_bounds__(64) void accept_kernel(int N, int n, int m, const T *xx, const T *xu, const T *cost_new,
                                                    T *xhat, T *uhat, T *cost, T *hist, int32_t *hist_len,
                                                    T tol_cost, T tol_osc, int32_t *outer_active)
{
    const int b = blockIdx.x;
    if (outer_active && outer_active[b] == 0) return;
    accept_body(b, N, n, m, xx, xu, cost_new, xhat, uhat, cost, hist, hist_len, tol_cost, tol_osc, outer_active);
}

template <typename T>
int launch_accept(const isls_accept_args &a, hipStream_t s)
{
    if (a.B < 0 || a.N < 1 || !a.xx || !a.xu || !a.cost_new || !a.xhat || !a.uhat || !a.cost) return ISLS_ERR_ARG;
    if (a.cost_hist && !a.hist_len) return ISLS_ERR_ARG;
    if (a.B == 0) return ISLS_OK;
    hipLaunchKernelGGL((accept_kernel<T>), dim3(a.B), dim3(64), 0, s, (int)a.N, (int)a.n, (int)a.m, (const T *)a.xx,
                       (const T *)a.xu, (const T *)a.cost_new, (T *)a.xhat, (T *)a.uhat, (T *)a.cost, (T *)a.cost_hist,
                       a.hist_len, (T)a.tol_cost, (T)a.tol_osc, a.outer_active);
    return check_launch();
}
template int launch_accept<double>(const isls_accept_args &, hipStream_t);
template int launch_accept<float>(const isls_accept_args &, hipStream_t);

// ------------------------------------------------------------------------------------------------
// out5 = { sum cost, max prim, max dual, #active, #status!=0 } over the local shard (single workgroup)
template <typename T>
__global__ __launch_bounds__(1024) void reduce_kernel(int B, const T *cost, const T *res, const int32_t *active,
                                                     const int32_t *status, T *out5, int row, int rows)
{
    // table form (row >= 0): out5 is a [rows,5] table, this shard's numbers go to `row`, the other rows are zeroed
    if (row >= 0) {
        for (int e = threadIdx.x; e < rows * 5; e += blockDim.x)
            if (e / 5 != row) out5[e] = T(0);
        out5 += row * 5;
    }
    __shared__ T sm[5][16];                                   // one partial per wavefront of the 1024-thread workgroup
    T cs = T(0), pm = T(0), dm = T(0), na = T(0), nf = T(0);
    // four trajectories per thread and trip, every load of the trip issued before the first use: the kernel is one workgroup
    // of dependent loads, so the trips -- not the bytes -- are its time (the sums keep their order: b ascending per thread)
    constexpr int U = 4;
    for (int b0 = threadIdx.x; b0 < B; b0 += U * blockDim.x) {
        T c[U], r0[U], r1[U];
        int ac[U], st[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int b = b0 + q * (int)blockDim.x, bc = b < B ? b : B - 1;
            c[q] = cost ? cost[bc] : T(0);
            r0[q] = res ? res[2 * bc] : T(0);
            r1[q] = res ? res[2 * bc + 1] : T(0);
            ac[q] = active ? active[bc] : 1;
            st[q] = status ? status[bc] : 0;
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (b0 + q * (int)blockDim.x < B) {
                cs += c[q];
                pm = r0[q] > pm ? r0[q] : pm;
                dm = r1[q] > dm ? r1[q] : dm;
                na += ac[q] != 0 ? T(1) : T(0);
                nf += st[q] != 0 ? T(1) : T(0);
            }
        }
    }
    cs = wave_sum(cs); na = wave_sum(na); nf = wave_sum(nf); pm = wave_max(pm); dm = wave_max(dm);
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) { sm[0][w] = cs; sm[1][w] = pm; sm[2][w] = dm; sm[3][w] = na; sm[4][w] = nf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        T o0 = T(0), o1 = T(0), o2 = T(0), o3 = T(0), o4 = T(0);
        for (int i = 0; i < (int)(blockDim.x / kWave); ++i) {
            o0 += sm[0][i]; o3 += sm[3][i]; o4 += sm[4][i];
            o1 = sm[1][i] > o1 ? sm[1][i] : o1;
            o2 = sm[2][i] > o2 ? sm[2][i] : o2;
        }
        out5[0] = o0; out5[1] = o1; out5[2] = o2; out5[3] = o3; out5[4] = o4;
    }
}

template <typename T>
int launch_reduce(int32_t B, const void *cost, const void *res, const int32_t *active, const int32_t *status,
                  void *out5, hipStream_t s, int row, int rows)
{
    if (B < 0 || !out5 || (row >= 0 && (rows < 1 || row >= rows))) return ISLS_ERR_ARG;
    // latency-bound (5 small loads per trajectory): the widest workgroup keeps the dependent iterations short
    hipLaunchKernelGGL((reduce_kernel<T>), dim3(1), dim3(B > 256 ? 1024 : 256), 0, s, (int)B, (const T *)cost, (const T *)res, active,
                       status, (T *)out5, row, rows);
    return check_launch();
}
template int launch_reduce<double>(int32_t, const void *, const void *, const int32_t *, const int32_t *, void *, hipStream_t, int, int);
template int launch_reduce<float>(int32_t, const void *, const void *, const int32_t *, const int32_t *, void *, hipStream_t, int, int);

// ------------------------------------------------------------------------------------------------
// start of an outer iteration: admm_active <- outer_active, lambda <- 0 (isls.py:414-415,482),
// previous residual norms <- 1e6 (admm.py:25-26), for the trajectories still iterating.
template <typename T>
__device__ __forceinline__ void outer_begin_body(int b, bool act, int N, int n, int m, int32_t *admm_active, T *lx, T *lu,
                                                 T *res_prev, int32_t *iters)
{
    if (threadIdx.x == 0 && admm_active) admm_active[b] = act ? 1 : 0;
    if (!act) return;
    if (threadIdx.x == 0 && iters) iters[b] = 0;
    if (lx) for (int e = threadIdx.x; e < N * n; e += kWave) lx[(int64_t)b * N * n + e] = T(0);
    if (lu) for (int e = threadIdx.x; e < N * m; e += kWave) lu[(int64_t)b * N * m + e] = T(0);
    if (res_prev && threadIdx.x < 2) res_prev[(int64_t)b * 2 + threadIdx.x] = T(1e6);
}

template <typename T>
__global__ __launch_bounds__(64) void outer_begin_kernel(int N, int n, int m, int32_t *admm_active,
                                                         const int32_t *outer_active, T *lx, T *lu, T *res_prev,
                                                         int32_t *iters)
{
    const int b = blockIdx.x;
    outer_begin_body(b, outer_active == nullptr || outer_active[b] != 0, N, n, m, admm_active, lx, lu, res_prev, iters);
}

template <typename T>
int launch_outer_begin(int32_t B, int32_t N, int32_t n, int32_t m, int32_t *admm_active, const int32_t *outer_active,
                       void *lx, void *lu, void *res_prev, int32_t *iters, hipStream_t s)
{
    if (B <= 0) return ISLS_OK;
    hipLaunchKernelGGL((outer_begin_kernel<T>), dim3(B), dim3(64), 0, s, (int)N, (int)n, (int)m, admm_active, outer_active,
                       (T *)lx, (T *)lu, (T *)res_prev, iters);
    return check_launch();
}
template int launch_outer_begin<double>(int32_t, int32_t, int32_t, int32_t, int32_t *, const int32_t *, void *, void *, void *, int32_t *, hipStream_t);
template int launch_outer_begin<float>(int32_t, int32_t, int32_t, int32_t, int32_t *, const int32_t *, void *, void *, void *, int32_t *, hipStream_t);

// ------------------------------------------------------------------------------------------------
// isls_outer_advance: the end of one outer iteration and the start of the next in ONE launch (one wavefront per trajectory):
// accept (nominal <- x-step, cost log, stop rules) -> start of the next iteration (admm_active, lambda, residual history)
// -> linearisation and cost expansion about the new nominal, for the trajectories still iterating.  The four stages were
// four launches that each re-read what the previous one had just written; here the new nominal is read from the x-step
// arrays (the same numbers; nothing this launch writes is read back by it) and the launch gaps are gone.
template <typename T>
struct AdvP {
    int N, n, m;
    const T *xx, *xu, *cost_new;
    T *xhat, *uhat, *cost, *hist;
    int32_t *hist_len;
    T tol_cost, tol_osc;
    int32_t *outer_active;
    int32_t *admm_active, *iters;
    T *lx, *lu, *res_prev;
    int has_lin, has_exp;
    LinP<T> lin;
    ExpP<T> exp;
};

template <typename T>
__global__ __launch_bounds__(64) void advance_kernel(AdvP<T> p)
{
    extern __shared__ __align__(16) unsigned char adv_smem[];
    const int b = blockIdx.x;
    bool act = p.outer_active == nullptr || p.outer_active[b] != 0;
    if (act) {
        const bool stopped = accept_body(b, p.N, p.n, p.m, p.xx, p.xu, p.cost_new, p.xhat, p.uhat, p.cost, p.hist, p.hist_len,
                                         p.tol_cost, p.tol_osc, p.outer_active);
        act = __builtin_amdgcn_readfirstlane(stopped ? 1 : 0) == 0;       // lane 0 evaluated the stop rules
    }
    outer_begin_body(b, act, p.N, p.n, p.m, p.admm_active, p.lx, p.lu, p.res_prev, p.iters);
    if (!act) return;
    if (p.has_lin) linearize_body(p.lin, b, reinterpret_cast<T *>(adv_smem));
    if (p.has_exp) expand_body(p.exp, b);
}

template <typename T>
int launch_advance(const isls_advance_args &a, hipStream_t s)
{
    const isls_accept_args &ac = a.accept;
    if (ac.B < 0 || ac.N < 1 || !ac.xx || !ac.xu || !ac.cost_new || !ac.xhat || !ac.uhat || !ac.cost) return ISLS_ERR_ARG;
    if (ac.cost_hist && !ac.hist_len) return ISLS_ERR_ARG;
    AdvP<T> p;
    p.N = ac.N; p.n = ac.n; p.m = ac.m;
    p.xx = (const T *)ac.xx; p.xu = (const T *)ac.xu; p.cost_new = (const T *)ac.cost_new;
    p.xhat = (T *)ac.xhat; p.uhat = (T *)ac.uhat; p.cost = (T *)ac.cost; p.hist = (T *)ac.cost_hist; p.hist_len = ac.hist_len;
    p.tol_cost = (T)ac.tol_cost; p.tol_osc = (T)ac.tol_osc; p.outer_active = ac.outer_active;
    p.admm_active = a.admm_active; p.iters = a.iters; p.lx = (T *)a.lx; p.lu = (T *)a.lu; p.res_prev = (T *)a.res_prev;
    size_t tab_words = 0;
    p.has_lin = a.lin.A != nullptr;
    p.has_exp = a.exp.c0x != nullptr;
    int rc;
    if (p.has_lin) {
        if ((rc = linearize_params<T>(a.lin, p.lin, &tab_words)) != ISLS_OK) return rc;
        if (a.lin.B != ac.B || a.lin.N != ac.N || a.lin.n != ac.n || a.lin.m != ac.m) return ISLS_ERR_ARG;
        // the nominal this launch writes is not read back by it: the same numbers come from the x-step arrays
        if (p.lin.xhat == p.xhat) p.lin.xhat = p.xx;
        if (p.lin.uhat == p.uhat) p.lin.uhat = p.xu;
    }
    if (p.has_exp) {
        if ((rc = expand_params<T>(a.exp, p.exp)) != ISLS_OK) return rc;
        if (a.exp.B != ac.B || a.exp.N != ac.N || a.exp.n != ac.n || a.exp.m != ac.m) return ISLS_ERR_ARG;
        if (p.exp.xhat == p.xhat) p.exp.xhat = p.xx;
        if (p.exp.uhat == p.uhat) p.exp.uhat = p.xu;
    }
    if (ac.B == 0) return ISLS_OK;
    hipLaunchKernelGGL((advance_kernel<T>), dim3(ac.B), dim3(64), tab_words * sizeof(T), s, p);
    return check_launch();
}
template int launch_advance<double>(const isls_advance_args &, hipStream_t);
template int launch_advance<float>(const isls_advance_args &, hipStream_t);

}  // namespace isls
