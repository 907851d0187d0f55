// generic.hip -- the recursive kernels for ANY (x_dim, u_dim) with n <= 16, m <= 8 (gfx950).
//
// The fast kernels (riccati.hip, riccati_ff*.hip, rollout_kernel.hpp) are templates over (n, m): a lane owns a matrix row in
// registers, which stops at n + m ~ 12 and needs one instantiation per pair (isls_common.hpp ISLS_FOR_EACH_DIMS).  The
// reference takes any dimensions (isls/base.py:11-14), so every other pair runs here: the dimensions are run-time values,
// the matrices of a step live in LDS, a workgroup (one 64-lane wavefront) owns ONE trajectory and its lanes share the
// entries of each product.  Same formulas, same order of operations as the fast kernels' array form (include/isls_hip.h;
// isls/isls.py:229-334, isls/sls.py:85-202); slower (one trajectory per wavefront, a barrier per product), never wrong.
// Not served here: the packed step records and what rides on them (first feed-forward pass inside the gain pass, time-parallel
// segments, ADMM update fused into the rollout) -- the launchers fall back to the plain sequence -- and the models with a
// fixed state dimension (they have their own instantiations).
#include "isls_common.hpp"

namespace isls {

constexpr int kGenMaxN = 16, kGenMaxM = 8;

bool dims_generic(int n, int m) { return n >= 1 && m >= 1 && n <= kGenMaxN && m <= kGenMaxM; }

// ------------------------------------------------------------------------------------------------------------------------
// Gain pass (isls/isls.py:245-308, isls/sls.py:98-162): Qxx = Cxx + (A'V)A, Qux = Cux + (B'V)A, Quu = Cuu + (B'V)B,
// K = -Quu^-1 Qux, V = Qxx + K'QuuK + Qux'K + K'Qux.
// ------------------------------------------------------------------------------------------------------------------------
template <typename T>
struct GenGainP {
    int B, N, n, m, mode;
    View<T> A, Bm, Cxx, Cuu, Cux;
    T *K, *Quu, *fac, *Qux;
    int32_t *status;
    const int32_t *active;
};

template <typename T>
__global__ __launch_bounds__(64) void gain_generic_kernel(GenGainP<T> p)
{
    constexpr int MN = kGenMaxN, MM = kGenMaxM;
    __shared__ T V[MN * MN], Vn[MN * MN], As[MN * MN], Bs[MN * MM], AtV[MN * MN], BtV[MM * MN];
    __shared__ T Qxx[MN * MN], Qux[MM * MN], Quu[MM * MM], U[MM * MM], rd[MM], inv[MM * MM], Kt[MM * MN];
    __shared__ int bad;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (p.active && p.active[b] == 0) return;
    const int N = p.N, n = p.n, m = p.m;
    const int64_t bN = (int64_t)b * N;
    for (int e = tid; e < m * n; e += kWave) { p.K[(bN + N - 1) * m * n + e] = T(0); p.Qux[(bN + N - 1) * m * n + e] = T(0); }
    for (int e = tid; e < m * m; e += kWave) { p.Quu[(bN + N - 1) * m * m + e] = T(0); p.fac[(bN + N - 1) * m * m + e] = T(0); }
    {
        const T *C = p.Cxx.at(b, N - 1);
        for (int e = tid; e < n * n; e += kWave) V[e] = C[e];
    }
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int t = N - 2; t >= 0; --t) {
        const T *A = p.A.at(b, t), *Bm = p.Bm.at(b, t);
        for (int e = tid; e < n * n; e += kWave) As[e] = A[e];
        for (int e = tid; e < n * m; e += kWave) Bs[e] = Bm[e];
        __syncthreads();
        for (int e = tid; e < n * n; e += kWave) {                 // A'V
            const int i = e / n, j = e - i * n;
            T s = T(0);
            for (int k = 0; k < n; ++k) s += As[k * n + i] * V[k * n + j];
            AtV[e] = s;
        }
        for (int e = tid; e < m * n; e += kWave) {                 // B'V
            const int r = e / n, j = e - r * n;
            T s = T(0);
            for (int k = 0; k < n; ++k) s += Bs[k * m + r] * V[k * n + j];
            BtV[e] = s;
        }
        __syncthreads();
        const T *Cxx = p.Cxx.at(b, t), *Cuu = p.Cuu.at(b, t), *Cux = p.Cux.p ? p.Cux.at(b, t) : nullptr;
        for (int e = tid; e < n * n; e += kWave) {
            const int i = e / n, j = e - i * n;
            T s = T(0);
            for (int k = 0; k < n; ++k) s += AtV[i * n + k] * As[k * n + j];
            Qxx[e] = Cxx[e] + s;
        }
        for (int e = tid; e < m * n; e += kWave) {
            const int r = e / n, j = e - r * n;
            T s = T(0);
            for (int k = 0; k < n; ++k) s += BtV[r * n + k] * As[k * n + j];
            Qux[e] = (Cux ? Cux[e] : T(0)) + s;
        }
        for (int e = tid; e < m * m; e += kWave) {
            const int r = e / m, c = e - r * m;
            T s = T(0);
            for (int k = 0; k < n; ++k) s += BtV[r * n + k] * Bs[k * m + c];
            Quu[e] = Cuu[e] + s;
        }
        __syncthreads();
        if (tid == 0) {                                            // upper Cholesky Quu = U'U (column order of dpotf2 'U')
            for (int j = 0; j < m; ++j) {
                T ajj = Quu[j * m + j];
                for (int k = 0; k < j; ++k) ajj -= U[k * m + j] * U[k * m + j];
                if (!(ajj > T(0))) bad = 1;
                const T d = sqrt(ajj);
                U[j * m + j] = d;
                rd[j] = T(1) / d;
                for (int c = j + 1; c < m; ++c) {
                    T s = Quu[j * m + c];
                    for (int k = 0; k < j; ++k) s -= U[k * m + j] * U[k * m + c];
                    U[j * m + c] = s * rd[j];
                }
            }
        }
        __syncthreads();
        if (bad) break;                                            // uniform: LinAlgError of the reference (isls.py:296)
        // x = (U'U)^-1 rhs for the n columns of Qux (CHOL) or the m unit vectors (INV: Quu^-1 column by column)
        const int ncols = p.mode == ISLS_SOLVE_CHOL ? n : m;
        if (tid < ncols) {
            T y[MM], x[MM];
            for (int i = 0; i < m; ++i) {
                T s = p.mode == ISLS_SOLVE_CHOL ? Qux[i * n + tid] : (i == tid ? T(1) : T(0));
                for (int k = 0; k < i; ++k) s -= U[k * m + i] * y[k];
                y[i] = s * rd[i];
            }
            for (int i = m - 1; i >= 0; --i) {
                T s = y[i];
                for (int k = i + 1; k < m; ++k) s -= U[i * m + k] * x[k];
                x[i] = s * rd[i];
            }
            for (int r = 0; r < m; ++r) {
                if (p.mode == ISLS_SOLVE_CHOL) Kt[r * n + tid] = -x[r];
                else inv[r * m + tid] = x[r];
            }
        }
        __syncthreads();
        if (p.mode != ISLS_SOLVE_CHOL) {                           // Kt = -Quu_inv.dot(Qux)   (sls.py:150)
            for (int e = tid; e < m * n; e += kWave) {
                const int r = e / n, j = e - r * n;
                T s = T(0);
                for (int c = 0; c < m; ++c) s += inv[r * m + c] * Qux[c * n + j];
                Kt[e] = -s;
            }
            __syncthreads();
        }
        const int64_t o = bN + t;
        for (int e = tid; e < m * n; e += kWave) { p.K[o * m * n + e] = Kt[e]; p.Qux[o * m * n + e] = Qux[e]; }
        for (int e = tid; e < m * m; e += kWave) {
            const int r = e / m, c = e - r * m;
            p.Quu[o * m * m + e] = Quu[e];
            // CHOL: upper factor with the diagonal stored as 1/U_ii, strict lower part zero; INV: Quu^-1
            p.fac[o * m * m + e] = p.mode == ISLS_SOLVE_CHOL ? (r == c ? rd[r] : (c > r ? U[e] : T(0))) : inv[e];
        }
        for (int e = tid; e < n * n; e += kWave) {                 // V = Qxx + (K'Quu)K + Qux'K + K'Qux   (isls.py:300 / sls.py:153)
            const int i = e / n, j = e - i * n;
            T t1 = T(0), t2 = T(0), t3 = T(0);
            for (int c = 0; c < m; ++c) {
                T w = T(0);
                for (int r = 0; r < m; ++r) w += Kt[r * n + i] * Quu[r * m + c];
                t1 += w * Kt[c * n + j];
            }
            for (int r = 0; r < m; ++r) {
                t2 += Qux[r * n + i] * Kt[r * n + j];
                t3 += Kt[r * n + i] * Qux[r * n + j];
            }
            Vn[e] = p.mode == ISLS_SOLVE_CHOL ? ((Qxx[e] + t1) + t2) + t3 : ((Qxx[e] + t2) + t3) + t1;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += kWave) V[e] = Vn[e];
        __syncthreads();
    }
    if (tid == 0 && bad && p.status) atomicOr(&p.status[b], ISLS_ST_NOT_PD);
}

template <typename T>
int launch_gain_generic(const isls_gain_args &a, hipStream_t s)
{
    if (!dims_generic(a.n, a.m)) return ISLS_ERR_UNSUPPORTED;
    if (a.rec || !a.Quu || !a.fac || !a.Qux) return ISLS_ERR_UNSUPPORTED;      // array form only: no packed records here
    GenGainP<T> p;
    p.B = a.B; p.N = a.N; p.n = a.n; p.m = a.m; p.mode = a.solve_mode;
    p.A = View<T>(a.A); p.Bm = View<T>(a.Bm); p.Cxx = View<T>(a.Cxx); p.Cuu = View<T>(a.Cuu); p.Cux = View<T>(a.Cux);
    p.K = (T *)a.K; p.Quu = (T *)a.Quu; p.fac = (T *)a.fac; p.Qux = (T *)a.Qux;
    p.status = a.status; p.active = a.active;
    hipLaunchKernelGGL((gain_generic_kernel<T>), dim3(a.B), dim3(64), 0, s, p);
    return check_launch();
}
template int launch_gain_generic<double>(const isls_gain_args &, hipStream_t);
template int launch_gain_generic<float>(const isls_gain_args &, hipStream_t);

// ------------------------------------------------------------------------------------------------------------------------
// Feed-forward pass (isls/isls.py:285-302, isls/sls.py:168-202) on the arrays of the gain pass.
// ------------------------------------------------------------------------------------------------------------------------
template <typename T>
struct GenFfP {
    int B, N, n, m, mode;
    View<T> A, Bm, c0x, c0u, Qr, Rr;
    const T *xhat, *uhat, *zx, *lx, *zu, *lu;
    const T *K, *Quu, *fac, *Qux;
    T *k;
    const int32_t *active;
};

// c_i = c0_i + 2 sum_j W[i,j] (hat_j - (z_j - l_j))   (the regulariser of isls/sls.py:132-137; W absent -> c0_i)
template <typename T>
__device__ __forceinline__ T gen_reg_grad(const View<T> &c0, const View<T> &W, const T *hat, const T *z, const T *l, int b, int t, int N,
                                          int d, int i)
{
    T c = c0.at(b, t)[i];
    if (W.p) {
        const T *w = W.at(b, t) + i * d;
        const int64_t o = ((int64_t)b * N + t) * d;
        T s = T(0);
        for (int j = 0; j < d; ++j) s += w[j] * ((hat ? hat[o + j] : T(0)) - (z[o + j] - l[o + j]));
        c += T(2) * s;
    }
    return c;
}

template <typename T>
__global__ __launch_bounds__(64) void ff_generic_kernel(GenFfP<T> p)
{
    constexpr int MN = kGenMaxN, MM = kGenMaxM;
    __shared__ T v[MN], vn[MN], cx[MN], cu[MM], qu[MM], kt[MM];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (p.active && p.active[b] == 0) return;
    const int N = p.N, n = p.n, m = p.m;
    const int64_t bN = (int64_t)b * N;
    if (tid < m) p.k[(bN + N - 1) * m + tid] = T(0);
    if (tid < n) v[tid] = gen_reg_grad(p.c0x, p.Qr, p.xhat, p.zx, p.lx, b, N - 1, N, n, tid);
    __syncthreads();
    for (int t = N - 2; t >= 0; --t) {
        const T *A = p.A.at(b, t), *Bm = p.Bm.at(b, t);
        const int64_t o = bN + t;
        const T *K = p.K + o * m * n, *Quu = p.Quu + o * m * m, *fac = p.fac + o * m * m, *Qux = p.Qux + o * m * n;
        if (tid < n) cx[tid] = gen_reg_grad(p.c0x, p.Qr, p.xhat, p.zx, p.lx, b, t, N, n, tid);
        else if (tid < n + m) {
            const int r = tid - n;
            T s = T(0);
            for (int k = 0; k < n; ++k) s += Bm[k * m + r] * v[k];
            const T c = gen_reg_grad(p.c0u, p.Rr, p.uhat, p.zu, p.lu, b, t, N, m, r);
            cu[r] = c;
            qu[r] = c + s;                                         // qu = cu + B'v
        }
        __syncthreads();
        if (tid == 0) {                                            // k = -Quu^-1 qu from the cached factor
            if (p.mode == ISLS_SOLVE_CHOL) {
                T y[MM], x[MM];
                for (int i = 0; i < m; ++i) {
                    T s = qu[i];
                    for (int k = 0; k < i; ++k) s -= fac[k * m + i] * y[k];
                    y[i] = s * fac[i * m + i];                     // the diagonal holds 1 / U_ii
                }
                for (int i = m - 1; i >= 0; --i) {
                    T s = y[i];
                    for (int k = i + 1; k < m; ++k) s -= fac[i * m + k] * x[k];
                    x[i] = s * fac[i * m + i];
                }
                for (int i = 0; i < m; ++i) kt[i] = -x[i];
            } else {
                for (int i = 0; i < m; ++i) {
                    T s = T(0);
                    for (int j = 0; j < m; ++j) s += fac[i * m + j] * qu[j];
                    kt[i] = -s;
                }
            }
        }
        __syncthreads();
        if (tid < n) {                                             // v = qx + K'qu + K'Quu k + Qux'k   (isls.py:302 / sls.py:200)
            const int i = tid;
            T s = T(0);
            for (int k = 0; k < n; ++k) s += A[k * n + i] * v[k];
            const T qx = cx[i] + s;
            T t_kqu = T(0), t_kquuk = T(0), t_quxk = T(0);
            for (int r = 0; r < m; ++r) t_kqu += K[r * n + i] * qu[r];
            for (int c = 0; c < m; ++c) {
                T kq = T(0);
                for (int r = 0; r < m; ++r) kq += K[r * n + i] * Quu[r * m + c];
                t_kquuk += kq * kt[c];
            }
            for (int r = 0; r < m; ++r) t_quxk += Qux[r * n + i] * kt[r];
            vn[i] = p.mode == ISLS_SOLVE_CHOL ? ((qx + t_kqu) + t_kquuk) + t_quxk : ((qx + t_quxk) + t_kqu) + t_kquuk;
        } else if (tid < n + m) {
            p.k[o * m + (tid - n)] = kt[tid - n];
        }
        __syncthreads();
        if (tid < n) v[tid] = vn[tid];
        __syncthreads();
    }
}

template <typename T>
int launch_ff_generic(const isls_ff_args &a, hipStream_t s)
{
    if (!dims_generic(a.n, a.m)) return ISLS_ERR_UNSUPPORTED;
    if (a.rec || a.lin_on || a._pad > 1 || a.Qr_term || !a.A.p || !a.Bm.p || !a.K || !a.Quu || !a.fac || !a.Qux) return ISLS_ERR_UNSUPPORTED;
    GenFfP<T> p;
    p.B = a.B; p.N = a.N; p.n = a.n; p.m = a.m; p.mode = a.solve_mode;
    p.A = View<T>(a.A); p.Bm = View<T>(a.Bm); p.c0x = View<T>(a.c0x); p.c0u = View<T>(a.c0u); p.Qr = View<T>(a.Qr); p.Rr = View<T>(a.Rr);
    p.xhat = (const T *)a.xhat; p.uhat = (const T *)a.uhat;
    p.zx = (const T *)a.zx; p.lx = (const T *)a.lx; p.zu = (const T *)a.zu; p.lu = (const T *)a.lu;
    p.K = (const T *)a.K; p.Quu = (const T *)a.Quu; p.fac = (const T *)a.fac; p.Qux = (const T *)a.Qux;
    p.k = (T *)a.k; p.active = a.active;
    hipLaunchKernelGGL((ff_generic_kernel<T>), dim3(a.B), dim3(64), 0, s, p);   // sequential over the horizon (a.seg is not used)
    return check_launch();
}
template int launch_ff_generic<double>(const isls_ff_args &, hipStream_t);
template int launch_ff_generic<float>(const isls_ff_args &, hipStream_t);

// ------------------------------------------------------------------------------------------------------------------------
// Line-search rollout (isls/isls.py:310-334, 357-369; cost isls/sls_base.py:25-44; AL terms isls/isls.py:471-477) for the
// dimension-free models (dense LTI, double integrator) and the via-point cost.  Lane = candidate.
// ------------------------------------------------------------------------------------------------------------------------
template <typename T>
struct GenRoP {
    int B, N, n, m, L, flags, model;
    const T *par;
    int64_t par_sb;
    const T *K, *k, *xhat, *uhat, *x0, *alphas, *Qtab, *ztab;
    int64_t Qtab_sb, ztab_sb;
    const int32_t *seq, *qnz;
    T u_std;
    View<T> wq, wr;
    const T *zx, *lx, *zu, *lu, *cost_cur;
    T *cost_all, *cost_new, *x_out, *u_out;
    int32_t *best, *status;
    const int32_t *active;
};

template <typename T>
__device__ __forceinline__ void gen_model_step(int model, const T *par, int n, int m, const T (&x)[kGenMaxN], const T (&u)[kGenMaxM],
                                               T (&xn)[kGenMaxN])
{
    if (model == ISLS_MODEL_DI) {
        const int d = n / 2;
#pragma unroll
        for (int i = 0; i < kGenMaxM; ++i)
            if (i < d) {
                T xv = T(0);                                       // x[d + i] by selects: a run-time index would send x to scratch
#pragma unroll
                for (int j = 0; j < kGenMaxN; ++j) xv = (j == d + i) ? x[j] : xv;
                xn[i] = (x[i] + par[0] * xv) + par[1] * u[i];
            }
#pragma unroll
        for (int i = 0; i < kGenMaxN; ++i)
            if (i >= d && i < n) {
                T ui = T(0);
#pragma unroll
                for (int r = 0; r < kGenMaxM; ++r) ui = (r == i - d) ? u[r] : ui;
                xn[i] = x[i] + par[2] * ui;
            }
        return;
    }
    const T *A = par, *Bm = par + n * n;
#pragma unroll
    for (int i = 0; i < kGenMaxN; ++i)
        if (i < n) {
            T s = T(0), r = T(0);
#pragma unroll
            for (int j = 0; j < kGenMaxN; ++j)
                if (j < n) s += A[i * n + j] * x[j];
#pragma unroll
            for (int j = 0; j < kGenMaxM; ++j)
                if (j < m) r += Bm[i * m + j] * u[j];
            xn[i] = s + r;
        }
}

// one closed-loop rollout of candidate `alpha`; WRITE: store x_t, u_t (the winner), else accumulate the costs
template <typename T, bool WRITE>
__device__ __forceinline__ void gen_rollout(const GenRoP<T> &p, int b, T alpha, T &plain, T &aug)
{
    constexpr int MN = kGenMaxN, MM = kGenMaxM;
    const int N = p.N, n = p.n, m = p.m;
    const int64_t bN = (int64_t)b * N;
    const bool absolute = (p.flags & ISLS_RO_ABSOLUTE) != 0;
    const T *xh = (!absolute && p.xhat) ? p.xhat + bN * n : nullptr, *uh = (!absolute && p.uhat) ? p.uhat + bN * m : nullptr;
    const T *par = p.par + (int64_t)b * p.par_sb;
    const T *Qtab = p.Qtab + (int64_t)b * p.Qtab_sb, *ztab = p.ztab + (int64_t)b * p.ztab_sb;
    const T *x0 = p.x0 ? p.x0 + (int64_t)b * n : p.xhat + bN * n;
    T x[MN], xn[MN], u[MM];
#pragma unroll
    for (int j = 0; j < MN; ++j) { x[j] = j < n ? x0[j] : T(0); xn[j] = T(0); }
    T cst = T(0), cu = T(0), ag = T(0);
    for (int t = 0; t < N; ++t) {
        const T *K = p.K + (bN + t) * m * n, *kk = p.k + (bN + t) * m;
#pragma unroll
        for (int r = 0; r < MM; ++r) {
            u[r] = T(0);
            if (r < m) {
                T s = T(0);
#pragma unroll
                for (int j = 0; j < MN; ++j)
                    if (j < n) s += (xh ? x[j] - xh[t * n + j] : x[j]) * K[r * n + j];
                u[r] = (s + alpha * kk[r]) + (uh ? uh[t * m + r] : T(0));
            }
        }
        if constexpr (WRITE) {
#pragma unroll
            for (int j = 0; j < MN; ++j)
                if (j < n) p.x_out[(bN + t) * n + j] = x[j];
#pragma unroll
            for (int r = 0; r < MM; ++r)
                if (r < m) p.u_out[(bN + t) * m + r] = u[r];
        } else {
            if (!p.qnz || p.qnz[t] != 0) {                          // (x - z)'Q(x - z)
                const T *Q = Qtab + (int64_t)p.seq[t] * n * n, *z = ztab + (int64_t)p.seq[t] * n;
#pragma unroll
                for (int i = 0; i < MN; ++i)
                    if (i < n) {
                        T s = T(0);
#pragma unroll
                        for (int j = 0; j < MN; ++j)
                            if (j < n) s += Q[i * n + j] * (x[j] - z[j]);
                        cst += (x[i] - z[i]) * s;
                    }
            }
#pragma unroll
            for (int r = 0; r < MM; ++r)
                if (r < m) cu += u[r] * (p.u_std * u[r]);
            if (p.wq.p) {
                const T *w = p.wq.at(b, t), *z = p.zx + (bN + t) * n, *l = p.lx + (bN + t) * n;
#pragma unroll
                for (int j = 0; j < MN; ++j)
                    if (j < n) { const T d = x[j] - (z[j] - l[j]); ag += (d * d) * w[j]; }
            }
            if (p.wr.p) {
                const T *w = p.wr.at(b, t), *z = p.zu + (bN + t) * m, *l = p.lu + (bN + t) * m;
#pragma unroll
                for (int r = 0; r < MM; ++r)
                    if (r < m) { const T d = u[r] - (z[r] - l[r]); ag += (d * d) * w[r]; }
            }
        }
        gen_model_step(p.model, par, n, m, x, u, xn);
#pragma unroll
        for (int j = 0; j < MN; ++j) x[j] = xn[j];
    }
    plain = cst + cu;                                              // sum over x, then += sum over u (sls_base.py:33-39)
    aug = plain + ag;
}

template <typename T>
__global__ __launch_bounds__(64) void rollout_generic_kernel(GenRoP<T> p)
{
    __shared__ T c_aug[64], c_pln[64];
    const int b = blockIdx.x, c = threadIdx.x, L = p.L;
    if (p.active && p.active[b] == 0) return;
    const bool absolute = (p.flags & ISLS_RO_ABSOLUTE) != 0;
    T plain = T(0), aug = T(0);
    if (c < L) gen_rollout<T, false>(p, b, absolute ? T(1) : p.alphas[c], plain, aug);
    c_aug[c] = aug;
    c_pln[c] = plain;
    __syncthreads();
    const bool nan_rule = (p.flags & ISLS_RO_NAN_TO_1E5) != 0;
    int ind = 0;
    bool nan_seen = false;
    T bestv = T(0), bestp = T(0);
    for (int l = 0; l < L; ++l) {                                  // np.argmin: the first NaN wins, else the first minimum
        T v = c_aug[l], pl = c_pln[l];
        const bool isn = v != v;
        nan_seen = nan_seen || isn;
        v = (isn && nan_rule) ? T(1e5) : v;
        pl = (isn && nan_rule) ? T(1e5) : pl;
        const bool take = l == 0 || (!(bestv != bestv) && (v != v || v < bestv));
        bestv = take ? v : bestv;
        bestp = take ? pl : bestp;
        ind = take ? l : ind;
    }
    if (c < L && p.cost_all) p.cost_all[(int64_t)b * L + c] = (aug != aug && nan_rule) ? T(1e5) : aug;
    bool accept = true;
    if (p.flags & ISLS_RO_ACCEPT_TEST) accept = (bestp - p.cost_cur[b]) < T(0);
    if (c == 0) {
        if (p.best) p.best[b] = ind;
        if (p.cost_new) p.cost_new[b] = accept ? bestp : p.cost_cur[b];
        if (p.status) {
            const int bits = (nan_seen ? ISLS_ST_NAN_COST : 0) | (accept ? 0 : ISLS_ST_LS_REJECT);
            if (bits) atomicOr(&p.status[b], bits);
        }
        if (accept) {                                              // x_noms[ind]: one more rollout of the winner
            T d0, d1;
            gen_rollout<T, true>(p, b, absolute ? T(1) : p.alphas[ind], d0, d1);
        }
    }
    if (!accept) {                                                 // the nominal is kept (isls.py:365-369)
        const int64_t bN = (int64_t)b * p.N;
        for (int e = c; e < p.N * p.n; e += kWave) p.x_out[bN * p.n + e] = p.xhat[bN * p.n + e];
        for (int e = c; e < p.N * p.m; e += kWave) p.u_out[bN * p.m + e] = p.uhat[bN * p.m + e];
    }
}

template <typename T>
int launch_rollout_generic(const isls_rollout_args &a, hipStream_t s)
{
    if (!dims_generic(a.n, a.m)) return ISLS_ERR_UNSUPPORTED;
    if (a.model != ISLS_MODEL_LTI && a.model != ISLS_MODEL_DI) return ISLS_ERR_UNSUPPORTED;   // the other models have fixed dimensions
    if (a.model == ISLS_MODEL_DI && (a.n != 2 * a.m)) return ISLS_ERR_UNSUPPORTED;
    if (a.cost_model != ISLS_COST_VIA) return ISLS_ERR_UNSUPPORTED;
    GenRoP<T> p;
    p.B = a.B; p.N = a.N; p.n = a.n; p.m = a.m; p.L = a.L; p.flags = a.flags; p.model = a.model;
    p.par = (const T *)a.model_par; p.par_sb = a.model_par_sb;
    p.K = (const T *)a.K; p.k = (const T *)a.k; p.xhat = (const T *)a.xhat; p.uhat = (const T *)a.uhat;
    p.x0 = (const T *)a.x0; p.alphas = (const T *)a.alphas;
    p.Qtab = (const T *)a.Qtab; p.ztab = (const T *)a.ztab; p.Qtab_sb = a.Qtab_sb; p.ztab_sb = a.ztab_sb;
    p.seq = a.seq; p.qnz = a.q_nonzero; p.u_std = (T)a.u_std;
    p.wq = View<T>(a.wq); p.wr = View<T>(a.wr);
    p.zx = (const T *)a.zx; p.lx = (const T *)a.lx; p.zu = (const T *)a.zu; p.lu = (const T *)a.lu;
    p.cost_cur = (const T *)a.cost_cur;
    p.cost_all = (T *)a.cost_all; p.cost_new = (T *)a.cost_new; p.x_out = (T *)a.x_out; p.u_out = (T *)a.u_out;
    p.best = a.best; p.status = a.status; p.active = a.active;
    hipLaunchKernelGGL((rollout_generic_kernel<T>), dim3(a.B), dim3(64), 0, s, p);
    return check_launch();
}
template int launch_rollout_generic<double>(const isls_rollout_args &, hipStream_t);
template int launch_rollout_generic<float>(const isls_rollout_args &, hipStream_t);

}  // namespace isls
