// projections.hpp -- device restatement of the row-wise projections of isls/projections.py and of
// project_set_convex (inner ADMM onto an intersection of sets), shared by project.hip and sls_admm.hip.
//
//   ISLS_SET_BOX       np.clip(y, lo, hi)                                   isls/projections.py:7-11
//   ISLS_SET_SOC_UNIT  (z,t) -> ||z|| <= t, project_soc_unit_batch          isls/projections.py:140-162
//   ISLS_SET_SQUARE    l <= ||W(y[:q]-c)||_inf <= u on the first q entries  isls/projections.py:256-266 and the
//                      keep-out rectangles of notebooks/Car/Iterative LQR with state constraints.ipynb cell 18
//   ISLS_SET_LINEAR    l <= a'y <= u, project_linear_batch                  isls/projections.py:30-43
//   ISLS_SET_QUADRATIC l <= y'y/2 <= u, project_quadratic_batch            isls/projections.py:91-105
//   project_set_convex                                                      isls/projections.py:289-374
// Every array of a row lives in registers: all loops run to the compile-time maxima with constant indices and
// are predicated on the runtime dimension (a runtime index into a per-lane array would go to scratch memory).
#pragma once
#include "isls_common.hpp"

namespace isls {

constexpr int kMaxRowDim = ISLS_MAX_ROW_DIM;   // d
constexpr int kMaxSetDim = ISLS_MAX_SET_DIM;   // dim_i of A_i y + b_i
constexpr int kMaxSets = ISLS_MAX_SETS;

// P: pointer type of the operands -- `const T *` (global memory) or the LDS form below (stage_sets_lds)
template <typename T, typename P = const T *>
struct CSet {
    int kind, dim;
    P A, b, par;                   // this problem's A [dim,d], b [dim], parameters
};
template <typename T> using LdsPtr = const __attribute__((address_space(3))) T *;
template <typename T> using CSetLds = CSet<T, LdsPtr<T>>;

// Words of a set's parameter block (include/isls_hip.h ISLS_SET_*): what stage_sets copies
template <typename P>
__device__ __forceinline__ int set_par_words(int kind, int dim, P par)
{
    switch (kind) {
        case ISLS_SET_BOX: return 2 * dim;
        case ISLS_SET_SQUARE: { const int q = (int)par[0]; return 3 + q + 2 * q * q; }
        case ISLS_SET_LINEAR: return 2 + dim;
        case ISLS_SET_QUADRATIC: return 2;
        case ISLS_SET_SHELL: return 2 + dim;
        case ISLS_SET_MULTILINEAR: { const int q = (int)par[0]; return 1 + 2 * q + q * dim; }
        default: return 0;                                     // ISLS_SET_SOC_UNIT: no parameters
    }
}
constexpr int kSetParMax = 3 + kMaxSetDim + 2 * kMaxSetDim * kMaxSetDim;                 // the keep-out square with q = dim
constexpr int kSetLdsWords = kMaxSetDim * kMaxRowDim + kMaxSetDim + kSetParMax;           // A | b | par of one set

// The sets of a problem are the same for all of its rows, and the inner iterations of project_set_convex read A_i, b_i and the
// primitive's parameters again and again: from global memory those are L2 round trips on every row's dependent chain (rocprofv3:
// the wavefronts of config 5 are parked on waits for 60 % of their life).  Every thread of the workgroup calls this once: the
// operands are copied into LDS and described by LDS-typed pointers (CSetLds).  A first form that only repointed the generic
// descriptors turned the reads into flat loads and ran SLOWER than the L2 hits (config 5 DI-1D fp32 7100 vs 7729 it/s, DI-3D
// 2751 vs 3246; config 4's row projection 197 vs 141 us).  With typed pointers (ds_read): config 5 DI-1D fp32 9234 vs 7745
// it/s, fp64 6717 vs 5660, DI-3D fp32 3740 vs 3260, fp64 1585 vs 1403 (same box, interleaved) -- on in sls_admm.hip; config 4's
// row projection 151 vs 141 us -- off in project.hip.  `lds` holds kMaxSets * kSetLdsWords words.
template <typename T>
__device__ __forceinline__ void stage_sets_lds(const CSet<T> (&sets)[kMaxSets], CSetLds<T> (&out)[kMaxSets], int nsets, int D, T *lds)
{
    const int tid = threadIdx.x, nt = blockDim.x;
#pragma unroll
    for (int s = 0; s < kMaxSets; ++s) {
        if (s < nsets) {
            T *base = lds + s * kSetLdsWords;
            const int dim = sets[s].dim;
            if (sets[s].A) for (int e = tid; e < dim * D; e += nt) base[e] = sets[s].A[e];
            if (sets[s].b) for (int e = tid; e < dim; e += nt) base[kMaxSetDim * kMaxRowDim + e] = sets[s].b[e];
            if (sets[s].par) {
                int np = set_par_words(sets[s].kind, dim, sets[s].par);
                np = np < kSetParMax ? np : kSetParMax;
                for (int e = tid; e < np; e += nt) base[kMaxSetDim * kMaxRowDim + kMaxSetDim + e] = sets[s].par[e];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kMaxSets; ++s) {
        // typed LDS pointers: the set loops are templates over the descriptor type, so these reads compile to ds_read
        const LdsPtr<T> base = (LdsPtr<T>)(lds + s * kSetLdsWords);
        out[s].kind = sets[s].kind;
        out[s].dim = sets[s].dim;
        out[s].A = base;
        out[s].b = base + kMaxSetDim * kMaxRowDim;
        out[s].par = base + kMaxSetDim * kMaxRowDim + kMaxSetDim;
    }
}

// numpy sign(): 0 for 0 (project_square_batch puts 0 on the arg-max entry of an all-zero row)
template <typename T> __device__ __forceinline__ T np_sign(T x) { return x > T(0) ? T(1) : (x < T(0) ? T(-1) : T(0)); }

// inverse of the d x d matrix M (SPD: I + rho sum A'A) by Gauss-Jordan without pivoting, in registers
template <typename T, int D>
__device__ __forceinline__ void invert_spd(T (&M)[D][D], T (&Inv)[D][D])
{
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Inv[i][j] = i == j ? T(1) : T(0);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const T piv = T(1) / M[k][k];
#pragma unroll
        for (int j = 0; j < D; ++j) { M[k][j] *= piv; Inv[k][j] *= piv; }
#pragma unroll
        for (int i = 0; i < D; ++i)
            if (i != k) {
                const T f = M[i][k];
#pragma unroll
                for (int j = 0; j < D; ++j) { M[i][j] -= f * M[k][j]; Inv[i][j] -= f * Inv[k][j]; }
            }
    }
}

// v[0..dim) <- primitive projection of v (in place)
template <typename T, typename P = const T *>
__device__ __forceinline__ void project_primitive(int kind, int dim, P par, T (&v)[kMaxSetDim])
{
    if (kind == ISLS_SET_BOX) {                                // par = lo[dim], hi[dim]
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < dim) {
                const T lo = par[i], hi = par[dim + i];
                const T a = v[i] < lo ? lo : v[i];             // np.clip = minimum(maximum(x, lo), hi)
                v[i] = a > hi ? hi : a;
            }
    } else if (kind == ISLS_SET_SOC_UNIT) {                    // z = v[:dim-1], t = v[dim-1]
        T t = T(0), ss = T(0);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) {
            if (i < dim - 1) ss += v[i] * v[i];
            if (i == dim - 1) t = v[i];
        }
        const T zn = sqrt(ss);
        const bool cond1 = (zn <= -t) || (t < T(0));
        const bool cond2 = (zn > t) || (zn > -t);
        const bool cond3 = zn <= t;
        const T tmp = (zn + t) / T(2);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) {
            if (i < dim) {
                const bool is_t = i == dim - 1;
                T o = v[i];
                if (cond2) o = is_t ? tmp : tmp * v[i] / (zn + T(1e-30));   // tmp[:,None] * z / (||z|| + 1e-30)
                if (cond1) o = T(0);
                if (cond3) o = v[i];
                v[i] = o;
            }
        }
    } else if (kind == ISLS_SET_LINEAR) {                      // par = l, u, a[dim]
        const T l = par[0], u = par[1];
        const auto a = par + 2;
        T atx = T(0), ata = T(0);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < dim) { atx += v[i] * a[i]; ata += a[i] * a[i]; }
        ata += T(1e-30);
        const bool hi = atx > u, lo = atx < l;
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < dim) {
                const T tmp = a[i] / ata;
                T o = v[i];
                if (hi) o = o - (atx - u) * tmp;
                if (lo) o = o - (atx - l) * tmp;
                v[i] = o;
            }
    } else if (kind == ISLS_SET_QUADRATIC) {                   // par = l, u
        const T l = par[0], u = par[1];
        T ss = T(0);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < dim) ss += v[i] * v[i];
        const T val = T(0.5) * ss, nrm = sqrt(ss);
        const bool hi = val > u, lo = l > val;
        const T su = sqrt(T(2) * u), sl = sqrt(T(2) * l);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < dim) {
                T o = v[i];
                if (hi) o = v[i] * su / nrm;
                if (lo) o = v[i] * sl / nrm;
                v[i] = o;
            }
    } else if (kind == ISLS_SET_SHELL) {                       // par = l, u, c[dim]: project_quadratic_batch(y - c, l, u) + c
        const T l = par[0], u = par[1];
        const auto c = par + 2;
        T w[kMaxSetDim], ss = T(0);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) {
            w[i] = i < dim ? v[i] - c[i] : T(0);
            ss += w[i] * w[i];
        }
        const T val = T(0.5) * ss, nrm = sqrt(ss);
        const bool hi = val > u, lo = l > val;
        const T su = sqrt(T(2) * u), sl = sqrt(T(2) * l);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < dim) {
                T o = w[i];
                if (hi) o = w[i] * su / nrm;
                if (lo) o = w[i] * sl / nrm;
                v[i] = o + c[i];
            }
    } else if (kind == ISLS_SET_MULTILINEAR) {                 // par = q, l[q], u[q], M[q*dim]   (project_multilinear)
        const int q = (int)par[0];
        const auto l = par + 1; const auto u = l + q; const auto Mm = u + q;
        T Ax[kMaxSetDim], G[kMaxSetDim][kMaxSetDim], Gi[kMaxSetDim][kMaxSetDim], mu[kMaxSetDim];
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < kMaxSetDim; ++j)
                if (i < q && j < dim) acc += Mm[i * dim + j] * v[j];
            Ax[i] = acc;
#pragma unroll
            for (int k = 0; k < kMaxSetDim; ++k) {             // M M' padded with the identity beyond q
                T g = (i == k && i >= q) ? T(1) : T(0);
#pragma unroll
                for (int j = 0; j < kMaxSetDim; ++j)
                    if (i < q && k < q && j < dim) g += Mm[i * dim + j] * Mm[k * dim + j];
                G[i][k] = g;
            }
        }
        invert_spd<T, kMaxSetDim>(G, Gi);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) {
            T t = Ax[i];
            if (i < q) {
                if (Ax[i] > u[i]) t = u[i];
                if (Ax[i] < l[i]) t = l[i];
            }
            Ax[i] = Ax[i] - t;
        }
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) {
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < kMaxSetDim; ++k) acc += Gi[i][k] * Ax[k];
            mu[i] = acc;
        }
#pragma unroll
        for (int j = 0; j < kMaxSetDim; ++j)
            if (j < dim) {
                T acc = T(0);
#pragma unroll
                for (int i = 0; i < kMaxSetDim; ++i)
                    if (i < q) acc += Mm[i * dim + j] * mu[i];
                v[j] = v[j] - acc;
            }
    } else if (kind == ISLS_SET_SQUARE) {                      // par = q, l, u, c[q], W[q*q], Winv[q*q]
        const int q = (int)par[0];
        const T l = par[1], u = par[2];
        const auto c = par + 3; const auto W = c + q; const auto Wi = W + q * q;
        T y[kMaxSetDim], w[kMaxSetDim];
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) y[i] = i < q ? v[i] - c[i] : T(0);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) {                 // w = y @ W.T
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < kMaxSetDim; ++j)
                if (i < q && j < q) acc += y[j] * W[i * q + j];
            w[i] = acc;
        }
        int jmax = 0;                                          // first arg-max of |w| (np.argmax)
        T amax = T(-1);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < q) {
                const T a = w[i] < T(0) ? -w[i] : w[i];
                if (a > amax) { amax = a; jmax = i; }
            }
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < q) {
                T o = w[i];
                if (amax < l && i == jmax) o = l * np_sign(w[i]);
                o = o > u ? u : o;                             // maximum(minimum(z, u), -u)
                o = o < -u ? -u : o;
                w[i] = o;
            }
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < q) {                                       // back: w @ Winv.T + c
                T acc = T(0);
#pragma unroll
                for (int j = 0; j < kMaxSetDim; ++j)
                    if (j < q) acc += w[j] * Wi[i * q + j];
                v[i] = acc + c[i];
            }
    }
}

// project_set_convex for ONE row x0[D] (isls/projections.py:289-374).  `block_max(a, b)` must return the maxima of
// a and b over every row of the same call (the reference stops all rows of a call together: np.max over rows and
// sets); rows that do not exist pass zeros.  Returns the number of iterations run.
template <typename T, int D, typename SetT, typename BlockMax>
__device__ __forceinline__ int project_set_convex_row(const T (&x0)[D], int nsets, const SetT (&sets)[kMaxSets], T rho,
                                                      int max_iter, T threshold, T (&x)[D], BlockMax &&block_max)
{
    T z[kMaxSets][kMaxSetDim], lmb[kMaxSets][kMaxSetDim];
    T M[D][D], Linv[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) M[i][j] = T(0);
#pragma unroll
    for (int s = 0; s < kMaxSets; ++s) {
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) { z[s][i] = T(0); lmb[s][i] = T(0); }
        if (s < nsets) {
            const auto A = sets[s].A; const auto b = sets[s].b;
            const int dim = sets[s].dim;
#pragma unroll
            for (int i = 0; i < kMaxSetDim; ++i)
                if (i < dim) {                                 // z_i = A_i x0 + b_i ; l_side_add += A_i' A_i
                    T acc = T(0);
#pragma unroll
                    for (int j = 0; j < D; ++j) acc += A[i * D + j] * x0[j];
                    z[s][i] = acc + b[i];
#pragma unroll
                    for (int j = 0; j < D; ++j)
#pragma unroll
                        for (int k = 0; k < D; ++k) M[j][k] += A[i * D + j] * A[i * D + k];
                }
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) M[i][j] = (i == j ? T(1) : T(0)) + rho * M[i][j];
    invert_spd<T, D>(M, Linv);
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = x0[j];

    T prim_g = T(1e5), dual_g = T(1e5);
    int it = 0;
    for (int j = 0; j < max_iter; ++j) {
        ++it;
        T rs[D];
#pragma unroll
        for (int k = 0; k < D; ++k) rs[k] = T(0);
#pragma unroll
        for (int s = 0; s < kMaxSets; ++s)
            if (s < nsets) {
                const auto A = sets[s].A; const auto b = sets[s].b;
                const int dim = sets[s].dim;
#pragma unroll
                for (int i = 0; i < kMaxSetDim; ++i)
                    if (i < dim) {
                        const T w = (-b[i] + z[s][i]) - lmb[s][i];
#pragma unroll
                        for (int k = 0; k < D; ++k) rs[k] += A[i * D + k] * w;
                    }
            }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < D; ++k) acc += Linv[i][k] * (x0[k] + rho * rs[k]);
            x[i] = acc;
        }
        const T prev_prim = prim_g, prev_dual = dual_g;
        T prim_m = T(0), dual_m = T(0);
#pragma unroll
        for (int s = 0; s < kMaxSets; ++s)
            if (s < nsets) {
                const auto A = sets[s].A; const auto b = sets[s].b;
                const int dim = sets[s].dim;
                T axb[kMaxSetDim], v[kMaxSetDim];
#pragma unroll
                for (int i = 0; i < kMaxSetDim; ++i) {
                    T acc = T(0);
                    if (i < dim) {
#pragma unroll
                        for (int k = 0; k < D; ++k) acc += A[i * D + k] * x[k];
                        acc += b[i];
                    }
                    axb[i] = acc;
                    v[i] = acc + lmb[s][i];
                }
                project_primitive<T>(sets[s].kind, dim, sets[s].par, v);
                T pn = T(0), dres[D];
#pragma unroll
                for (int k = 0; k < D; ++k) dres[k] = T(0);
#pragma unroll
                for (int i = 0; i < kMaxSetDim; ++i)
                    if (i < dim) {
                        const T pr = axb[i] - v[i];
                        const T dz = v[i] - z[s][i];
#pragma unroll
                        for (int k = 0; k < D; ++k) dres[k] += A[i * D + k] * dz;
                        lmb[s][i] += pr;
                        z[s][i] = v[i];
                        pn += pr * pr;
                    }
                T dn = T(0);
#pragma unroll
                for (int k = 0; k < D; ++k) dn += (rho * dres[k]) * (rho * dres[k]);
                pn = sqrt(pn);
                dn = sqrt(dn);
                prim_m = pn > prim_m ? pn : prim_m;
                dual_m = dn > dual_m ? dn : dual_m;
            }
        block_max(prim_m, dual_m);                             // -> maxima over every row of the call
        prim_g = prim_m;
        dual_g = dual_m;
        if (prim_g < threshold && dual_g < threshold) break;
        if (j != max_iter - 1) {
            const T pc = fabs(prev_prim - prim_g) / (prev_prim + T(1e-30));
            const T dc = fabs(prev_dual - dual_g) / (prev_dual + T(1e-30));
            if (pc < T(1e-5) && dc < T(1e-5)) break;
        }
    }
    return it;
}

// project_set_convex_dykstra for ONE row (isls/projections.py:465-504): every set acts on the row itself.  block_max as
// above (one value is enough: the summed squared change of the corrections of a pass).  Returns the passes run.
template <typename T, int D, typename SetT, typename BlockMax>
__device__ __forceinline__ int dykstra_row(const T (&x0)[D], int nsets, const SetT (&sets)[kMaxSets], int max_iter, T tol,
                                           T (&x)[D], BlockMax &&block_max)
{
    T u[D], z[kMaxSets][D];
#pragma unroll
    for (int j = 0; j < D; ++j) u[j] = x0[j];
#pragma unroll
    for (int s = 0; s < kMaxSets; ++s)
#pragma unroll
        for (int j = 0; j < D; ++j) z[s][j] = T(0);
    int k = 0;
    T cmax = T(10);
    while (k <= max_iter && cmax >= tol) {                     // np.any(cI >= tol)
        T cI = T(0);
#pragma unroll
        for (int s = 0; s < kMaxSets; ++s)
            if (s < nsets) {
                T v[kMaxSetDim], prev_u[D], nn = T(0);
#pragma unroll
                for (int j = 0; j < kMaxSetDim; ++j) v[j] = T(0);
#pragma unroll
                for (int j = 0; j < D; ++j) { prev_u[j] = u[j]; v[j] = prev_u[j] - z[s][j]; }
                project_primitive<T>(sets[s].kind, D, sets[s].par, v);
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const T prev_z = z[s][j];
                    const T zn = v[j] - (prev_u[j] - prev_z);
                    u[j] = v[j];
                    z[s][j] = zn;
                    nn += (prev_z - zn) * (prev_z - zn);
                }
                const T nr = sqrt(nn);
                cI += nr * nr;                                 // np.linalg.norm(...)**2
            }
        T dummy = T(0);
        block_max(cI, dummy);
        cmax = cI;
        ++k;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = u[j];
    return k;
}

// project_soc for ONE row (isls/projections.py:163-234): A y + b in the second-order cone by ADMM.
template <typename T, int D, typename SetT, typename BlockMax>
__device__ __forceinline__ int project_soc_row(const T (&z0)[D], const SetT &st, T rho, int max_iter, T tol, T (&zo)[D],
                                               BlockMax &&block_max)
{
    const auto A = st.A; const auto b = st.b;
    const int dim = st.dim;
    T M[D][D], Linv[D][D], z[D], lmb[kMaxSetDim];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            T acc = T(0);
#pragma unroll
            for (int i = 0; i < kMaxSetDim; ++i)
                if (i < dim) acc += A[i * D + j] * A[i * D + k];
            M[j][k] = (j == k ? T(1) : T(0)) + rho * acc;
        }
    invert_spd<T, D>(M, Linv);
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = z0[j];
#pragma unroll
    for (int i = 0; i < kMaxSetDim; ++i) lmb[i] = T(0);
    T prim_g = T(1e5), dual_g = T(1e5);
    int it = 0;
    for (int j = 0; j < max_iter; ++j) {
        ++it;
        T x[kMaxSetDim], zp[D], rs[D];
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i) {
            T acc = T(0);
            if (i < dim) {
#pragma unroll
                for (int k = 0; k < D; ++k) acc += A[i * D + k] * z[k];
                acc = (acc + b[i]) + lmb[i];
            }
            x[i] = acc;
        }
        project_primitive<T>(ISLS_SET_SOC_UNIT, dim, (const T *)nullptr, x);
#pragma unroll
        for (int k = 0; k < D; ++k) { zp[k] = z[k]; rs[k] = T(0); }
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < dim) {
                const T w = (-b[i] + x[i]) - lmb[i];
#pragma unroll
                for (int k = 0; k < D; ++k) rs[k] += A[i * D + k] * w;
            }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < D; ++k) acc += Linv[i][k] * (z0[k] + rho * rs[k]);
            z[i] = acc;
        }
        T pn = T(0), dn = T(0);
#pragma unroll
        for (int i = 0; i < kMaxSetDim; ++i)
            if (i < dim) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < D; ++k) acc += A[i * D + k] * z[k];
                const T pr = (acc + b[i]) - x[i];
                lmb[i] += pr;
                pn += pr * pr;
            }
#pragma unroll
        for (int k = 0; k < D; ++k) dn += (rho * (z[k] - zp[k])) * (rho * (z[k] - zp[k]));
        pn = sqrt(pn);
        dn = sqrt(dn);
        const T prev_p = prim_g, prev_d = dual_g;
        block_max(pn, dn);
        prim_g = pn;
        dual_g = dn;
        if (prim_g < tol && dual_g < tol) break;
        if (j != max_iter - 1) {
            const T pc = fabs(prev_p - prim_g) / (prev_p + T(1e-30)), dc = fabs(prev_d - dual_g) / (prev_d + T(1e-30));
            if (pc < T(1e-5) && dc < T(1e-5)) break;
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) zo[j] = z[j];
    return it;
}

}  // namespace isls
