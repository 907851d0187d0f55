/*
 * isls_oracle_impl.h -- body of the CPU oracle, included twice by isls_oracle.c (REAL = double / float).
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar, per-trajectory C restatement of the reference's
 * algorithm (chenjianxing1/iLQR-ADMM, python package `isls`) for the hot path of SURVEY.md section 8.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the product
 * (libisls_hip.so + ilqr-admm_amd/isls) never links, imports or falls back to it.
 *
 * Pinned: tests/test_oracle_golden.py checks every function below against golden vectors produced
 * by running the reference itself in the build container (tests/golden/make_golden.py).
 *
 * Every function cites the reference lines it restates.  Operation order follows the reference
 * (association of matrix products, order of the sums); LAPACK calls (dposv, dgesv) are replaced
 * by the unblocked textbook algorithms LAPACK itself uses at these sizes.
 */

#define MAXN 16
#define MAXM 8

/* object (b,t) of a strided view */
#define VIEW(v, b, t) (((const REAL *)(v).p) + (int64_t)(b) * (v).sb + (int64_t)(t) * (v).st)

/* ---- tiny dense helpers (row-major) ---------------------------------------------------------- */
/* C[p x r] = A^T[p x q] * B[q x r] where A is stored q x p */
static void FN(matTmul)(const REAL *A, const REAL *B, REAL *C, int q, int p, int r)
{
    for (int i = 0; i < p; ++i)
        for (int j = 0; j < r; ++j) {
            REAL s = 0;
            for (int k = 0; k < q; ++k) s += A[k * p + i] * B[k * r + j];
            C[i * r + j] = s;
        }
}
/* C[p x r] = A[p x q] * B[q x r] */
static void FN(matmul)(const REAL *A, const REAL *B, REAL *C, int p, int q, int r)
{
    for (int i = 0; i < p; ++i)
        for (int j = 0; j < r; ++j) {
            REAL s = 0;
            for (int k = 0; k < q; ++k) s += A[i * q + k] * B[k * r + j];
            C[i * r + j] = s;
        }
}

/* Upper Cholesky A = U^T U, unblocked (LAPACK dpotf2 'U' order).  Returns 0 or 1 if not PD. */
static int FN(chol_upper)(const REAL *A, REAL *U, int m)
{
    for (int i = 0; i < m * m; ++i) U[i] = 0;
    for (int j = 0; j < m; ++j) {
        REAL ajj = A[j * m + j];
        for (int k = 0; k < j; ++k) ajj -= U[k * m + j] * U[k * m + j];
        if (!(ajj > 0)) return 1;
        ajj = SQRT(ajj);
        U[j * m + j] = ajj;
        for (int c = j + 1; c < m; ++c) {
            REAL s = A[j * m + c];
            for (int k = 0; k < j; ++k) s -= U[k * m + j] * U[k * m + c];
            U[j * m + c] = s / ajj;
        }
    }
    return 0;
}
/* solve U^T U x = b (dpotrs 'U': forward with U^T, backward with U) */
static void FN(chol_solve)(const REAL *U, const REAL *b, REAL *x, int m)
{
    REAL y[MAXM];
    for (int i = 0; i < m; ++i) {
        REAL s = b[i];
        for (int k = 0; k < i; ++k) s -= U[k * m + i] * y[k];
        y[i] = s / U[i * m + i];
    }
    for (int i = m - 1; i >= 0; --i) {
        REAL s = y[i];
        for (int k = i + 1; k < m; ++k) s -= U[i * m + k] * x[k];
        x[i] = s / U[i * m + i];
    }
}
/* inverse by LU with partial pivoting applied to the identity (numpy.linalg.inv = dgesv(A, I)) */
static int FN(inv_lu)(const REAL *A, REAL *Ainv, int m)
{
    REAL LU[MAXM * MAXM];
    int piv[MAXM];
    for (int i = 0; i < m * m; ++i) LU[i] = A[i];
    for (int j = 0; j < m; ++j) {
        int p = j;
        REAL best = FABS(LU[j * m + j]);
        for (int i = j + 1; i < m; ++i)
            if (FABS(LU[i * m + j]) > best) { best = FABS(LU[i * m + j]); p = i; }
        piv[j] = p;
        if (best == 0) return 1;
        if (p != j)
            for (int c = 0; c < m; ++c) { REAL tmp = LU[j * m + c]; LU[j * m + c] = LU[p * m + c]; LU[p * m + c] = tmp; }
        for (int i = j + 1; i < m; ++i) {
            LU[i * m + j] /= LU[j * m + j];
            for (int c = j + 1; c < m; ++c) LU[i * m + c] -= LU[i * m + j] * LU[j * m + c];
        }
    }
    for (int col = 0; col < m; ++col) {
        REAL e[MAXM];
        for (int i = 0; i < m; ++i) e[i] = (i == col) ? 1 : 0;
        for (int j = 0; j < m; ++j)
            if (piv[j] != j) { REAL tmp = e[j]; e[j] = e[piv[j]]; e[piv[j]] = tmp; }
        for (int i = 0; i < m; ++i)
            for (int k = 0; k < i; ++k) e[i] -= LU[i * m + k] * e[k];
        for (int i = m - 1; i >= 0; --i) {
            for (int k = i + 1; k < m; ++k) e[i] -= LU[i * m + k] * e[k];
            e[i] /= LU[i * m + i];
        }
        for (int i = 0; i < m; ++i) Ainv[i * m + col] = e[i];
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Gain pass.  isls/isls.py:245-308 (CHOL) and isls/sls.py:98-162 (INV), K-dependent part.
 * ------------------------------------------------------------------------------------------- */
int FN(oracle_riccati_gain)(const isls_gain_args *a)
{
    const int B = a->B, N = a->N, n = a->n, m = a->m;
    if (B < 0 || N < 1 || n < 1 || m < 1 || n > MAXN || m > MAXM) return ISLS_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        if (a->active && !a->active[b]) continue;
        REAL *K = (REAL *)a->K + (int64_t)b * N * m * n;
        REAL *Quu_o = (REAL *)a->Quu + (int64_t)b * N * m * m;
        REAL *fac_o = (REAL *)a->fac + (int64_t)b * N * m * m;
        REAL *Qux_o = (REAL *)a->Qux + (int64_t)b * N * m * n;
        REAL V[MAXN * MAXN], AtV[MAXN * MAXN], BtV[MAXM * MAXN];
        REAL Qxx[MAXN * MAXN], Qux[MAXM * MAXN], Quu[MAXM * MAXM];
        REAL U[MAXM * MAXM], Kt[MAXM * MAXN], KtQuu[MAXN * MAXM], T1[MAXN * MAXN], T2[MAXN * MAXN], T3[MAXN * MAXN];
        /* terminal: K[N-1]=0 (isls.py:245), V = Cxx[N-1] (isls.py:251,257) */
        for (int i = 0; i < m * n; ++i) { K[(N - 1) * m * n + i] = 0; Qux_o[(N - 1) * m * n + i] = 0; }
        for (int i = 0; i < m * m; ++i) { Quu_o[(N - 1) * m * m + i] = 0; fac_o[(N - 1) * m * m + i] = 0; }
        {
            const REAL *C = VIEW(a->Cxx, b, N - 1);
            for (int i = 0; i < n * n; ++i) V[i] = C[i];
        }
        int bad = 0;
        for (int t = N - 2; t >= 0; --t) {
            const REAL *A = VIEW(a->A, b, t), *Bm = VIEW(a->Bm, b, t);
            const REAL *Cxx = VIEW(a->Cxx, b, t), *Cuu = VIEW(a->Cuu, b, t);
            const REAL *Cux = a->Cux.p ? VIEW(a->Cux, b, t) : 0;
            /* Qxx = Cxx + (A'V)A ; Qux = Cux + (B'V)A ; Quu = Cuu + (B'V)B   (isls.py:288-290) */
            FN(matTmul)(A, V, AtV, n, n, n);
            FN(matTmul)(Bm, V, BtV, n, m, n);
            FN(matmul)(AtV, A, Qxx, n, n, n);
            FN(matmul)(BtV, A, Qux, m, n, n);
            FN(matmul)(BtV, Bm, Quu, m, n, m);
            for (int i = 0; i < n * n; ++i) Qxx[i] = Cxx[i] + Qxx[i];
            for (int i = 0; i < m * n; ++i) Qux[i] = (Cux ? Cux[i] : (REAL)0) + Qux[i];
            for (int i = 0; i < m * m; ++i) Quu[i] = Cuu[i] + Quu[i];
            REAL *fac = fac_o + (int64_t)t * m * m;
            if (a->solve_mode == ISLS_SOLVE_CHOL) {
                /* sol = -solve(Quu, [Qux qu], assume_a="pos")  (isls.py:296-298) */
                if (FN(chol_upper)(Quu, U, m)) { bad = 1; break; }
                for (int j = 0; j < n; ++j) {
                    REAL rhs[MAXM], x[MAXM];
                    for (int r = 0; r < m; ++r) rhs[r] = Qux[r * n + j];
                    FN(chol_solve)(U, rhs, x, m);
                    for (int r = 0; r < m; ++r) Kt[r * n + j] = -x[r];
                }
                for (int i = 0; i < m; ++i)
                    for (int j = 0; j < m; ++j) fac[i * m + j] = (i == j) ? (REAL)1 / U[i * m + i] : U[i * m + j];
            } else {
                /* Quu_inv = inv(Quu); Kt = -Quu_inv.dot(Qux)  (sls.py:149-150) */
                if (FN(inv_lu)(Quu, fac, m)) { bad = 1; break; }
                FN(matmul)(fac, Qux, Kt, m, m, n);
                for (int i = 0; i < m * n; ++i) Kt[i] = -Kt[i];
            }
            /* V update */
            FN(matTmul)(Kt, Quu, KtQuu, m, n, m);  /* K'Quu   n x m */
            FN(matmul)(KtQuu, Kt, T1, n, m, n);    /* (K'Quu)K       */
            FN(matTmul)(Qux, Kt, T2, m, n, n);     /* Qux'K          */
            FN(matTmul)(Kt, Qux, T3, m, n, n);     /* K'Qux          */
            if (a->solve_mode == ISLS_SOLVE_CHOL)   /* isls.py:300: Qxx + K'QuuK + Qux'K + K'Qux */
                for (int i = 0; i < n * n; ++i) V[i] = ((Qxx[i] + T1[i]) + T2[i]) + T3[i];
            else                                    /* sls.py:153:  Qxx + Qux'K + K'Qux + K'QuuK */
                for (int i = 0; i < n * n; ++i) V[i] = ((Qxx[i] + T2[i]) + T3[i]) + T1[i];
            for (int i = 0; i < m * n; ++i) { K[(int64_t)t * m * n + i] = Kt[i]; Qux_o[(int64_t)t * m * n + i] = Qux[i]; }
            for (int i = 0; i < m * m; ++i) Quu_o[(int64_t)t * m * m + i] = Quu[i];
        }
        if (bad && a->status) a->status[b] |= ISLS_ST_NOT_PD;
    }
    return ISLS_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Feed-forward pass.  isls/isls.py:252-302 (v,k part; dposv re-factorises Quu every call, so does
 * this oracle) and isls/sls.py:168-202 (INV: uses the cached Quu_inv = fac).
 * ------------------------------------------------------------------------------------------- */
static void FN(reg_grad)(const isls_ff_args *a, int b, int t, REAL *cx, REAL *cu)
{
    const int N = a->N, n = a->n, m = a->m;
    const REAL *c0x = VIEW(a->c0x, b, t), *c0u = VIEW(a->c0u, b, t);
    for (int i = 0; i < n; ++i) cx[i] = c0x[i];
    for (int i = 0; i < m; ++i) cu[i] = c0u[i];
    if (a->Qr.p) {
        const REAL *Qr = (a->Qr_term && t == N - 1) ? (const REAL *)a->Qr_term + (int64_t)b * a->Qr.sb : VIEW(a->Qr, b, t);
        const REAL *xh = a->xhat ? (const REAL *)a->xhat + ((int64_t)b * N + t) * n : 0;
        const REAL *z = (const REAL *)a->zx + ((int64_t)b * N + t) * n, *l = (const REAL *)a->lx + ((int64_t)b * N + t) * n;
        REAL d[MAXN];
        for (int j = 0; j < n; ++j) d[j] = (xh ? xh[j] : (REAL)0) - (z[j] - l[j]);
        for (int i = 0; i < n; ++i) {
            REAL s = 0;
            for (int j = 0; j < n; ++j) s += Qr[i * n + j] * d[j];
            cx[i] += 2 * s;
        }
    }
    if (a->Rr.p) {
        const REAL *Rr = VIEW(a->Rr, b, t);
        const REAL *uh = a->uhat ? (const REAL *)a->uhat + ((int64_t)b * N + t) * m : 0;
        const REAL *z = (const REAL *)a->zu + ((int64_t)b * N + t) * m, *l = (const REAL *)a->lu + ((int64_t)b * N + t) * m;
        REAL d[MAXM];
        for (int j = 0; j < m; ++j) d[j] = (uh ? uh[j] : (REAL)0) - (z[j] - l[j]);
        for (int i = 0; i < m; ++i) {
            REAL s = 0;
            for (int j = 0; j < m; ++j) s += Rr[i * m + j] * d[j];
            cu[i] += 2 * s;
        }
    }
}

int FN(oracle_riccati_ff)(const isls_ff_args *a)
{
    const int B = a->B, N = a->N, n = a->n, m = a->m;
    if (B < 0 || N < 1 || n < 1 || m < 1 || n > MAXN || m > MAXM) return ISLS_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        if (a->active && !a->active[b]) continue;
        REAL *kout = (REAL *)a->k + (int64_t)b * N * m;
        REAL v[MAXN], cx[MAXN], cu[MAXM], qx[MAXN], qu[MAXM], kt[MAXM], U[MAXM * MAXM], w[MAXM];
        for (int i = 0; i < m; ++i) kout[(N - 1) * m + i] = 0;
        FN(reg_grad)(a, b, N - 1, v, cu);
        for (int t = N - 2; t >= 0; --t) {
            const REAL *A = VIEW(a->A, b, t), *Bm = VIEW(a->Bm, b, t);
            const REAL *K = (const REAL *)a->K + ((int64_t)b * N + t) * m * n;
            const REAL *Quu = (const REAL *)a->Quu + ((int64_t)b * N + t) * m * m;
            const REAL *fac = (const REAL *)a->fac + ((int64_t)b * N + t) * m * m;
            const REAL *Qux = (const REAL *)a->Qux + ((int64_t)b * N + t) * m * n;
            FN(reg_grad)(a, b, t, cx, cu);
            for (int i = 0; i < n; ++i) { REAL s = 0; for (int k = 0; k < n; ++k) s += A[k * n + i] * v[k]; qx[i] = cx[i] + s; }
            for (int i = 0; i < m; ++i) { REAL s = 0; for (int k = 0; k < n; ++k) s += Bm[k * m + i] * v[k]; qu[i] = cu[i] + s; }
            if (a->solve_mode == ISLS_SOLVE_CHOL) {
                if (FN(chol_upper)(Quu, U, m)) break;
                FN(chol_solve)(U, qu, kt, m);
                for (int i = 0; i < m; ++i) kt[i] = -kt[i];
            } else {
                for (int i = 0; i < m; ++i) { REAL s = 0; for (int j = 0; j < m; ++j) s += fac[i * m + j] * qu[j]; kt[i] = -s; }
            }
            /* w = (K'Quu) k is evaluated as K'(row) after KtQuu, like Kt.T.dot(Quu).dot(kt) */
            for (int i = 0; i < n; ++i) {
                REAL t_kqu = 0, t_kquuk = 0, t_quxk = 0;
                for (int r = 0; r < m; ++r) t_kqu += K[r * n + i] * qu[r];
                for (int c = 0; c < m; ++c) {
                    REAL kq = 0;
                    for (int r = 0; r < m; ++r) kq += K[r * n + i] * Quu[r * m + c];
                    w[c] = kq;
                }
                for (int c = 0; c < m; ++c) t_kquuk += w[c] * kt[c];
                for (int r = 0; r < m; ++r) t_quxk += Qux[r * n + i] * kt[r];
                if (a->solve_mode == ISLS_SOLVE_CHOL)  /* isls.py:302 */
                    v[i] = ((qx[i] + t_kqu) + t_kquuk) + t_quxk;
                else                                    /* sls.py:200 */
                    v[i] = ((qx[i] + t_quxk) + t_kqu) + t_kquuk;
            }
            for (int i = 0; i < m; ++i) kout[(int64_t)t * m + i] = kt[i];
        }
    }
    return ISLS_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Built-in forward models (specs: SURVEY Appendix A; notebooks).
 * ------------------------------------------------------------------------------------------- */
static REAL FN(pymod)(REAL a, REAL b) /* numpy `%` on floats: fmod + sign fix */
{
    REAL r = FMOD(a, b);
    if (r != 0 && ((b < 0) != (r < 0))) r += b;
    return r;
}

static void FN(model_step)(int model, const REAL *par, int n, int m, const REAL *x, const REAL *u, REAL *xn)
{
    if (model == ISLS_MODEL_LTI) { /* x.dot(A.T) + u.dot(B.T), isls/sls_base.py:49-53 */
        const REAL *A = par, *Bm = par + n * n;
        for (int i = 0; i < n; ++i) {
            REAL s = 0, r = 0;
            for (int j = 0; j < n; ++j) s += A[i * n + j] * x[j];
            for (int j = 0; j < m; ++j) r += Bm[i * m + j] * u[j];
            xn[i] = s + r;
        }
    } else if (model == ISLS_MODEL_DI) { /* same LTI map through its Kronecker structure (isls/utils.py:266-276) */
        const int d = n / 2;
        for (int i = 0; i < d; ++i) {
            xn[i] = (x[i] + par[0] * x[d + i]) + par[1] * u[i];
            xn[d + i] = x[d + i] + par[2] * u[i];
        }
    } else if (model == ISLS_MODEL_ARM3R) { /* 3DoF notebook cell 9 + closed-form FK */
        const REAL dt = par[0];
        REAL c = 0, ex = 0, ey = 0;
        for (int j = 0; j < 3; ++j) {
            xn[j] = x[j] + x[3 + j] * dt + (REAL)0.5 * u[j] * (dt * dt);
            xn[3 + j] = x[3 + j] + u[j] * dt;
        }
        for (int j = 0; j < 3; ++j) { c += xn[j]; ex += COS(c); ey += SIN(c); }
        xn[6] = ex; xn[7] = ey; xn[8] = 0;
    } else if (model == ISLS_MODEL_TASSA) { /* notebooks/Tutorial.ipynb cell 8 */
        const REAL dt = par[0], d = par[1];
        const REAL f = dt * x[3];
        const REAL sw = SIN(u[0]) * f;
        const REAL b = (f * COS(u[0]) + d) - SQRT(d * d - sw * sw);
        xn[0] = x[0] + b * COS(x[2]);
        xn[1] = x[1] + b * SIN(x[2]);
        xn[2] = x[2] + ASIN(sw / d);
        xn[3] = x[3] + u[1] * dt;
    } else { /* car-simple, Car notebooks cell 6 */
        const REAL dt = par[0];
        xn[0] = x[0] + dt * x[3] * COS(x[2]);
        xn[1] = x[1] + dt * x[3] * SIN(x[2]);
        xn[2] = FN(pymod)(x[2] + dt * x[3] * u[0], (REAL)(2 * 3.14159265358979323846));
        xn[3] = x[3] + dt * u[1];
    }
}

/* ---------------------------------------------------------------------------------------------
 * Line-search rollout.  isls/isls.py:310-334 (rollout_DP), 357-369 (candidates, NaN rule, arg-min,
 * acceptance), isls/sls_base.py:25-44 (cost), isls/isls.py:471-477 (AL terms, no acceptance test).
 * ------------------------------------------------------------------------------------------- */
static void FN(rollout_one)(const isls_rollout_args *a, int b, REAL alpha, REAL *xs, REAL *us, REAL *cost, REAL *aug)
{
    const int N = a->N, n = a->n, m = a->m;
    const REAL *K = (const REAL *)a->K + (int64_t)b * N * m * n, *k = (const REAL *)a->k + (int64_t)b * N * m;
    const REAL *xh = a->xhat ? (const REAL *)a->xhat + (int64_t)b * N * n : 0;
    const REAL *uh = a->uhat ? (const REAL *)a->uhat + (int64_t)b * N * m : 0;
    const REAL *par = (const REAL *)a->model_par + (int64_t)b * a->model_par_sb;
    const REAL *Qtab = (const REAL *)a->Qtab + (int64_t)b * a->Qtab_sb;
    const REAL *ztab = (const REAL *)a->ztab + (int64_t)b * a->ztab_sb;
    const int absolute = (a->flags & ISLS_RO_ABSOLUTE) != 0;
    REAL x[MAXN], xn[MAXN], u[MAXM];
    REAL cst = 0, ag = 0;
    if (a->x0) for (int i = 0; i < n; ++i) x[i] = ((const REAL *)a->x0)[(int64_t)b * n + i];
    else for (int i = 0; i < n; ++i) x[i] = xh[i];
    for (int t = 0; t < N; ++t) {
        for (int r = 0; r < m; ++r) { /* u = dx @ K.T + k[:,i] + u_nom[i]  (isls.py:329) */
            REAL s = 0;
            for (int j = 0; j < n; ++j) s += ((absolute || !xh) ? x[j] : x[j] - xh[t * n + j]) * K[((int64_t)t * m + r) * n + j];
            u[r] = (s + alpha * k[t * m + r]) + ((absolute || !uh) ? (REAL)0 : uh[t * m + r]);
        }
        if (xs) { for (int i = 0; i < n; ++i) xs[t * n + i] = x[i]; for (int i = 0; i < m; ++i) us[t * m + i] = u[i]; }
        if (cost && a->cost_model == ISLS_COST_PHUBER) { /* Tutorial.ipynb cell 14: running + final pseudo-Huber terms */
            const REAL *cp = (const REAL *)a->cost_par, *cx = cp + m, *px = cx + n, *cf = px + n, *pf = cf + n;
            for (int i = 0; i < n; ++i) cst += cx[i] * (SQRT(x[i] * x[i] + px[i] * px[i]) - px[i]);
            if (t == N - 1) for (int i = 0; i < n; ++i) cst += cf[i] * (SQRT(x[i] * x[i] + pf[i] * pf[i]) - pf[i]);
        } else if (cost && (!a->q_nonzero || a->q_nonzero[t])) {
            const REAL *Q = Qtab + (int64_t)a->seq[t] * n * n, *z = ztab + (int64_t)a->seq[t] * n;
            REAL d[MAXN];
            for (int i = 0; i < n; ++i) d[i] = x[i] - z[i];
            for (int i = 0; i < n; ++i) { REAL s = 0; for (int j = 0; j < n; ++j) s += Q[i * n + j] * d[j]; cst += d[i] * s; }
        }
        if (aug && a->wq.p) {
            const REAL *w = VIEW(a->wq, b, t);
            const REAL *z = (const REAL *)a->zx + ((int64_t)b * N + t) * n, *l = (const REAL *)a->lx + ((int64_t)b * N + t) * n;
            for (int i = 0; i < n; ++i) { REAL d = x[i] - (z[i] - l[i]); ag += (d * d) * w[i]; }
        }
        if (aug && a->wr.p) {
            const REAL *w = VIEW(a->wr, b, t);
            const REAL *z = (const REAL *)a->zu + ((int64_t)b * N + t) * m, *l = (const REAL *)a->lu + ((int64_t)b * N + t) * m;
            for (int i = 0; i < m; ++i) { REAL d = u[i] - (z[i] - l[i]); ag += (d * d) * w[i]; }
        }
        FN(model_step)(a->model, par, n, m, x, u, xn);   /* x = f(x,u), isls.py:332 */
        for (int i = 0; i < n; ++i) x[i] = xn[i];
    }
    /* *cost is the STATE part only; the caller adds the control part from the stored controls so
     * that the order is the reference's: sum over x first, then += sum over u (sls_base.py:33-39). */
    if (cost) *cost = cst;
    if (aug) *aug = ag;
}

int FN(oracle_rollout_ls)(const isls_rollout_args *a)
{
    const int B = a->B, N = a->N, n = a->n, m = a->m, L = a->L;
    if (B < 0 || N < 1 || n < 1 || m < 1 || n > MAXN || m > MAXM || L < 1 || L > 64) return ISLS_ERR_ARG;
    int rc = ISLS_OK;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        if (a->active && !a->active[b]) continue;
        REAL *xs = (REAL *)malloc(sizeof(REAL) * (size_t)N * (n + m));
        REAL *us = xs + (size_t)N * n;
        REAL costs[64], augs[64];
        const REAL ustd = (REAL)a->u_std;
        int nan_seen = 0;
        for (int l = 0; l < L; ++l) {
            REAL alpha = (a->flags & ISLS_RO_ABSOLUTE) ? (REAL)1 : ((const REAL *)a->alphas)[l];
            REAL cst, ag;
            FN(rollout_one)(a, b, alpha, xs, us, &cst, &ag);
            REAL cu = 0;
            if (a->cost_model == ISLS_COST_PHUBER) {
                const REAL *cuw = (const REAL *)a->cost_par;
                for (int i = 0; i < N * m; ++i) cu += cuw[i % m] * (us[i] * us[i]);
            } else
                for (int i = 0; i < N * m; ++i) cu += us[i] * (ustd * us[i]);
            costs[l] = cst + cu;
            augs[l] = costs[l] + ag;
            if (augs[l] != augs[l]) {
                nan_seen = 1;
                if (a->flags & ISLS_RO_NAN_TO_1E5) { augs[l] = (REAL)1e5; costs[l] = (REAL)1e5; }
            }
        }
        int ind = 0; /* np.argmin: first minimum; NaN propagates as minimum in numpy */
        for (int l = 1; l < L; ++l) {
            if (augs[ind] != augs[ind]) break;
            if (augs[l] != augs[l] || augs[l] < augs[ind]) ind = l;
        }
        if (a->cost_all) for (int l = 0; l < L; ++l) ((REAL *)a->cost_all)[(int64_t)b * L + l] = augs[l];
        int accept = 1;
        if (a->flags & ISLS_RO_ACCEPT_TEST) accept = (costs[ind] - ((const REAL *)a->cost_cur)[b]) < 0;
        REAL *xo = (REAL *)a->x_out + (int64_t)b * N * n, *uo = (REAL *)a->u_out + (int64_t)b * N * m;
        if (accept) {
            REAL alpha = (a->flags & ISLS_RO_ABSOLUTE) ? (REAL)1 : ((const REAL *)a->alphas)[ind];
            FN(rollout_one)(a, b, alpha, xo, uo, 0, 0);
            if (a->cost_new) ((REAL *)a->cost_new)[b] = costs[ind];
        } else {
            const REAL *xh = (const REAL *)a->xhat + (int64_t)b * N * n, *uh = (const REAL *)a->uhat + (int64_t)b * N * m;
            for (int i = 0; i < N * n; ++i) xo[i] = xh[i];
            for (int i = 0; i < N * m; ++i) uo[i] = uh[i];
            if (a->cost_new) ((REAL *)a->cost_new)[b] = ((const REAL *)a->cost_cur)[b];
        }
        if (a->best) a->best[b] = ind;
        if (a->status) a->status[b] |= (nan_seen ? ISLS_ST_NAN_COST : 0) | (accept ? 0 : ISLS_ST_LS_REJECT);
        free(xs);
    }
    return rc;
}

/* ---------------------------------------------------------------------------------------------
 * ADMM update.  isls/admm.py:43-85, box projection isls/projections.py:7-11.
 * ------------------------------------------------------------------------------------------- */
int FN(oracle_project_rows)(const isls_project_args *a);

static void FN(admm_block)(int N, int d, int proj, REAL relax, const REAL *x, REAL *z, REAL *l,
                           const isls_view *lo, const isls_view *hi, const REAL *work, int b, REAL *prim2, REAL *dual2)
{
    REAL p2 = 0, d2 = 0;
    for (int t = 0; t < N; ++t) {
        const REAL *lo_t = proj == ISLS_PROJ_BOX ? VIEW(*lo, b, t) : 0, *hi_t = proj == ISLS_PROJ_BOX ? VIEW(*hi, b, t) : 0;
        for (int i = 0; i < d; ++i) {
            const int e = t * d + i;
            REAL zp = z[e];
            REAL zz = relax * x[e] + (1 - relax) * zp;   /* admm.py:48 */
            REAL arg = zz + l[e];
            REAL zn = arg;
            if (proj == ISLS_PROJ_BOX) { zn = arg < lo_t[i] ? lo_t[i] : arg; zn = zn > hi_t[i] ? hi_t[i] : zn; } /* np.clip */
            else if (proj == ISLS_PROJ_SETS) zn = work[e];  /* project_set_convex of the argument, done for all rows before */
            REAL r = x[e] - zn;                            /* admm.py:51 */
            l[e] += r;                                     /* admm.py:52 */
            z[e] = zn;
            p2 += r * r;
            d2 += (zn - zp) * (zn - zp);
        }
    }
    *prim2 = p2; *dual2 = d2;
}

int FN(oracle_admm_update)(const isls_admm_args *a)
{
    const int B = a->B, N = a->N, n = a->n, m = a->m;
    if (B < 0 || N < 1 || n < 1 || m < 1) return ISLS_ERR_ARG;
    for (int blk = 0; blk < 2; ++blk) {   /* ISLS_PROJ_SETS: argument -> work, project_set_convex over the time steps */
        const int isx = blk == 0, d = isx ? n : m, col0 = isx ? a->x_col0 : a->u_col0;
        if (!(isx ? a->zx : a->zu) || (isx ? a->proj_x : a->proj_u) != ISLS_PROJ_SETS) continue;
        const isls_project_args *ps = isx ? a->x_sets : a->u_sets;
        REAL *work = (REAL *)(isx ? a->x_work : a->u_work);
        if (!ps || !work) return ISLS_ERR_ARG;
        const REAL *x = (const REAL *)(isx ? a->xx : a->xu), *z = (const REAL *)(isx ? a->zx : a->zu), *l = (const REAL *)(isx ? a->lx : a->lu);
        for (int b = 0; b < B; ++b) {
            if (a->active && !a->active[b]) continue;
            for (int64_t e = (int64_t)b * N * d; e < (int64_t)(b + 1) * N * d; ++e)
                work[e] = ((REAL)a->relax * x[e] + (1 - (REAL)a->relax) * z[e]) + l[e];
        }
        isls_project_args pr = *ps;
        pr.P = B; pr.R = N; pr.y_in = work + col0; pr.y_out = work + col0;
        pr.in_sp = pr.out_sp = (int64_t)N * d; pr.in_sr = pr.out_sr = d; pr.iters = 0; pr.active = a->active;
        int rc = FN(oracle_project_rows)(&pr);
        if (rc != ISLS_OK) return rc;
    }
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        if (a->active && !a->active[b]) continue;
        REAL prim = 0, dual = 0, p2, d2;
        if (a->zx) {
            FN(admm_block)(N, n, a->proj_x, (REAL)a->relax, (const REAL *)a->xx + (int64_t)b * N * n,
                           (REAL *)a->zx + (int64_t)b * N * n, (REAL *)a->lx + (int64_t)b * N * n, &a->x_lo, &a->x_hi,
                           a->proj_x == ISLS_PROJ_SETS ? (const REAL *)a->x_work + (int64_t)b * N * n : 0, b, &p2, &d2);
            prim += SQRT(p2); dual += SQRT(d2);
        }
        if (a->zu) {
            FN(admm_block)(N, m, a->proj_u, (REAL)a->relax, (const REAL *)a->xu + (int64_t)b * N * m,
                           (REAL *)a->zu + (int64_t)b * N * m, (REAL *)a->lu + (int64_t)b * N * m, &a->u_lo, &a->u_hi,
                           a->proj_u == ISLS_PROJ_SETS ? (const REAL *)a->u_work + (int64_t)b * N * m : 0, b, &p2, &d2);
            prim += SQRT(p2); dual += SQRT(d2);
        }
        REAL *res = (REAL *)a->res + (int64_t)b * 2;
        REAL *prev = a->res_prev ? (REAL *)a->res_prev + (int64_t)b * 2 : 0;
        if (a->active && prev) {
            int stop = 0;
            if (prim < (REAL)a->tol_abs && dual < (REAL)a->tol_abs) stop = 1;            /* admm.py:72 */
            else {
                REAL pc = FABS(prev[0] - prim) / (prev[0] + (REAL)1e-30);               /* admm.py:78-80 */
                REAL dc = FABS(prev[1] - dual) / (prev[1] + (REAL)1e-30);
                if (pc < (REAL)a->tol_rel && dc < (REAL)a->tol_rel) stop = 1;
            }
            if (stop) a->active[b] = 0;
        }
        res[0] = prim; res[1] = dual;
        if (prev) { prev[0] = prim; prev[1] = dual; }
        if (a->iters) a->iters[b] += 1;
    }
    return ISLS_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Quadratic expansion.  isls/isls.py:263-271 (Cts None branch) + regulariser (isls/sls.py:132-137);
 * cost: isls/sls_base.py:25-44.
 * ------------------------------------------------------------------------------------------- */
int FN(oracle_expand_quadratic)(const isls_expand_args *a)
{
    const int B = a->B, N = a->N, n = a->n, m = a->m;
    if (B < 0 || N < 1 || n < 1 || m < 1 || n > MAXN || m > MAXM) return ISLS_ERR_ARG;
    const REAL ustd = (REAL)a->u_std;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        if (a->active && !a->active[b]) continue;
        REAL cx_sum = 0, cu_sum = 0;
        if (a->cost_model == ISLS_COST_PHUBER) { /* what Tutorial.ipynb cell 16 gets from autograd: ph' = x/s, ph'' = p^2/s^3 */
            const REAL *cu = (const REAL *)a->cost_par, *cx = cu + m, *px = cx + n, *cf = px + n, *pf = cf + n;
            for (int t = 0; t < N; ++t) {
                const REAL *xh = a->xhat ? (const REAL *)a->xhat + ((int64_t)b * N + t) * n : 0;
                const REAL *uh = a->uhat ? (const REAL *)a->uhat + ((int64_t)b * N + t) * m : 0;
                REAL *c0x = (REAL *)a->c0x + ((int64_t)b * N + t) * n, *c0u = (REAL *)a->c0u + ((int64_t)b * N + t) * m;
                const REAL *Qr = a->Qr.p ? VIEW(a->Qr, b, t) : 0, *Rr = a->Rr.p ? VIEW(a->Rr, b, t) : 0;
                for (int i = 0; i < n; ++i) {
                    REAL x = xh ? xh[i] : (REAL)0, s1 = SQRT(x * x + px[i] * px[i]);
                    REAL g = cx[i] * (x / s1), h = cx[i] * ((px[i] * px[i]) / (s1 * s1 * s1)), c = cx[i] * (s1 - px[i]);
                    if (t == N - 1) {
                        REAL s2 = SQRT(x * x + pf[i] * pf[i]);
                        g += cf[i] * (x / s2); h += cf[i] * ((pf[i] * pf[i]) / (s2 * s2 * s2)); c += cf[i] * (s2 - pf[i]);
                    }
                    c0x[i] = g; cx_sum += c;
                    if (a->Cxx) {
                        REAL *C = (REAL *)a->Cxx + ((int64_t)b * N + t) * n * n + i * n;
                        for (int j = 0; j < n; ++j) C[j] = (j == i ? h : (REAL)0) + (Qr ? 2 * Qr[i * n + j] : (REAL)0);
                    }
                }
                for (int i = 0; i < m; ++i) {
                    REAL uu = uh ? uh[i] : (REAL)0;
                    c0u[i] = 2 * (cu[i] * uu); cu_sum += cu[i] * (uu * uu);
                    if (a->Cuu) {
                        REAL *C = (REAL *)a->Cuu + ((int64_t)b * N + t) * m * m + i * m;
                        for (int j = 0; j < m; ++j) C[j] = (j == i ? 2 * cu[i] : (REAL)0) + (Rr ? 2 * Rr[i * m + j] : (REAL)0);
                    }
                }
            }
            if (a->cost) ((REAL *)a->cost)[b] = cx_sum + cu_sum;
            continue;
        }
        const REAL *Qtab = (const REAL *)a->Qtab + (int64_t)b * a->Qtab_sb, *ztab = (const REAL *)a->ztab + (int64_t)b * a->ztab_sb;
        for (int t = 0; t < N; ++t) {
            const REAL *Q = Qtab + (int64_t)a->seq[t] * n * n, *z = ztab + (int64_t)a->seq[t] * n;
            const REAL *xh = a->xhat ? (const REAL *)a->xhat + ((int64_t)b * N + t) * n : 0;
            const REAL *uh = a->uhat ? (const REAL *)a->uhat + ((int64_t)b * N + t) * m : 0;
            REAL d[MAXN];
            for (int i = 0; i < n; ++i) d[i] = (xh ? xh[i] : (REAL)0) - z[i];
            REAL *c0x = (REAL *)a->c0x + ((int64_t)b * N + t) * n, *c0u = (REAL *)a->c0u + ((int64_t)b * N + t) * m;
            for (int i = 0; i < n; ++i) {
                REAL s = 0;
                for (int j = 0; j < n; ++j) s += Q[i * n + j] * d[j];
                c0x[i] = 2 * s;
                cx_sum += d[i] * s;
            }
            for (int i = 0; i < m; ++i) {
                REAL uu = uh ? uh[i] : (REAL)0;
                c0u[i] = 2 * (ustd * uu);
                cu_sum += uu * (ustd * uu);
            }
            if (a->Cxx) {
                REAL *C = (REAL *)a->Cxx + ((int64_t)b * N + t) * n * n;
                const REAL *Qr = a->Qr.p ? VIEW(a->Qr, b, t) : 0;
                for (int i = 0; i < n * n; ++i) C[i] = 2 * Q[i] + (Qr ? 2 * Qr[i] : (REAL)0);
            }
            if (a->Cuu) {
                REAL *C = (REAL *)a->Cuu + ((int64_t)b * N + t) * m * m;
                const REAL *Rr = a->Rr.p ? VIEW(a->Rr, b, t) : 0;
                for (int i = 0; i < m; ++i)
                    for (int j = 0; j < m; ++j) C[i * m + j] = ((i == j) ? 2 * ustd : (REAL)0) + (Rr ? 2 * Rr[i * m + j] : (REAL)0);
            }
        }
        if (a->cost) ((REAL *)a->cost)[b] = cx_sum + cu_sum;
    }
    return ISLS_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Linearisation of the built-in models (the notebooks' get_AB callbacks).
 * ------------------------------------------------------------------------------------------- */
int FN(oracle_linearize)(const isls_linearize_args *a)
{
    const int B = a->B, N = a->N, n = a->n, m = a->m;
    if (B < 0 || N < 1) return ISLS_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        if (a->active && !a->active[b]) continue;
        const REAL *par = (const REAL *)a->model_par + (int64_t)b * a->model_par_sb;
        for (int t = 0; t < N; ++t) {
            REAL *A = (REAL *)a->A + ((int64_t)b * N + t) * n * n, *Bm = (REAL *)a->Bm + ((int64_t)b * N + t) * n * m;
            const REAL *x = (const REAL *)a->xhat + ((int64_t)b * N + t) * n, *u = (const REAL *)a->uhat + ((int64_t)b * N + t) * m;
            for (int i = 0; i < n * n; ++i) A[i] = 0;
            for (int i = 0; i < n * m; ++i) Bm[i] = 0;
            if (a->model == ISLS_MODEL_LTI) {
                for (int i = 0; i < n * n; ++i) A[i] = par[i];
                for (int i = 0; i < n * m; ++i) Bm[i] = par[n * n + i];
            } else if (a->model == ISLS_MODEL_DI) {
                const int d = n / 2;
                for (int i = 0; i < n; ++i) A[i * n + i] = 1;
                for (int i = 0; i < d; ++i) { A[i * n + d + i] = par[0]; Bm[i * m + i] = par[1]; Bm[(d + i) * m + i] = par[2]; }
            } else if (a->model == ISLS_MODEL_ARM3R) {
                const REAL dt = par[0];
                REAL q[3], c = 0, sn[3], cs[3];
                for (int j = 0; j < 3; ++j) {
                    q[j] = x[j] + x[3 + j] * dt + (REAL)0.5 * u[j] * (dt * dt);
                    c += q[j]; sn[j] = SIN(c); cs[j] = COS(c);
                }
                for (int j = 0; j < 3; ++j) {
                    A[j * 9 + j] = 1; A[j * 9 + 3 + j] = dt; A[(3 + j) * 9 + 3 + j] = 1;
                    Bm[j * 3 + j] = (REAL)0.5 * (dt * dt); /* dt^2/2! */
                    Bm[(3 + j) * 3 + j] = dt;
                }
                for (int j = 0; j < 3; ++j) {
                    REAL j0 = 0, j1 = 0; /* J[0,j] = -sum_{i>=j} sin c_i, J[1,j] = sum_{i>=j} cos c_i */
                    for (int i = j; i < 3; ++i) { j0 += sn[i]; j1 += cs[i]; }
                    j0 = -j0;
                    A[6 * 9 + j] = j0; A[7 * 9 + j] = j1;
                    A[6 * 9 + 3 + j] = j0 * dt; A[7 * 9 + 3 + j] = j1 * dt;
                    Bm[6 * 3 + j] = ((REAL)0.5 * j0) * (dt * dt); Bm[7 * 3 + j] = ((REAL)0.5 * j1) * (dt * dt);
                }
            } else if (a->model == ISLS_MODEL_TASSA) { /* Tutorial.ipynb cell 8, differentiated by hand */
                const REAL dt = par[0], d = par[1];
                const REAL f = dt * x[3], sw = SIN(u[0]), cw = COS(u[0]);
                const REAL r = SQRT(d * d - (sw * f) * (sw * f));
                const REAL bb = (f * cw + d) - r, dbdf = cw + (sw * sw * f) / r, dbdw = -f * sw + (sw * cw * f * f) / r;
                const REAL st = SIN(x[2]), ct = COS(x[2]);
                for (int i = 0; i < 4; ++i) A[i * 4 + i] = 1;
                A[0 * 4 + 2] = -bb * st;
                A[1 * 4 + 2] = bb * ct;
                A[0 * 4 + 3] = (dbdf * dt) * ct;
                A[1 * 4 + 3] = (dbdf * dt) * st;
                A[2 * 4 + 3] = (sw / r) * dt;
                Bm[0 * 2 + 0] = dbdw * ct;
                Bm[1 * 2 + 0] = dbdw * st;
                Bm[2 * 2 + 0] = (cw * f) / r;
                Bm[3 * 2 + 1] = dt;
            } else {
                const REAL dt = par[0];
                for (int i = 0; i < 4; ++i) A[i * 4 + i] = 1;
                A[0 * 4 + 2] = -dt * x[3] * SIN(x[2]);
                A[1 * 4 + 2] = dt * x[3] * COS(x[2]);
                A[0 * 4 + 3] = dt * COS(x[2]);
                A[1 * 4 + 3] = dt * SIN(x[2]);
                A[2 * 4 + 3] = dt * u[0];
                Bm[2 * 2 + 0] = dt * x[3];
                Bm[3 * 2 + 1] = dt;
            }
        }
    }
    return ISLS_OK;
}

/* End of an outer iteration: nominal <- x-step, cost_log tail, stop rules.  isls/isls.py:488-499,
 * isls/isls_base.py:80-85. */
int FN(oracle_accept_step)(const isls_accept_args *a)
{
    const int B = a->B, N = a->N, n = a->n, m = a->m;
    for (int b = 0; b < B; ++b) {
        if (a->outer_active && !a->outer_active[b]) continue;
        memcpy((REAL *)a->xhat + (int64_t)b * N * n, (const REAL *)a->xx + (int64_t)b * N * n, sizeof(REAL) * (size_t)N * n);
        memcpy((REAL *)a->uhat + (int64_t)b * N * m, (const REAL *)a->xu + (int64_t)b * N * m, sizeof(REAL) * (size_t)N * m);
        REAL *cost = (REAL *)a->cost;
        const REAL prev = cost[b], cur = ((const REAL *)a->cost_new)[b];
        cost[b] = cur;
        int stop = 0;
        if (a->tol_cost >= 0 && FABS(cur - prev) < (REAL)a->tol_cost) stop = 1;
        if (a->cost_hist) {
            REAL *h = (REAL *)a->cost_hist + (int64_t)b * 8;
            int len = a->hist_len[b];
            if (len < 8) h[len++] = cur;
            else { for (int i = 0; i < 7; ++i) h[i] = h[i + 1]; h[7] = cur; }
            a->hist_len[b] = len;
            if (!stop && a->tol_osc >= 0 && len >= 5) {
                REAL a4 = 0, b4 = 0;
                for (int i = len - 4; i < len; ++i) a4 += h[i];
                for (int i = 0; i < len - 4; ++i) b4 += h[i];
                if (FABS(a4 / 4 - b4 / (REAL)(len - 4)) < (REAL)a->tol_osc) stop = 1;
            }
        }
        if (stop && a->outer_active) a->outer_active[b] = 0;
    }
    return ISLS_OK;
}

int FN(oracle_reduce_convergence)(int32_t B, const void *cost, const void *res, const int32_t *active,
                                  const int32_t *status, void *out5)
{
    REAL *o = (REAL *)out5;
    REAL cs = 0, pm = 0, dm = 0, na = 0, nf = 0;
    for (int b = 0; b < B; ++b) {
        if (cost) cs += ((const REAL *)cost)[b];
        if (res) { REAL p = ((const REAL *)res)[2 * b], d = ((const REAL *)res)[2 * b + 1]; if (p > pm) pm = p; if (d > dm) dm = d; }
        if (active) na += active[b] ? 1 : 0; else na += 1;
        if (status) nf += status[b] ? 1 : 0;
    }
    o[0] = cs; o[1] = pm; o[2] = dm; o[3] = na; o[4] = nf;
    return ISLS_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Row-wise projections: isls/projections.py 7-11 (project_bound), 140-162 (project_soc_unit_batch),
 * 256-266 (project_square_batch; affine keep-out form of the car notebook, cell 18) and 289-374
 * (project_set_convex, all rows of one problem stopped together).
 * ------------------------------------------------------------------------------------------- */
static REAL FN(npsign)(REAL x) { return x > 0 ? (REAL)1 : (x < 0 ? (REAL)-1 : (REAL)0); }

static void FN(primitive)(int kind, int dim, const REAL *par, REAL *v)
{
    if (kind == ISLS_SET_BOX) {
        for (int i = 0; i < dim; ++i) {
            REAL a = v[i] < par[i] ? par[i] : v[i];
            v[i] = a > par[dim + i] ? par[dim + i] : a;
        }
    } else if (kind == ISLS_SET_SOC_UNIT) {
        REAL ss = 0, t = v[dim - 1];
        for (int i = 0; i < dim - 1; ++i) ss += v[i] * v[i];
        REAL zn = SQRT(ss);
        int cond1 = (zn <= -t) || (t < 0), cond2 = (zn > t) || (zn > -t), cond3 = zn <= t;
        REAL tmp = (zn + t) / 2, o[ISLS_MAX_SET_DIM];
        for (int i = 0; i < dim; ++i) o[i] = v[i];
        if (cond2) { for (int i = 0; i < dim - 1; ++i) o[i] = tmp * v[i] / (zn + (REAL)1e-30); o[dim - 1] = tmp; }
        if (cond1) for (int i = 0; i < dim; ++i) o[i] = 0;
        if (cond3) for (int i = 0; i < dim; ++i) o[i] = v[i];
        for (int i = 0; i < dim; ++i) v[i] = o[i];
    } else if (kind == ISLS_SET_LINEAR) {     /* project_linear_batch, isls/projections.py:30-43 */
        REAL l = par[0], u = par[1], atx = 0, ata = 0;
        const REAL *a = par + 2;
        for (int i = 0; i < dim; ++i) { atx += v[i] * a[i]; ata += a[i] * a[i]; }
        ata += (REAL)1e-30;
        int hi = atx > u, lo = atx < l;
        for (int i = 0; i < dim; ++i) {
            REAL tmp = a[i] / ata, o = v[i];
            if (hi) o = o - (atx - u) * tmp;
            if (lo) o = o - (atx - l) * tmp;
            v[i] = o;
        }
    } else if (kind == ISLS_SET_QUADRATIC) {  /* project_quadratic_batch, isls/projections.py:91-105 */
        REAL l = par[0], u = par[1], ss = 0;
        for (int i = 0; i < dim; ++i) ss += v[i] * v[i];
        REAL val = (REAL)0.5 * ss, nrm = SQRT(ss);
        int hi = val > u, lo = l > val;
        REAL su = SQRT(2 * u), sl = SQRT(2 * l);
        for (int i = 0; i < dim; ++i) {
            REAL o = v[i];
            if (hi) o = v[i] * su / nrm;
            if (lo) o = v[i] * sl / nrm;
            v[i] = o;
        }
    } else if (kind == ISLS_SET_SHELL) {      /* project_quadratic_batch(y - c, l, u) + c  (obstacle notebook, cell 12) */
        REAL l = par[0], u = par[1], ss = 0, w[ISLS_MAX_SET_DIM];
        const REAL *c = par + 2;
        for (int i = 0; i < dim; ++i) { w[i] = v[i] - c[i]; ss += w[i] * w[i]; }
        REAL val = (REAL)0.5 * ss, nrm = SQRT(ss);
        int hi = val > u, lo = l > val;
        REAL su = SQRT(2 * u), sl = SQRT(2 * l);
        for (int i = 0; i < dim; ++i) {
            REAL o = w[i];
            if (hi) o = w[i] * su / nrm;
            if (lo) o = w[i] * sl / nrm;
            v[i] = o + c[i];
        }
    } else if (kind == ISLS_SET_MULTILINEAR) { /* project_multilinear, isls/projections.py:46-61 */
        int q = (int)par[0];
        const REAL *l = par + 1, *u = l + q, *Mm = u + q;
        REAL Ax[ISLS_MAX_SET_DIM], G[ISLS_MAX_SET_DIM * ISLS_MAX_SET_DIM], Gi[ISLS_MAX_SET_DIM * ISLS_MAX_SET_DIM], mu[ISLS_MAX_SET_DIM];
        for (int i = 0; i < q; ++i) { REAL acc = 0; for (int j = 0; j < dim; ++j) acc += Mm[i * dim + j] * v[j]; Ax[i] = acc; }
        for (int i = 0; i < q; ++i)
            for (int k = 0; k < q; ++k) { REAL acc = 0; for (int j = 0; j < dim; ++j) acc += Mm[i * dim + j] * Mm[k * dim + j]; G[i * q + k] = acc; }
        if (FN(inv_lu)(G, Gi, q)) return;
        for (int i = 0; i < q; ++i) {
            REAL t = Ax[i];
            if (Ax[i] > u[i]) t = u[i];
            if (Ax[i] < l[i]) t = l[i];
            Ax[i] = Ax[i] - t;
        }
        for (int i = 0; i < q; ++i) { REAL acc = 0; for (int k = 0; k < q; ++k) acc += Gi[i * q + k] * Ax[k]; mu[i] = acc; }
        for (int j = 0; j < dim; ++j) { REAL acc = 0; for (int i = 0; i < q; ++i) acc += Mm[i * dim + j] * mu[i]; v[j] = v[j] - acc; }
    } else if (kind == ISLS_SET_SQUARE) {
        int q = (int)par[0];
        REAL l = par[1], u = par[2];
        const REAL *c = par + 3, *W = c + q, *Wi = W + q * q;
        REAL y[ISLS_MAX_SET_DIM], w[ISLS_MAX_SET_DIM];
        for (int i = 0; i < q; ++i) y[i] = v[i] - c[i];
        for (int i = 0; i < q; ++i) { REAL acc = 0; for (int j = 0; j < q; ++j) acc += y[j] * W[i * q + j]; w[i] = acc; }
        int jm = 0;
        REAL am = -1;
        for (int i = 0; i < q; ++i) { REAL a = FABS(w[i]); if (a > am) { am = a; jm = i; } }
        if (am < l) w[jm] = l * FN(npsign)(w[jm]);
        for (int i = 0; i < q; ++i) { REAL o = w[i] > u ? u : w[i]; w[i] = o < -u ? -u : o; }
        for (int i = 0; i < q; ++i) { REAL acc = 0; for (int j = 0; j < q; ++j) acc += w[j] * Wi[i * q + j]; v[i] = acc + c[i]; }
    }
}

int FN(oracle_project_rows)(const isls_project_args *a)
{
    const int P = a->P, R = a->R, d = a->d, ns = a->nsets;
    if (P < 0 || R < 1 || d < 1 || d > ISLS_MAX_ROW_DIM || ns < 1 || ns > ISLS_MAX_SETS || !a->y_in || !a->y_out) return ISLS_ERR_ARG;
    const int alg = a->algorithm;
    const int direct = alg == ISLS_PROJ_ALG_ADMM && ns == 1 && a->sets[0].A == 0;
    const REAL rho = (REAL)a->rho, thr = (REAL)a->threshold;
    const int32_t *mask = a->row_mask;
    if (alg == ISLS_PROJ_ALG_SOC && (ns != 1 || a->sets[0].kind != ISLS_SET_SOC_UNIT || !a->sets[0].A || !a->sets[0].b)) return ISLS_ERR_ARG;
    int rc = ISLS_OK;
#pragma omp parallel for schedule(dynamic)
    for (int p = 0; p < P; ++p) {
        if (a->active && !a->active[p]) continue;
        const REAL *yin = (const REAL *)a->y_in + (int64_t)p * a->in_sp;
        REAL *yout = (REAL *)a->y_out + (int64_t)p * a->out_sp;
        if (mask)                                                  /* rows outside the mask pass through */
            for (int r = 0; r < R; ++r)
                if (!mask[r])
                    for (int j = 0; j < d; ++j) yout[(int64_t)r * a->out_sr + j] = yin[(int64_t)r * a->in_sr + j];
        if (alg == ISLS_PROJ_ALG_DYKSTRA) {                        /* project_set_convex_dykstra, isls/projections.py:465-504 */
            REAL *uu = (REAL *)malloc(sizeof(REAL) * (size_t)R * ISLS_MAX_ROW_DIM * (1 + ISLS_MAX_SETS));
            REAL *zz = uu + (size_t)R * ISLS_MAX_ROW_DIM;
            for (int r = 0; r < R; ++r) {
                for (int j = 0; j < d; ++j) uu[r * ISLS_MAX_ROW_DIM + j] = yin[(int64_t)r * a->in_sr + j];
                for (int i = 0; i < ISLS_MAX_SETS * ISLS_MAX_ROW_DIM; ++i) zz[(size_t)r * ISLS_MAX_SETS * ISLS_MAX_ROW_DIM + i] = 0;
            }
            int k = 0;
            REAL cmax = 10;
            while (k <= a->max_iter && cmax >= thr) {              /* np.any(cI >= tol) */
                cmax = 0;
                for (int r = 0; r < R; ++r) {
                    if (mask && !mask[r]) continue;
                    REAL *u = uu + r * ISLS_MAX_ROW_DIM, *z = zz + (size_t)r * ISLS_MAX_SETS * ISLS_MAX_ROW_DIM, cI = 0;
                    for (int s = 0; s < ns; ++s) {
                        const REAL *par = a->sets[s].par ? (const REAL *)a->sets[s].par + (int64_t)p * a->sets[s].par_sp : 0;
                        REAL v[ISLS_MAX_SET_DIM], prev_u[ISLS_MAX_ROW_DIM], nn = 0;
                        for (int j = 0; j < d; ++j) { prev_u[j] = u[j]; v[j] = prev_u[j] - z[s * ISLS_MAX_ROW_DIM + j]; }
                        FN(primitive)(a->sets[s].kind, d, par, v);
                        for (int j = 0; j < d; ++j) {
                            const REAL prev_z = z[s * ISLS_MAX_ROW_DIM + j];
                            const REAL zn = v[j] - (prev_u[j] - prev_z);
                            u[j] = v[j];
                            z[s * ISLS_MAX_ROW_DIM + j] = zn;
                            nn += (prev_z - zn) * (prev_z - zn);
                        }
                        cI += SQRT(nn) * SQRT(nn);                  /* np.linalg.norm(...)**2 */
                    }
                    if (cI > cmax) cmax = cI;
                }
                ++k;
            }
            for (int r = 0; r < R; ++r)
                if (!mask || mask[r])
                    for (int j = 0; j < d; ++j) yout[(int64_t)r * a->out_sr + j] = uu[r * ISLS_MAX_ROW_DIM + j];
            if (a->iters) a->iters[p] = k;
            free(uu);
            continue;
        }
        if (alg == ISLS_PROJ_ALG_SOC) {                            /* project_soc, isls/projections.py:163-234 */
            const int dm = a->sets[0].dim;
            const REAL *A0 = (const REAL *)a->sets[0].A + (int64_t)p * a->sets[0].A_sp, *b0 = (const REAL *)a->sets[0].b + (int64_t)p * a->sets[0].b_sp;
            REAL M[ISLS_MAX_ROW_DIM * ISLS_MAX_ROW_DIM], Li[ISLS_MAX_ROW_DIM * ISLS_MAX_ROW_DIM];
            for (int j = 0; j < d; ++j)
                for (int k = 0; k < d; ++k) {
                    REAL acc = 0;
                    for (int i = 0; i < dm; ++i) acc += A0[i * d + j] * A0[i * d + k];
                    M[j * d + k] = (j == k ? (REAL)1 : (REAL)0) + rho * acc;
                }
            if (FN(inv_lu)(M, Li, d)) { rc = ISLS_ERR_ARG; continue; }
            REAL *zz = (REAL *)malloc(sizeof(REAL) * (size_t)R * (ISLS_MAX_ROW_DIM + ISLS_MAX_SET_DIM));
            REAL *ll = zz + (size_t)R * ISLS_MAX_ROW_DIM;
            for (int r = 0; r < R; ++r) {
                for (int j = 0; j < d; ++j) zz[r * ISLS_MAX_ROW_DIM + j] = yin[(int64_t)r * a->in_sr + j];
                for (int i = 0; i < ISLS_MAX_SET_DIM; ++i) ll[r * ISLS_MAX_SET_DIM + i] = 0;
            }
            REAL prim_g = (REAL)1e5, dual_g = (REAL)1e5;
            int it = 0;
            for (int j = 0; j < a->max_iter; ++j) {
                ++it;
                REAL prev_p = prim_g, prev_d = dual_g, pm = 0, dmx = 0;
                for (int r = 0; r < R; ++r) {
                    if (mask && !mask[r]) continue;
                    const REAL *z0 = yin + (int64_t)r * a->in_sr;
                    REAL *z = zz + r * ISLS_MAX_ROW_DIM, *lm = ll + r * ISLS_MAX_SET_DIM;
                    REAL x[ISLS_MAX_SET_DIM], zp[ISLS_MAX_ROW_DIM], rs[ISLS_MAX_ROW_DIM], pn = 0, dn = 0;
                    for (int i = 0; i < dm; ++i) {
                        REAL acc = 0;
                        for (int k = 0; k < d; ++k) acc += A0[i * d + k] * z[k];
                        x[i] = (acc + b0[i]) + lm[i];
                    }
                    FN(primitive)(ISLS_SET_SOC_UNIT, dm, 0, x);
                    for (int k = 0; k < d; ++k) { zp[k] = z[k]; rs[k] = 0; }
                    for (int i = 0; i < dm; ++i) {
                        const REAL w = (-b0[i] + x[i]) - lm[i];
                        for (int k = 0; k < d; ++k) rs[k] += A0[i * d + k] * w;
                    }
                    for (int i = 0; i < d; ++i) {
                        REAL acc = 0;
                        for (int k = 0; k < d; ++k) acc += Li[i * d + k] * (z0[k] + rho * rs[k]);
                        z[i] = acc;
                    }
                    for (int i = 0; i < dm; ++i) {
                        REAL acc = 0;
                        for (int k = 0; k < d; ++k) acc += A0[i * d + k] * z[k];
                        const REAL pr = (acc + b0[i]) - x[i];
                        lm[i] += pr;
                        pn += pr * pr;
                    }
                    for (int k = 0; k < d; ++k) dn += (rho * (z[k] - zp[k])) * (rho * (z[k] - zp[k]));
                    pn = SQRT(pn); dn = SQRT(dn);
                    if (pn > pm) pm = pn;
                    if (dn > dmx) dmx = dn;
                }
                prim_g = pm; dual_g = dmx;
                if (prim_g < thr && dual_g < thr) break;
                if (j != a->max_iter - 1) {
                    REAL pc = FABS(prev_p - prim_g) / (prev_p + (REAL)1e-30), dc = FABS(prev_d - dual_g) / (prev_d + (REAL)1e-30);
                    if (pc < (REAL)1e-5 && dc < (REAL)1e-5) break;
                }
            }
            for (int r = 0; r < R; ++r)
                if (!mask || mask[r])
                    for (int j = 0; j < d; ++j) yout[(int64_t)r * a->out_sr + j] = zz[r * ISLS_MAX_ROW_DIM + j];
            if (a->iters) a->iters[p] = it;
            free(zz);
            continue;
        }
        if (direct) {
            const REAL *par = a->sets[0].par ? (const REAL *)a->sets[0].par + (int64_t)p * a->sets[0].par_sp : 0;
            for (int r = 0; r < R; ++r) {
                if (mask && !mask[r]) continue;
                REAL v[ISLS_MAX_SET_DIM];
                for (int j = 0; j < d; ++j) v[j] = yin[(int64_t)r * a->in_sr + j];
                FN(primitive)(a->sets[0].kind, d, par, v);
                for (int j = 0; j < d; ++j) yout[(int64_t)r * a->out_sr + j] = v[j];
            }
            if (a->iters) a->iters[p] = 0;
            continue;
        }
        const REAL *A[ISLS_MAX_SETS], *b[ISLS_MAX_SETS], *par[ISLS_MAX_SETS];
        int dim[ISLS_MAX_SETS];
        for (int s = 0; s < ns; ++s) {
            A[s] = (const REAL *)a->sets[s].A + (int64_t)p * a->sets[s].A_sp;
            b[s] = (const REAL *)a->sets[s].b + (int64_t)p * a->sets[s].b_sp;
            par[s] = a->sets[s].par ? (const REAL *)a->sets[s].par + (int64_t)p * a->sets[s].par_sp : 0;
            dim[s] = a->sets[s].dim;
        }
        /* l_side_inv = inv(I + rho sum A'A)  (np.linalg.inv: LU with partial pivoting) */
        REAL M[ISLS_MAX_ROW_DIM * ISLS_MAX_ROW_DIM], Linv[ISLS_MAX_ROW_DIM * ISLS_MAX_ROW_DIM];
        for (int i = 0; i < d * d; ++i) M[i] = 0;
        for (int s = 0; s < ns; ++s)
            for (int j = 0; j < d; ++j)
                for (int k = 0; k < d; ++k) {
                    REAL acc = 0;
                    for (int i = 0; i < dim[s]; ++i) acc += A[s][i * d + j] * A[s][i * d + k];
                    M[j * d + k] += acc;
                }
        for (int j = 0; j < d; ++j)
            for (int k = 0; k < d; ++k) M[j * d + k] = (j == k ? (REAL)1 : (REAL)0) + rho * M[j * d + k];
        if (FN(inv_lu)(M, Linv, d)) { rc = ISLS_ERR_ARG; continue; }
        const int SD = ISLS_MAX_SETS * ISLS_MAX_SET_DIM;
        REAL *x = (REAL *)malloc(sizeof(REAL) * (size_t)R * (ISLS_MAX_ROW_DIM + 2 * SD));
        REAL *z = x + (size_t)R * ISLS_MAX_ROW_DIM, *lm = z + (size_t)R * SD;
        for (int r = 0; r < R; ++r) {
            const REAL *x0 = yin + (int64_t)r * a->in_sr;
            for (int j = 0; j < d; ++j) x[r * ISLS_MAX_ROW_DIM + j] = x0[j];
            for (int s = 0; s < ns; ++s)
                for (int i = 0; i < dim[s]; ++i) {
                    REAL acc = 0;
                    for (int j = 0; j < d; ++j) acc += A[s][i * d + j] * x0[j];
                    z[r * SD + s * ISLS_MAX_SET_DIM + i] = acc + b[s][i];
                    lm[r * SD + s * ISLS_MAX_SET_DIM + i] = 0;
                }
        }
        REAL prim_g = (REAL)1e5, dual_g = (REAL)1e5;
        int it = 0;
        for (int j = 0; j < a->max_iter; ++j) {
            ++it;
            REAL prev_p = prim_g, prev_d = dual_g, pm = 0, dm = 0;
            for (int r = 0; r < R; ++r) {
                if (mask && !mask[r]) continue;
                const REAL *x0 = yin + (int64_t)r * a->in_sr;
                REAL *xr = x + r * ISLS_MAX_ROW_DIM, *zr = z + r * SD, *lr = lm + r * SD;
                REAL rs[ISLS_MAX_ROW_DIM];
                for (int k = 0; k < d; ++k) rs[k] = 0;
                for (int s = 0; s < ns; ++s)
                    for (int i = 0; i < dim[s]; ++i) {
                        REAL w = (-b[s][i] + zr[s * ISLS_MAX_SET_DIM + i]) - lr[s * ISLS_MAX_SET_DIM + i];
                        for (int k = 0; k < d; ++k) rs[k] += A[s][i * d + k] * w;
                    }
                for (int i = 0; i < d; ++i) {
                    REAL acc = 0;
                    for (int k = 0; k < d; ++k) acc += Linv[i * d + k] * (x0[k] + rho * rs[k]);
                    xr[i] = acc;
                }
                for (int s = 0; s < ns; ++s) {
                    REAL axb[ISLS_MAX_SET_DIM], v[ISLS_MAX_SET_DIM], dres[ISLS_MAX_ROW_DIM], pn = 0, dn = 0;
                    for (int i = 0; i < dim[s]; ++i) {
                        REAL acc = 0;
                        for (int k = 0; k < d; ++k) acc += A[s][i * d + k] * xr[k];
                        axb[i] = acc + b[s][i];
                        v[i] = axb[i] + lr[s * ISLS_MAX_SET_DIM + i];
                    }
                    FN(primitive)(a->sets[s].kind, dim[s], par[s], v);
                    for (int k = 0; k < d; ++k) dres[k] = 0;
                    for (int i = 0; i < dim[s]; ++i) {
                        REAL pr = axb[i] - v[i], dz = v[i] - zr[s * ISLS_MAX_SET_DIM + i];
                        for (int k = 0; k < d; ++k) dres[k] += A[s][i * d + k] * dz;
                        lr[s * ISLS_MAX_SET_DIM + i] += pr;
                        zr[s * ISLS_MAX_SET_DIM + i] = v[i];
                        pn += pr * pr;
                    }
                    for (int k = 0; k < d; ++k) dn += (rho * dres[k]) * (rho * dres[k]);
                    pn = SQRT(pn); dn = SQRT(dn);
                    if (pn > pm) pm = pn;
                    if (dn > dm) dm = dn;
                }
            }
            prim_g = pm; dual_g = dm;
            if (prim_g < thr && dual_g < thr) break;
            if (j != a->max_iter - 1) {
                REAL pc = FABS(prev_p - prim_g) / (prev_p + (REAL)1e-30), dc = FABS(prev_d - dual_g) / (prev_d + (REAL)1e-30);
                if (pc < (REAL)1e-5 && dc < (REAL)1e-5) break;
            }
        }
        for (int r = 0; r < R; ++r)
            if (!mask || mask[r])
                for (int j = 0; j < d; ++j) yout[(int64_t)r * a->out_sr + j] = x[r * ISLS_MAX_ROW_DIM + j];
        if (a->iters) a->iters[p] = it;
        free(x);
    }
    if (rc == ISLS_OK && a->next) {                                /* next stage, in place on this stage's output */
        isls_project_args nx = *a->next;
        nx.P = P; nx.R = R; nx.d = d; nx.y_in = a->y_out; nx.y_out = a->y_out;
        nx.in_sp = nx.out_sp = a->out_sp; nx.in_sr = nx.out_sr = a->out_sr; nx.iters = 0; nx.active = a->active;
        rc = FN(oracle_project_rows)(&nx);
    }
    return rc;
}

/* ---------------------------------------------------------------------------------------------
 * Closed loop of a dense causal controller about a nominal (iSLSBase.get_trajectory_sls,
 * isls/isls_base.py:28-42, noise_scale = 0): x_vec holds x_j - x_nom[j] for j <= i, zeros beyond;
 * u_i = (x_vec @ K.T + k)[i] + u_nom[i] ; x_{i+1} = forward_model(x_i, u_i).
 * ------------------------------------------------------------------------------------------- */
int FN(oracle_dense_closed_loop)(const isls_dense_loop_args *a)
{
    const int M = a->M, N = a->N, n = a->n, m = a->m;
    if (M < 0 || N < 1 || n < 1 || m < 1 || n > MAXN || m > MAXM) return ISLS_ERR_ARG;
    const REAL *K = (const REAL *)a->K, *k = (const REAL *)a->k, *xh = (const REAL *)a->xhat, *uh = (const REAL *)a->uhat;
#pragma omp parallel for schedule(static)
    for (int s = 0; s < M; ++s) {
        REAL *xs = (REAL *)a->x_log + (int64_t)s * N * n, *us = (REAL *)a->u_log + (int64_t)s * N * m;
        REAL x[MAXN], xn[MAXN], u[MAXM];
        for (int j = 0; j < n; ++j) x[j] = ((const REAL *)a->x0)[(int64_t)s * n + j];
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < n; ++j) xs[i * n + j] = x[j];
            for (int r = 0; r < m; ++r) {
                const REAL *Kr = K + (int64_t)(i * m + r) * N * n;
                REAL acc = 0;
                for (int j = 0; j < (i + 1) * n; ++j) acc += (xs[j] - (xh ? xh[j] : (REAL)0)) * Kr[j];
                u[r] = (acc + k[i * m + r]) + (uh ? uh[i * m + r] : (REAL)0);
                us[i * m + r] = u[r];
            }
            FN(model_step)(a->model, (const REAL *)a->model_par, n, m, x, u, xn);
            for (int j = 0; j < n; ++j) x[j] = xn[j];
        }
    }
    return ISLS_OK;
}

/* ---------------------------------------------------------------------------------------------
 * SLS-ADMM, control chance constraints: the loop of SLS.ADMM_SLS, isls/sls.py:363-449 (project_u only;
 * f_argmin 375-384, z/lambda update 398-403, Rr-scaled Frobenius residuals 405-412, stop rules 415-430).
 * ------------------------------------------------------------------------------------------- */
int FN(oracle_sls_admm)(const isls_sls_admm_args *a)
{
    const int P = a->P, R = a->R, D = a->D;
    if (P < 0 || R < 1 || D < 1 || D > ISLS_MAX_ROW_DIM || a->max_iter < 1 || !a->Linv || !a->r_side || !a->rr || !a->x_u) return ISLS_ERR_ARG;
    const REAL *Linv = (const REAL *)a->Linv, *rr = (const REAL *)a->rr;
    const REAL alpha = (REAL)a->alpha, tol = (REAL)a->tol;
    int rc_all = ISLS_OK;
#pragma omp parallel for schedule(dynamic)
    for (int p = 0; p < P; ++p) {
        const REAL *rside = (const REAL *)a->r_side + (int64_t)p * R * D;
        REAL *buf = (REAL *)calloc((size_t)R * D * 6, sizeof(REAL));
        REAL *x = buf, *z = x + R * D, *lm = z + R * D, *rhs = lm + R * D, *v = rhs + R * D, *zn = v + R * D;
        isls_project_args pr = a->proj;
        pr.P = 1; pr.R = R; pr.d = D; pr.y_in = v; pr.y_out = zn; pr.in_sp = pr.out_sp = 0; pr.in_sr = pr.out_sr = D;
        pr.iters = 0; pr.active = 0;
        for (int s = 0; s < pr.nsets; ++s) {               /* this problem's operands */
            if (pr.sets[s].A) pr.sets[s].A = (const REAL *)pr.sets[s].A + (int64_t)p * pr.sets[s].A_sp;
            if (pr.sets[s].b) pr.sets[s].b = (const REAL *)pr.sets[s].b + (int64_t)p * pr.sets[s].b_sp;
            if (pr.sets[s].par) pr.sets[s].par = (const REAL *)pr.sets[s].par + (int64_t)p * pr.sets[s].par_sp;
            pr.sets[s].A_sp = pr.sets[s].b_sp = pr.sets[s].par_sp = 0;
        }
        REAL prim = (REAL)1e6, dual = (REAL)1e6;
        int it = 0;
        for (int j = 0; j < a->max_iter; ++j) {
            ++it;
            for (int i = 0; i < R; ++i)
                for (int c = 0; c < D; ++c) rhs[i * D + c] = rside[i * D + c] + rr[i] * (z[i * D + c] - lm[i * D + c]);
            for (int i = 0; i < R; ++i)
                for (int c = 0; c < D; ++c) {
                    REAL acc = 0;
                    for (int k = 0; k < R; ++k) acc += Linv[(int64_t)i * R + k] * rhs[k * D + c];
                    x[i * D + c] = acc;
                }
            for (int e = 0; e < R * D; ++e) v[e] = (alpha * x[e] + (1 - alpha) * z[e]) + lm[e];
            int rc = FN(oracle_project_rows)(&pr);
            if (rc != ISLS_OK) { rc_all = rc; break; }
            REAL prev_p = prim, prev_d = dual, p2 = 0, d2 = 0;
            for (int i = 0; i < R; ++i)
                for (int c = 0; c < D; ++c) {
                    const int e = i * D + c;
                    REAL prs = x[e] - zn[e], dz = zn[e] - z[e];
                    lm[e] += prs;
                    z[e] = zn[e];
                    p2 += (rr[i] * prs) * (rr[i] * prs);
                    d2 += (rr[i] * dz) * (rr[i] * dz);
                }
            prim = SQRT(p2); dual = SQRT(d2);
            if (a->logs) { REAL *lg = (REAL *)a->logs + ((int64_t)p * a->max_iter + j) * 2; lg[0] = prim; lg[1] = dual; }
            if (prim < tol && dual < tol) break;
            REAL pc = FABS(prev_p - prim) / (prev_p + (REAL)1e-30), dc = FABS(prev_d - dual) / (prev_d + (REAL)1e-30);
            if (pc < (REAL)a->rel_tol && dc < (REAL)a->rel_tol) break;
        }
        for (int e = 0; e < R * D; ++e) {
            ((REAL *)a->x_u)[(int64_t)p * R * D + e] = x[e];
            if (a->z) ((REAL *)a->z)[(int64_t)p * R * D + e] = z[e];
            if (a->lmb) ((REAL *)a->lmb)[(int64_t)p * R * D + e] = lm[e];
        }
        if (a->iters) a->iters[p] = it;
        free(buf);
    }
    return rc_all;
}

/* Closed loop of the dense causal controller, isls/sls_base.py:91-105 (noise_scale = 0). */
int FN(oracle_sls_closed_loop)(int32_t M, int32_t N, int32_t n, int32_t m, const void *A_, const void *B_, const void *K_,
                               const void *k_, const void *x0_, void *x_log, void *u_log)
{
    const REAL *A = (const REAL *)A_, *B = (const REAL *)B_, *K = (const REAL *)K_, *k = (const REAL *)k_, *x0 = (const REAL *)x0_;
    if (M < 0 || N < 1 || n < 1 || m < 1 || !A || !B || !K || !k || !x0 || !x_log || !u_log) return ISLS_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int s = 0; s < M; ++s) {
        REAL *xs = (REAL *)x_log + (int64_t)s * N * n, *us = (REAL *)u_log + (int64_t)s * N * m;
        for (int j = 0; j < n; ++j) xs[j] = x0[(int64_t)s * n + j];
        for (int i = 0; i < N; ++i) {
            for (int r = 0; r < m; ++r) {
                const REAL *Kr = K + (int64_t)(i * m + r) * N * n;
                REAL acc = 0;
                for (int j = 0; j < (i + 1) * n; ++j) acc += xs[j] * Kr[j];
                us[i * m + r] = acc + k[i * m + r];
            }
            if (i + 1 < N)
                for (int a = 0; a < n; ++a) {
                    REAL acc = 0;
                    for (int j = 0; j < n; ++j) acc += A[a * n + j] * xs[i * n + j];
                    for (int r = 0; r < m; ++r) acc += B[a * m + r] * us[i * m + r];
                    xs[(i + 1) * n + a] = acc;
                }
        }
    }
    return ISLS_OK;
}

/* One outer iteration: gain -> J x [ff -> rollout -> update]  (SURVEY 3.3 / 8d metric definition). */
int FN(oracle_ilqr_admm_outer)(const isls_outer_args *a)
{
    int rc;
    const int B = a->gain.B, N = a->admm.N, n = a->admm.n, m = a->admm.m;
    /* start of an outer iteration: admm_active <- outer_active, lambda <- 0 (isls.py:414-415,482),
     * previous residual norms <- 1e6 (admm.py:25-26) for the trajectories still iterating */
    for (int b = 0; b < B && !a->begin_done; ++b) {
        const int act = a->outer_active ? (a->outer_active[b] != 0) : 1;
        if (a->admm.active) a->admm.active[b] = act;
        if (!act) continue;
        if (a->admm.iters) a->admm.iters[b] = 0;
        if (a->admm.lx) for (int e = 0; e < N * n; ++e) ((REAL *)a->admm.lx)[(int64_t)b * N * n + e] = 0;
        if (a->admm.lu) for (int e = 0; e < N * m; ++e) ((REAL *)a->admm.lu)[(int64_t)b * N * m + e] = 0;
        if (a->admm.res_prev) { ((REAL *)a->admm.res_prev)[2 * b] = (REAL)1e6; ((REAL *)a->admm.res_prev)[2 * b + 1] = (REAL)1e6; }
    }
    if (!a->skip_gain && (rc = FN(oracle_riccati_gain)(&a->gain)) != ISLS_OK) return rc;
    for (int j = 0; j < a->J; ++j) {
        if ((rc = FN(oracle_riccati_ff)(&a->ff)) != ISLS_OK) return rc;
        if ((rc = FN(oracle_rollout_ls)(&a->ro)) != ISLS_OK) return rc;
        isls_admm_args ad = a->admm;
        if ((rc = FN(oracle_admm_update)(&ad)) != ISLS_OK) return rc;
        if (a->log) memcpy((REAL *)a->log + (int64_t)j * B * 2, ad.res, sizeof(REAL) * (size_t)B * 2);
    }
    return ISLS_OK;
}

/* isls_outer_advance_* (include/isls_hip.h): the end of an outer iteration and the start of the next, as the plain sequence
 * of the four steps the reference takes there -- accept (isls.py:488-499), the ADMM restart (isls.py:414-415,482,
 * admm.py:25-26), linearisation and cost expansion about the new nominal (isls.py:61-66,95-102) for the trajectories that go on. */
int FN(oracle_outer_advance)(const isls_advance_args *a)
{
    int rc;
    const isls_accept_args *ac = &a->accept;
    const int B = ac->B, N = ac->N, n = ac->n, m = ac->m;
    if ((rc = FN(oracle_accept_step)(ac)) != ISLS_OK) return rc;
    for (int b = 0; b < B; ++b) {
        const int act = ac->outer_active ? (ac->outer_active[b] != 0) : 1;
        if (a->admm_active) a->admm_active[b] = act;
        if (!act) continue;
        if (a->iters) a->iters[b] = 0;
        if (a->lx) for (int e = 0; e < N * n; ++e) ((REAL *)a->lx)[(int64_t)b * N * n + e] = 0;
        if (a->lu) for (int e = 0; e < N * m; ++e) ((REAL *)a->lu)[(int64_t)b * N * m + e] = 0;
        if (a->res_prev) { ((REAL *)a->res_prev)[2 * b] = (REAL)1e6; ((REAL *)a->res_prev)[2 * b + 1] = (REAL)1e6; }
    }
    if (a->lin.A) {
        isls_linearize_args l = a->lin;
        l.active = ac->outer_active;
        if ((rc = FN(oracle_linearize)(&l)) != ISLS_OK) return rc;
    }
    if (a->exp.c0x) {
        isls_expand_args e = a->exp;
        e.active = ac->outer_active;
        if ((rc = FN(oracle_expand_quadratic)(&e)) != ISLS_OK) return rc;
    }
    return ISLS_OK;
}

#undef VIEW
#undef MAXN
#undef MAXM
