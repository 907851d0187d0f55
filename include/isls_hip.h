/*
 * isls_hip.h -- C ABI of libisls_hip.so: the MI355X (gfx950) batched DP-form iLQR-ADMM hot path.
 *
 * The reference (chenjianxing1/iLQR-ADMM, package `isls`) is pure Python; it has NO FFI/plugin
 * boundary.  Its hot path is the set of Python functions cited next to every entry point below;
 * each entry point is the batched (B independent trajectories) device replacement of one of them.
 * `INTEGRATION.md` shows the ctypes stub a maintainer of the reference would add to bind them.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (torch tensors in our host code);
 *     the library allocates no device memory, keeps no global state, and is re-entrant per stream
 *     (the one process-wide input: experiment switches read from the environment once -- ISLS_*_TPW trajectories per
 *     wavefront, ISLS_GAIN_FF=0 / ISLS_FF_V2=0 / ISLS_COL_ROWS=0 select the previous form of a kernel; results do not
 *     depend on them beyond rounding);
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: ISLS_OK or a negative ISLS_ERR_* (argument / unsupported-size / launch error);
 *     numerical trouble is reported per trajectory in the int32 `status[B]` bit mask instead;
 *   - dense arrays are C-contiguous row-major exactly like the reference's numpy arrays with a
 *     leading batch axis:  K[B,N,m,n], k[B,N,m], x[B,N,n], u[B,N,m] ...
 *   - `isls_view` = strided per-timestep operand: object (b,t) starts at p + b*sb + t*st (strides in
 *     ELEMENTS).  sb==0 shares it over the batch, st==0 over the horizon (LTI fast path);
 *   - `_f64` entry points read/write double, `_f32` float; struct scalars are always double;
 *   - `active` (nullable) : trajectories with active[b]==0 are skipped entirely (frozen: converged
 *     or failed, SURVEY section 5 "failure detection").
 *
 * (n,m) with fast (templated) kernels: {2,1} {4,2} {6,3} {9,3} (the reference notebooks' systems) and {3,1} {6,2} {2,2} {3,3} (with
 * them every get_double_integrator_AB(nb_dim <= 3, nb_deriv <= 3) system): isls_dims_supported(n, m).  Every other pair with
 * n <= 16, m <= 8 runs the generic kernels (isls_dims_generic); beyond that ISLS_ERR_UNSUPPORTED.  1 <= L <= 64.
 */
#ifndef ISLS_HIP_H
#define ISLS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISLS_VERSION 107   /* 107: isls_gain_args.lin_on; 106: isls_ff_args.lin_on (model-structured feed-forward pass); 105: isls_outer_advance_*, isls_outer_args.begin_done; 104: isls_riccati_gain_ff_*; 103: project_rows: Dykstra / project_soc algorithms, shell + multilinear sets, row masks; 102: timing context, reduce table */

#define ISLS_OK 0
#define ISLS_ERR_ARG (-1)
#define ISLS_ERR_UNSUPPORTED (-2)
#define ISLS_ERR_LAUNCH (-3)

/* status[b] bits */
#define ISLS_ST_NOT_PD 1     /* Quu not positive definite (reference: LinAlgError from dposv, isls/isls.py:296) */
#define ISLS_ST_NAN_COST 2   /* a line-search candidate produced a NaN cost (isls/isls.py:362) */
#define ISLS_ST_LS_REJECT 4  /* no candidate improved the cost (fp_success False, isls/isls.py:365-372) */

/* Quu solve: iLQR path uses Cholesky (isls/isls.py:296), SLS.solve_dp an explicit inverse (isls/sls.py:149-151) */
#define ISLS_SOLVE_CHOL 0
#define ISLS_SOLVE_INV 1

/* built-in forward models f(x,u) (reference: user callbacks in the notebooks) */
#define ISLS_MODEL_LTI 0    /* x+ = A x + B u ; par = [A(n*n), B(n*m)]            (isls/sls_base.py:49-53)       */
#define ISLS_MODEL_ARM3R 1  /* planar 3R arm, n=9 m=3 ; par = [dt]                 (3DoF notebooks cells 9-10)    */
#define ISLS_MODEL_CAR 2    /* car-simple, n=4 m=2 ; par = [dt]                    (Car notebooks cell 6)         */
#define ISLS_MODEL_DI 3     /* double integrator, n=2d m=d: the LTI model of get_double_integrator_AB(d, 2, dt)
                               (isls/utils.py:266-276) evaluated through its Kronecker structure
                               A=[[I,aI],[0,I]], B=[[b0 I],[b1 I]]: p+ = p + a v + b0 u, v+ = v + b1 u;
                               par = [a, b0, b1] = [A[0,d], B[0,0], B[d,0]]. Same values as ISLS_MODEL_LTI on
                               those matrices (the skipped terms are exact zeros), a sixth of the multiplies */

#define ISLS_MODEL_TASSA 4  /* Tassa car-parking model, n=4 m=2 ; par = [dt, d]      (notebooks/Tutorial.ipynb cell 8):
                               f = dt v, b = f cos w + d - sqrt(d^2 - (f sin w)^2), x+ = x + b cos th, y+ = y + b sin th,
                               th+ = th + asin(f sin w / d), v+ = v + a dt ; state [x,y,th,v], control [w,a]              */

/* cost models of the line search and of the cost expansion */
#define ISLS_COST_VIA 0     /* via-point quadratic (isls/sls_base.py:25-44): Qtab, ztab, seq, u_std                       */
#define ISLS_COST_PHUBER 1  /* control-quadratic + pseudo-Huber state cost (notebooks/Tutorial.ipynb cell 14):
                               sum_t [ sum_i cu_i u_ti^2 + sum_i cx_i ph(x_ti, px_i) ] + sum_i cf_i ph(x_{N-1,i}, pf_i),
                               ph(x,p) = sqrt(x^2 + p^2) - p ; cost_par = [cu(m), cx(n), px(n), cf(n), pf(n)]
                               (built into the kernels of ISLS_MODEL_TASSA; Qtab/ztab/seq/u_std are then ignored)         */

/* rollout flags */
#define ISLS_RO_NAN_TO_1E5 1   /* costs[isnan] = 1e5            (iterate_once_dp only, isls/isls.py:362)          */
#define ISLS_RO_ACCEPT_TEST 2  /* accept iff cost_best < cost_cur (isls/isls.py:365-369); else nominal is kept    */
#define ISLS_RO_ABSOLUTE 4     /* u = K x + k (xhat=uhat=0, alpha forced to 1): SLSBase.get_trajectory_dp         */

/* projections (z-step of ADMM) */
#define ISLS_PROJ_NONE 0
#define ISLS_PROJ_BOX 1        /* np.clip(x, lo, hi)   isls/projections.py:7-11 ; bounds per (t, dim), +-inf = free */
#define ISLS_PROJ_SETS 2       /* project_set_convex over the time steps, isls/projections.py:289-374 (isls_admm_args) */

typedef struct isls_view {
    const void *p;
    int64_t sb, st;
} isls_view;

/* ---------------------------------------------------------------------------------------------
 * Riccati backward pass, gain part.
 * Replaces iSLS.backward_pass_DP (isls/isls.py:229-308; K and the factors that do not depend on the
 * linear cost terms) and SLS.solve_dp(return_Qs=True) (isls/sls.py:85-166).
 *   V_{N-1} = Cxx[N-1];  for t = N-2..0:
 *     Qxx = Cxx + A'VA, Qux = Cux + B'VA, Quu = Cuu + B'VB
 *     K_t = -Quu^{-1} Qux   (Cholesky U'U, or explicit inverse in ISLS_SOLVE_INV mode)
 *     V   = Qxx + K'Quu K + Qux'K + K'Qux                         (no symmetrisation, isls.py:300-301)
 * Outputs (dense): K[B,N,m,n] (K[N-1]=0), Quu[B,N,m,m], Qux[B,N,m,n] and
 *   fac[B,N,m,m]: CHOL mode: upper factor U with the DIAGONAL STORED AS 1/U_ii (strict lower = 0);
 *                 INV  mode: Quu^{-1} (the reference's Quu_inv_log).
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_gain_args {
    int32_t B, N, n, m;
    int32_t solve_mode;
    int32_t _pad;
    isls_view A;    /* [.,.,n,n] */
    isls_view Bm;   /* [.,.,n,m] */
    isls_view Cxx;  /* [.,.,n,n] */
    isls_view Cuu;  /* [.,.,m,m] */
    isls_view Cux;  /* [.,.,m,n] ; p==NULL -> 0 */
    void *K, *Quu, *fac, *Qux; /* Quu, fac, Qux may all be NULL when `rec` is given (their consumers then read the records) */
    int32_t *status;       /* [B] OR-ed in */
    const int32_t *active; /* nullable */
    void *rec;             /* nullable: packed step records [A+B K | B | K | fac] (row-major blocks, RW = n*n+2*n*m+m*m words)
                            * for the feed-forward pass (isls_ff_args.rec).  Opaque to the caller: an allocation of
                            * isls_ff_record_elems(B,N,n,m) elements, laid out [ceil(B/T)][N][T][RW] with T = 64/(n+m)
                            * trajectories per wavefront, so that a wavefront streams one contiguous burst per step;
                            * steps t = N-1 are not written.  Scratch semantics: the places of trajectories that are
                            * inactive (or past the batch in the last wavefront) are overwritten too, with a copy of
                            * another trajectory's record -- a consumer must use the same `active` mask as this pass   */
    int32_t lin_on;        /* != 0 (rec given, Quu / fac / Qux NULL; else ISLS_ERR_UNSUPPORTED): A and Bm are what isls_linearize_*
                            * wrote for `lin_model`.  The records are then written LEAN -- [K | fac | model words] only, at their own
                            * dense stride in the same buffer -- for feed-forward passes with the same hint (isls_ff_args.lin_on); a
                            * reader without it would misread them (the entry points that see both blocks check this).
                            * ISLS_MODEL_DI: the pass also neither loads nor stages A, Bm and evaluates [A B]'V [A B] and A + B K
                            * from the two non-zero entries of every column -- the same sums in the same order: fp64 results
                            * bit-identical to the dense pass, fp32 equal to rounding.  ISLS_MODEL_ARM3R (n=9, m=3) and ISLS_MODEL_CAR (n=4, m=2): dense
                            * arithmetic, lean records with the model words behind fac.  Other models: ISLS_ERR_UNSUPPORTED */
    int32_t lin_model;
    const void *lin_par;   /* isls_linearize_args.model_par of that model */
    int64_t lin_par_sb;    /* its batch stride in words (0: shared) */
} isls_gain_args;

/* elements of the packed-record buffer (see isls_gain_args.rec) */
int64_t isls_ff_record_elems(int32_t B, int32_t N, int32_t n, int32_t m);

int isls_riccati_gain_f64(const isls_gain_args *a, void *stream);
int isls_riccati_gain_f32(const isls_gain_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Riccati backward pass, feed-forward part (once per ADMM iteration).
 * Replaces the v/k recursion of iSLS.backward_pass_DP (isls/isls.py:285-302) and SLS.solve_dp_ff
 * (isls/sls.py:168-202) with the cached K, Quu, fac, Qux of the gain pass:
 *   cx_t = c0x_t + 2 Qr_t (xhat_t - (zx_t - lx_t)),  cu_t = c0u_t + 2 Rr_t (uhat_t - (zu_t - lu_t))
 *   v_{N-1} = cx_{N-1};  for t = N-2..0:
 *     qx = cx + A'v, qu = cu + B'v, k_t = -Quu^{-1} qu, v = qx + K'qu + K'Quu k + Qux'k
 * (the ADMM regulariser of isls/sls.py:132-137 and of O2).  xhat/uhat NULL -> 0 (absolute
 * coordinates, SLS path).  Qr.p / Rr.p NULL -> that term (and zx,lx / zu,lu) is unused.
 * Output k[B,N,m] (k[N-1]=0).
 *
 * Time-parallel form (optional, `seg`).  The recursion is affine in v: v_t = Phi_t v_{t+1} + g_t and
 * k_t = Gamma_t v_{t+1} + h_t, where Phi_t, Gamma_t depend only on A, B and the gain-pass outputs.  With
 * the N-1 steps cut into nseg segments of seg_len steps, every segment recurses concurrently from
 * v_in = 0 (the last one from the terminal gradient), the true segment inputs follow from nseg-2 small
 * mat-vecs with the segment transfer matrices Psi_s = Phi_{t0} ... Phi_{t1-1}, and k_t += G_t v_in(seg(t))
 * with G_t = Gamma_t Phi_{t+1} ... Phi_{t1-1}.  G and Psi are produced once per gain pass by
 * isls_riccati_ff_prepare_* and reused by every ADMM iteration; the result equals the sequential
 * recursion up to rounding (a different association of the same sums).
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_ffseg {
    int32_t nseg;     /* <= 1 or G == NULL: sequential recursion over the whole horizon            */
    int32_t seg_len;  /* steps per segment; (nseg, seg_len) as returned by isls_ff_segments()       */
    void *G;          /* [B,N,m,n]     k_t correction operators          (prepare -> ff)           */
    void *Psi;        /* [B,nseg,n,n]  segment transfer matrices         (prepare -> ff)           */
    void *v;          /* [B,nseg,n]    scratch of the ff pass (segment-start values of v)          */
} isls_ffseg;

/* Effective segmentation of a horizon of N steps for a requested segment count: returns nseg (>= 1)
 * and stores the segment length; every segment is non-empty. */
int32_t isls_ff_segments(int32_t N, int32_t nseg_requested, int32_t *seg_len);

typedef struct isls_ff_args {
    int32_t B, N, n, m;
    int32_t solve_mode;
    int32_t _pad;                   /* ncol: 0 / 1 = one pass.  C > 1: the C feedback columns of isls_admm (isls/isls.py:578-590) in ONE
                                     * launch on the same records: zx, lx, zu, lu and k hold C blocks [C,B,N,.], seg.v holds [C,B,nseg,n],
                                     * c0x / c0u act on column 0 only (the other columns carry no cost gradient).  Record path with
                                     * time-invariant Qr / Rr only (else ISLS_ERR_UNSUPPORTED: call once per column) */
    isls_view A, Bm;
    isls_view c0x; /* [.,.,n] */
    isls_view c0u; /* [.,.,m] */
    isls_view Qr;  /* [.,.,n,n] nullable */
    isls_view Rr;  /* [.,.,m,m] nullable */
    const void *xhat, *uhat;        /* [B,N,n], [B,N,m] nullable */
    const void *zx, *lx, *zu, *lu;  /* ADMM consensus / scaled dual, dense */
    const void *K, *Quu, *fac, *Qux;
    void *k;
    const int32_t *active;
    isls_ffseg seg;                 /* zero-initialised => sequential */
    const void *rec;                /* nullable: the packed records the gain pass wrote (isls_gain_args.rec).  The pass then
                                     * reads them instead of A, Bm, K, Quu, fac, Qux and evaluates the same recursion as
                                     * v = cx + K'cu + (A + B K)'v, k = -Quu^-1 (cu + B'v): one contiguous 81-word burst per
                                     * step instead of 108 words in six streams (n=6, m=3); results equal up to rounding */
    const void *Qr_term;            /* nullable: the state weight block [n,n] of the LAST step (batch stride Qr.sb), replacing Qr
                                     * there -- a weight that is the same at every step but the terminal one (a terminal state
                                     * constraint: notebooks/3DoF robot/State and control bound constraints.ipynb cell 22) is
                                     * then passed with a zero time stride and keeps the one-hand-off record kernel.  Record
                                     * path with time-invariant Qr / Rr only (else ISLS_ERR_UNSUPPORTED: pass the full [N,n,n]) */
    int32_t lin_on;                 /* != 0 (record path only): A and Bm are what isls_linearize_* wrote for `lin_model`, and the gain pass
                                     * that wrote `rec` had the same hint (isls_gain_args.lin_on): the records are LEAN -- only the tail
                                     * [K | fac | model words] of every record, 28 instead of 82 words per step at n=6, m=3, 42 instead of
                                     * 150 at n=9 -- and the pass evaluates (A + B K)'v = A'v + K'(B'v) from the model's structure:
                                     * ISLS_MODEL_DI (A = [I aI; 0 I], B = [b0 I; b1 I]) or ISLS_MODEL_ARM3R (A = [I dtI 0; 0 I 0; J dtJ 0],
                                     * B = [hI; dtI; hJ], the six words of J = A[6:8,0:3] kept behind fac) or ISLS_MODEL_CAR (A = I + {a02, a12, a03, a13,
                                     * a23}, B = {b20, b31 = dt}: those six behind fac).  Only the one-hand-off form
                                     * reads them: time-varying Qr / Rr or a time-parallel `seg` are ISLS_ERR_UNSUPPORTED, another model
                                     * too.  The same products in another association: results equal up to rounding.  The caller must
                                     * NOT set it for A, Bm of its own (get_AB callbacks) */
    int32_t lin_model;
    const void *lin_par;            /* isls_linearize_args.model_par of that model */
    int64_t lin_par_sb;             /* batch stride of lin_par in words (0: shared) */
} isls_ff_args;

int isls_riccati_ff_f64(const isls_ff_args *a, void *stream);
int isls_riccati_ff_f32(const isls_ff_args *a, void *stream);

/* Gain pass and the first feed-forward pass in one launch: iSLS.backward_pass_DP computes K and k in the same backward
 * sweep (isls/isls.py:285-302); so does this entry, for the linear terms `ff` describes.  The outer driver uses it for the
 * first of its J feed-forward passes.  Conditions (else ISLS_ERR_UNSUPPORTED): g->rec != NULL and ff->rec == g->rec,
 * g->Quu == g->fac == g->Qux == NULL (records only), Qr / Rr absent or time-invariant (st == 0), same B, N, n, m,
 * solve_mode and active as `g`.  ff->seg is ignored (the sweep is sequential); outputs: everything `g` writes, and ff->k. */
int isls_riccati_gain_ff_f64(const isls_gain_args *g, const isls_ff_args *ff, void *stream);
int isls_riccati_gain_ff_f32(const isls_gain_args *g, const isls_ff_args *ff, void *stream);

/* Operators of the time-parallel feed-forward pass (see isls_ffseg): run after the gain pass whenever
 * A, B, K, Quu, fac or Qux changed.  Writes seg.G for t < (nseg-1)*seg_len and seg.Psi for 1 <= s <= nseg-2. */
typedef struct isls_ff_prepare_args {
    int32_t B, N, n, m;
    int32_t solve_mode;
    int32_t _pad;
    isls_view A, Bm;
    const void *K, *Quu, *fac, *Qux;
    const int32_t *active;
    isls_ffseg seg;
    const void *rec;       /* nullable: the gain pass's packed records (isls_gain_args.rec); A .. Qux are then not read */
} isls_ff_prepare_args;

int isls_riccati_ff_prepare_f64(const isls_ff_prepare_args *a, void *stream);
int isls_riccati_ff_prepare_f32(const isls_ff_prepare_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Forward line-search rollout + cost + arg-min + winner trajectory.
 * Replaces iSLS.rollout_DP (isls/isls.py:310-334), the candidate costs / arg-min / acceptance of
 * iSLS.iterate_once_dp (isls.py:357-369), the quadratic via-point cost SLSBase.compute_cost
 * (isls/sls_base.py:25-44, no 1/2) and the augmented-Lagrangian terms of the ilqr_admm line search
 * (isls/isls.py:471-476, `(dx*dx)@Qr` precedence => weights are the ROW SUMS of Qr: wq, wr).
 *   candidate l: x_0 = xhat_0 ; u_t = K_t (x_t - xhat_t) + alpha_l k_t + uhat_t ; x_{t+1} = f(x_t,u_t)
 *   cost_l  = sum_t (x_t-z_t)'Q_t(x_t-z_t) + u_std |u_t|^2           (Q_t=Qtab[seq[t]], z_t=ztab[seq[t]])
 *   aug_l   = cost_l + sum_t wq_t.(x_t-(zx_t-lx_t))^2 + wr_t.(u_t-(zu_t-lu_t))^2
 *   best = first arg-min of aug ; x_out,u_out = trajectory of `best` ; cost_new = cost_best (plain)
 * Outputs: cost_all[B,L] (aug, nullable), best[B], cost_new[B], x_out[B,N,n], u_out[B,N,m].
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_rollout_args {
    int32_t B, N, n, m, L;
    int32_t model;
    int32_t flags;
    int32_t nvia;
    const void *model_par;
    int64_t model_par_sb;           /* 0 = shared */
    const void *K, *k, *xhat, *uhat;
    const void *x0;                 /* [B,n] nullable: default xhat[:,0] */
    const void *alphas;             /* [L] */
    const void *Qtab;               /* [.,nvia,n,n] */
    int64_t Qtab_sb;
    const void *ztab;               /* [.,nvia,n] */
    int64_t ztab_sb;
    const int32_t *seq;             /* [N] */
    const int32_t *q_nonzero;       /* [N] nullable hint: 0 where Q_t == 0 for every trajectory (term skipped) */
    double u_std;
    isls_view wq;                   /* [.,.,n] nullable */
    isls_view wr;                   /* [.,.,m] nullable */
    const void *zx, *lx, *zu, *lu;
    const void *cost_cur;           /* [B], ISLS_RO_ACCEPT_TEST only */
    void *cost_all;
    int32_t *best;
    void *cost_new;
    void *x_out, *u_out;
    int32_t *status;
    const int32_t *active;
    int32_t cost_model;             /* ISLS_COST_* */
    int32_t _pad2;
    const void *cost_par;           /* ISLS_COST_PHUBER: [m + 4n] shared by the batch */
} isls_rollout_args;

int isls_rollout_ls_f64(const isls_rollout_args *a, void *stream);
int isls_rollout_ls_f32(const isls_rollout_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Row-wise projections onto an intersection of sets.
 * Replaces isls/projections.py: project_bound (7-11), project_soc_unit_batch (140-162),
 * project_square_batch (256-266, with the affine keep-out transform of the car notebook) and
 * project_set_convex (289-374): the inner ADMM
 *   x = (I + rho sum A_i'A_i)^-1 (x0 + rho sum A_i'(z_i - b_i - l_i)),  z_i = P_i(A_i x + b_i + l_i),
 *   l_i += A_i x + b_i - z_i
 * run for every row of a problem at once and stopped for all rows of that problem together (max-norm
 * residuals below `threshold`, or both relative changes below 1e-5), like one call of the reference.
 * P problems x R rows (<= 1024) of dimension d (<= ISLS_MAX_ROW_DIM); rows are addressed through
 * (sp, sr) element strides so that a coordinate block of a [B,N,n] state array can be projected in place.
 * nsets == 1 with sets[0].A == NULL applies the primitive directly (no iteration, dim == d).  `algorithm` selects
 * project_set_convex_dykstra or project_soc instead of project_set_convex; `row_mask` restricts a call to some rows.
 * ------------------------------------------------------------------------------------------- */
#define ISLS_MAX_ROW_DIM 4
#define ISLS_MAX_SET_DIM 5
#define ISLS_MAX_SETS 4
#define ISLS_SET_BOX 1       /* par: lo[dim], hi[dim]                                                  */
#define ISLS_SET_SOC_UNIT 2  /* (z, t) = first dim-1 entries, last entry; no parameters                */
#define ISLS_SET_SQUARE 3    /* par: q, l, u, c[q], W[q*q], Winv[q*q]: l <= ||W(y[:q]-c)||_inf <= u    */
#define ISLS_SET_LINEAR 4    /* par: l, u, a[dim]: l <= a'y <= u        (project_linear_batch, projections.py:30-43) */
#define ISLS_SET_QUADRATIC 5 /* par: l, u: l <= y'y/2 <= u              (project_quadratic_batch, projections.py:91-105) */
#define ISLS_SET_SHELL 6     /* par: l, u, c[dim]: l <= |y-c|^2/2 <= u, project_quadratic_batch(y - c, l, u) + c -- the spherical
                                keep-out shells of notebooks/Double integrator/...spherical obstacle avoidance.ipynb cell 12  */
#define ISLS_SET_MULTILINEAR 7 /* par: q, l[q], u[q], M[q*dim]: l <= M y <= u, y - M'(M M')^-1 (M y - clip(M y, l, u))
                                  (project_multilinear, projections.py:46-61); q <= ISLS_MAX_SET_DIM                           */

/* algorithm of isls_project_rows_* over the sets of a call */
#define ISLS_PROJ_ALG_ADMM 0     /* project_set_convex: consensus ADMM over A_i y + b_i in C_i   (projections.py:289-374)       */
#define ISLS_PROJ_ALG_DYKSTRA 1  /* project_set_convex_dykstra: alternating projections with corrections (projections.py:465-504);
                                    every set acts on the row itself (A, b unused, dim == d); threshold = tol on the summed
                                    squared correction change of a pass, at most max_iter + 1 passes                            */
#define ISLS_PROJ_ALG_SOC 2      /* project_soc: A y + b in the second-order cone by ADMM (projections.py:163-234); one set of
                                    kind ISLS_SET_SOC_UNIT with A [dim,d], b [dim]; threshold = tol                             */

typedef struct isls_cset {
    int32_t kind, dim;          /* primitive and dimension of A y + b                               */
    const void *A;              /* [dim, d] row-major; NULL only for the direct form                */
    const void *b;              /* [dim]                                                            */
    const void *par;            /* primitive parameters                                             */
    int64_t A_sp, b_sp, par_sp; /* element strides between problems (0 = shared by all problems)    */
} isls_cset;

typedef struct isls_project_args {
    int32_t P, R, d, nsets;
    int32_t max_iter;
    int32_t algorithm;          /* ISLS_PROJ_ALG_* (0 = project_set_convex)                         */
    double rho, threshold;
    isls_cset sets[ISLS_MAX_SETS];
    const void *y_in;           /* rows y_in[p*in_sp + r*in_sr + 0..d)   */
    int64_t in_sp, in_sr;
    void *y_out;                /* may alias y_in                          */
    int64_t out_sp, out_sr;
    int32_t *iters;             /* [P] nullable: inner iterations run      */
    const int32_t *active;      /* [P] nullable                            */
    const int32_t *row_mask;    /* [R] nullable, shared by the problems: rows with row_mask[r] == 0 pass through unchanged and
                                 * take no part in the stop rule -- a constraint set that touches only some time steps (the
                                 * state-bounds notebook projects its last two rows, each with its own set: one call per set) */
    const struct isls_project_args *next;   /* nullable: a further stage applied in place to this stage's output (same P, R, rows,
                                 * active; its own sets / algorithm / row_mask): project_set_convex followed by Dykstra in the
                                 * obstacle notebook, or one stage per group of rows                                          */
} isls_project_args;

int isls_project_rows_f64(const isls_project_args *a, void *stream);
int isls_project_rows_f32(const isls_project_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * ADMM z/dual update with projection, residuals and the two stop rules.
 * Replaces the body of ADMM() after the x-step (isls/admm.py:43-85) with project_x/project_u given
 * as descriptors (box: isls/projections.py:7-11; ISLS_PROJ_SETS: project_set_convex, 289-374):
 *   z_prev = z ; z = Proj(relax*x + (1-relax)*z + lmb) ; r = x - z ; lmb += r
 *   prim = |r_x| + |r_u| ; dual = |z_x - z_prev_x| + |z_u - z_prev_u|        (unscaled 2-norms)
 *   stop  if prim<tol_abs and dual<tol_abs, else if both relative changes (vs res_prev, +1e-30) < tol_rel
 * res[B,2] receives (prim,dual); res_prev[B,2] is read then overwritten with them (init 1e6, admm.py:25-26);
 * active[b] is cleared when a stop rule fires (nullable => no stop rule evaluated); iters[b] counts the
 * executed ADMM iterations of trajectory b (the length of the reference's `logs`, admm.py:71).
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_admm_args {
    int32_t B, N, n, m;
    int32_t proj_x, proj_u;
    double relax, tol_abs, tol_rel;
    const void *xx, *xu;
    void *zx, *lx, *zu, *lu;   /* a NULL z pointer disables that block (project_x / project_u False) */
    isls_view x_lo, x_hi;      /* [.,.,n] */
    isls_view u_lo, u_hi;      /* [.,.,m] */
    void *res, *res_prev;
    int32_t *active;
    int32_t *iters;            /* [B] nullable: incremented for every trajectory updated by this call */
    /* ISLS_PROJ_SETS: the z block is project_set_convex (isls_project_rows) over the N rows (time steps) of
     * the coordinate block [col0, col0+d) of relax*x + (1-relax)*z + lmb, the other coordinates pass through
     * (the state constraint of notebooks/Car/Iterative LQR with state constraints.ipynb cell 18).  Of the
     * descriptor only d, nsets, sets, rho, max_iter, threshold are used. */
    const isls_project_args *x_sets, *u_sets;
    int32_t x_col0, u_col0;
    void *x_work, *u_work;     /* [B,N,n] / [B,N,m] caller-owned scratch (argument of the projection) */
} isls_admm_args;

int isls_admm_update_f64(const isls_admm_args *a, void *stream);
int isls_admm_update_f32(const isls_admm_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * SLS-ADMM with chance constraints on the controls (config 5 of BASELINE.json).
 * Replaces the iteration of SLS.ADMM_SLS (isls/sls.py:319-454, project_u only) with project_u given as
 * set descriptors: project_set_convex (isls/projections.py:289-374) over the R = N*m rows
 * y = [d_u, phi_u] (dimension D = 1 + p) of one problem per call of the reference:
 *   x_u = Linv (r_side + rr .* (z - lmb)) ;  z = Proj(alpha x_u + (1-alpha) z + lmb) ;  lmb += x_u - z
 *   prim = |rr .* (x_u - z)|_F , dual = |rr .* (z - z_prev)|_F
 *   stop if both < tol, else if both relative changes (+1e-30) < rel_tol = 1e-2      (sls.py:417-430)
 * (once the primal residual has reached rounding level its relative change is noise, so the iteration at which
 * the second rule fires is not reproducible to the last step, in the reference either; rel_tol = 0 disables it)
 * z and lmb start at zero.  Linv [R,R] = (Su'Q Su + R + Rr)^-1 (symmetric) is shared by the P problems
 * (same dynamics and weights), r_side [P,R,D] = [Su'Q xd_p , -Su'Q Sx] and the constraint sets may differ
 * per problem.  rr [R] is the diagonal of Rr.  logs [P,max_iter,2] receives (prim, dual) per executed
 * iteration (rows beyond iters[p] are left untouched).
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_sls_admm_args {
    int32_t P, R, D, max_iter;
    double alpha, tol;
    double rel_tol;           /* relative-change stop rule; the reference hard-codes 1e-2 (sls.py:426)   */
    const void *Linv;
    const void *r_side;
    const void *rr;
    isls_project_args proj;   /* nsets, sets, rho, max_iter, threshold are used (d = D)          */
    void *x_u;                /* [P,R,D] out: du = x_u[:,:,0], phi_u[:, :p] = x_u[:,:,1:]         */
    void *z, *lmb;            /* [P,R,D] out, nullable                                            */
    void *logs;               /* nullable                                                         */
    int32_t *iters;           /* [P] nullable                                                     */
} isls_sls_admm_args;

int isls_sls_admm_f64(const isls_sls_admm_args *a, void *stream);
int isls_sls_admm_f32(const isls_sls_admm_args *a, void *stream);

/* Closed loop of the dense causal SLS controller for M initial states (Monte-Carlo evaluation of the notebooks):
 *   u_i = k_i + sum_{j<=i} K[i, j] x_j ,  x_{i+1} = A x_i + B u_i          (isls/sls_base.py:91-105, noise_scale = 0)
 * A [n,n], B [n,m], K [N m, N n], k [N m], x0 [M,n] -> x_log [M,N,n], u_log [M,N,m]. */
int isls_sls_closed_loop_f64(int32_t M, int32_t N, int32_t n, int32_t m, const void *A, const void *B, const void *K,
                             const void *k, const void *x0, void *x_log, void *u_log, void *stream);
int isls_sls_closed_loop_f32(int32_t M, int32_t N, int32_t n, int32_t m, const void *A, const void *B, const void *K,
                             const void *k, const void *x0, void *x_log, void *u_log, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Feedback columns of iSLS.isls_admm (isls/isls.py:503-712) in DP form.
 * The reference solves, per outer iteration, the dense normal equations
 *   [d_u, phi_u] = (Su'Q Su + R + Su'Qr Su + Rr)^-1 (r_side + Su'Qr x_reg + Rr u_reg)      (isls.py:571,580-588)
 *   [d_x, phi_x] = Su [d_u, phi_u] + [0, Sx]                                                (isls.py:589-590)
 * for the C = 1 + dim columns at once.  Column c is the minimiser of one time-varying LQ problem about the
 * nominal: column 0 with the cost gradients of the nominal and delta x_0 = 0, column j >= 1 with no cost
 * gradient and delta x_0 = e_j (first `dim` states), each with its own ADMM target (z - lmb)[c].  All columns
 * share the gains K_t of ONE Riccati gain pass (isls_riccati_gain_*); their feed-forward terms k[c] come from
 * C feed-forward passes (isls_riccati_ff_* with xhat = uhat = NULL and, for c >= 1, zero c0x / c0u), and this
 * entry point rolls them through the linearised dynamics:
 *   dx_0 = e_c ; du_t = K_t dx_t + k[c]_t ; dx_{t+1} = A_t dx_t + B_t du_t           (t < N-1)
 *   du_{N-1} = -Cuu_{N-1}^-1 ( [c == 0] c0u_{N-1} - 2 Rr_{N-1} (zu - lu)[c]_{N-1} )
 * The last control never acts on the state (SURVEY 8a quirk i): the dense form gives it the minimiser of its
 * own cost term, reproduced by the second line.  Arrays with a leading C are column-major: [C,B,N,.].
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_columns_args {
    int32_t B, N, n, m;
    int32_t C;              /* 1 + dim                                                            */
    int32_t _pad;
    isls_view A, Bm;        /* [.,.,n,n], [.,.,n,m]                                               */
    isls_view Cuu;          /* [.,.,m,m] control Hessian of the gain pass (2 R + 2 Rr)            */
    isls_view c0u;          /* [.,.,m]   control gradient at the nominal (2 R uhat)               */
    isls_view Rr;           /* [.,.,m,m] nullable (project_u False)                               */
    const void *K;          /* [B,N,m,n]                                                          */
    const void *k;          /* [C,B,N,m]                                                          */
    const void *zu, *lu;    /* [C,B,N,m] nullable with Rr                                         */
    void *dx, *du;          /* [C,B,N,n], [C,B,N,m] out                                           */
    const int32_t *active;  /* [B] nullable                                                       */
} isls_columns_args;

int isls_columns_rollout_f64(const isls_columns_args *a, void *stream);
int isls_columns_rollout_f32(const isls_columns_args *a, void *stream);

/* Monte-Carlo closed loop of a dense causal controller about a nominal through a built-in forward model
 * (iSLSBase.get_trajectory_sls, isls/isls_base.py:28-42, noise_scale = 0), M initial states:
 *   dx_j = x_j - xhat_j (j <= i) ; u_i = (K [dx_0 .. dx_i, 0 ..] + k)_i + uhat_i ; x_{i+1} = f(x_i, u_i)
 * K [N m, N n] (lower block triangular is not assumed, entries right of block i are not read), k [N m], the nominal
 * xhat [N,n] / uhat [N,m] of ONE problem (NULL = 0: absolute form), x0 [M,n] -> x_log [M,N,n], u_log [M,N,m]. */
typedef struct isls_dense_loop_args {
    int32_t M, N, n, m;
    int32_t model, _pad;
    const void *model_par;
    const void *K, *k;
    const void *xhat, *uhat;
    const void *x0;
    void *x_log, *u_log;
} isls_dense_loop_args;

int isls_dense_closed_loop_f64(const isls_dense_loop_args *a, void *stream);
int isls_dense_closed_loop_f32(const isls_dense_loop_args *a, void *stream);

/* z / dual step of isls_admm around the row projection (isls/isls.py:626-665), in two phases:
 *   phase 0:  z_prev <- z ;  work[b, t*d+i, c] <- relax x + (1-relax) z + lmb  (+ nom[b,t,i] for c == 0)
 *             ... project the rows of `work` (isls_project_rows_* on [B, N*d, C], or the caller's function) ...
 *   phase 1:  z <- work (- nom for c == 0) ; r = x - z ; lmb += r
 *             prim = |Qr r_x|_F + |Rr r_u|_F , dual = |Qr (z_x - z_prev_x)|_F + |Rr (z_u - z_prev_u)|_F
 *             stop if prim < tol_abs and dual < tol_abs, else if both relative changes (+1e-30) < tol_rel (1e-3)
 * `nom` is the nominal the notebooks add to the d column before projecting and remove afterwards
 * (notebooks/3DoF robot/State bounds and robust control bounds.ipynb cell 25); NULL when the caller's projection
 * does that itself.  A block with x == NULL is absent (project_x / project_u False).  res / res_prev / active /
 * iters as in isls_admm_args. */
typedef struct isls_columns_admm_args {
    int32_t B, N, n, m;
    int32_t C, phase;
    double relax, tol_abs, tol_rel;
    const void *xx, *xu;          /* [C,B,N,n], [C,B,N,m] x-step (nullable per block)             */
    void *zx, *lx, *zu, *lu;      /* [C,B,N,.]                                                    */
    void *zx_prev, *zu_prev;      /* [C,B,N,.] scratch between the phases                         */
    const void *x_nom, *u_nom;    /* [B,N,n], [B,N,m] nullable                                    */
    void *x_work, *u_work;        /* [B,N*n,C], [B,N*m,C] projection argument / result            */
    isls_view Qr, Rr;             /* residual weights, [.,.,n,n] / [.,.,m,m]                      */
    void *res, *res_prev;         /* [B,2]                                                        */
    int32_t *active;              /* [B] nullable                                                 */
    int32_t *iters;               /* [B] nullable                                                 */
} isls_columns_admm_args;

int isls_columns_admm_f64(const isls_columns_admm_args *a, void *stream);
int isls_columns_admm_f32(const isls_columns_admm_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * One ADMM iteration of isls_admm (isls/isls.py:578-665) enqueued as a whole -- the feedback-column counterpart of
 * isls_ilqr_admm_outer_*:
 *   x-step   : the C feed-forward passes (`ff` in the column layout of ff._pad = C: one launch on the packed records where that
 *              form applies, else -- or when ff._pad == 1 asks for it -- one pass per column:
 *              the driver then offsets zx, lx, zu, lu, k per column and hands columns >= 1 the zero vectors zero_x / zero_u as
 *              cost gradients), then isls_columns_rollout (`cols`);
 *   line search on column 0 (`ls`, skipped when ls.L == 0: SLS.ADMM_SLS has none): isls_rollout_ls of u_nom + alpha d_u with
 *              zero gains (ls.K a zero array, ls.k = du[0]), then  du[0] <- alpha* du[0],  dx[0] <- x_out - xhat  for the active
 *              problems (isls.py:593-606);
 *   z-step   : isls_columns_admm phase 0, the row projections (proj_x / proj_u, nullable: block absent), phase 1 (`admm`;
 *              skipped when both blocks are absent: the unconstrained problem has no z-step);
 *   `log`    : nullable [B,2], receives admm.res behind the z-step;  `any_active`: see the field.
 * Nothing is exchanged with the host between the launches.
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_columns_iteration_args {
    isls_ff_args ff;
    isls_columns_args cols;
    isls_rollout_args ls;
    isls_columns_admm_args admm;
    const isls_project_args *proj_x, *proj_u;
    const void *zero_x, *zero_u;    /* [n], [m] zeros (cost gradients of the columns >= 1 in the one-pass-per-column form) */
    void *log;
    int32_t *any_active;            /* nullable: one word, set to 1 when admm.active still holds a problem behind the z-step, else 0 --
                                     * the flag a host loop follows (copied to pinned memory, read when it has landed) instead of
                                     * reducing the mask itself */
} isls_columns_iteration_args;

int isls_columns_iteration_f64(const isls_columns_iteration_args *a, void *stream);
int isls_columns_iteration_f32(const isls_columns_iteration_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Quadratic via-point cost expansion about the nominal (the `Cts is None` branch of
 * backward_pass_DP, isls/isls.py:263-271, written out as arrays, plus the ADMM regulariser):
 *   Cxx[b,t] = 2 Q_t (+ 2 Qr_t) ; Cuu[b,t] = 2 u_std I (+ 2 Rr_t)
 *   c0x[b,t] = 2 Q_t (xhat_t - z_t) ; c0u[b,t] = 2 u_std uhat_t ; cost[b] = compute_cost(xhat,uhat)
 * Cxx/Cuu may be NULL (skip).  xhat/uhat NULL -> 0 (absolute coordinates).
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_expand_args {
    int32_t B, N, n, m;
    int32_t nvia, _pad;
    const void *Qtab; int64_t Qtab_sb;
    const void *ztab; int64_t ztab_sb;
    const int32_t *seq;
    double u_std;
    isls_view Qr, Rr;               /* nullable */
    const void *xhat, *uhat;
    void *Cxx, *Cuu;                /* [B,N,n,n], [B,N,m,m] nullable */
    void *c0x, *c0u;                /* [B,N,n], [B,N,m] */
    void *cost;                     /* [B] nullable */
    const int32_t *active;
    int32_t cost_model;             /* ISLS_COST_VIA: as above; ISLS_COST_PHUBER: gradient / (diagonal) Hessian of the
                                       pseudo-Huber cost about the nominal, the get_Cs callback of Tutorial.ipynb cell 16 */
    int32_t _pad2;
    const void *cost_par;
    const int32_t *q_nonzero;       /* [N] nullable hint as in isls_rollout_args: 0 where Q_t == 0 for every trajectory */
} isls_expand_args;

int isls_expand_quadratic_f64(const isls_expand_args *a, void *stream);
int isls_expand_quadratic_f32(const isls_expand_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Linearisation of the built-in models along the nominal: the user's get_AB(xhat,uhat) callback
 * (isls/isls.py:61-66,95; notebooks).  A[B,N,n,n], B[B,N,n,m] dense outputs.
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_linearize_args {
    int32_t B, N, n, m;
    int32_t model, _pad;
    const void *model_par; int64_t model_par_sb;
    const void *xhat, *uhat;
    void *A, *Bm;
    const int32_t *active;
} isls_linearize_args;

int isls_linearize_f64(const isls_linearize_args *a, void *stream);
int isls_linearize_f32(const isls_linearize_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * End of an outer iteration: nominal_values <- last x-step of the ADMM (isls/isls.py:488; the setter
 * isls/isls_base.py:80-85 appends the new cost to cost_log) followed by the two outer stop rules
 * (isls/isls.py:493-499), per trajectory, for the trajectories with outer_active[b] != 0:
 *   xhat <- xx, uhat <- xu, prev = cost, cost <- cost_new, cost_hist <- push(cost)
 *   stop if |cost - prev| < tol_cost, or |mean(hist[-4:]) - mean(hist[-8:-4])| < tol_osc (needs >= 5 entries)
 * cost_hist[B,8] holds the tail of cost_log (oldest first), hist_len[B] its fill (<= 8).  A negative
 * tolerance disables that rule.  Stopped trajectories get outer_active[b] <- 0.
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_accept_args {
    int32_t B, N, n, m;
    const void *xx, *xu, *cost_new;
    void *xhat, *uhat, *cost;
    void *cost_hist;        /* nullable (then the oscillation rule is skipped) */
    int32_t *hist_len;
    double tol_cost, tol_osc;
    int32_t *outer_active;  /* nullable => every trajectory is updated, no stop rule */
} isls_accept_args;

int isls_accept_step_f64(const isls_accept_args *a, void *stream);
int isls_accept_step_f32(const isls_accept_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Convergence reduction over the local batch shard: out[5] = { sum cost, max prim, max dual,
 * #active, #status!=0 } -- the 40-byte payload of the per-iteration RCCL all-reduce.
 * Batched analogue of the scalar tests at isls/isls.py:125-132,493-499 and isls/admm.py:72-85.
 * ------------------------------------------------------------------------------------------- */
int isls_reduce_convergence_f64(int32_t B, const void *cost, const void *res, const int32_t *active,
                                const int32_t *status, void *out5, void *stream);
int isls_reduce_convergence_f32(int32_t B, const void *cost, const void *res, const int32_t *active,
                                const int32_t *status, void *out5, void *stream);
/* The same reduction written straight into the [W,5] table of the all-reduce: row `rank` receives the five numbers, the
 * other W-1 rows are zeroed (so that one sum-all-reduce of the table gathers every rank's row; isls/shard.py). */
int isls_reduce_convergence_table_f64(int32_t B, const void *cost, const void *res, const int32_t *active,
                                      const int32_t *status, void *table, int32_t rank, int32_t world, void *stream);
int isls_reduce_convergence_table_f32(int32_t B, const void *cost, const void *res, const int32_t *active,
                                      const int32_t *status, void *table, int32_t rank, int32_t world, void *stream);

/* ---------------------------------------------------------------------------------------------
 * One outer DP-form iLQR-ADMM iteration enqueued as a whole (SURVEY 3.3 with the dense solve
 * replaced by the Riccati pass): gain -> J x [ ff -> rollout/line-search -> ADMM update ].
 * The sub-structs are used as given; between inner iterations nothing is exchanged with the host.
 * `log` (nullable) receives res per inner iteration: log[j,B,2].  Before the first inner iteration,
 * for the trajectories with outer_active[b]!=0 (nullable => all): admm.active[b] <- 1, lambda <- 0
 * (isls/isls.py:414-415,482), res_prev <- 1e6 (isls/admm.py:25-26); the others get admm.active[b] <- 0.
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_outer_args {
    isls_gain_args gain;
    isls_ff_args ff;
    isls_rollout_args ro;
    isls_admm_args admm;
    int32_t J;
    int32_t skip_gain;          /* reuse the cached factors (is_dynamics_linear && is_cost_quadratic);
                                 * when the gain pass runs and ff.seg is set, the ff operators are prepared too */
    int32_t begin_done;         /* 1: the start-of-iteration resets described above have been made already (by
                                 * isls_outer_advance_* at the end of the previous iteration) and are skipped here */
    int32_t _pad;
    void *log;
    const int32_t *outer_active;
    void *timing;               /* nullable: isls_timing_create() context that records the kernel-family durations */
} isls_outer_args;

int isls_ilqr_admm_outer_f64(const isls_outer_args *a, void *stream);
int isls_ilqr_admm_outer_f32(const isls_outer_args *a, void *stream);

/* ---------------------------------------------------------------------------------------------
 * End of one outer iteration and start of the next in ONE launch: what the reference does between two x-step solves
 * (isls/isls.py:488-499, then 61-66 / 95-102 / 414-415 of the next pass through the loop), per trajectory:
 *   1. `accept`  : isls_accept_step_* semantics (nominal <- x-step, cost log, the two stop rules -> outer_active);
 *   2. for the trajectories still iterating afterwards: admm_active <- 1, lambda <- 0, res_prev <- 1e6, iters <- 0 (the others:
 *      admm_active <- 0) -- the resets isls_ilqr_admm_outer_* starts with (pass begin_done = 1 there);
 *   3. `lin`     : isls_linearize_* about the new nominal   (lin.A == NULL: skipped; lin.active is ignored);
 *   4. `exp`     : isls_expand_quadratic_* about it         (exp.c0x == NULL: skipped; exp.active is ignored).
 * Same results as the four calls in that order with active = accept.outer_active.  lin / exp normally name accept.xhat /
 * accept.uhat as their nominal.  admm_active, lx, lu, res_prev, iters are nullable.
 * ------------------------------------------------------------------------------------------- */
typedef struct isls_advance_args {
    isls_accept_args accept;
    isls_linearize_args lin;
    isls_expand_args exp;
    int32_t *admm_active;       /* [B] */
    int32_t *iters;             /* [B] */
    void *lx, *lu;              /* [B,N,n], [B,N,m] scaled duals of the ADMM */
    void *res_prev;             /* [B,2] */
} isls_advance_args;

int isls_outer_advance_f64(const isls_advance_args *a, void *stream);
int isls_outer_advance_f32(const isls_advance_args *a, void *stream);

int isls_version(void);
/* 1 when the kernels are instantiated for state dimension n and control dimension m (the pairs are compile-time template
 * arguments: csrc/isls_common.hpp ISLS_FOR_EACH_DIMS), else 0: every entry point returns ISLS_ERR_UNSUPPORTED for other pairs. */
int32_t isls_dims_supported(int32_t n, int32_t m);
/* 1 when (n, m) is served at all: every pair with n <= 16, m <= 8 (the reference takes any dimensions, isls/base.py:11-14).  Pairs
 * without an instantiation run the generic kernels (csrc/generic.hip: dimensions at run time, one trajectory per wavefront,
 * matrices in LDS; array form only -- no packed records, no time-parallel segments, dense LTI / double-integrator model,
 * via-point cost): isls_riccati_gain / isls_riccati_ff / isls_rollout_ls / isls_admm_update / isls_ilqr_admm_outer and the
 * streaming kernels work, the isls_columns_* entry points of isls_admm do not (ISLS_ERR_UNSUPPORTED). */
int32_t isls_dims_generic(int32_t n, int32_t m);
const char *isls_error_string(int code);
/* Per-kernel-family durations for bench.py: HIP events recorded on the launch stream around the launches that
 * isls_ilqr_admm_outer_* enqueues, kept in a CALLER-OWNED context (the library itself has no state): create one, put it
 * into isls_outer_args.timing, read it after the timed region, destroy it.  kind: 0 gain, 1 ff, 2 rollout, 3 admm,
 * 4 ff_prepare.  A context serves one host thread at a time. */
void *isls_timing_create(void);
void isls_timing_destroy(void *timing);
int isls_timing_reset(void *timing);                 /* start a new measurement window                          */
int isls_timing_pause(void *timing, int paused);     /* suspend / resume recording without resetting the window */
double isls_timing_read_ms(void *timing, int kind, int *count);   /* summed ms and launch count of one family; synchronises */

#ifdef __cplusplus
}
#endif
#endif /* ISLS_HIP_H */
