// rollout.hip -- forward line-search rollout + cost + arg-min + winner trajectory on gfx950.
//
// Reference semantics: iSLS.rollout_DP (isls/isls.py:310-334), iterate_once_dp candidates / NaN rule /
// arg-min / acceptance (isls.py:357-369), SLSBase.compute_cost (isls/sls_base.py:25-44), AL terms of
// the ilqr_admm line search (isls.py:471-477), SLSBase.get_trajectory_dp (sls_base.py:76-89).
//
// Mapping: one 64-lane wavefront per workgroup, cut into TPW = 64/GL slots (GL = max(L,8) lanes);
// slot s owns trajectory blockIdx.x*TPW+s and lane c of the slot owns line-search candidate c (its
// state x lives in registers for the whole horizon).  The per-step operands shared by the candidates
// of a trajectory (K_t, k_t, xhat_t, uhat_t, the ADMM targets z-lambda and AL weights) are fetched by
// the slot's lanes one element each, ONE STEP AHEAD, into a double-buffered LDS record and read back
// as broadcasts.  After the arg-min the slot replays the winning step size and streams x_t,u_t out.
#include <type_traits>

#include "isls_common.hpp"

namespace isls {

template <typename T>
struct RoP {
    int B, N, L, flags;
    const T *par;
    int64_t par_sb;
    const T *K, *k, *xhat, *uhat, *x0, *alphas;
    const T *Qtab, *ztab;
    int64_t Qtab_sb, ztab_sb;
    const int32_t *seq, *qnz;
    T u_std;
    View<T> wq, wr;
    const T *zx, *lx, *zu, *lu, *cost_cur;
    T *cost_all, *cost_new, *x_out, *u_out;
    int32_t *best, *status;
    const int32_t *active;
};

// ---- built-in forward models -------------------------------------------------
template <typename T, int NX, int NU, int MODEL>
struct Model;

template <typename T, int NX, int NU>
struct Model<T, NX, NU, ISLS_MODEL_LTI> {      // x+ = A x + B u   (isls/sls_base.py:49-53)
    T A[NX][NX], Bm[NX][NU];
    __device__ __forceinline__ void load(const T *par)
    {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j < NX; ++j) A[i][j] = par[i * NX + j];
#pragma unroll
            for (int j = 0; j < NU; ++j) Bm[i][j] = par[NX * NX + i * NU + j];
        }
    }
    __device__ __forceinline__ void step(const T (&x)[NX], const T (&u)[NU], T (&xn)[NX]) const
    {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            T s = T(0), r = T(0);
#pragma unroll
            for (int j = 0; j < NX; ++j) s += A[i][j] * x[j];
#pragma unroll
            for (int j = 0; j < NU; ++j) r += Bm[i][j] * u[j];
            xn[i] = s + r;
        }
    }
};

template <typename T>
struct Model<T, 9, 3, ISLS_MODEL_ARM3R> {      // planar 3R arm, state [q, qd, ee]  (3DoF notebooks cell 9)
    T dt;
    __device__ __forceinline__ void load(const T *par) { dt = par[0]; }
    __device__ __forceinline__ void step(const T (&x)[9], const T (&u)[3], T (&xn)[9]) const
    {
        T c = T(0), ex = T(0), ey = T(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            xn[j] = x[j] + x[3 + j] * dt + T(0.5) * u[j] * (dt * dt);
            xn[3 + j] = x[3 + j] + u[j] * dt;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            c += xn[j];
            ex += cos(c);
            ey += sin(c);
        }
        xn[6] = ex; xn[7] = ey; xn[8] = T(0);
    }
};

template <typename T>
struct Model<T, 4, 2, ISLS_MODEL_CAR> {        // car-simple [x, y, theta, v]  (Car notebooks cell 6)
    T dt;
    __device__ __forceinline__ void load(const T *par) { dt = par[0]; }
    __device__ __forceinline__ void step(const T (&x)[4], const T (&u)[2], T (&xn)[4]) const
    {
        xn[0] = x[0] + dt * x[3] * cos(x[2]);
        xn[1] = x[1] + dt * x[3] * sin(x[2]);
        xn[2] = py_mod(x[2] + dt * x[3] * u[0], T(2 * 3.14159265358979323846));
        xn[3] = x[3] + dt * u[1];
    }
};

constexpr int kRolloutDepth = 6;   // steps of record elements in flight per lane

template <typename T, int NX, int NU, int MODEL>
__global__ __launch_bounds__(64) void rollout_kernel(RoP<T> p)
{
    constexpr int D = kRolloutDepth;
    // record layout (doubles): K | xh | rx | wq | k | uh | ru | wr
    constexpr int O_K = 0, O_XH = O_K + NU * NX, O_RX = O_XH + NX, O_WQ = O_RX + NX, O_KK = O_WQ + NX,
                  O_UH = O_KK + NU, O_RU = O_UH + NU, O_WR = O_RU + NU, REC = O_WR + NU;
    constexpr int MINGL = 8, MAXJ = (REC + MINGL - 1) / MINGL, MAXTPW = kWave / MINGL;
    constexpr int OUTW = NX + NU;
    constexpr int SLOT = 2 * REC + 2 * OUTW + 2 * kWave;      // 2 records, 2 out buffers, aug[] + plain[] costs
    __shared__ T lds[MAXTPW * SLOT];

    const int L = p.L, N = p.N;
    const int GL = L > MINGL ? L : MINGL, TPW = kWave / GL;
    const int lane = threadIdx.x;
    const int s = lane / GL, c = lane - s * GL;
    const int b = blockIdx.x * TPW + s;
    const bool inslot = s < TPW;
    const bool valid = inslot && b < p.B && (p.active == nullptr || p.active[b] != 0);
    const bool cand = valid && c < L;
    const int bb = valid ? b : 0;
    const int64_t bN = (int64_t)bb * N;
    T *slot = lds + (inslot ? s : TPW - 1) * SLOT;
    T *recs = slot, *outs = slot + 2 * REC, *c_aug = outs + 2 * OUTW, *c_pln = c_aug + kWave;
    const bool absolute = (p.flags & ISLS_RO_ABSOLUTE) != 0;
    const bool has_xh = !absolute && p.xhat != nullptr, has_uh = !absolute && p.uhat != nullptr;
    const bool has_wq = p.wq.p != nullptr, has_wr = p.wr.p != nullptr;

    // ---- per-lane load plan for the record elements e = c + GL*j ------------------------------------
    const T *pa[MAXJ], *pb[MAXJ];
    int stp[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int e = c + GL * j;
        pa[j] = nullptr; pb[j] = nullptr; stp[j] = 0;
        if (!valid || e >= REC) continue;
        if (e < O_XH) { pa[j] = p.K + bN * NU * NX + e; stp[j] = NU * NX; }
        else if (e < O_RX) { if (has_xh) { pa[j] = p.xhat + bN * NX + (e - O_XH); stp[j] = NX; } }
        else if (e < O_WQ) { if (has_wq) { pa[j] = p.zx + bN * NX + (e - O_RX); pb[j] = p.lx + bN * NX + (e - O_RX); stp[j] = NX; } }
        else if (e < O_KK) { if (has_wq) { pa[j] = p.wq.at(bb, 0) + (e - O_WQ); stp[j] = (int)p.wq.st; } }
        else if (e < O_UH) { pa[j] = p.k + bN * NU + (e - O_KK); stp[j] = NU; }
        else if (e < O_RU) { if (has_uh) { pa[j] = p.uhat + bN * NU + (e - O_UH); stp[j] = NU; } }
        else if (e < O_WR) { if (has_wr) { pa[j] = p.zu + bN * NU + (e - O_RU); pb[j] = p.lu + bN * NU + (e - O_RU); stp[j] = NU; } }
        else { if (has_wr) { pa[j] = p.wr.at(bb, 0) + (e - O_WR); stp[j] = (int)p.wr.st; } }
    }
    // ring of D steps of record elements in flight per lane (HBM latency >> one step of math)
    // Loads are raw and unconditional (absent elements point at K[b,0,0,0], a valid word, with stride 0) so
    // that no branch and no arithmetic sits behind a load; z - lambda and the masking happen in put().
    struct Stage {
        T a[MAXJ], b[MAXJ];
    };
    bool has_a[MAXJ], has_b[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        has_a[j] = pa[j] != nullptr;
        has_b[j] = pb[j] != nullptr;
        if (!has_a[j]) { pa[j] = p.K + bN * NU * NX; stp[j] = 0; }
        if (!has_b[j]) pb[j] = pa[j];
    }
    auto fetch = [&](int t, Stage &g) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int64_t o = (int64_t)t * stp[j];
            g.a[j] = pa[j][o];
            if (has_b[j]) g.b[j] = pb[j][o];                    // only the z/lambda pairs carry a second word
        }
    };
    auto put = [&](T *rec, const Stage &g) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int e = c + GL * j;
            const T v = has_a[j] ? (has_b[j] ? g.a[j] - g.b[j] : g.a[j]) : T(0);      // z - lambda
            if (valid && e < REC) rec[e] = v;
        }
    };

    Model<T, NX, NU, MODEL> model;
    model.load(p.par + (int64_t)bb * p.par_sb);
    const T *Qtab = p.Qtab + (int64_t)bb * p.Qtab_sb, *ztab = p.ztab + (int64_t)bb * p.ztab_sb;
    const T ustd = p.u_std;
    T x_init[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) x_init[j] = p.x0 ? p.x0[(int64_t)bb * NX + j] : (p.xhat ? p.xhat[bN * NX + j] : T(0));

    // step t of the winner, staged in LDS by lane 0 of the slot, leaves as one contiguous store per array
    auto stream_out = [&](int t) {
#pragma unroll
        for (int j = 0; j < (OUTW + MINGL - 1) / MINGL; ++j) {
            const int e = c + GL * j;
            if (valid && e < OUTW) {
                const T v = outs[(t & 1) * OUTW + e];
                if (e < NX) p.x_out[(bN + t) * NX + e] = v;
                else p.u_out[(bN + t) * NU + (e - NX)] = v;
            }
        }
    };

    // One pass over the horizon.  SEARCH: every candidate lane accumulates its costs.  Otherwise the
    // slot replays alpha_w and streams the trajectory (or the kept nominal when `keep`) to x_out/u_out.
    auto roll = [&](auto search_tag, T alpha, bool keep, T &cst, T &cu, T &ag) {
        constexpr bool SEARCH = decltype(search_tag)::value;
        T x[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) x[j] = x_init[j];
        cst = T(0); cu = T(0); ag = T(0);
        Stage ring[D] = {};
#pragma unroll
        for (int d = 0; d < D; ++d)
            fetch(d < N ? d : N - 1, ring[d]);                 // unconditional (clamped): exact vmcnt bookkeeping
        auto step = [&](int t, Stage &g) {
            T *rec = recs + (t & 1) * REC;
            put(rec, g);
            slot_sync();                                      // record(t) (and out(t-1)) visible
            fetch(t + D < N ? t + D : N - 1, g);               // refill this ring entry (clamped, unconditional)
            if (!SEARCH && t > 0) stream_out(t - 1);
            // u = (x - xhat) K' + alpha k + uhat            (isls.py:328-329)
            T u[NU];
#pragma unroll
            for (int r = 0; r < NU; ++r) {
                T acc = T(0);
#pragma unroll
                for (int j = 0; j < NX; ++j) acc += (x[j] - rec[O_XH + j]) * rec[O_K + r * NX + j];
                u[r] = (acc + alpha * rec[O_KK + r]) + rec[O_UH + r];
            }
            if (SEARCH) {
                if (p.qnz == nullptr || p.qnz[t] != 0) {       // (x-z)'Q(x-z), skipped where Q_t == 0
                    const T *Q = Qtab + (int64_t)p.seq[t] * NX * NX, *z = ztab + (int64_t)p.seq[t] * NX;
                    T d[NX];
#pragma unroll
                    for (int j = 0; j < NX; ++j) d[j] = x[j] - z[j];
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        T acc = T(0);
#pragma unroll
                        for (int j = 0; j < NX; ++j) acc += Q[i * NX + j] * d[j];
                        cst += d[i] * acc;
                    }
                }
#pragma unroll
                for (int r = 0; r < NU; ++r) cu += u[r] * (ustd * u[r]);
                if (has_wq) {
#pragma unroll
                    for (int j = 0; j < NX; ++j) { const T d = x[j] - rec[O_RX + j]; ag += (d * d) * rec[O_WQ + j]; }
                }
                if (has_wr) {
#pragma unroll
                    for (int r = 0; r < NU; ++r) { const T d = u[r] - rec[O_RU + r]; ag += (d * d) * rec[O_WR + r]; }
                }
            } else if (c == 0 && valid) {
                T *o = outs + (t & 1) * OUTW;
#pragma unroll
                for (int j = 0; j < NX; ++j) o[j] = keep ? rec[O_XH + j] : x[j];
#pragma unroll
                for (int r = 0; r < NU; ++r) o[NX + r] = keep ? rec[O_UH + r] : u[r];
            }
            T xn[NX];
            model.step(x, u, xn);                              // x = f(x, u)   (isls.py:332)
#pragma unroll
            for (int j = 0; j < NX; ++j) x[j] = xn[j];
        };
        int tb = 0;
        for (; tb + D <= N; tb += D) {                         // full groups: branch-free, exact vmcnt bookkeeping
#pragma unroll
            for (int d = 0; d < D; ++d) step(tb + d, ring[d]);
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (tb + d < N) step(tb + d, ring[d]);
        slot_sync();   
        if (!SEARCH) stream_out(N - 1);
    };

    // ---- search pass --------------------------------------------------------------------------------
    const T my_alpha = absolute ? T(1) : ((c < L) ? p.alphas[c] : T(0));
    T cst, cu, ag;
    roll(std::true_type{}, my_alpha, false, cst, cu, ag);
    const T plain = cst + cu;                                  // sum over x, then += sum over u (sls_base.py:33-39)
    const T aug = plain + ag;
    if (cand) { c_aug[c] = aug; c_pln[c] = plain; }
    slot_sync();   
    // first arg-min with numpy's NaN semantics; optional costs[isnan] = 1e5 (isls.py:362)
    const bool nan_rule = (p.flags & ISLS_RO_NAN_TO_1E5) != 0;
    int ind = 0;
    bool nan_seen = false;
    T bestv = T(0), bestp = T(0);
    for (int l = 0; l < L; ++l) {
        T v = c_aug[l], pl = c_pln[l];
        if (v != v) {
            nan_seen = true;
            if (nan_rule) { v = T(1e5); pl = T(1e5); }
        }
        if (l == 0) { bestv = v; bestp = pl; }
        else if (!(bestv != bestv) && (v != v || v < bestv)) { bestv = v; bestp = pl; ind = l; }
        if (cand && l == c && p.cost_all) p.cost_all[(int64_t)b * L + c] = v;
    }
    bool accept = true;
    if (p.flags & ISLS_RO_ACCEPT_TEST) accept = (bestp - p.cost_cur[bb]) < T(0);   // isls.py:365-367
    if (valid && c == 0) {
        if (p.best) p.best[b] = ind;
        if (p.cost_new) p.cost_new[b] = accept ? bestp : p.cost_cur[bb];
        if (p.status) {
            const int bits = (nan_seen ? ISLS_ST_NAN_COST : 0) | (accept ? 0 : ISLS_ST_LS_REJECT);
            if (bits) atomicOr(&p.status[b], bits);
        }
    }
    // ---- winner pass ----------------------------------------------------------------------------------
    const T alpha_w = absolute ? T(1) : p.alphas[ind];
    slot_sync();   
    roll(std::false_type{}, alpha_w, !accept, cst, cu, ag);
}

template <typename T>
int launch_rollout(const isls_rollout_args &a, hipStream_t s)
{
    if (a.B < 0 || a.N < 1 || a.L < 1 || a.L > 64) return ISLS_ERR_ARG;
    if (!a.model_par || !a.K || !a.k || !a.alphas || !a.Qtab || !a.ztab || !a.seq || !a.x_out || !a.u_out) return ISLS_ERR_ARG;
    if (!(a.flags & ISLS_RO_ABSOLUTE) && (!a.xhat || !a.uhat)) return ISLS_ERR_ARG;
    if ((a.flags & ISLS_RO_ABSOLUTE) && !a.x0) return ISLS_ERR_ARG;
    if ((a.flags & ISLS_RO_ACCEPT_TEST) && !a.cost_cur) return ISLS_ERR_ARG;
    if (a.wq.p && (!a.zx || !a.lx)) return ISLS_ERR_ARG;
    if (a.wr.p && (!a.zu || !a.lu)) return ISLS_ERR_ARG;
    if (a.wq.sb != 0 || a.wr.sb != 0) {
        /* per-trajectory AL weights are supported through the view's batch stride */
    }
    if (a.B == 0) return ISLS_OK;
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
    const int GL = a.L > 8 ? a.L : 8, TPW = kWave / GL;
    const int grid = (a.B + TPW - 1) / TPW;
#define LAUNCH(NX_, NU_, MODEL_) \
    hipLaunchKernelGGL((rollout_kernel<T, NX_, NU_, MODEL_>), dim3(grid), dim3(64), 0, s, p)
    if (a.model == ISLS_MODEL_LTI) {
        if (a.n == 6 && a.m == 3) LAUNCH(6, 3, ISLS_MODEL_LTI);
        else if (a.n == 2 && a.m == 1) LAUNCH(2, 1, ISLS_MODEL_LTI);
        else if (a.n == 4 && a.m == 2) LAUNCH(4, 2, ISLS_MODEL_LTI);
        else if (a.n == 9 && a.m == 3) LAUNCH(9, 3, ISLS_MODEL_LTI);
        else return ISLS_ERR_UNSUPPORTED;
    } else if (a.model == ISLS_MODEL_ARM3R) {
        if (a.n == 9 && a.m == 3) LAUNCH(9, 3, ISLS_MODEL_ARM3R);
        else return ISLS_ERR_UNSUPPORTED;
    } else if (a.model == ISLS_MODEL_CAR) {
        if (a.n == 4 && a.m == 2) LAUNCH(4, 2, ISLS_MODEL_CAR);
        else return ISLS_ERR_UNSUPPORTED;
    } else {
        return ISLS_ERR_UNSUPPORTED;
    }
#undef LAUNCH
    return check_launch();
}
template int launch_rollout<double>(const isls_rollout_args &, hipStream_t);
template int launch_rollout<float>(const isls_rollout_args &, hipStream_t);

}  // namespace isls
