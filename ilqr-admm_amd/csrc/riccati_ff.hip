// riccati_ff.hip -- feed-forward half of the backward Riccati pass (once per ADMM iteration) on gfx950.
//
// Reference semantics: the v/k recursion of iSLS.backward_pass_DP (isls/isls.py:285-302) and SLS.solve_dp_ff
// (isls/sls.py:168-202) with the factors cached by the gain pass; formulas in include/isls_hip.h.
//
// This pass is HBM-bound by its arithmetic (≈1 KB per step and trajectory against ≈130 FMAs), so the kernel is
// organised around the memory stream and around a short instruction path:
//   * one 64-lane wavefront per workgroup, TPW = 64/(n+m) slots of G = n+m lanes, slot = trajectory;
//   * every lane keeps a ring of D steps of its operands in flight in registers.  All loads are
//     `global_load v, v_off32, s[base]` : a per-array SGPR base that moves by one step per iteration and a
//     per-lane 32-bit offset that never changes; no load sits behind a branch or an arithmetic use;
//   * staging into the slot's LDS record is UNCONDITIONAL: every staged element has a precomputed
//     destination word (surplus elements and idle lanes point at a dump word), so there is no exec-mask
//     scaffolding around the ds_writes;
//   * LDS hand-offs inside the wave use slot_sync() (no s_barrier, no vmcnt(0) drain of the ring);
//   * two hand-offs per step: (a) record + v  ->  q ; (b) qu  ->  k, v'.
//
// The N-1 sequential steps are the real limit at the reference batch (585 wavefronts of 100 dependent steps on
// 1024 SIMDs), so the pass also exists in a TIME-PARALLEL form (isls_ffseg in include/isls_hip.h): blockIdx.y
// cuts the horizon into segments that recurse concurrently from v_in = 0, ff_stitch_kernel chains the true
// segment inputs through the transfer matrices and adds G_t v_in to k_t, and ff_prepare_kernel produces those
// operators once per gain pass by pushing the n unit vectors through the homogeneous recursion.
#include "isls_common.hpp"

namespace isls {

#ifndef ISLS_FF_DEPTH
#define ISLS_FF_DEPTH 4
#define ISLS_FF_GROUP 2
#endif
#ifndef ISLS_FF_SEG_DEPTH
#define ISLS_FF_SEG_DEPTH 2
#define ISLS_FF_SEG_GROUP 2
#endif
constexpr int kFfDepth = ISLS_FF_DEPTH;         // steps of operands in flight per lane (sequential form, <= 1 wave per SIMD)
constexpr int kFfGroup = ISLS_FF_GROUP;         // ... fetched in groups of this many consecutive steps
constexpr int kFfSegDepth = ISLS_FF_SEG_DEPTH;  // ... in the time-parallel form, where kFfSegOcc co-resident waves hide the latency
constexpr int kFfSegGroup = ISLS_FF_SEG_GROUP;
#ifndef ISLS_FF_SEG_OCC
#define ISLS_FF_SEG_OCC 2
#endif
constexpr int kFfSegOcc = ISLS_FF_SEG_OCC;

template <typename T>
struct FfP {
    int B, N, mode, tpw;
    int nseg, seg_len;             // time-parallel segments (1 = the whole horizon)
    T *vseg;                       // [B,nseg,NX] v at the first step of every segment (nseg > 1)
    View<T> A, Bm, c0x, c0u, Qr, Rr;
    const T *xhat, *uhat, *zx, *lx, *zu, *lu;
    const T *K, *Quu, *fac, *Qux;
    T *k;
    const int32_t *active;
};

// D = steps of operands in flight per lane, OCC = wavefronts per SIMD the register budget must allow: the
// sequential form runs <= 1 wave per SIMD and hides HBM latency with a deep ring; the segmented form has
// nseg times the waves and trades ring depth for co-residency.
// ROWC: the ADMM weights Qr, Rr do not depend on the time step (stride 0: the usual rho * I): the lane's own weight row is
// then loaded once instead of riding in every ring entry (NX loads and NX registers per step and entry less).
template <typename T, int NX, int NU, int D, int OCC, int FG, bool ROWC>
__global__ __launch_bounds__(64, OCC) void riccati_ff_kernel(FfP<T> p)
{
    constexpr int G = NX + NU, W = NX + NU, MAXTPW = kWave / G;
    // slot record (elements): AB[NX][W] | K[NU][NX] | Qux[NU][NX] | Quu[NU][NU] | fac[NU][NU] | d[W] | v[NX] | qu[NU] | kt[NU] | dump
    constexpr int AB_OFF = 0, K_OFF = AB_OFF + NX * W, QUX_OFF = K_OFF + NU * NX, QUU_OFF = QUX_OFF + NU * NX,
                  FAC_OFF = QUU_OFF + NU * NU, D_OFF = FAC_OFF + NU * NU, V_OFF = D_OFF + W, QU_OFF = V_OFF + NX,
                  KT_OFF = QU_OFF + NU, DUMP_OFF = KT_OFF + NU;
    constexpr int SLOT = ((DUMP_OFF + 1) | 1);
    constexpr int JA = (NX * NX + G - 1) / G, JB = (NX * NU + G - 1) / G, JK = (NU * NX + G - 1) / G,
                  JU = (NU * NU + G - 1) / G;
    __shared__ T lds[(MAXTPW + 1) * SLOT];                     // + one dump slot for the lanes beyond the last slot
    const int TPW = p.tpw;                                     // trajectories per wavefront (<= 64/G), chosen by the launcher

    const int lane = threadIdx.x;
    const int s = lane / G, i = lane - s * G;
    const int b = blockIdx.x * TPW + s;
    const bool inslot = s < TPW;
    const bool inbatch = inslot && b < p.B;
    const bool valid = inbatch && (p.active == nullptr || p.active[b] != 0);
    const int N = p.N;
    // segment of the horizon handled by this workgroup: steps t_hi .. t_lo of the recursion t = N-2 .. 0
    const int seg = blockIdx.y;
    const bool last = seg == p.nseg - 1;
    const int t_lo = seg * p.seg_len, t_hi = last ? N - 2 : t_lo + p.seg_len - 1;
    const int sl = inbatch ? s : 0;                            // idle lanes shadow the block's first trajectory (loads only)
    T *rec = lds + (inslot ? s : TPW) * SLOT;
    const bool xl = i < NX;
    const int iu = xl ? 0 : i - NX;
    const bool hasx = p.Qr.p != nullptr, hasu = p.Rr.p != nullptr;
    const bool hasreg = xl ? hasx : hasu;
    const int b0 = blockIdx.x * TPW;                           // first trajectory of this workgroup (uniform)

    // ---- load plan: uniform bases (block, step) + per-lane 32-bit element offsets --------------------------
    const T *bA = p.A.at(b0, 0), *bB = p.Bm.at(b0, 0);
    const T *bK = p.K + (int64_t)b0 * N * NU * NX, *bQ = p.Qux + (int64_t)b0 * N * NU * NX;
    const T *bU = p.Quu + (int64_t)b0 * N * NU * NU, *bF = p.fac + (int64_t)b0 * N * NU * NU;
    const uint32_t oA = (uint32_t)(sl * p.A.sb), oB = (uint32_t)(sl * p.Bm.sb);
    const uint32_t oK = (uint32_t)sl * N * NU * NX, oU = (uint32_t)sl * N * NU * NU;
    // own scalar entries: gradient c0, (xhat|uhat), z, lambda and own row of Qr|Rr through per-lane pointers
    const T *pc0 = (xl ? p.c0x.at(b0 + sl, 0) + i : p.c0u.at(b0 + sl, 0) + iu);
    const int64_t c0st = xl ? p.c0x.st : p.c0u.st;
    const int dd = xl ? NX : NU;                               // entries per step of the own dense vector
    const int64_t ovec = ((int64_t)(b0 + sl) * N) * dd + (xl ? i : iu);
    const T *ph = xl ? p.xhat : p.uhat, *pz = xl ? p.zx : p.zu, *pl = xl ? p.lx : p.lu;
    const T *prow = hasreg ? (xl ? p.Qr.at(b0 + sl, 0) + i * NX : p.Rr.at(b0 + sl, 0) + iu * NU) : nullptr;
    const int64_t rowst = xl ? p.Qr.st : p.Rr.st;
    const int lim = xl ? NX : NU;

    // ---- staging destinations inside the slot record (dump word for surplus elements) ------------------------
    int dA[JA], dB[JB], dK[JK], dQ[JK], dU[JU], dF[JU];
#pragma unroll
    for (int j = 0; j < JA; ++j) { const int e = i + G * j; dA[j] = e < NX * NX ? AB_OFF + (e / NX) * W + (e % NX) : DUMP_OFF; }
#pragma unroll
    for (int j = 0; j < JB; ++j) { const int e = i + G * j; dB[j] = e < NX * NU ? AB_OFF + (e / NU) * W + NX + (e % NU) : DUMP_OFF; }
#pragma unroll
    for (int j = 0; j < JK; ++j) { const int e = i + G * j; dK[j] = e < NU * NX ? K_OFF + e : DUMP_OFF; dQ[j] = e < NU * NX ? QUX_OFF + e : DUMP_OFF; }
#pragma unroll
    for (int j = 0; j < JU; ++j) { const int e = i + G * j; dU[j] = e < NU * NU ? QUU_OFF + e : DUMP_OFF; dF[j] = e < NU * NU ? FAC_OFF + e : DUMP_OFF; }

    struct Stage {
        T ra[JA], rb[JB], rk[JK], rq[JK], ruu[JU], rf[JU];
        T c0, hv, zv, lv, rrow[ROWC ? 1 : NX];
    };
    auto fetch = [&](int t, Stage &g) {
        const T *a = bA + (int64_t)t * p.A.st, *bm = bB + (int64_t)t * p.Bm.st;
        const T *kk = bK + (int64_t)t * (NU * NX), *qq = bQ + (int64_t)t * (NU * NX);
        const T *uu = bU + (int64_t)t * (NU * NU), *ff = bF + (int64_t)t * (NU * NU);
#pragma unroll
        for (int j = 0; j < JA; ++j) { const int e = i + G * j; g.ra[j] = a[oA + (uint32_t)(e < NX * NX ? e : NX * NX - 1)]; }
#pragma unroll
        for (int j = 0; j < JB; ++j) { const int e = i + G * j; g.rb[j] = bm[oB + (uint32_t)(e < NX * NU ? e : NX * NU - 1)]; }
#pragma unroll
        for (int j = 0; j < JK; ++j) {
            const uint32_t e = (uint32_t)(i + G * j < NU * NX ? i + G * j : NU * NX - 1);
            g.rk[j] = kk[oK + e];
            g.rq[j] = qq[oK + e];
        }
#pragma unroll
        for (int j = 0; j < JU; ++j) {
            const uint32_t e = (uint32_t)(i + G * j < NU * NU ? i + G * j : NU * NU - 1);
            g.ruu[j] = uu[oU + e];
            g.rf[j] = ff[oU + e];
        }
        g.c0 = pc0[(int64_t)t * c0st];
        if (hasreg) {
            const int64_t e = ovec + (int64_t)t * dd;
            g.zv = pz[e];
            g.lv = pl[e];
            g.hv = ph ? ph[e] : T(0);
            if constexpr (!ROWC) {
                const T *q = prow + (int64_t)t * rowst;
#pragma unroll
                for (int j = 0; j < NX; ++j) g.rrow[j] = q[j < lim ? j : lim - 1];
            }
        } else {
            g.hv = g.zv = g.lv = T(0);
            if constexpr (!ROWC) {
#pragma unroll
                for (int j = 0; j < NX; ++j) g.rrow[j] = T(0);
            }
        }
    };
    // cx_i / cu_i = c0 + 2 * (row of Qr/Rr) . d       (isls/sls.py:132-137; O2 of SURVEY 8c)
    const int doff = D_OFF + (xl ? 0 : NX);
    auto reg_grad = [&](T c0v, const T (&row)[NX]) -> T {
        T sacc = T(0);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const T dj = rec[doff + j];                        // j >= lim reads a neighbour word, discarded below
            sacc += (j < lim) ? row[j] * dj : T(0);
        }
        return hasreg ? c0v + T(2) * sacc : c0v;
    };

    // time-invariant weights (ROWC): the lane's own row of Qr / Rr, loaded once
    T rowc[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) rowc[j] = (ROWC && hasreg) ? prow[j < lim ? j : lim - 1] : T(0);

    // ---- terminal step: v = cx[N-1], k[N-1] = 0 (last segment); the others start from v_in = 0 ----------
    T vcur;
    {
        Stage term;
        fetch(N - 1, term);
        rec[D_OFF + i] = hasreg ? term.hv - (term.zv - term.lv) : T(0);
        slot_sync();
        T rowt[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) rowt[j] = ROWC ? rowc[j] : term.rrow[ROWC ? 0 : j];
        const T cterm = reg_grad(term.c0, rowt);
        vcur = last ? cterm : T(0);
        rec[xl ? V_OFF + i : DUMP_OFF] = vcur;
        if (valid && !xl && last) p.k[((int64_t)b * N + N - 1) * NU + iu] = T(0);
        slot_sync();
    }
    Stage ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(t_hi - d > t_lo ? t_hi - d : t_lo, ring[d]);

    T *kout = p.k + (int64_t)(valid ? b : 0) * N * NU + iu;
    const bool kstore = valid && !xl;

    // Groups of D steps with the ring index a compile-time constant; the last group is padded with dead steps
    // (t < t_lo: loads clamped, v and k untouched), so the body exists once and every VMEM instruction of the loop
    // is unconditional (exact vmcnt bookkeeping: D steps of loads really stay in flight).
    for (int tb = t_hi; tb >= t_lo; tb -= D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = tb - d;
            const bool live = t >= t_lo;
            Stage &g = ring[d];
            // (a) stage the operands of step t (unconditional) and publish d_i = xhat_i - (z_i - lambda_i)
#pragma unroll
            for (int j = 0; j < JA; ++j) rec[dA[j]] = g.ra[j];
#pragma unroll
            for (int j = 0; j < JB; ++j) rec[dB[j]] = g.rb[j];
#pragma unroll
            for (int j = 0; j < JK; ++j) { rec[dK[j]] = g.rk[j]; rec[dQ[j]] = g.rq[j]; }
#pragma unroll
            for (int j = 0; j < JU; ++j) { rec[dU[j]] = g.ruu[j]; rec[dF[j]] = g.rf[j]; }
            rec[D_OFF + i] = hasreg ? g.hv - (g.zv - g.lv) : T(0);
            const T c0_now = g.c0;
            T row_now[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) row_now[j] = ROWC ? rowc[j] : g.rrow[ROWC ? 0 : j];
            slot_sync();
            // refill: FG consecutive steps at once, on the last ring entry of each group of FG -- one burst of FG steps
            // per stream instead of FG separate requests (tools/streambench.hip: 3.8 -> 4.7 TB/s for FG = 4)
            if constexpr (FG == 1) {
                fetch(t - D > t_lo ? t - D : t_lo, g);             // (clamped, unconditional)
            } else if ((d % FG) == FG - 1) {
#pragma unroll
                for (int q = FG - 1; q >= 0; --q) fetch(t + q - D > t_lo ? t + q - D : t_lo, ring[d - q >= 0 ? d - q : 0]);
            }

            // q_i = c_i + ([A B]' v)_i        (isls.py:285-286)
            const T ci = reg_grad(c0_now, row_now);
            T sacc = T(0);
#pragma unroll
            for (int k = 0; k < NX; ++k) sacc += rec[AB_OFF + k * W + i] * rec[V_OFF + k];
            const T qi = ci + sacc;
            rec[xl ? DUMP_OFF : QU_OFF + iu] = qi;
            slot_sync();                                           // (b) qu visible; every lane has read v

            // k_t = -Quu^{-1} qu  (every lane), then v_i for x-lanes
            T qu[NU], kt[NU];
#pragma unroll
            for (int r = 0; r < NU; ++r) qu[r] = rec[QU_OFF + r];
            if (p.mode == ISLS_SOLVE_CHOL) {
                T U[NU][NU], rd[NU], x[NU];
#pragma unroll
                for (int r = 0; r < NU; ++r) {
#pragma unroll
                    for (int c = 0; c < NU; ++c) U[r][c] = rec[FAC_OFF + r * NU + c];
                    rd[r] = U[r][r];
                }
                chol_solve<NU>(U, rd, qu, x);
#pragma unroll
                for (int r = 0; r < NU; ++r) kt[r] = -x[r];
            } else {
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    T acc = T(0);
#pragma unroll
                    for (int c = 0; c < NU; ++c) acc += rec[FAC_OFF + r * NU + c] * qu[c];
                    kt[r] = -acc;
                }
            }
            // v_i = qx_i + (K'qu)_i + (K'Quu k)_i + (Qux'k)_i ; u-lanes evaluate the same expression on their
            // (clamped) column and throw it away: no divergence
            const int ic = xl ? i : 0;
            T t_kqu = T(0), t_kquuk = T(0), t_quxk = T(0);
            T Kc[NU];
#pragma unroll
            for (int r = 0; r < NU; ++r) { Kc[r] = rec[K_OFF + r * NX + ic]; t_kqu += Kc[r] * qu[r]; }
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                T w = T(0);
#pragma unroll
                for (int r = 0; r < NU; ++r) w += Kc[r] * rec[QUU_OFF + r * NU + c];
                t_kquuk += w * kt[c];
            }
#pragma unroll
            for (int r = 0; r < NU; ++r) t_quxk += rec[QUX_OFF + r * NX + ic] * kt[r];
            const T vnew = (p.mode == ISLS_SOLVE_CHOL) ? ((qi + t_kqu) + t_kquuk) + t_quxk      // isls.py:302
                                                       : ((qi + t_quxk) + t_kqu) + t_kquuk;     // sls.py:200
            vcur = live ? vnew : vcur;                             // dead (padding) steps leave v alone
            rec[xl ? V_OFF + i : DUMP_OFF] = vcur;                 // safe: all reads of v happened before (b)
            // k_t leaves through LDS: every lane holds the same kt[], u-lane r needs entry r -- a per-lane LDS
            // address does that for free, whereas a select chain over kt[] becomes a scratch-memory array
#pragma unroll
            for (int r = 0; r < NU; ++r) rec[KT_OFF + r] = kt[r];
            slot_sync();
            const T kv = rec[KT_OFF + iu];
            if (kstore && live) kout[(int64_t)t * NU] = kv;
        }
    }
    if (seg > 0 && valid && xl) p.vseg[((int64_t)b * p.nseg + seg) * NX + i] = vcur;   // v0 at the segment start
}

template <typename T>
int launch_ff(const isls_ff_args &a, hipStream_t s)
{
    if (a.B < 0 || a.N < 1 || !a.c0x.p || !a.c0u.p || !a.k) return ISLS_ERR_ARG;
    if (!a.rec && (!a.A.p || !a.Bm.p || !a.K || !a.Quu || !a.fac || !a.Qux)) return ISLS_ERR_ARG;
    if (a.Qr.p && (!a.zx || !a.lx)) return ISLS_ERR_ARG;
    if (a.Rr.p && (!a.zu || !a.lu)) return ISLS_ERR_ARG;
    if (a.solve_mode != ISLS_SOLVE_CHOL && a.solve_mode != ISLS_SOLVE_INV) return ISLS_ERR_ARG;
    if (a.lin_on && !a.rec) return ISLS_ERR_ARG;              // the hint describes the records' layout
    // per-lane offsets inside one workgroup are 32-bit element counts
    if ((int64_t)a.N * a.n * a.n * 64 >= ((int64_t)1 << 31) || a.A.sb * 64 >= ((int64_t)1 << 31) || a.Bm.sb * 64 >= ((int64_t)1 << 31))
        return ISLS_ERR_UNSUPPORTED;
    if (a.B == 0) return ISLS_OK;
    if (!dims_supported(a.n, a.m)) return launch_ff_generic<T>(a, s);          // generic.hip
    if ((a._pad > 1 || a.Qr_term) && !a.rec) return ISLS_ERR_UNSUPPORTED;     // feedback columns in one launch, terminal weight block: record path only
    if (a.rec) {                                               // packed records of the gain pass: riccati_ffrec.hip
        const bool sg = ff_seg_enabled(a.seg) && a.N > 2;
        if (sg && (a.seg.nseg > 16 || !a.seg.Psi || !a.seg.v || (int64_t)a.seg.nseg * a.seg.seg_len < a.N - 1 ||
                   (int64_t)(a.seg.nseg - 1) * a.seg.seg_len >= a.N - 1))
            return ISLS_ERR_ARG;
        const int rc = launch_ff_record<T>(a, s);
        if (rc != ISLS_OK || !sg) return rc;
        return launch_ff_stitch<T>(a, s);
    }
    FfP<T> p;
    p.B = a.B; p.N = a.N; p.mode = a.solve_mode;
    p.A = View<T>(a.A); p.Bm = View<T>(a.Bm); p.c0x = View<T>(a.c0x); p.c0u = View<T>(a.c0u);
    p.Qr = View<T>(a.Qr); p.Rr = View<T>(a.Rr);
    p.xhat = (const T *)a.xhat; p.uhat = (const T *)a.uhat;
    p.zx = (const T *)a.zx; p.lx = (const T *)a.lx; p.zu = (const T *)a.zu; p.lu = (const T *)a.lu;
    p.K = (const T *)a.K; p.Quu = (const T *)a.Quu; p.fac = (const T *)a.fac; p.Qux = (const T *)a.Qux;
    p.k = (T *)a.k; p.active = a.active;
    const bool segmented = ff_seg_enabled(a.seg) && a.N > 2;
    p.nseg = segmented ? a.seg.nseg : 1;
    p.seg_len = segmented ? a.seg.seg_len : (a.N > 1 ? a.N - 1 : 1);
    p.vseg = segmented ? (T *)a.seg.v : nullptr;
    if (segmented && (a.seg.nseg > 16 || !a.seg.Psi || !a.seg.v || (int64_t)a.seg.nseg * a.seg.seg_len < a.N - 1 ||
                      (int64_t)(a.seg.nseg - 1) * a.seg.seg_len >= a.N - 1))
        return ISLS_ERR_ARG;
    // weights that do not depend on the time step (or are absent) are loaded once per lane instead of once per step
    const bool rowc = (!a.Qr.p || a.Qr.st == 0) && (!a.Rr.p || a.Rr.st == 0);
#define CALL(NX_, NU_)                                                                                 \
    {                                                                                                  \
        p.tpw = pick_tpw(a.B, kWave / (NX_ + NU_), "ISLS_FF_TPW");                                     \
        const int grid = (a.B + p.tpw - 1) / p.tpw;                                                    \
        if (segmented && rowc)                                                                         \
            hipLaunchKernelGGL((riccati_ff_kernel<T, NX_, NU_, kFfSegDepth, kFfSegOcc, kFfSegGroup, true>), dim3(grid, p.nseg), dim3(64), 0, s, p); \
        else if (segmented)                                                                            \
            hipLaunchKernelGGL((riccati_ff_kernel<T, NX_, NU_, kFfSegDepth, kFfSegOcc, kFfSegGroup, false>), dim3(grid, p.nseg), dim3(64), 0, s, p); \
        else if (rowc)                                                                                 \
            hipLaunchKernelGGL((riccati_ff_kernel<T, NX_, NU_, kFfDepth, 1, kFfGroup, true>), dim3(grid), dim3(64), 0, s, p);  \
        else                                                                                           \
            hipLaunchKernelGGL((riccati_ff_kernel<T, NX_, NU_, kFfDepth, 1, kFfGroup, false>), dim3(grid), dim3(64), 0, s, p); \
    }
    ISLS_DISPATCH_DIMS(a.n, a.m, CALL)
#undef CALL
    const int rc = check_launch();
    if (rc != ISLS_OK || !segmented) return rc;
    return launch_ff_stitch<T>(a, s);
}
template int launch_ff<double>(const isls_ff_args &, hipStream_t);
template int launch_ff<float>(const isls_ff_args &, hipStream_t);

}  // namespace isls
