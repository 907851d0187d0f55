// riccati_ffrec.hip -- feed-forward pass on the packed step records of the gain pass (isls_ff_args.rec).
//
// Same recursion as riccati_ff.hip (isls/isls.py:285-302, isls/sls.py:168-202), with the gain-dependent part folded into
// operators once per gain pass:
//     qu = cu + B'v,  k = -Quu^-1 qu,  v' = (cx + A'v) + K'qu + K'Quu k + Qux'k
// and K'Quu k = -K'qu, Qux' = -K'Quu, hence   v' = cx + K'cu + Phi'v   with   Phi = A + B K  (the closed-loop matrix).
// The gain pass writes, per trajectory and step, ONE contiguous record  [Phi (n x n) | B (n x m) | K (m x n) | fac (m x m)]
// -- 81 words at n=6, m=3 instead of the 108 words of A, B, K, Quu, fac, Qux in six separate arrays -- so this pass
// streams one burst per (trajectory, step), stages 25 % fewer words through LDS and, because v' no longer waits for k,
// has two wave-internal hand-offs per step instead of three.  k_t itself is computed exactly as before (qu, then the
// cached factor), off the critical path of v.  The result equals riccati_ff_kernel's up to rounding (a different
// association of the same sums); tests compare both against the oracle at the same tolerance.
//
// Structure (slots, register ring, unconditional staging with dump words, padded dead steps, time-parallel segments via
// blockIdx.y) is that of riccati_ff.hip; see there for the reasoning.
#include "isls_common.hpp"

namespace isls {

#ifndef ISLS_FFREC_DEPTH
#define ISLS_FFREC_DEPTH 4
#endif
#ifndef ISLS_FFREC_SEG_DEPTH
#define ISLS_FFREC_SEG_DEPTH 2
#endif
#ifndef ISLS_FFREC_SEG_OCC
#define ISLS_FFREC_SEG_OCC 2
#endif
#ifndef ISLS_FFREC_GROUP
#define ISLS_FFREC_GROUP 1   // refilling the ring in groups of 2 / 4 steps measured 0.49 / 0.51 ms against 0.475 ms for single steps
#endif

template <typename T>
struct FfRecP {
    int B, N, mode, tpw;
    int nseg, seg_len;
    T *vseg;
    View<T> c0x, c0u, Qr, Rr;
    const T *xhat, *uhat, *zx, *lx, *zu, *lu;
    const T *rec;                  // [ceil(B/TPW)][N][TPW][RW]: the records of a wavefront's trajectories are contiguous per step
    T *k;
    const int32_t *active;
};

// FG: ring entries are refilled in groups of FG consecutive steps -- one burst of FG records (FG x 648 B at n=6, m=3) per
// trajectory instead of FG separate requests (larger DRAM bursts; same idea as kFfGroup in riccati_ff.hip)
template <typename T, int NX, int NU, int D, int OCC, int FG, bool ROWC>
__global__ __launch_bounds__(64, OCC) void riccati_ffrec_kernel(FfRecP<T> p)
{
    constexpr int G = NX + NU, W = NX + NU, MAXTPW = kWave / G;
    // packed record (HBM and LDS): Phi[NX][NX] | B[NX][NU] | K[NU][NX] | fac[NU][NU]
    constexpr int PHI_OFF = 0, B_OFF = PHI_OFF + NX * NX, K_OFF = B_OFF + NX * NU, FAC_OFF = K_OFF + NU * NX, RW = rec_stride(NX, NU);
    // slot: record (padded to an even word count) | d[W] | v[NX] | cu[NU] | qu[NU] | dump pair
    constexpr int D_OFF = RW, V_OFF = D_OFF + W, CU_OFF = V_OFF + NX, QU_OFF = CU_OFF + NU, DUMP_OFF = (QU_OFF + NU + 1) & ~1;
    constexpr int SLOT = DUMP_OFF + 2;                         // even: every slot's record starts on a 16-byte boundary
    __shared__ __align__(16) T lds[(MAXTPW + 1) * SLOT];       // + one dump slot for the lanes beyond the last slot
    const int TPW = p.tpw;

    const int lane = threadIdx.x;
    const int s = lane / G, i = lane - s * G;
    const int b = blockIdx.x * TPW + s;
    const bool inslot = s < TPW;
    const bool inbatch = inslot && b < p.B;
    const bool valid = inbatch && (p.active == nullptr || p.active[b] != 0);
    const int N = p.N;
    const int seg = blockIdx.y;
    const bool last = seg == p.nseg - 1;
    const int t_lo = seg * p.seg_len, t_hi = last ? N - 2 : t_lo + p.seg_len - 1;
    const int sl = inbatch ? s : 0;                            // idle lanes shadow the block's first trajectory (loads only)
    T *rec = lds + (inslot ? s : TPW) * SLOT;
    const bool xl = i < NX;
    const int iu = xl ? 0 : i - NX;
    const bool hasx = p.Qr.p != nullptr, hasu = p.Rr.p != nullptr;
    const bool hasreg = xl ? hasx : hasu;
    const int b0 = blockIdx.x * TPW;

    // ---- load plan: the records are blocked by wavefront, [block][t][slot][RW] with RW even, so the MAXTPW records of a
    // step are one contiguous, 16-byte aligned run of MAXTPW*RW words: lane l fetches the word PAIRS l, l+64, ... (fully
    // coalesced 1-KB requests, half as many as with single words -- the pass is bound by the requests it can keep in
    // flight) and drops pair w into slot w / RW of the LDS (surplus pairs into the dump pair)
    typedef T V2 __attribute__((ext_vector_type(2)));
    constexpr int BW = MAXTPW * RW, NP = BW / 2, JR = (NP + kWave - 1) / kWave;
    const T *bR = p.rec + (int64_t)blockIdx.x * N * BW;
    uint32_t oR[JR];
    int dR[JR];
#pragma unroll
    for (int j = 0; j < JR; ++j) {
        const int w = 2 * (lane + kWave * j);
        oR[j] = (uint32_t)(w < BW ? w : BW - 2);
        dR[j] = w < BW ? (w / RW) * SLOT + (w % RW) : MAXTPW * SLOT + DUMP_OFF;
    }
    const T *pc0 = (xl ? p.c0x.at(b0 + sl, 0) + i : p.c0u.at(b0 + sl, 0) + iu);
    const int64_t c0st = xl ? p.c0x.st : p.c0u.st;
    const int dd = xl ? NX : NU;
    const int64_t ovec = ((int64_t)(b0 + sl) * N) * dd + (xl ? i : iu);
    const T *ph = xl ? p.xhat : p.uhat, *pz = xl ? p.zx : p.zu, *pl = xl ? p.lx : p.lu;
    const T *prow = hasreg ? (xl ? p.Qr.at(b0 + sl, 0) + i * NX : p.Rr.at(b0 + sl, 0) + iu * NU) : nullptr;
    const int64_t rowst = xl ? p.Qr.st : p.Rr.st;
    const int lim = xl ? NX : NU;

    struct Stage {
        V2 rr[JR];
        T c0, hv, zv, lv, rrow[ROWC ? 1 : NX];
    };
    auto fetch_vec = [&](int t, Stage &g) {
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
    auto fetch = [&](int t, Stage &g) {
        const T *r = bR + (int64_t)t * BW;
#pragma unroll
        for (int j = 0; j < JR; ++j) g.rr[j] = *reinterpret_cast<const V2 *>(r + oR[j]);
        fetch_vec(t, g);
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
    T rowc[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) rowc[j] = (ROWC && hasreg) ? prow[j < lim ? j : lim - 1] : T(0);

    // ---- terminal step: v = cx[N-1], k[N-1] = 0 (last segment); the others start from v_in = 0 ----------------------
    T vcur;
    {
        Stage term;
        fetch_vec(N - 1, term);
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
    // own column of the stacked [Phi | B]: x-lane i -> Phi[:, i], u-lane r -> B[:, r]
    const int cbase = xl ? PHI_OFF + i : B_OFF + iu, cstr = xl ? NX : NU;
    const int ic = xl ? i : 0;

    for (int tb = t_hi; tb >= t_lo; tb -= D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = tb - d;
            const bool live = t >= t_lo;
            Stage &g = ring[d];
            // (a) stage the record of step t (unconditional) and publish d_i = xhat_i - (z_i - lambda_i)
#pragma unroll
            for (int j = 0; j < JR; ++j) *reinterpret_cast<V2 *>(lds + dR[j]) = g.rr[j];
            rec[D_OFF + i] = hasreg ? g.hv - (g.zv - g.lv) : T(0);
            const T c0_now = g.c0;
            T row_now[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) row_now[j] = ROWC ? rowc[j] : g.rrow[ROWC ? 0 : j];
            slot_sync();
            if constexpr (FG == 1) {
                fetch(t - D > t_lo ? t - D : t_lo, g);             // refill (clamped, unconditional)
            } else if ((d % FG) == FG - 1) {
#pragma unroll
                for (int q = FG - 1; q >= 0; --q) fetch(t + q - D > t_lo ? t + q - D : t_lo, ring[d - q >= 0 ? d - q : 0]);
            }

            // c_i, then the lane's column of [Phi | B] against v:  x-lanes cx_i + (Phi'v)_i,  u-lanes qu_r = cu_r + (B'v)_r
            const T ci = reg_grad(c0_now, row_now);
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < NX; ++k) acc += rec[cbase + k * cstr] * rec[V_OFF + k];
            const T qi = ci + acc;
            rec[xl ? DUMP_OFF : CU_OFF + iu] = ci;
            rec[xl ? DUMP_OFF : QU_OFF + iu] = qi;
            slot_sync();                                           // (b) cu, qu visible; every lane has read v

            // v_i = (cx_i + (Phi'v)_i) + (K'cu)_i     (u-lanes evaluate it on a clamped column and discard it)
            T kcu = T(0);
#pragma unroll
            for (int r = 0; r < NU; ++r) kcu += rec[K_OFF + r * NX + ic] * rec[CU_OFF + r];
            const T vnew = qi + kcu;
            vcur = live ? vnew : vcur;
            rec[xl ? V_OFF + i : DUMP_OFF] = vcur;                 // read again only after the next (a)
            // k_t = -Quu^{-1} qu from the cached factor (every lane; u-lane r keeps entry r)
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
                    T a2 = T(0);
#pragma unroll
                    for (int c = 0; c < NU; ++c) a2 += rec[FAC_OFF + r * NU + c] * qu[c];
                    kt[r] = -a2;
                }
            }
            T kv = kt[0];
#pragma unroll
            for (int r = 1; r < NU; ++r) kv = (iu == r) ? kt[r] : kv;
            if (kstore && live) kout[(int64_t)t * NU] = kv;
        }
    }
    if (seg > 0 && valid && xl) p.vseg[((int64_t)b * p.nseg + seg) * NX + i] = vcur;   // v0 at the segment start
}

template <typename T>
int launch_ff_record(const isls_ff_args &a, hipStream_t s)
{
    if ((int64_t)a.N * rec_stride(a.n, a.m) * 64 >= ((int64_t)1 << 31)) return ISLS_ERR_UNSUPPORTED;
    FfRecP<T> p;
    p.B = a.B; p.N = a.N; p.mode = a.solve_mode;
    p.c0x = View<T>(a.c0x); p.c0u = View<T>(a.c0u); p.Qr = View<T>(a.Qr); p.Rr = View<T>(a.Rr);
    p.xhat = (const T *)a.xhat; p.uhat = (const T *)a.uhat;
    p.zx = (const T *)a.zx; p.lx = (const T *)a.lx; p.zu = (const T *)a.zu; p.lu = (const T *)a.lu;
    p.rec = (const T *)a.rec; p.k = (T *)a.k; p.active = a.active;
    const bool segmented = ff_seg_enabled(a.seg) && a.N > 2;
    p.nseg = segmented ? a.seg.nseg : 1;
    p.seg_len = segmented ? a.seg.seg_len : (a.N > 1 ? a.N - 1 : 1);
    p.vseg = segmented ? (T *)a.seg.v : nullptr;
    const bool rowc = (!a.Qr.p || a.Qr.st == 0) && (!a.Rr.p || a.Rr.st == 0);
#define CALL(NX_, NU_)                                                                                                  \
    {                                                                                                                   \
        p.tpw = kWave / (NX_ + NU_);            /* the record layout is blocked by the gain pass's slots per wavefront */ \
        const int grid = (a.B + p.tpw - 1) / p.tpw;                                                                     \
        if (segmented && rowc)                                                                                          \
            hipLaunchKernelGGL((riccati_ffrec_kernel<T, NX_, NU_, ISLS_FFREC_SEG_DEPTH, ISLS_FFREC_SEG_OCC, ISLS_FFREC_GROUP, true>), dim3(grid, p.nseg), dim3(64), 0, s, p);  \
        else if (segmented)                                                                                             \
            hipLaunchKernelGGL((riccati_ffrec_kernel<T, NX_, NU_, ISLS_FFREC_SEG_DEPTH, ISLS_FFREC_SEG_OCC, ISLS_FFREC_GROUP, false>), dim3(grid, p.nseg), dim3(64), 0, s, p); \
        else if (rowc)                                                                                                  \
            hipLaunchKernelGGL((riccati_ffrec_kernel<T, NX_, NU_, ISLS_FFREC_DEPTH, 1, ISLS_FFREC_GROUP, true>), dim3(grid), dim3(64), 0, s, p);   \
        else                                                                                                            \
            hipLaunchKernelGGL((riccati_ffrec_kernel<T, NX_, NU_, ISLS_FFREC_DEPTH, 1, ISLS_FFREC_GROUP, false>), dim3(grid), dim3(64), 0, s, p);  \
    }
    ISLS_DISPATCH_DIMS(a.n, a.m, CALL)
#undef CALL
    return check_launch();
}
template int launch_ff_record<double>(const isls_ff_args &, hipStream_t);
template int launch_ff_record<float>(const isls_ff_args &, hipStream_t);

}  // namespace isls
