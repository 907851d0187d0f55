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
    int rev;                       // 1: the grid walks the trajectory blocks from the last to the first (see launch_ff_record)
    const T *Qr_term;              // nullable: weight block of the terminal step (isls_ff_args.Qr_term), batch stride Qr.sb
    int ncol;                      // feedback columns in one launch (blockIdx.z): zx, lx, zu, lu, k, vseg hold ncol blocks, c0 acts on column 0
    // model-structured form (isls_ff_args.lin_on): only [K | fac] of a record is read; A'v and B'v come from the model
    const T *lin_par;              // model parameters (isls_linearize_args.model_par), batch stride lin_par_sb
    int64_t lin_par_sb;
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
        for (int j = 0; j < JR; ++j) {
            if constexpr (ISLS_NT_FFREC) g.rr[j] = ld_stream(reinterpret_cast<const V2 *>(r + oR[j]));
            else g.rr[j] = *reinterpret_cast<const V2 *>(r + oR[j]);
        }
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

// ---------------------------------------------------------------------------------------------------------------------
// Second form of the pass (time-invariant or absent Qr / Rr rows): ONE wave-internal hand-off and ONE batch of LDS reads
// per step.  A wavefront with the SIMD to itself pays the whole LDS latency at every wait, so the step is organised around
// its only true dependence, v:
//   * cu is evaluated by every lane from the published d_u and c0u (m^2 multiply-adds) instead of being handed from the
//     u-lanes to the x-lanes, so v' = (cx_i + (Phi'v)_i) + (K'cu)_i needs no second hand-off;
//   * k_t = -Quu^-1 (cu + B'v) is off the recursion: the u-lanes publish qu_r with the step's v, and the NEXT step of the
//     loop turns it into k_t with the cached factor it kept in registers;
//   * every memory instruction is unconditional (lanes without a regularised block load c0 again, surplus lanes repeat the
//     last lane, slots without a trajectory shadow the first valid one, every lane of a slot stores an entry of k_t), so
//     the compiler's vmcnt bookkeeping stays exact and D steps of records really are in flight.
// The sums of the dense form (LIN = 0) are those of riccati_ffrec_kernel in the same order: its results are bit-identical to it.
// LIN (isls_ff_args.lin_on): 0 = the whole record, Phi'v from its [Phi | B] block.  Otherwise A and B are the linearisation of a
// built-in model whose structure the pass knows, the records are the LEAN ones the gain pass wrote under the same hint
// ([K | fac | model words] at stride rec_lean_stride: 28 instead of 82 words per step at n = 6, m = 3; 42 instead of 150 at
// n = 9) and the pass evaluates  Phi'v = A'v + K'(B'v)  with
//   LIN_DI    (ISLS_MODEL_DI):    A = [I aI; 0 I], B = [b0 I; b1 I]:  (A'v)_i = v_i (+ a v_{i-d}),  (B'v)_r = b0 v_r + b1 v_{d+r}
//   LIN_ARM3R (ISLS_MODEL_ARM3R): A = [I dtI 0; 0 I 0; J dtJ 0], B = [hI; dtI; hJ], h = dt^2/2, J = A[6:8, 0:3] (6 words per step,
//             which the gain pass puts behind fac: rec_model_words):  jv = J'v_ee,  (A'v)_q = v_q + jv,  (A'v)_qd = dt (v_q + jv) + v_qd,  (A'v)_ee = 0,
//             B'v = h (v_q + jv) + dt v_qd
//   LIN_CAR   (ISLS_MODEL_CAR):   A = I + {a02, a12, a03, a13, a23}, B = {b20, b31 = dt} (the six entries that vary ride behind
//             fac):  (A'v)_0 = v_0, (A'v)_1 = v_1, (A'v)_2 = a02 v_0 + a12 v_1 + v_2, (A'v)_3 = a03 v_0 + a13 v_1 + a23 v_2 + v_3,
//             B'v = (b20 v_2, dt v_3)
// -- the same products as the dense form in another association (results equal up to rounding, same tolerance against the oracle).
constexpr int LIN_NONE = 0, LIN_DI = 1, LIN_ARM3R = 2, LIN_CAR = 3;

template <typename T, int NX, int NU, int D, int OCC, int MODE, int LIN = LIN_NONE>
__global__ __launch_bounds__(64, OCC) void riccati_ffrec2_kernel(FfRecP<T> p)
{
    static_assert(LIN != LIN_DI || NX == 2 * NU, "double integrator: n = 2 d, m = d");
    static_assert(LIN != LIN_ARM3R || (NX == 9 && NU == 3), "planar 3R arm: n = 9, m = 3");
    static_assert(LIN != LIN_CAR || (NX == 4 && NU == 2), "car: n = 4, m = 2");
    constexpr bool LEAN = LIN != LIN_NONE;
    constexpr int G = NX + NU, W = NX + NU, TPW = kWave / G;
    // the structured forms read the LEAN records the gain pass writes under the same hint: [K | fac | model words] at stride
    // rec_lean_stride; the dense form the whole records at rec_stride
    constexpr int RW = LEAN ? rec_lean_stride(NX, NU) : rec_stride(NX, NU);   // words between the records of consecutive slots in HBM
    constexpr int NJ = (LIN == LIN_ARM3R || LIN == LIN_CAR) ? 6 : 0;   // the model words behind fac the form reads (rec_model_words)
    static_assert(NJ <= rec_model_words(NX, NU), "the records of this pair carry no model words");
    constexpr int SW = RW;                                     // words of a record staged through LDS
    constexpr int PHI_OFF = 0, B_OFF = PHI_OFF + NX * NX, K_OFF = LEAN ? 0 : B_OFF + NX * NU, FAC_OFF = K_OFF + NU * NX;
    constexpr int J_OFF = FAC_OFF + NU * NU;
    // slot: record (tail: K | fac | J) | d[W] | c0u[NU] | v[NX] | qu[NU] | dump pair
    constexpr int D_OFF = SW, C0U_OFF = D_OFF + W, V_OFF = C0U_OFF + NU, QU_OFF = V_OFF + NX, DUMP_OFF = (QU_OFF + NU + 1) & ~1;
    constexpr int SLOT = DUMP_OFF + 2;                         // even: every slot's record starts on a 16-byte boundary
    __shared__ __align__(16) T lds[(TPW + 1) * SLOT];
    typedef T V2 __attribute__((ext_vector_type(2)));

    const int lane = threadIdx.x;
    const int bx = p.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;   // trajectory block of this wavefront
    const bool inslot = lane / G < TPW;
    const int s = inslot ? lane / G : TPW - 1, i = inslot ? lane - (lane / G) * G : G - 1;   // surplus lanes repeat the last lane
    const int b = bx * TPW + s;
    const bool valid = b < p.B && (p.active == nullptr || p.active[b] != 0);
    const unsigned long long vmask = __ballot(valid);
    if (vmask == 0ull) return;
    const int bsh = __builtin_amdgcn_readlane(b, __builtin_ctzll(vmask));
    const int bb = valid ? b : bsh;                            // slots without a trajectory shadow the first valid one
    const int N = p.N;
    const int seg = blockIdx.y;
    const int col = blockIdx.z;                                // feedback column (isls_admm): own targets and k, shared records
    const T c0m = col == 0 ? T(1) : T(0);                      // the cost gradients belong to column 0
    const bool last = seg == p.nseg - 1;
    const int t_lo = seg * p.seg_len, t_hi = last ? N - 2 : t_lo + p.seg_len - 1;
    T *rec = lds + s * SLOT;
    const bool xl = i < NX;
    const int iu = xl ? 0 : i - NX;
    const bool hasx = p.Qr.p != nullptr, hasu = p.Rr.p != nullptr;
    const bool hasreg = xl ? hasx : hasu;

    // records: blocked by wavefront, [block][t][slot][RW]; lane l fetches the 16-byte pairs l, l+64, ... of the step's run.
    // A slot that shadows another trajectory still reads its own place in the run (whatever the gain pass left there) but
    // restages the shadowed slot's record below, so it computes exactly what that slot computes.
    constexpr int BW = TPW * RW, NP = TPW * SW / 2, JR = (NP + kWave - 1) / kWave;
    const T *bR = p.rec + (int64_t)bx * N * BW;
    uint32_t oR[JR];
    int dR[JR];
#pragma unroll
    for (int j = 0; j < JR; ++j) {
        const int q = lane + kWave * j;
        const int w = 2 * (q < NP ? q : NP - 1);               // word within the staged words of the wavefront's slots
        oR[j] = (uint32_t)w;                                   // (SW == RW in both layouts: the staged words are the record)
        dR[j] = q < NP ? (w / SW) * SLOT + (w % SW) : TPW * SLOT + DUMP_OFF;
    }
    // where this lane READS its slot's record: its own slot, or the shadowed one
    const int ssh = (bb - bx * TPW);                   // slot of the trajectory the lane computes (== s when valid)
    const T *rrec = lds + ssh * SLOT;
    // vectors: own component of c0, xhat/uhat, z, lambda (lanes without a regularised block load c0 again and drop it)
    const T *pc0 = xl ? p.c0x.at(bb, 0) + i : p.c0u.at(bb, 0) + iu;
    const int64_t c0st = xl ? p.c0x.st : p.c0u.st;
    const int dd = xl ? NX : NU;
    const int64_t ovec = ((int64_t)col * p.B + bb) * N * dd + (xl ? i : iu);
    const T *phat = xl ? p.xhat : p.uhat;
    const bool hash = hasreg && phat != nullptr;
    const T *ph = hash ? phat + ovec : pc0;
    const T *pz = hasreg ? (xl ? p.zx : p.zu) + ovec : pc0, *pl = hasreg ? (xl ? p.lx : p.lu) + ovec : pc0;
    const int64_t vst = hasreg ? dd : c0st, hst = hash ? dd : c0st;
    const T dmask = hasreg ? T(1) : T(0), hmask = hash ? T(1) : T(0);
    // 2 * rows of the ADMM weights: the lane's own row of Qr (x-lanes) and all of Rr (every lane evaluates cu)
    T qrow[NX], rr2[NU][NU];
#pragma unroll
    for (int j = 0; j < NX; ++j) qrow[j] = (xl && hasx) ? T(2) * p.Qr.at(bb, 0)[i * NX + j] : T(0);
#pragma unroll
    for (int r = 0; r < NU; ++r) {
#pragma unroll
        for (int c = 0; c < NU; ++c) rr2[r][c] = hasu ? T(2) * p.Rr.at(bb, 0)[r * NU + c] : T(0);
    }
    const T xmask = xl ? T(1) : T(0);
    // structured forms: (A'v)_i = v_i * sown + cv * v_o + cj * jv_o with lane constants; w = B'v by every lane
    const T *lpar = LEAN ? p.lin_par + (int64_t)bb * p.lin_par_sb : nullptr;
    T la = T(0), lb0 = T(0), lb1 = T(0), ldt = T(0), lh = T(0);
    if constexpr (LIN == LIN_DI) { la = lpar[0]; lb0 = lpar[1]; lb1 = lpar[2]; }
    if constexpr (LIN == LIN_ARM3R) { ldt = lpar[0]; lh = T(0.5) * (ldt * ldt); }
    if constexpr (LIN == LIN_CAR) ldt = lpar[0];
    const int lo = LIN == LIN_DI ? ((xl && i >= NU) ? i - NU : 0) : ((xl && i < 6) ? i % 3 : 0);   // the other entry of v the lane needs
    const T sown = LIN == LIN_ARM3R ? ((xl && i < 6) ? T(1) : T(0)) : T(1);
    const T cv = LIN == LIN_DI ? ((xl && i >= NU) ? la : T(0)) : ((xl && i >= 3 && i < 6) ? ldt : T(0));
    const T cj = LIN == LIN_ARM3R ? ((xl && i < 3) ? T(1) : ((xl && i < 6) ? ldt : T(0))) : T(0);

    struct Stage {
        V2 rr[JR];
        T c0, hv, zv, lv;
    };
    auto fetch = [&](int tq, Stage &g) {
        const int t = __builtin_amdgcn_readfirstlane(tq);
        const T *r = bR + (int64_t)t * BW;
#pragma unroll
        for (int j = 0; j < JR; ++j) {                          // the records stream through once per pass: 78 -> 71 us with `nt`
            if constexpr (ISLS_NT_FFREC) g.rr[j] = ld_stream(reinterpret_cast<const V2 *>(r + oR[j]));
            else g.rr[j] = *reinterpret_cast<const V2 *>(r + oR[j]);
        }
        g.c0 = pc0[(int64_t)t * c0st];
        g.hv = ph[(int64_t)t * hst];
        g.zv = pz[(int64_t)t * vst];
        g.lv = pl[(int64_t)t * vst];
    };
    const int d_dst = D_OFF + i;                               // own d component (x-lanes: d_x[i]; u-lanes: d_u[r] at D_OFF + NX + r)
    const int c_dst = xl ? DUMP_OFF : C0U_OFF + iu;            // u-lanes publish c0u_r
    const int o_dst = xl ? V_OFF + i : QU_OFF + iu;            // x-lanes publish v_i, u-lanes qu_r
    const int cbase = xl ? PHI_OFF + i : B_OFF + iu, cstr = xl ? NX : NU;   // own column of [Phi | B] (dense form)
    const int ic = xl ? i : 0;
    const int ku = xl ? i % NU : iu;                           // the entry of k this lane stores (x-lanes: copies)
    T *const kbase = p.k + ((int64_t)col * p.B + bb) * N * NU + ku;

    // ---- terminal step: v = cx[N-1] (last segment; the others start from v_in = 0), qu = 0 -> k[t_hi + 1] ... -------------
    T vcur;
    {
        Stage term;
        fetch(N - 1, term);
        rec[d_dst] = dmask * (hmask * term.hv - (term.zv - term.lv));
        slot_sync();
        T sacc = T(0);
        const T *qT = (p.Qr_term && xl) ? p.Qr_term + (int64_t)bb * p.Qr.sb + i * NX : nullptr;   // the terminal step's own weight row
#pragma unroll
        for (int j = 0; j < NX; ++j) sacc += (qT ? T(2) * qT[j] : qrow[j]) * rrec[D_OFF + j];
        const T cterm = c0m * term.c0 + sacc;                  // u-lanes: unused
        vcur = (last && xl) ? cterm : T(0);
        rec[o_dst] = xl ? vcur : T(0);                         // v, and qu = 0: the first k the loop emits is k[N-1] = 0 (last segment)
        slot_sync();
    }
    Stage ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        fetch(t_hi - d > t_lo ? t_hi - d : t_lo, ring[d]);
        __builtin_amdgcn_sched_barrier(0);                     // issue order = consumption order: vmcnt is in order, and a prologue
    }                                                          // the scheduler shuffled costs a vmcnt(0) at the top of every trip
    T fac_h[NU][NU];                                           // cached factor of the step whose qu is pending
#pragma unroll
    for (int r = 0; r < NU; ++r) {
#pragma unroll
        for (int c = 0; c < NU; ++c) fac_h[r][c] = (r == c) ? T(1) : T(0);
    }
    // k of the step before: its row index.  The first iteration of a segment but the last has no pending step: it stores a
    // throw-away value into its OWN step's entry, which the next iteration overwrites (same lane, same address, in order).
    int tprev = last ? N - 1 : t_hi;

    for (int tb = t_hi; tb >= t_lo; tb -= D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = tb - d;
            const bool live = t >= t_lo;                       // uniform
            Stage &g = ring[d];
            // stage the records of the step and publish d_i (u-lanes: c0u_r as well)
#pragma unroll
            for (int j = 0; j < JR; ++j) *reinterpret_cast<V2 *>(lds + dR[j]) = g.rr[j];
            rec[d_dst] = dmask * (hmask * g.hv - (g.zv - g.lv));
            rec[c_dst] = c0m * g.c0;
            const T c0_own = c0m * g.c0;
            slot_sync();
            fetch(t - D > t_lo ? t - D : t_lo, g);             // refill (clamped, unconditional)
            // ---- one batch of reads ----
            T dx[NX], du[NU], c0u[NU], col[NX], vv[NX], kcol[NU], facn[NU][NU], qup[NU];
            T v_own = T(0), v_oth = T(0), jm[NJ > 0 ? NJ : 1];
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                dx[j] = rrec[D_OFF + j];
                vv[j] = rrec[V_OFF + j];
                if constexpr (!LEAN) col[j] = rrec[cbase + j * cstr];
            }
            if constexpr (LEAN) {
                v_own = rrec[V_OFF + ic];
                v_oth = rrec[V_OFF + lo];
#pragma unroll
                for (int e = 0; e < NJ; ++e) jm[e] = rrec[J_OFF + e];
            }
#pragma unroll
            for (int r = 0; r < NU; ++r) {
                du[r] = rrec[D_OFF + NX + r]; c0u[r] = rrec[C0U_OFF + r]; kcol[r] = rrec[K_OFF + r * NX + ic]; qup[r] = rrec[QU_OFF + r];
#pragma unroll
                for (int c = 0; c < NU; ++c) facn[r][c] = rrec[FAC_OFF + r * NU + c];
            }
            __builtin_amdgcn_sched_barrier(0);
            // cu (every lane), c_i, the lane's column of [Phi | B] against v
            T cu[NU];
#pragma unroll
            for (int r = 0; r < NU; ++r) {
                T sacc = T(0);
#pragma unroll
                for (int c = 0; c < NU; ++c) sacc += rr2[r][c] * du[c];
                cu[r] = c0u[r] + sacc;
            }
            T sx = T(0);
#pragma unroll
            for (int j = 0; j < NX; ++j) sx += qrow[j] * dx[j];
            T ci = c0_own + sx;                                // x-lanes: cx_i
#pragma unroll
            for (int r = 0; r < NU; ++r) ci = (!xl && iu == r) ? cu[r] : ci;
            T qi, vnew;
            if constexpr (!LEAN) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < NX; ++k) acc += col[k] * vv[k];
                qi = ci + acc;                                 // x-lanes: cx_i + (Phi'v)_i; u-lanes: qu_r
                T kcu = T(0);
#pragma unroll
                for (int r = 0; r < NU; ++r) kcu += kcol[r] * cu[r];
                vnew = qi + kcu;
            } else {
                T w[NU], jvo = T(0);                           // w = B'v; jvo = (J'v_ee)_o of the lane's own o
                if constexpr (LIN == LIN_DI) {
#pragma unroll
                    for (int r = 0; r < NU; ++r) w[r] = lb0 * vv[r] + lb1 * vv[NU + r];
                } else if constexpr (LIN == LIN_CAR) {
                    w[0] = jm[5] * vv[2];
                    w[1] = ldt * vv[3];
                } else {
#pragma unroll
                    for (int r = 0; r < NU; ++r) {
                        const T jv = jm[r] * vv[6] + jm[3 + r] * vv[7];
                        w[r] = lh * (vv[r] + jv) + ldt * vv[3 + r];
                        jvo = (lo == r) ? jv : jvo;
                    }
                }
                T atv = (sown * v_own + cv * v_oth) + cj * jvo;            // x-lanes: (A'v)_i
                if constexpr (LIN == LIN_CAR) {
                    const T a2 = (jm[0] * vv[0] + jm[1] * vv[1]) + vv[2];
                    const T a3 = ((jm[2] * vv[0] + jm[3] * vv[1]) + jm[4] * vv[2]) + vv[3];
                    atv = (i == 2) ? a2 : ((i == 3) ? a3 : v_own);
                }
                T wu = w[0];
#pragma unroll
                for (int r = 1; r < NU; ++r) wu = (iu == r) ? w[r] : wu;
                qi = ci + (xl ? atv : wu);                     // x-lanes: cx_i + (A'v)_i; u-lanes: qu_r = cu_r + (B'v)_r
                T kq = T(0);
#pragma unroll
                for (int r = 0; r < NU; ++r) kq += kcol[r] * (cu[r] + w[r]);   // (K' qu)_i = (K'cu)_i + (K'B'v)_i
                vnew = qi + kq;
            }
            vcur = live ? vnew : vcur;
            // k of the previous step from its cached factor and qu (this step's hand-off delivered qu)
            T kt[NU];
            if constexpr (MODE == ISLS_SOLVE_CHOL) {
                T rd[NU], x[NU];
#pragma unroll
                for (int r = 0; r < NU; ++r) rd[r] = fac_h[r][r];
                chol_solve<NU>(fac_h, rd, qup, x);
#pragma unroll
                for (int r = 0; r < NU; ++r) kt[r] = T(0) - x[r];     // -x, and +0 for the k[N-1] = 0 the first iteration emits
            } else {
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    T a2 = T(0);
#pragma unroll
                    for (int c = 0; c < NU; ++c) a2 += fac_h[r][c] * qup[c];
                    kt[r] = T(0) - a2;
                }
            }
            T kv = kt[0];
#pragma unroll
            for (int r = 1; r < NU; ++r) kv = (ku == r) ? kt[r] : kv;
            kbase[(int64_t)tprev * NU] = kv;
            // publish v_i / qu_r (dead steps publish what is there already: vcur, and the pending qu again)
            T outv = xl ? vcur : qi;
#pragma unroll
            for (int r = 0; r < NU; ++r) outv = (!live && !xl && iu == r) ? qup[r] : outv;
            rec[o_dst] = outv;
            if (live) {                                        // uniform: scalar bookkeeping only
                tprev = t;
#pragma unroll
                for (int r = 0; r < NU; ++r) {
#pragma unroll
                    for (int c = 0; c < NU; ++c) fac_h[r][c] = facn[r][c];
                }
            }
        }
    }
    // ---- epilogue: k of the segment's first step from the last hand-off ---------------------------------------------------
    slot_sync();
    {
        T qup[NU], kt[NU];
#pragma unroll
        for (int r = 0; r < NU; ++r) qup[r] = rrec[QU_OFF + r];
        if constexpr (MODE == ISLS_SOLVE_CHOL) {
            T rd[NU], x[NU];
#pragma unroll
            for (int r = 0; r < NU; ++r) rd[r] = fac_h[r][r];
            chol_solve<NU>(fac_h, rd, qup, x);
#pragma unroll
            for (int r = 0; r < NU; ++r) kt[r] = -x[r];
        } else {
#pragma unroll
            for (int r = 0; r < NU; ++r) {
                T a2 = T(0);
#pragma unroll
                for (int c = 0; c < NU; ++c) a2 += fac_h[r][c] * qup[c];
                kt[r] = -a2;
            }
        }
        T kv = kt[0];
#pragma unroll
        for (int r = 1; r < NU; ++r) kv = (ku == r) ? kt[r] : kv;
        kbase[(int64_t)tprev * NU] = kv;
    }
    if (seg > 0 && valid && xl) p.vseg[(((int64_t)col * p.B + b) * p.nseg + seg) * NX + i] = vcur;   // v0 at the segment start
}

static bool v2_on_()
{
    static const bool on = [] { const char *e = getenv("ISLS_FF_V2"); return !e || atoi(e) != 0; }();
    return on;
}

#ifndef ISLS_FF2_LEAN_DEPTH
#define ISLS_FF2_LEAN_DEPTH 3      // ring depth of the model-structured form (a third of the record words per entry)
#endif
// the one-hand-off kernel in the form `lin` selects (the structured forms exist for the dimensions their models have)
template <typename T, int NX, int NU, int D, int DL, int OCC, int MODE>
static void launch_ffrec2(int lin, dim3 grid, hipStream_t s, const FfRecP<T> &p)
{
    if constexpr (NX == 2 * NU) {
        if (lin == LIN_DI) {
            hipLaunchKernelGGL((riccati_ffrec2_kernel<T, NX, NU, DL, OCC, MODE, LIN_DI>), grid, dim3(64), 0, s, p);
            return;
        }
    }
    if constexpr (NX == 9 && NU == 3) {
        if (lin == LIN_ARM3R) {
            hipLaunchKernelGGL((riccati_ffrec2_kernel<T, NX, NU, DL, OCC, MODE, LIN_ARM3R>), grid, dim3(64), 0, s, p);
            return;
        }
    }
    if constexpr (NX == 4 && NU == 2) {
        if (lin == LIN_CAR) {
            hipLaunchKernelGGL((riccati_ffrec2_kernel<T, NX, NU, DL, OCC, MODE, LIN_CAR>), grid, dim3(64), 0, s, p);
            return;
        }
    }
    hipLaunchKernelGGL((riccati_ffrec2_kernel<T, NX, NU, D, OCC, MODE, LIN_NONE>), grid, dim3(64), 0, s, p);
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
    // EXPERIMENT (ISLS_FF_REV = 1: every pass walks the blocks backwards; 2: consecutive passes alternate)
    p.Qr_term = (const T *)a.Qr_term;
    if (p.Qr_term && !(rowc && v2_on_() && a.Qr.p)) return ISLS_ERR_UNSUPPORTED;
    p.ncol = a._pad > 1 ? a._pad : 1;
    if (p.ncol > 1 && !(rowc && v2_on_())) return ISLS_ERR_UNSUPPORTED;   // columns ride on the one-hand-off kernel only
    static const int rev_mode = [] { const char *e = getenv("ISLS_FF_REV"); return e ? atoi(e) : 0; }();
    static int rev_count = 0;
    p.rev = rev_mode == 1 ? 1 : (rev_mode == 2 ? (rev_count++ & 1) : 0);
#ifndef ISLS_FF2_SEQ_DEPTH
#define ISLS_FF2_SEQ_DEPTH 3
#endif
#ifndef ISLS_FF2_SEG_DEPTH
#define ISLS_FF2_SEG_DEPTH 2
#endif
    const bool v2_on = v2_on_();
    // model-structured form (isls_ff_args.lin_on): the records are the LEAN ones the gain pass wrote under the same hint, which
    // only the one-hand-off kernel reads -- time-varying weights, the time-parallel form (its operators come from the dense
    // records) or ISLS_FF_V2 = 0 are ISLS_ERR_UNSUPPORTED, not a fall-back
    int lin = LIN_NONE;
    if (a.lin_on) {
        if (a.lin_model == ISLS_MODEL_DI) { if (a.n != 2 * a.m) return ISLS_ERR_ARG; lin = LIN_DI; }
        else if (a.lin_model == ISLS_MODEL_ARM3R) { if (a.n != 9 || a.m != 3) return ISLS_ERR_ARG; lin = LIN_ARM3R; }
        else if (a.lin_model == ISLS_MODEL_CAR) { if (a.n != 4 || a.m != 2) return ISLS_ERR_ARG; lin = LIN_CAR; }
        else return ISLS_ERR_UNSUPPORTED;
        if (!a.lin_par) return ISLS_ERR_ARG;
        if (!(rowc && v2_on) || segmented) return ISLS_ERR_UNSUPPORTED;
    }
    p.lin_par = (const T *)a.lin_par; p.lin_par_sb = a.lin_par_sb;
#define LAUNCH2(NX_, NU_, MODE_)                                                                                        \
    {                                                                                                                   \
        if (segmented) launch_ffrec2<T, NX_, NU_, ISLS_FF2_SEG_DEPTH, ISLS_FF2_SEG_DEPTH, (NX_ * NX_ > 64 ? 1 : 2), MODE_>(lin, dim3(grid, p.nseg, p.ncol), s, p); /* n = 9: 256 registers spill */ \
        else launch_ffrec2<T, NX_, NU_, ISLS_FF2_SEQ_DEPTH, ISLS_FF2_LEAN_DEPTH, 1, MODE_>(lin, dim3(grid, 1, p.ncol), s, p);                  \
    }
#define CALL(NX_, NU_)                                                                                                  \
    {                                                                                                                   \
        p.tpw = kWave / (NX_ + NU_);            /* the record layout is blocked by the gain pass's slots per wavefront */ \
        const int grid = (a.B + p.tpw - 1) / p.tpw;                                                                     \
        if (rowc && v2_on) {                                                                                            \
            if (a.solve_mode == ISLS_SOLVE_CHOL) LAUNCH2(NX_, NU_, ISLS_SOLVE_CHOL)                                     \
            else LAUNCH2(NX_, NU_, ISLS_SOLVE_INV)                                                                      \
        } else if (segmented && rowc)                                                                                   \
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
#undef LAUNCH2
    return check_launch();
}
template int launch_ff_record<double>(const isls_ff_args &, hipStream_t);
template int launch_ff_record<float>(const isls_ff_args &, hipStream_t);

}  // namespace isls
