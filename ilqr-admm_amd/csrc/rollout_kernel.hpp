// rollout_kernel.hpp -- forward line-search rollout + cost + arg-min + winner trajectory on gfx950 (kernel template).
//
// Reference semantics: iSLS.rollout_DP (isls/isls.py:310-334), iterate_once_dp candidates / NaN rule /
// arg-min / acceptance (isls.py:357-369), SLSBase.compute_cost (isls/sls_base.py:25-44), AL terms of
// the ilqr_admm line search (isls.py:471-477), SLSBase.get_trajectory_dp (sls_base.py:76-89).
//
// Mapping: one 64-lane wavefront per workgroup, cut into TPW = 64/GL slots (GL = max(L,8) lanes);
// slot s owns trajectory blockIdx.x*TPW+s and lane c of the slot owns line-search candidate c (its
// state x lives in registers for the whole horizon).
//   SEARCH  : the per-step operands shared by the candidates of a trajectory (K_t, k_t, xhat_t, uhat_t, the
//             ADMM targets z-lambda and time-varying AL weights) are fetched by the slot's lanes as PAIRS of
//             adjacent words (one 16-byte load per lane and step at n=6, m=3, fp64), D steps ahead (register
//             ring, unconditional loads, running pointers), dropped into a double-buffered LDS record and read
//             back as broadcasts.  The load plan only covers arrays that are present and vary over time: absent
//             operands stay zero in the record, time-invariant AL weights are written once.
//             The steady-state loop runs branch-free groups of D steps; the last 2D-1 steps at most go through
//             a general tail (clamped refills, padded dead steps).
//             Every S steps each candidate drops its state into an LDS checkpoint.
//   ARG-MIN : first minimum over the slot's candidates (numpy NaN semantics, optional NaN rule/accept test).
//   WINNER  : the reference returns x_noms[ind].  The search RECORDS the controls of one candidate per trajectory -- the winner
//             of the previous line search of that trajectory (isls_rollout_args.best as it stands at launch) -- in the slot's
//             LDS, step by step.  When that candidate wins again (the rule in the ADMM iterations of an outer iteration, where
//             the full step keeps winning) the winner's u_t are already on chip: the fused ADMM update sweeps them at once, and
//             x_t, where somebody will read it, follows from the candidates' checkpoints (every S steps) by NSEG lanes
//             stepping the model through their segments -- no operand of the search is fetched a second time.
//             Otherwise the winner is REPLAYED: the horizon is cut into NSEG segments of S steps and the lanes of segment sg
//             re-run steps [sg S, (sg+1) S) of candidate `ind` from its checkpoint, fetching K_t, k_t, xhat_t, uhat_t again.
//             Either way the trajectory is collected in LDS (x over the dead records and checkpoints), leaves in one coalesced
//             sweep, and the element-wise ADMM update of the outer driver rides on it.
#pragma once
// Ablation switches (wrong results by design; tools/ab_build.sh + tools/kbench.py): -DISLS_RO_EXP_NOREPLAY
// (no winner replay), _NOREFILL (replay without its operand refills), _NOXCHG (no exchange of the control rows), _NOSTAGE
// (no stage writes), _HALFLDS (the search reads every second operand pair from the LDS: 78.6 -> 72.3 us, i.e. the search is not
// bound by the LDS pipeline); -DISLS_DIAG prints cycle stamps of the phases.

#include <type_traits>

#include "isls_common.hpp"

namespace isls {

template <typename T>
struct RoP {
    int B, N, L, flags, nseg, seg_len;
    int seg_lanes;                 // lanes per replay segment: 1, or NU (one control row per lane)
    int stage_on;                  // winner trajectory collected in LDS and written out in one sweep (it fits the slot)
    int tpw;                       // trajectories (slots) per wavefront: 64 / max(L, 8), or fewer when that keeps the stage on chip
    int u_off;                     // element offset of the winner's u rows [N][NU] in the slot's region (behind x [N][NX] and, with
                                   // the recording on, behind the search's records and checkpoints: they are written DURING the search)
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
    int cost_model;
    const T *cpar;                 // ISLS_COST_PHUBER parameters [NU + 4 NX]
    // z / dual update of the ADMM (isls_admm_update semantics, admm.hip) fused behind the winner replay by the outer
    // driver: the winner's x_t, u_t are in LDS, so the update costs two reads and two writes per element and saves a
    // launch and a second pass over x, u
    int fa_on, fa_proj_x, fa_proj_u;
    int fa_last;                   // fused form: 0 = more ADMM iterations follow in this outer iteration, so x_out / u_out are
                                   // needed only for the trajectories that stop now (nothing else reads them in between)
    T fa_relax, fa_tol_abs, fa_tol_rel;
    T *fa_zx, *fa_lx, *fa_zu, *fa_lu, *fa_res, *fa_res_prev;
    View<T> fa_xlo, fa_xhi, fa_ulo, fa_uhi;
    int32_t *fa_active, *fa_iters;
};

// ---- built-in forward models -------------------------------------------------
// sin_cos: isls_common.hpp
template <typename T, int NX, int NU, int MODEL>
struct Model;

template <typename T, int NX, int NU>
struct Model<T, NX, NU, ISLS_MODEL_LTI> {      // x+ = A x + B u   (isls/sls_base.py:49-53)
    // [A B] lives in the slot's LDS (NX x (NX+NU) words, read back as broadcasts): 54 doubles in registers per
    // lane would push the kernel past 256 VGPRs, i.e. down to one wavefront per SIMD
    static constexpr int LDS_WORDS = NX * (NX + NU);
    const T *ab;
    __device__ __forceinline__ void load(const T *par, T *lds_words, int c, int GL)
    {
        for (int e = c; e < NX * NX; e += GL) lds_words[(e / NX) * (NX + NU) + e % NX] = par[e];
        for (int e = c; e < NX * NU; e += GL) lds_words[(e / NU) * (NX + NU) + NX + e % NU] = par[NX * NX + e];
        ab = lds_words;
    }
    __device__ __forceinline__ void step(const T (&x)[NX], const T (&u)[NU], T (&xn)[NX]) const
    {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            T s = T(0), r = T(0);
#pragma unroll
            for (int j = 0; j < NX; ++j) s += ab[i * (NX + NU) + j] * x[j];
#pragma unroll
            for (int j = 0; j < NU; ++j) r += ab[i * (NX + NU) + NX + j] * u[j];
            xn[i] = s + r;
        }
    }
};

template <typename T, int NX, int NU>
struct Model<T, NX, NU, ISLS_MODEL_DI> {       // double integrator through its Kronecker structure (see isls_hip.h)
    static_assert(NX == 2 * NU, "double integrator: n = 2 d, m = d");
    static constexpr int LDS_WORDS = 0;
    T a, b0, b1;
    __device__ __forceinline__ void load(const T *par, T *, int, int) { a = par[0]; b0 = par[1]; b1 = par[2]; }
    __device__ __forceinline__ void step(const T (&x)[NX], const T (&u)[NU], T (&xn)[NX]) const
    {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            xn[i] = (x[i] + a * x[NU + i]) + b0 * u[i];
            xn[NU + i] = x[NU + i] + b1 * u[i];
        }
    }
};

template <typename T>
struct Model<T, 9, 3, ISLS_MODEL_ARM3R> {      // planar 3R arm, state [q, qd, ee]  (3DoF notebooks cell 9)
    static constexpr int LDS_WORDS = 0;
    T dt;
    __device__ __forceinline__ void load(const T *par, T *, int, int) { dt = par[0]; }
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
            T sn, cs;
            sin_cos(c, sn, cs);                               // one range reduction for both (the same values as sin / cos)
            ex += cs;
            ey += sn;
        }
        xn[6] = ex; xn[7] = ey; xn[8] = T(0);
    }
};

template <typename T>
struct Model<T, 4, 2, ISLS_MODEL_CAR> {        // car-simple [x, y, theta, v]  (Car notebooks cell 6)
    static constexpr int LDS_WORDS = 0;
    T dt;
    __device__ __forceinline__ void load(const T *par, T *, int, int) { dt = par[0]; }
    __device__ __forceinline__ void step(const T (&x)[4], const T (&u)[2], T (&xn)[4]) const
    {
        T sn, cs;
        sin_cos(x[2], sn, cs);
        xn[0] = x[0] + dt * x[3] * cs;
        xn[1] = x[1] + dt * x[3] * sn;
        xn[2] = py_mod(x[2] + dt * x[3] * u[0], T(2 * 3.14159265358979323846));
        xn[3] = x[3] + dt * u[1];
    }
};

template <typename T>
struct Model<T, 4, 2, ISLS_MODEL_TASSA> {      // Tassa car-parking [x, y, theta, v], u = [w, a]  (Tutorial.ipynb cell 8)
    static constexpr int LDS_WORDS = 0;
    T dt, d;
    __device__ __forceinline__ void load(const T *par, T *, int, int) { dt = par[0]; d = par[1]; }
    __device__ __forceinline__ void step(const T (&x)[4], const T (&u)[2], T (&xn)[4]) const
    {
        const T f = dt * x[3];
        T su, cu_, s2, c2;
        sin_cos(u[0], su, cu_);
        sin_cos(x[2], s2, c2);
        const T sw = su * f;
        const T b = (f * cu_ + d) - sqrt(d * d - sw * sw);
        xn[0] = x[0] + b * c2;
        xn[1] = x[1] + b * s2;
        xn[2] = x[2] + asin(sw / d);
        xn[3] = x[3] + u[1] * dt;
    }
};

#ifndef ISLS_RO_SW
#define ISLS_RO_SW 16
#endif
#ifndef ISLS_RO_HOIST_WR
#define ISLS_RO_HOIST_WR 0
#endif
// EXPERIMENT (off): the augmented-Lagrangian operands of the u block read with the step's first batch of LDS reads instead of
// behind `if (has_wr)` -- one exposed LDS round trip per step less, 12 registers more: the two-wavefront form of n = 6, m = 3
// reaches 256 registers and spills 64 B, line search 70.2 -> 72.2 us, outer iteration +15-20 us (tools/kbench.py, same box)
constexpr bool kHoistWr = ISLS_RO_HOIST_WR != 0;
constexpr int kRolloutDepth = 2;   // steps of record words in flight per lane (D = 2..5 ran within 3 %: issue bound; 2 is leanest)
constexpr int kMaxSeg = 16;        // winner replay: at most this many segments

template <int NX, int NU>
struct RoLayout {
    // words a lane fetches per load: adjacent pairs wherever every per-step group has at least two words
    static constexpr int W = NU >= 2 ? 2 : 1;
    // record (elements): K | xh | rx | wq | k | uh | ru | wr | dump (W words); with pair loads every group starts on an even
    // word and the record on a 16-byte boundary, so the candidates read it back two words at a time (ds_read_b128: half the
    // LDS cycles of ds_read2_b64 -- the LDS pipeline of a CU is shared by all its wavefronts and ran at ~50 %)
    __host__ __device__ static constexpr int even(int g) { return W == 2 ? g + (g & 1) : g; }
    static constexpr int O_K = 0, O_XH = O_K + even(NU * NX), O_RX = O_XH + even(NX), O_WQ = O_RX + even(NX), O_KK = O_WQ + even(NX),
                         O_UH = O_KK + even(NU), O_RU = O_UH + even(NU), O_WR = O_RU + even(NU), REC = O_WR + even(NU), O_DUMP = REC,
                         RECP = W == 2 ? REC + 2 : ((REC + 1) | 1);
    __host__ __device__ static constexpr int npair(int g) { return (g + W - 1) / W; }
    // loads per step of a slot: everything present and time-varying (upper bound of the plan; see pairs_needed)
    static constexpr int MAXPAIRS = npair(NU * NX) + 3 * npair(NX) + 4 * npair(NU);
    static constexpr int BPAIRS = npair(NX) + npair(NU);        // the two groups with a second operand (z - lambda)
    static_assert(BPAIRS <= 8, "the z - lambda groups must fit the first load slot of the narrowest slot width");
    // slot (elements): aug[GL] | plain[GL] | model | region
    //   region during the search: 2 records | checkpoints [L+1][nseg][NX] (row L: dump for idle lanes) | ... | u [N][NU]
    //   region after it         : the winner's trajectory x [N][NX] | ... | u [N][NU] (the "stage"), written out in one sweep
    //   with the stage on, the u rows sit behind BOTH (u_off): the search records the predicted winner's controls there
    __host__ __device__ static constexpr int mdl_elems(int mdlw) { return mdlw + (mdlw & 1); }   // model words of the slot ([A B] of an LTI model), even
    __host__ __device__ static constexpr int search_elems(int L, int nseg) { return 2 * RECP + (L + 1) * nseg * NX + 1; }
    __host__ __device__ static constexpr int u_off(int L, int nseg, int N)
    {
        return search_elems(L, nseg) > N * NX ? search_elems(L, nseg) : N * NX;
    }
    __host__ __device__ static constexpr int region_elems(int L, int nseg, int N, bool stage_on)
    {
        return stage_on ? u_off(L, nseg, N) + N * NU + 1 : search_elems(L, nseg);
    }
    __host__ __device__ static constexpr int slot_elems(int L, int GL, int nseg, int N, bool stage_on, int mdlw)
    {
        const int e = 2 * GL + mdl_elems(mdlw) + region_elems(L, nseg, N, stage_on);
        return W == 2 ? e + (e & 1) : e;                       // even: the records of every slot stay 16-byte aligned
    }
    // pairs the load plan really has for a launch (the launcher picks the kernel's slot count JM from it)
    __host__ __device__ static constexpr int pairs_needed(bool has_xh, bool has_uh, bool has_wq, bool wq_var, bool has_wr, bool wr_var)
    {
        return npair(NU * NX) + npair(NU) + (has_xh ? npair(NX) : 0) + (has_uh ? npair(NU) : 0) + (has_wq ? npair(NX) : 0) +
               (has_wq && wq_var ? npair(NX) : 0) + (has_wr ? npair(NU) : 0) + (has_wr && wr_var ? npair(NU) : 0);
    }
};

// W adjacent words as one load (8-byte aligned 16-byte loads are legal global accesses on gfx950)
template <typename T, int W>
struct alignas(sizeof(T)) RoVec {
    T v[W];
};

// one staging load of the search (W adjacent words); ISLS_NT_RO_LD: as a streaming access
template <typename T, int W>
__device__ __forceinline__ RoVec<T, W> ro_load(const char *ptr)
{
    if constexpr (ISLS_NT_RO_LD) {
        RoVec<T, W> r;
        if constexpr (W == 2) {
            typedef T V2 __attribute__((ext_vector_type(2)));
            typedef V2 V2u __attribute__((aligned(sizeof(T))));
            const V2 v = ld_stream(reinterpret_cast<const V2u *>(ptr));
            r.v[0] = v.x;
            r.v[1] = v.y;
        } else {
            r.v[0] = ld_stream(reinterpret_cast<const T *>(ptr));
        }
        return r;
    } else {
        return *reinterpret_cast<const RoVec<T, W> *>(ptr);
    }
}

// CNT words of a record group (CNT even with pair loads: groups are padded) read back as aligned pairs: one ds_read_b128
// per two doubles
template <int W, int CNT, typename T>
__device__ __forceinline__ void ro_read(const T *src, T (&out)[CNT])
{
    if constexpr (W == 2) {
        typedef T V2 __attribute__((ext_vector_type(2)));
        static_assert(CNT % 2 == 0, "padded group");
#pragma unroll
        for (int i = 0; i < CNT / 2; ++i) {
#ifdef ISLS_RO_EXP_HALFLDS
            const V2 v = *reinterpret_cast<const V2 *>(src + 2 * (i & ~1));   // ablation: every second pair read (wrong results)
#else
            const V2 v = *reinterpret_cast<const V2 *>(src + 2 * i);
#endif
            out[2 * i] = v.x;
            out[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CNT; ++i) out[i] = src[i];
    }
}

// load-plan walk: the lane's pair index q is counted down through the groups that are present; the group it lands in
// fixes the lane's source pointer(s), step stride and record word
template <typename T>
struct RoPlan {
    int q;
    bool done = false;
    const T *a = nullptr, *b = nullptr;
    int st = 0, off = 0;
};
template <int W, typename T>
__device__ __forceinline__ void ro_plan_group(RoPlan<T> &pl, bool present, const T *ga, const T *gb, int g, int o, int gst)
{
    if (pl.done || !present) return;
    const int np = (g + W - 1) / W;
    if (pl.q < np) {
        int e = pl.q * W;
        if (W == 2 && e + 1 >= g) e = g - 2;                    // odd group: the last pair starts one word early
        pl.a = ga + e;
        pl.b = gb ? gb + e : nullptr;
        pl.st = gst;
        pl.off = o + e;
        pl.done = true;
    } else {
        pl.q -= np;
    }
}

// first set bit at position >= from in the 256-bit mask (m0 = bits 0..63, ...); 256 when there is none
__device__ __forceinline__ int ro_next_bit(unsigned long long m0, unsigned long long m1, unsigned long long m2,
                                           unsigned long long m3, int from)
{
    for (int w = from >> 6; w < 4; ++w) {
        unsigned long long mm = w == 0 ? m0 : (w == 1 ? m1 : (w == 2 ? m2 : m3));
        if (w == (from >> 6)) mm &= ~0ull << (from & 63);
        if (mm) return w * 64 + __builtin_ctzll(mm);
    }
    return 256;
}

// Winner replay: RL lanes per segment (RL = 1: the segment lane does everything; RL = NU: lane r owns control row r --
// its row of K_t, k_t[r], uhat_t[r] -- the RL lanes exchange u through the LDS and advance the state redundantly).  With
// RL = NU the lanes of a segment read adjacent rows (one contiguous run per step instead of a scattered gather per lane:
// the gather form is bound by the address path of the vector cache, ~2.7k cycles per iteration at n=6, m=3) and a lane
// carries 2n+2 instead of nm+n+2m operand words per step, so the ring reaches four iterations ahead.
template <typename T, int NX, int RPL>
struct RoWOp {
    T K[RPL * NX], k[RPL], uh[RPL], xh[NX];
};
// Raw, unconditional loads (absent xhat / uhat: the caller passes a pointer into K and multiplies the words by zero where it
// uses them): a branch in the fetch, even a wave-uniform one, makes the compiler wait for vmcnt(0) in the replay loop and
// the ring of operands in flight is gone.
template <typename T, int NX, int NU, int RPL>
__device__ __forceinline__ void ro_wfetch(RoWOp<T, NX, RPL> &o, const T *wK, const T *wk, const T *wxh, const T *wuh, int i, int last)
{
    const int ii = i < last ? i : last;
    const T *qK = wK + (int64_t)ii * (NU * NX), *qk = wk + (int64_t)ii * NU;
#pragma unroll
    for (int e = 0; e < RPL * NX; ++e) o.K[e] = qK[e];
#pragma unroll
    for (int r = 0; r < RPL; ++r) o.k[r] = qk[r];
    const T *qx = wxh + (int64_t)ii * NX;
#pragma unroll
    for (int j = 0; j < NX; ++j) o.xh[j] = qx[j];
    const T *qu = wuh + (int64_t)ii * NU;
#pragma unroll
    for (int r = 0; r < RPL; ++r) o.uh[r] = qu[r];
}

template <typename T, int NX, int NU, int MODEL, int RL, int WD, bool STAGE>
__device__ __forceinline__ void ro_replay(const Model<T, NX, NU, MODEL> &model, int c, int ind, bool valid, bool accept, bool stage_on,
                                          int N, int NSEG, int S, int64_t bN, T alpha_w, const T *pK, const T *pk, const T *pxh,
                                          const T *puh, const T *ck, T *stage, int uoff, T *ubuf, int gl, T *x_out, T *u_out)
{
    constexpr int RPL = NU / RL;                               // control rows per lane
    static_assert(RPL * RL == NU, "lanes per segment must divide the control dimension");
    using WOp = RoWOp<T, NX, RPL>;
    const int nl = NSEG * RL;
    const int cc = c < nl ? c : 0;
    const int sg = cc / RL, r0 = (cc - sg * RL) * RPL;
    const int t0 = sg * S, t1 = (t0 + S < N) ? t0 + S : N;
    T xw[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) xw[j] = ck[(ind * NSEG + sg) * NX + j];
    slot_sync();                                               // every checkpoint is read before the stage overwrites it
    // every lane runs the loop: lanes past the last segment repeat lane 0 (same loads, same LDS writes), so that no memory
    // instruction of the loop is conditional
    {
        const bool wl = valid;
        const T *wK = pK + (bN + t0) * NU * NX + r0 * NX, *wk = pk + (bN + t0) * NU + r0;
        const bool has_xh = pxh != nullptr, has_uh = puh != nullptr;      // uniform
        const T xhm = has_xh ? T(1) : T(0), uhm = has_uh ? T(1) : T(0);
        const T *wxh = has_xh ? pxh + (bN + t0) * NX : pK + bN * NU * NX, *wuh = has_uh ? puh + (bN + t0) * NU + r0 : pK + bN * NU * NX;
        const int last = N - 1 - t0;                           // iterations beyond it repeat step N-1 (loads only)
        WOp ring[WD];
#pragma unroll
        for (int d = 0; d < WD; ++d) {
            ro_wfetch<T, NX, NU, RPL>(ring[d], wK, wk, wxh, wuh, d, last);
            __builtin_amdgcn_sched_barrier(0);
        }
        T *xo = x_out + (bN + t0) * NX, *uo = u_out + (bN + t0) * NU + r0;   // direct stores when the stage does not fit
        // every lane has a dump word of its own (the slot's plain-cost words, free by now) for the writes it must not make:
        // LDS writes of several lanes to ONE address are a bank conflict of that many ways, and these sit on the step's
        // critical path
        T *const dumpw = ubuf + gl + (c < gl ? c : 0);
        const int sdump = (int)(dumpw - stage);
        T *ub = ubuf + sg * NU;
        T *const ubw = c < nl ? ub + r0 : dumpw;               // lanes past the last segment repeat lane 0's arithmetic, not its writes
        const int ubs = c < nl ? 1 : 0;
        for (int i0 = 0; i0 < S; i0 += WD) {
#pragma unroll
            for (int d = 0; d < WD; ++d) {
                const int i = i0 + d, t = t0 + i;
                const bool in = wl && i < S && t < t1;
                const WOp o = ring[d];
#ifndef ISLS_RO_EXP_NOREFILL
                ro_wfetch<T, NX, NU, RPL>(ring[d], wK, wk, wxh, wuh, i + WD, last);
#endif
                T uown[RPL], u[NU];
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    T acc = T(0);
#pragma unroll
                    for (int j = 0; j < NX; ++j) acc += fma(-xhm, o.xh[j], xw[j]) * o.K[r * NX + j];   // x - xhat, one rounding
                    uown[r] = fma(uhm, o.uh[r], acc + alpha_w * o.k[r]);                                // ... + uhat
                }
#ifdef ISLS_RO_EXP_NOXCHG
                if constexpr (false) {
#else
                if constexpr (RL > 1) {                        // the segment's lanes swap their rows of u
#endif
#pragma unroll
                    for (int r = 0; r < RPL; ++r) ubw[r * ubs] = uown[r];
                    slot_sync();
#pragma unroll
                    for (int r = 0; r < NU; ++r) u[r] = ub[r];
                    slot_sync();                               // read before the next iteration's rows land
                } else {
#pragma unroll
                    for (int r = 0; r < NU; ++r) u[r] = uown[r < RPL ? r : 0];
                }
#ifdef ISLS_RO_EXP_NOSTAGE
                if constexpr (false) {
#else
                if constexpr (STAGE) {
#endif
                    const bool own = in && c < nl;
                    const bool xown = own && (RL == 1 || r0 == 0);     // the first lane of a segment stores x_t, every lane its rows of u_t
                    T *sx = stage + (xown ? t * NX : sdump), *su = stage + (own ? uoff + t * NU + r0 : sdump);
                    const int sstx = xown ? 1 : 0, sstu = own ? 1 : 0;
#pragma unroll
                    for (int j = 0; j < NX; ++j) sx[j * sstx] = xw[j];
#pragma unroll
                    for (int r = 0; r < RPL; ++r) su[r * sstu] = uown[r];
                } else if (in && c < nl) {                     // long horizons: straight to HBM
                    if (RL == 1 || r0 == 0) {
#pragma unroll
                        for (int j = 0; j < NX; ++j) xo[i * NX + j] = accept ? xw[j] : xhm * o.xh[j];
                    }
#pragma unroll
                    for (int r = 0; r < RPL; ++r) uo[i * NU + r] = accept ? uown[r] : uhm * o.uh[r];
                }
                T xn[NX];
                model.step(xw, u, xn);
#pragma unroll
                for (int j = 0; j < NX; ++j) xw[j] = xn[j];
            }
        }
    }
}

// x_t of a recorded winner (its u_t sit in the stage's u rows): lane sg < NSEG steps the model through segment sg from the
// winner's checkpoint with the recorded controls -- LDS only, no operand of the search is fetched again.
template <typename T, int NX, int NU, int MODEL>
__device__ __forceinline__ void ro_xreplay(const Model<T, NX, NU, MODEL> &model, int c, int ind, int N, int NSEG, int S, const T *ck,
                                           T *stage, const T *ustage)
{
    const int sg = c < NSEG ? c : 0;
    const int t0 = sg * S, t1 = (t0 + S < N) ? t0 + S : N;
    T xw[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) xw[j] = ck[(ind * NSEG + sg) * NX + j];
    slot_sync();                                               // every checkpoint is read before x_t overwrites it
    const bool own = c < NSEG;
    for (int t = t0; t < t0 + S; ++t) {
        const int tt = t < N ? t : N - 1;
        T u[NU], xn[NX];
#pragma unroll
        for (int r = 0; r < NU; ++r) u[r] = ustage[tt * NU + r];
        if (own && t < t1) {
#pragma unroll
            for (int j = 0; j < NX; ++j) stage[t * NX + j] = xw[j];
        }
        model.step(xw, u, xn);
#pragma unroll
        for (int j = 0; j < NX; ++j) xw[j] = xn[j];
    }
}

#define ISLS_RO_LD(ptr) ro_load<T, W>(ptr)
template <typename T, int NX, int NU, int MODEL, int JM, int OCC>
__global__ __launch_bounds__(64, OCC) void rollout_kernel(RoP<T> p)
{
    using LY = RoLayout<NX, NU>;
    using Vec = RoVec<T, LY::W>;
    constexpr int D = kRolloutDepth, W = LY::W;
    constexpr int O_K = LY::O_K, O_XH = LY::O_XH, O_RX = LY::O_RX, O_WQ = LY::O_WQ, O_KK = LY::O_KK, O_UH = LY::O_UH,
                  O_RU = LY::O_RU, O_WR = LY::O_WR, REC = LY::REC, O_DUMP = LY::O_DUMP, RECP = LY::RECP;
    static_assert(D == 2, "the record double buffer is indexed by the ring position");
    extern __shared__ __align__(16) unsigned char ro_smem[];
    T *lds = reinterpret_cast<T *>(ro_smem);

    const int L = p.L, N = p.N, NSEG = p.nseg, S = p.seg_len;
    const int GL = L > 8 ? L : 8, TPW = p.tpw;
    const bool stage_on = p.stage_on != 0;
    constexpr int MDLW = Model<T, NX, NU, MODEL>::LDS_WORDS;
    const int SLOT = LY::slot_elems(L, GL, NSEG, N, stage_on, MDLW);
    const int lane = threadIdx.x;
    // lanes beyond TPW*GL join the last slot as extra idle candidate lanes (c >= GL): they help nobody and
    // write only dump words, but need no slot of their own
    const int s = (lane / GL < TPW) ? lane / GL : TPW - 1, c = lane - s * GL;
    const int b = blockIdx.x * TPW + s;
    const bool inbatch = b < p.B;
    const bool valid = inbatch && (p.active == nullptr || p.active[b] != 0);
    const bool cand = valid && c < L;
    const int bb = inbatch ? b : blockIdx.x * TPW;             // idle lanes shadow the block's first trajectory (loads only)
    const int64_t bN = (int64_t)bb * N;
    T *slot = lds + s * SLOT;
    T *c_aug = slot, *c_pln = c_aug + GL, *mdl = c_pln + GL;
    T *recs = static_cast<T *>(__builtin_assume_aligned(mdl + LY::mdl_elems(MDLW), W == 2 ? 2 * sizeof(T) : sizeof(T)));
    T *ck = recs + 2 * RECP;
    T *stage = recs;                                           // winner trajectory: x over the dead records + checkpoints,
    const int uoff = p.u_off;                                  // u behind them (recorded during the search)
    T *ustage = stage + uoff;
    const bool absolute = (p.flags & ISLS_RO_ABSOLUTE) != 0;
    const bool has_xh = !absolute && p.xhat != nullptr, has_uh = !absolute && p.uhat != nullptr;
    const bool has_wq = p.wq.p != nullptr, has_wr = p.wr.p != nullptr;
    const bool wq_var = has_wq && p.wq.st != 0, wr_var = has_wr && p.wr.st != 0;
    // (tested once per step each; forcing them into scalar registers with readfirstlane measured SLOWER: 73.1 vs 71.9 us per launch)
    // ---- load plan: pair q = c + GL*j of the slot's per-step operand list (groups in the order below; the two groups with
    // a second operand come first so that only load slot 0 carries one).  A pair is W adjacent words of one array; the
    // last pair of an odd group overlaps its predecessor by one word (same value written twice) so that no load ever
    // leaves its array.  Lanes without a pair load K[b,0] with stride 0 and write the dump words.
    const T *pa[JM], *pb0;
    int64_t stp[JM];                                           // step stride in bytes
    int dst[JM];                                               // record word of the pair (second word follows)
    T mb0 = T(0);
    pb0 = nullptr;
    static_for<JM>([&](auto J) {
        constexpr int j = decltype(J)::value;
        RoPlan<T> pl;
        pl.q = c < GL ? c + GL * j : (1 << 20);                // extra idle lanes stage nothing
        ro_plan_group<W>(pl, has_wq, p.zx + bN * NX, p.lx + bN * NX, NX, O_RX, NX);
        ro_plan_group<W>(pl, has_wr, p.zu + bN * NU, p.lu + bN * NU, NU, O_RU, NU);
        ro_plan_group<W>(pl, true, p.K + bN * NU * NX, (const T *)nullptr, NU * NX, O_K, NU * NX);
        ro_plan_group<W>(pl, true, p.k + bN * NU, (const T *)nullptr, NU, O_KK, NU);
        ro_plan_group<W>(pl, has_xh, p.xhat + bN * NX, (const T *)nullptr, NX, O_XH, NX);
        ro_plan_group<W>(pl, has_uh, p.uhat + bN * NU, (const T *)nullptr, NU, O_UH, NU);
        ro_plan_group<W>(pl, wq_var, wq_var ? p.wq.at(bb, 0) : (const T *)nullptr, (const T *)nullptr, NX, O_WQ, (int)p.wq.st);
        ro_plan_group<W>(pl, wr_var, wr_var ? p.wr.at(bb, 0) : (const T *)nullptr, (const T *)nullptr, NU, O_WR, (int)p.wr.st);
        pa[j] = pl.done ? pl.a : p.K + bN * NU * NX;
        stp[j] = pl.done ? (int64_t)pl.st * (int64_t)sizeof(T) : 0;
        dst[j] = pl.done ? pl.off : O_DUMP;
        if (j == 0) {
            const bool hb = pl.done && pl.b != nullptr;
            pb0 = hb ? pl.b : pa[0];
            mb0 = hb ? T(1) : T(0);
        }
    });
    // both records start as zeros (absent operands are never written); time-invariant AL weights go in once
    for (int e = c; e < 2 * RECP; e += GL) recs[e] = T(0);
    slot_sync();
    if (has_wq && !wq_var) {
        for (int e = c; e < NX; e += GL) { const T w = p.wq.at(bb, 0)[e]; recs[O_WQ + e] = w; recs[RECP + O_WQ + e] = w; }
    }
    if (has_wr && !wr_var) {
        for (int e = c; e < NU; e += GL) { const T w = p.wr.at(bb, 0)[e]; recs[O_WR + e] = w; recs[RECP + O_WR + e] = w; }
    }
    Model<T, NX, NU, MODEL> model;
    model.load(p.par + (int64_t)bb * p.par_sb, mdl, c < GL ? c : 0, GL);
    slot_sync();
    const T *Qtab = p.Qtab + (int64_t)bb * p.Qtab_sb, *ztab = p.ztab + (int64_t)bb * p.ztab_sb;
    const T ustd = p.u_std;
    // pseudo-Huber cost model: compiled into the kernels of the Tassa model only, selected at run time
    constexpr bool kHasPH = MODEL == ISLS_MODEL_TASSA;
    const bool phuber = kHasPH && p.cost_model == ISLS_COST_PHUBER;
    T ph_cu[NU], ph_cx[NX], ph_px[NX], ph_cf[NX], ph_pf[NX];
#pragma unroll
    for (int r = 0; r < NU; ++r) ph_cu[r] = phuber ? p.cpar[r] : T(0);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        ph_cx[j] = phuber ? p.cpar[NU + j] : T(0);
        ph_px[j] = phuber ? p.cpar[NU + NX + j] : T(1);
        ph_cf[j] = phuber ? p.cpar[NU + 2 * NX + j] : T(0);
        ph_pf[j] = phuber ? p.cpar[NU + 3 * NX + j] : T(1);
    }
    const T *x0p = p.x0 ? p.x0 + (int64_t)bb * NX : (p.xhat ? p.xhat + bN * NX : nullptr);

#ifdef ISLS_DIAG
    const unsigned long long tstart_ = __builtin_readcyclecounter();
#endif
    // kernel-argument fields used inside the step bodies are copied to locals first
    const int32_t *const seqp = p.seq, *const qnzp = p.qnz;
    // per-step "Q_t != 0" hints as ballot masks in SGPRs (a scalar load per step would sit on the critical path)
    const bool use_mask = qnzp != nullptr && N <= 256;
    unsigned long long qm0 = ~0ull, qm1 = ~0ull, qm2 = ~0ull, qm3 = ~0ull;
    if (use_mask) {
        qm0 = __ballot(lane < N && qnzp[lane < N ? lane : 0] != 0);
        qm1 = __ballot(lane + 64 < N && qnzp[lane + 64 < N ? lane + 64 : 0] != 0);
        qm2 = __ballot(lane + 128 < N && qnzp[lane + 128 < N ? lane + 128 : 0] != 0);
        qm3 = __ballot(lane + 192 < N && qnzp[lane + 192 < N ? lane + 192 : 0] != 0);
    }
    int next_q = use_mask ? ro_next_bit(qm0, qm1, qm2, qm3, 0) : 0;   // next step with a non-zero Q_t (256: none left)
    // ---- recording: the candidate that won this trajectory's previous line search drops its u_t into the stage's u rows as
    // the search goes; every other lane writes the same values to a dump word of its own (the slot's cost words, unused until
    // the arg-min): the stores are unconditional
    int pred = 0;
    if (stage_on && p.best) {
        pred = p.best[bb];
        pred = pred < 0 ? 0 : (pred < L ? pred : L - 1);
    }
    const bool is_pred = stage_on && c == pred;
    T *const udump = c_aug + (c < 2 * GL ? c : c % (2 * GL));  // the slot's 2 GL cost words (extra idle lanes of a narrow wavefront share)
    T *urw = is_pred ? ustage : udump;                         // running: u_t of the step the search is at
    const int ust = is_pred ? 1 : 0, uadv = is_pred ? NU : 0;
    // ================================ SEARCH ==========================================================
    const T alpha = absolute ? T(1) : p.alphas[c < L ? c : 0];
    T x[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = x0p ? x0p[j] : T(0);
    T cst = T(0), cu = T(0), ag = T(0);
    {
        // Ring of record pairs D steps ahead (raw loads; z - lambda is formed when staging).  Running source pointers:
        // the fetches walk t = 0, 1, 2, ..., so a pointer advances by its stride after every fetch until it sits on step
        // N-1 (the tail's padding fetches repeat that step).
        Vec ra[D][JM], rb[D];
        const char *ca[JM], *cb;
#pragma unroll
        for (int j = 0; j < JM; ++j) ca[j] = reinterpret_cast<const char *>(pa[j]);
        cb = reinterpret_cast<const char *>(pb0);
        const int64_t stpb = stp[0];
        int tf = 0;                                            // step the pointers sit on
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int j = 0; j < JM; ++j) ra[d][j] = ISLS_RO_LD(ca[j]);
            rb[d] = ISLS_RO_LD(cb);
            const int64_t adv = tf < N - 1 ? 1 : 0;
            tf += (int)adv;
#pragma unroll
            for (int j = 0; j < JM; ++j) ca[j] += adv * stp[j];
            cb += adv * stpb;
            __builtin_amdgcn_sched_barrier(0);                  // keep the issue order = consumption order (vmcnt is in-order)
        }
        T *ckc = ck + (c < L ? c : L) * NSEG * NX;             // this candidate's checkpoints (row L = dump)
        int next_ck = 0, seg = 0;

// One step of the search.  TAILF = false: steady state (the step and its refill are inside the horizon: no predicates, pointers
// always advance).  TAILF = true: the last steps (refills clamped to step N-1, dead padding steps leave the sums alone).
#define ISLS_RO_STEP(d, TAILF)                                                                                              \
    {                                                                                                                       \
        const int t = tb + d;                                                                                               \
        const bool live = TAILF ? (t < N) : true;                                                                           \
        T *rec = static_cast<T *>(__builtin_assume_aligned(recs + d * RECP, W == 2 ? 2 * sizeof(T) : sizeof(T)));           \
        {                                                                                                                   \
            T *w0 = rec + dst[0];                             /* slot 0: z - lambda where the pair has a second operand */  \
            _Pragma("unroll") for (int h = 0; h < W; ++h) w0[h] = fma(-mb0, rb[d].v[h], ra[d][0].v[h]);                     \
        }                                                                                                                   \
        _Pragma("unroll") for (int j = 1; j < JM; ++j) {                                                                    \
            T *wj = rec + dst[j];                                                                                           \
            _Pragma("unroll") for (int h = 0; h < W; ++h) wj[h] = ra[d][j].v[h];                                            \
        }                                                                                                                   \
        slot_sync();                                          /* record(t) visible to the slot */                           \
        if (!TAILF || t + D < N) {                            /* refill this ring entry with step t + D (the tail skips */  \
            _Pragma("unroll") for (int j = 0; j < JM; ++j)    /* fetches nobody consumes: the wave would wait for them)  */  \
                ra[d][j] = ISLS_RO_LD(ca[j]);                                                                               \
            rb[d] = ISLS_RO_LD(cb);                                                                                         \
            if (TAILF) {                                                                                                    \
                const int64_t adv = tf < N - 1 ? 1 : 0;                                                                     \
                tf += (int)adv;                                                                                             \
                _Pragma("unroll") for (int j = 0; j < JM; ++j) ca[j] += adv * stp[j];                                       \
                cb += adv * stpb;                                                                                           \
            } else {                                                                                                        \
                ++tf;                                                                                                       \
                _Pragma("unroll") for (int j = 0; j < JM; ++j) ca[j] += stp[j];                                             \
                cb += stpb;                                                                                                 \
            }                                                                                                               \
        }                                                                                                                   \
        if (t == next_ck && live) {                           /* uniform: state of every candidate at a segment start */    \
            _Pragma("unroll") for (int j = 0; j < NX; ++j) ckc[seg * NX + j] = x[j];                                        \
            ++seg;                                                                                                          \
            next_ck += S;                                                                                                   \
        }                                                                                                                   \
        /* u = (x - xhat) K' + alpha k + uhat            (isls.py:328-329) */                                               \
        T u[NU];                                                                                                            \
        T rru_h[LY::even(NU)], rwr_h[LY::even(NU)];           /* kHoistWr: the AL operands of the u block ride in the first batch */ \
        {                                                                                                                   \
            T rK[LY::even(NU * NX)], rxh[LY::even(NX)], rk[LY::even(NU)], ruh[LY::even(NU)];                                \
            ro_read<W>(rec + O_K, rK);                                                                                      \
            ro_read<W>(rec + O_XH, rxh);                                                                                    \
            ro_read<W>(rec + O_KK, rk);                                                                                     \
            ro_read<W>(rec + O_UH, ruh);                                                                                    \
            if constexpr (kHoistWr) {                                                                                       \
                ro_read<W>(rec + O_RU, rru_h);                                                                              \
                ro_read<W>(rec + O_WR, rwr_h);                                                                              \
            }                                                                                                               \
            _Pragma("unroll") for (int r = 0; r < NU; ++r) {                                                                \
                T acc = T(0);                                                                                               \
                _Pragma("unroll") for (int j = 0; j < NX; ++j) acc += (x[j] - rxh[j]) * rK[r * NX + j];                     \
                u[r] = (acc + alpha * rk[r]) + ruh[r];                                                                      \
            }                                                                                                               \
        }                                                                                                                   \
        {                                                     /* the predicted winner records u_t (dead steps: dump word) */ \
            T *uw = (!TAILF || live) ? urw : udump;                                                                         \
            const int us_ = (!TAILF || live) ? ust : 0;                                                                     \
            _Pragma("unroll") for (int r = 0; r < NU; ++r) uw[r * us_] = u[r];                                              \
            urw += uadv;                                                                                                    \
        }                                                                                                                   \
        T cst1 = cst, cu1 = cu, ag1 = ag;                                                                                   \
        bool nz = use_mask ? t == next_q : live;              /* (x-z)'Q(x-z), skipped where Q_t == 0 (uniform test) */     \
        if (use_mask && nz) next_q = ro_next_bit(qm0, qm1, qm2, qm3, t + 1);                                                \
        if (kHasPH && phuber) {                               /* sum_i cu_i u_i^2 + cx_i ph(x_i,px_i) (+ final term) */     \
            nz = false;                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < NX; ++j)                                                                  \
                cst1 += ph_cx[j] * (sqrt(x[j] * x[j] + ph_px[j] * ph_px[j]) - ph_px[j]);                                    \
            if (t == N - 1) {                                                                                               \
                _Pragma("unroll") for (int j = 0; j < NX; ++j)                                                              \
                    cst1 += ph_cf[j] * (sqrt(x[j] * x[j] + ph_pf[j] * ph_pf[j]) - ph_pf[j]);                                \
            }                                                                                                               \
            _Pragma("unroll") for (int r = 0; r < NU; ++r) cu1 += ph_cu[r] * (u[r] * u[r]);                                 \
        }                                                                                                                   \
        if (nz) {                                                                                                           \
            const int sq = seqp[t];                                                                                         \
            const T *Q = Qtab + (int64_t)sq * NX * NX, *z = ztab + (int64_t)sq * NX;                                        \
            T dq[NX];                                                                                                       \
            _Pragma("unroll") for (int j = 0; j < NX; ++j) dq[j] = x[j] - z[j];                                             \
            _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                                \
                T acc = T(0);                                                                                               \
                _Pragma("unroll") for (int j = 0; j < NX; ++j) acc += Q[i * NX + j] * dq[j];                                \
                cst1 += dq[i] * acc;                                                                                        \
            }                                                                                                               \
        }                                                                                                                   \
        if (!(kHasPH && phuber)) {                                                                                          \
            _Pragma("unroll") for (int r = 0; r < NU; ++r) cu1 += u[r] * (ustd * u[r]);                                     \
        }                                                                                                                   \
        if (has_wq) {                                                                                                       \
            T rrx[LY::even(NX)], rwq[LY::even(NX)];                                                                         \
            ro_read<W>(rec + O_RX, rrx);                                                                                    \
            ro_read<W>(rec + O_WQ, rwq);                                                                                    \
            _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                                \
                const T df = x[j] - rrx[j];                                                                                 \
                ag1 += (df * df) * rwq[j];                                                                                  \
            }                                                                                                               \
        }                                                                                                                   \
        if (has_wr) {                                                                                                       \
            T rru[LY::even(NU)], rwr[LY::even(NU)];                                                                         \
            if constexpr (!kHoistWr) {                                                                                      \
                ro_read<W>(rec + O_RU, rru);                                                                                \
                ro_read<W>(rec + O_WR, rwr);                                                                                \
            }                                                                                                               \
            _Pragma("unroll") for (int r = 0; r < NU; ++r) {                                                                \
                const T df = u[r] - (kHoistWr ? rru_h[r] : rru[r]);                                                         \
                ag1 += (df * df) * (kHoistWr ? rwr_h[r] : rwr[r]);                                                          \
            }                                                                                                               \
        }                                                                                                                   \
        if (TAILF) {                                          /* dead (padding) steps leave the sums alone */               \
            cst = live ? cst1 : cst;                                                                                        \
            cu = live ? cu1 : cu;                                                                                           \
            ag = live ? ag1 : ag;                                                                                           \
        } else {                                                                                                            \
            cst = cst1; cu = cu1; ag = ag1;                                                                                 \
        }                                                                                                                   \
        T xn[NX];                                                                                                           \
        model.step(x, u, xn);                                 /* x = f(x, u)   (isls.py:332) */                             \
        _Pragma("unroll") for (int j = 0; j < NX; ++j) x[j] = xn[j];                                                        \
    }

        int tb = 0;
        for (; tb + 2 * D <= N - 1; tb += D) {                 // steady state: every refill of the group is a new step < N-1
            ISLS_RO_STEP(0, false)
            ISLS_RO_STEP(1, false)
        }
        for (; tb < N; tb += D) {                              // at most 2D steps: clamped refills, padded with dead steps
            ISLS_RO_STEP(0, true)
            ISLS_RO_STEP(1, true)
        }
#undef ISLS_RO_STEP
    }

#ifdef ISLS_DIAG
    const unsigned long long tsearch_ = __builtin_readcyclecounter();
#endif
    // ================================ ARG-MIN =========================================================
    const T plain = cst + cu;                                  // sum over x, then += sum over u (sls_base.py:33-39)
    const T aug = plain + ag;
    if (c < GL) { c_aug[c] = aug; c_pln[c] = plain; }          // lanes L <= c < GL write words nobody reads
    slot_sync();
    // first arg-min with numpy's NaN semantics; optional costs[isnan] = 1e5 (isls.py:362)
    const bool nan_rule = (p.flags & ISLS_RO_NAN_TO_1E5) != 0;
    int ind = 0;
    bool nan_seen = false;
    T bestv = T(0), bestp = T(0);
    for (int l0 = 0; l0 < L; l0 += 4) {                          // four candidates per round trip to the LDS, select form
        T va[4], pa4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int l = l0 + q < L ? l0 + q : L - 1;
            va[q] = c_aug[l];
            pa4[q] = c_pln[l];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int l = l0 + q;
            T v = va[q], pl = pa4[q];
            const bool isn = v != v;
            nan_seen = nan_seen || (isn && l < L);
            v = (isn && nan_rule) ? T(1e5) : v;
            pl = (isn && nan_rule) ? T(1e5) : pl;
            // np.argmin: the first NaN wins, else the first minimum
            const bool take = l < L && (l == 0 || (!(bestv != bestv) && (v != v || v < bestv)));
            bestv = take ? v : bestv;
            bestp = take ? pl : bestp;
            ind = take ? l : ind;
        }
    }
    if (cand && p.cost_all) p.cost_all[(int64_t)b * L + c] = (aug != aug && nan_rule) ? T(1e5) : aug;
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

#ifdef ISLS_DIAG
    const unsigned long long targmin_ = __builtin_readcyclecounter();
#endif
    // ================================ WINNER ==========================================================
    // Recorded winner (every trajectory of the wavefront won by the candidate that recorded its controls): u_t is in the
    // stage already; x_t is produced on demand by ro_xreplay below.  Otherwise the whole wavefront replays: the lanes of
    // segment sg re-run steps [sg*S, min((sg+1)*S, N)) of candidate `ind` from its checkpoint, fetching their operands (K_t,
    // k_t, xhat_t, uhat_t) themselves several iterations ahead.  That loop has no exec-mask branch (steps past the end of
    // the last segment compute on clamped operands and write the dump word), so the loads of the next iterations stay in
    // flight behind the current one.  x_t, u_t go to the stage in LDS (or straight to HBM when the stage does not fit the slot;
    // a trajectory that hit its prediction gets the same values again).
#ifdef ISLS_RO_EXP_ALLMISS
    const bool any_miss = true;                                // A/B build: the winner is always replayed
#else
    const bool any_miss = !stage_on || __ballot(valid && ind != pred) != 0ull;   // uniform
#endif
    // x_t is needed now unless the fused ADMM update only sweeps u and further ADMM iterations follow (then only a trajectory
    // that stops in this update is written out: decided behind the sweep)
    const bool x_early = !p.fa_on || p.fa_last || p.fa_zx != nullptr;   // uniform
    bool x_ready = true;
#ifndef ISLS_RO_EXP_NOREPLAY
    if (any_miss) {
        const T alpha_w = absolute ? T(1) : p.alphas[ind];
        const T *pxh = has_xh ? p.xhat : nullptr, *puh = has_uh ? p.uhat : nullptr;
        if (p.seg_lanes > 1 && stage_on)
            ro_replay<T, NX, NU, MODEL, NU, 4, true>(model, c, ind, valid, accept, stage_on, N, NSEG, S, bN, alpha_w, p.K, p.k, pxh, puh, ck,
                                                     stage, uoff, c_aug, GL, p.x_out, p.u_out);
        else if (p.seg_lanes > 1)
            ro_replay<T, NX, NU, MODEL, NU, 4, false>(model, c, ind, valid, accept, stage_on, N, NSEG, S, bN, alpha_w, p.K, p.k, pxh, puh, ck,
                                                      stage, uoff, c_aug, GL, p.x_out, p.u_out);
        else if (stage_on)
            ro_replay<T, NX, NU, MODEL, 1, 2, true>(model, c, ind, valid, accept, stage_on, N, NSEG, S, bN, alpha_w, p.K, p.k, pxh, puh, ck,
                                                    stage, uoff, c_aug, GL, p.x_out, p.u_out);
        else
            ro_replay<T, NX, NU, MODEL, 1, 2, false>(model, c, ind, valid, accept, stage_on, N, NSEG, S, bN, alpha_w, p.K, p.k, pxh, puh, ck,
                                                     stage, uoff, c_aug, GL, p.x_out, p.u_out);
    } else if (x_early) {
        ro_xreplay<T, NX, NU, MODEL>(model, c, ind, N, NSEG, S, ck, stage, ustage);
    } else {
        x_ready = false;
    }
#endif
#ifdef ISLS_DIAG
    const unsigned long long twloop_ = __builtin_readcyclecounter();
#endif
    // the winner's trajectory (or the kept nominal when the acceptance test failed: isls.py:365-369) leaves in one coalesced
    // sweep -- behind the fused ADMM update when that may tell that nobody will read it (fa_last == 0 and the trajectory goes on)
    slot_sync();
    bool write_out = stage_on && valid && c < GL;
    bool fa_stop = false;
#define ISLS_RO_WRITE_OUT()                                                                  \
    if (write_out) {                                                                         \
        T *xo = p.x_out + bN * NX, *uo = p.u_out + bN * NU;                                  \
        if (accept) {                                                                        \
            for (int e = c; e < N * NX; e += GL) xo[e] = stage[e];                           \
            for (int e = c; e < N * NU; e += GL) uo[e] = ustage[e];                          \
        } else {                                                                             \
            for (int e = c; e < N * NX; e += GL) xo[e] = p.xhat[bN * NX + e];                \
            for (int e = c; e < N * NU; e += GL) uo[e] = p.uhat[bN * NU + e];                \
        }                                                                                    \
    }
    if (!p.fa_on || p.fa_last) {                               // uniform
        ISLS_RO_WRITE_OUT()
        write_out = false;
    }
#ifdef ISLS_DIAG
    const unsigned long long twinner_ = __builtin_readcyclecounter();
#endif
    if (p.fa_on) {
        // Fused ADMM update (isls_admm_update semantics, admm.py:43-85): the slot's lanes sweep the flat [N*d] blocks,
        // x / u from the stage, z / lambda / bounds coalesced from HBM -- independent iterations, no per-step round trip.
        T prim = T(0), dual = T(0);
        // kernel-argument fields are copied to locals before the static_for body captures them
        T *const fa_zx = p.fa_zx, *const fa_lx = p.fa_lx, *const fa_zu = p.fa_zu, *const fa_lu = p.fa_lu;
        const int fa_proj_x = p.fa_proj_x, fa_proj_u = p.fa_proj_u;
        const T fa_relax = p.fa_relax;
        const View<T> fa_xlo = p.fa_xlo, fa_xhi = p.fa_xhi, fa_ulo = p.fa_ulo, fa_uhi = p.fa_uhi;
        static_for<2>([&](auto BLK) {
            constexpr bool isx = decltype(BLK)::value == 0;
            constexpr int d = isx ? NX : NU;
            T *zz = isx ? fa_zx : fa_zu, *ll = isx ? fa_lx : fa_lu;
            if (zz == nullptr) return;                         // uniform
            const int cnt = N * d, proj = isx ? fa_proj_x : fa_proj_u;
            const T *src = isx ? stage : ustage;
            const T *lo_p = (isx ? fa_xlo : fa_ulo).at(b, 0), *hi_p = (isx ? fa_xhi : fa_uhi).at(b, 0);
            const int lo_st = (int)(isx ? fa_xlo : fa_ulo).st, hi_st = (int)(isx ? fa_xhi : fa_uhi).st;
            T p2 = T(0), d2 = T(0);
            if (valid && c < GL) {
                const int64_t o = bN * d;
                // chunks of SW elements per lane: all loads of a chunk are issued before its first store, so a lane pays one
                // HBM round trip per chunk instead of one per element (the stores to z / lambda keep the compiler from
                // hoisting the next element's loads); SW covers the whole block of the headline sizes
                constexpr int SW = ISLS_RO_SW;
                for (int e0 = c; e0 < cnt; e0 += GL * SW) {
                    T zp[SW], lv[SW], lo_v[SW], hi_v[SW];
#pragma unroll
                    for (int q = 0; q < SW; ++q) {
                        const int e = e0 + GL * q < cnt ? e0 + GL * q : cnt - 1;      // clamped (surplus results are dropped)
                        zp[q] = zz[o + e];
                        lv[q] = ll[o + e];
                        if (proj == ISLS_PROJ_BOX) {
                            const int t = e / d, i = e - t * d;
                            lo_v[q] = lo_p[t * lo_st + i];
                            hi_v[q] = hi_p[t * hi_st + i];
                        } else {
                            lo_v[q] = hi_v[q] = T(0);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < SW; ++q) {
                        const int e = e0 + GL * q;
                        if (e < cnt) {
                            const T xv = src[e];
                            const T arg = (fa_relax * xv + (T(1) - fa_relax) * zp[q]) + lv[q];
                            T zn = arg;
                            if (proj == ISLS_PROJ_BOX) {
                                zn = arg < lo_v[q] ? lo_v[q] : arg;
                                zn = zn > hi_v[q] ? hi_v[q] : zn;
                            }
                            const T rr = xv - zn;
                            ll[o + e] = lv[q] + rr;
                            zz[o + e] = zn;
                            p2 += rr * rr;
                            d2 += (zn - zp[q]) * (zn - zp[q]);
                        }
                    }
                }
            }
            if (c < GL) { c_aug[c] = p2; c_pln[c] = d2; }
            slot_sync();
            T sp = T(0), sd = T(0);
            for (int l = 0; l < GL; ++l) { sp += c_aug[l]; sd += c_pln[l]; }
            slot_sync();
            prim += sqrt(sp);
            dual += sqrt(sd);
        });
        {
            // every lane of the slot evaluates the stop rule (same sums): the lanes need it to decide about the write-out
            const T *prevr = p.fa_res_prev ? p.fa_res_prev + (int64_t)bb * 2 : nullptr;
            if (p.fa_active && prevr) {
                const T p0 = prevr[0], p1 = prevr[1];
                if (prim < p.fa_tol_abs && dual < p.fa_tol_abs) fa_stop = true;
                else {
                    const T pc = fabs(p0 - prim) / (p0 + T(1e-30));
                    const T dc = fabs(p1 - dual) / (p1 + T(1e-30));
                    fa_stop = pc < p.fa_tol_rel && dc < p.fa_tol_rel;
                }
            }
            slot_sync();                                       // res_prev is read by every lane before lane 0 overwrites it
        }
        if (valid && c == 0) {
            T *res = p.fa_res + (int64_t)b * 2;
            T *prev = p.fa_res_prev ? p.fa_res_prev + (int64_t)b * 2 : nullptr;
            if (fa_stop) p.fa_active[b] = 0;
            res[0] = prim; res[1] = dual;
            if (prev) { prev[0] = prim; prev[1] = dual; }
            if (p.fa_iters) p.fa_iters[b] += 1;
        }
        if (!fa_stop) write_out = false;                       // the trajectory goes on: its next x-step overwrites x_out / u_out
        if (!x_ready && __ballot(write_out) != 0ull) {         // a recorded winner stops here: its x_t after all (uniform)
            ro_xreplay<T, NX, NU, MODEL>(model, c, ind, N, NSEG, S, ck, stage, ustage);
            slot_sync();
        }
        ISLS_RO_WRITE_OUT()
    }
#undef ISLS_RO_WRITE_OUT
#ifdef ISLS_DIAG
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 700))
        printf("ro diag block %d: search %llu argmin %llu replay %llu writeout %llu sweep %llu cycles (N=%d nseg=%d S=%d JM=%d)\n",
               blockIdx.x, tsearch_ - tstart_, targmin_ - tsearch_, twloop_ - targmin_, twinner_ - twloop_,
               __builtin_readcyclecounter() - twinner_, N, NSEG, S, JM);
#endif
}

// Launch of one (n, m, model) family: geometry of the winner replay, LDS budget, load-slot count.  Instantiated once per
// family in rollout_<family>.hip; the dispatcher in rollout.hip picks the family.
template <typename T, int NX, int NU, int MODEL>
int launch_rollout_family(RoP<T> &p, const isls_rollout_args &a, hipStream_t s, bool want_fused, bool *stage_ok);

#define ISLS_ROLLOUT_FAMILY_DECL(NX_, NU_, MODEL_)                                                                                         \
    extern template int launch_rollout_family<double, NX_, NU_, MODEL_>(RoP<double> &, const isls_rollout_args &, hipStream_t, bool, bool *); \
    extern template int launch_rollout_family<float, NX_, NU_, MODEL_>(RoP<float> &, const isls_rollout_args &, hipStream_t, bool, bool *);

template <typename T, int NX, int NU, int MODEL>
int launch_rollout_family_impl(RoP<T> &p, const isls_rollout_args &a, hipStream_t s, bool want_fused, bool *stage_ok)
{
    using LY = RoLayout<NX, NU>;
    const int GL = a.L > 8 ? a.L : 8;
    int TPW = kWave / GL;
    // two waves per SIMD where a batch of 4096 launches more waves than SIMDs (slots of >= 16 lanes), the whole register
    // file for the 8-lane slots (8 trajectories per wave) and the 9-state models in 16-lane slots (4 per wave)
    const bool occ2 = GL >= 32 || (GL >= 16 && NX < 9);
    // winner replay geometry: NSEG segments of S steps; a segment is replayed by one lane or by NU lanes (one control row
    // each, ro_replay).  An iteration of the one-lane form is a scattered gather (~4x the time of the row form's), so the
    // row form wins unless it gets far fewer segments
    bool stage_on = true;
    constexpr int MDLW = Model<T, NX, NU, MODEL>::LDS_WORDS;
    auto smem_bytes = [&](int ns) { return (size_t)TPW * LY::slot_elems(a.L, GL, ns, a.N, stage_on, MDLW) * sizeof(T); };
    // the winner's trajectory is collected in LDS when that keeps the workgroup under 28 KB (>= 5 per CU) ...
    size_t lds_limit = 26 * 1024 + 512;                        // <= 26.5 KB per wavefront keeps 6 workgroups per CU
    if (smem_bytes(1) > 28 * 1024) {
        // ... longer horizons (the car's N = 200) get FEWER trajectories per wavefront instead, as long as every wavefront of
        // the launch is still resident (two per SIMD at most, the CU's 160 KB shared by its wavefronts): the stage is what
        // lets a recorded winner skip the replay.  Beyond that the trajectory is stored step by step.
        // (only where the element-wise ADMM update can then ride on the launch -- config 3: 10 x (181 + 47) -> 10 x 218 us;
        // without that the extra wavefronts cost more than the replay they save: config 4, 199 vs 190 us per launch)
        stage_on = false;
        const int full = TPW;
        for (int t2 = want_fused ? full - 1 : 0; t2 >= 1; --t2) {
            const int waves = (a.B + t2 - 1) / t2, per_cu = (waves + 255) / 256;
            if (waves > (occ2 ? 2048 : 1024)) break;
            const size_t lim = (size_t)(160 * 1024) / per_cu - 256;
            const size_t need = (size_t)t2 * LY::slot_elems(a.L, GL, 1, a.N, true, MDLW) * sizeof(T);
            if (need <= lim && need <= 60 * 1024) {
                TPW = t2;
                stage_on = true;
                lds_limit = lim < 60 * 1024 ? lim : 60 * 1024;
                break;
            }
        }
    }
    const int grid = (a.B + TPW - 1) / TPW;
    p.tpw = TPW;
    auto fit = [&](int ns) {
        if (ns > a.N) ns = a.N;
        if (ns > kMaxSeg) ns = kMaxSeg;
        while (ns > 1 && smem_bytes(ns) > lds_limit) --ns;
        return ns < 1 ? 1 : ns;
    };
    const int ns1 = fit(GL), nsr = NU > 1 ? fit(GL / NU) : 0;
    const int it1 = (a.N + ns1 - 1) / ns1, itr = nsr ? (a.N + nsr - 1) / nsr : (1 << 30);
    const bool rows = nsr >= 1 && itr < 4 * it1;
    const int nseg = rows ? nsr : ns1;
    p.seg_lanes = rows ? NU : 1;
    p.seg_len = (a.N + nseg - 1) / nseg;
    p.nseg = (a.N + p.seg_len - 1) / p.seg_len;                // drop empty trailing segments
    p.stage_on = stage_on ? 1 : 0;
    p.u_off = LY::u_off(a.L, p.nseg, a.N);
    if (stage_ok) *stage_ok = stage_on;
    if (!stage_on) p.fa_on = 0;                                 // the fused update reads x, u from the stage: the caller runs it as its own launch
    const size_t smem = smem_bytes(p.nseg);
    if (smem > 64 * 1024) return ISLS_ERR_UNSUPPORTED;
    const bool absolute = (a.flags & ISLS_RO_ABSOLUTE) != 0;
    const int pairs = LY::pairs_needed(!absolute && a.xhat, !absolute && a.uhat, a.wq.p != nullptr, a.wq.p && a.wq.st != 0,
                                       a.wr.p != nullptr, a.wr.p && a.wr.st != 0);
    const int jm = (pairs + GL - 1) / GL;
    // instantiated load-slot counts: 1, 2, 3 and the family's maximum for its narrowest slots of that occupancy class
    constexpr int J2MAX = (LY::MAXPAIRS + 15) / 16, J1MAX = (LY::MAXPAIRS + 7) / 8;
    if (occ2) {
        if (jm <= 1) hipLaunchKernelGGL((rollout_kernel<T, NX, NU, MODEL, 1, 2>), dim3(grid), dim3(64), smem, s, p);
        else if (jm <= 2) hipLaunchKernelGGL((rollout_kernel<T, NX, NU, MODEL, (J2MAX >= 2 ? 2 : 1), 2>), dim3(grid), dim3(64), smem, s, p);
        else if (jm <= J2MAX) hipLaunchKernelGGL((rollout_kernel<T, NX, NU, MODEL, J2MAX, 2>), dim3(grid), dim3(64), smem, s, p);
        else return ISLS_ERR_UNSUPPORTED;
    } else {
        if (jm <= 1) hipLaunchKernelGGL((rollout_kernel<T, NX, NU, MODEL, 1, 1>), dim3(grid), dim3(64), smem, s, p);
        else if (jm <= 2) hipLaunchKernelGGL((rollout_kernel<T, NX, NU, MODEL, (J1MAX >= 2 ? 2 : 1), 1>), dim3(grid), dim3(64), smem, s, p);
        else if (jm <= 3) hipLaunchKernelGGL((rollout_kernel<T, NX, NU, MODEL, (J1MAX >= 3 ? 3 : 1), 1>), dim3(grid), dim3(64), smem, s, p);
        else if (jm <= J1MAX) hipLaunchKernelGGL((rollout_kernel<T, NX, NU, MODEL, J1MAX, 1>), dim3(grid), dim3(64), smem, s, p);
        else return ISLS_ERR_UNSUPPORTED;
    }
    return check_launch();
}

#define ISLS_ROLLOUT_FAMILY_DEFINE(NX_, NU_, MODEL_)                                                                                    \
    template <>                                                                                                                          \
    int launch_rollout_family<double, NX_, NU_, MODEL_>(RoP<double> & p, const isls_rollout_args &a, hipStream_t s, bool f, bool *ok)    \
    {                                                                                                                                    \
        return launch_rollout_family_impl<double, NX_, NU_, MODEL_>(p, a, s, f, ok);                                                     \
    }                                                                                                                                    \
    template <>                                                                                                                          \
    int launch_rollout_family<float, NX_, NU_, MODEL_>(RoP<float> & p, const isls_rollout_args &a, hipStream_t s, bool f, bool *ok)      \
    {                                                                                                                                    \
        return launch_rollout_family_impl<float, NX_, NU_, MODEL_>(p, a, s, f, ok);                                                      \
    }

}  // namespace isls
