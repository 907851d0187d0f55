// rollout.hip -- forward line-search rollout + cost + arg-min + winner trajectory on gfx950.
//
// Reference semantics: iSLS.rollout_DP (isls/isls.py:310-334), iterate_once_dp candidates / NaN rule /
// arg-min / acceptance (isls.py:357-369), SLSBase.compute_cost (isls/sls_base.py:25-44), AL terms of
// the ilqr_admm line search (isls.py:471-477), SLSBase.get_trajectory_dp (sls_base.py:76-89).
//
// Mapping: one 64-lane wavefront per workgroup, cut into TPW = 64/GL slots (GL = max(L,8) lanes);
// slot s owns trajectory blockIdx.x*TPW+s and lane c of the slot owns line-search candidate c (its
// state x lives in registers for the whole horizon).
//   SEARCH  : the per-step operands shared by the candidates of a trajectory (K_t, k_t, xhat_t, uhat_t, the
//             ADMM targets z-lambda and the AL weights) are fetched by the slot's lanes one element each,
//             D steps ahead (register ring, unconditional loads), into a double-buffered LDS record and read
//             back as broadcasts.  Every S steps each candidate drops its state into an LDS checkpoint.
//   ARG-MIN : first minimum over the slot's candidates (numpy NaN semantics, optional NaN rule/accept test).
//   WINNER  : the reference returns x_noms[ind]; re-running the winner sequentially would cost another N
//             dependent steps, so the horizon is cut into NSEG segments of S steps and lane c < NSEG replays
//             segment c from the winner's checkpoint: N/NSEG dependent steps instead of N (the kernels are
//             bound by the latency of the per-step chain, not by throughput).
#include <type_traits>

#include "isls_common.hpp"

namespace isls {

template <typename T>
struct RoP {
    int B, N, L, flags, nseg, seg_len;
    int stage_on;                  // winner trajectory collected in LDS and written out in one sweep (it fits the slot)
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
    // z / dual update of the ADMM (isls_admm_update semantics, admm.hip) fused into the winner replay by the outer
    // driver: the winner lanes hold x_t, u_t in registers, so the update costs two reads and two writes per element and
    // saves a launch and a second pass over x, u
    int fa_on, fa_proj_x, fa_proj_u;
    T fa_relax, fa_tol_abs, fa_tol_rel;
    T *fa_zx, *fa_lx, *fa_zu, *fa_lu, *fa_res, *fa_res_prev;
    View<T> fa_xlo, fa_xhi, fa_ulo, fa_uhi;
    int32_t *fa_active, *fa_iters;
};

// ---- built-in forward models -------------------------------------------------
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
            ex += cos(c);
            ey += sin(c);
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
        xn[0] = x[0] + dt * x[3] * cos(x[2]);
        xn[1] = x[1] + dt * x[3] * sin(x[2]);
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
        const T sw = sin(u[0]) * f;
        const T b = (f * cos(u[0]) + d) - sqrt(d * d - sw * sw);
        xn[0] = x[0] + b * cos(x[2]);
        xn[1] = x[1] + b * sin(x[2]);
        xn[2] = x[2] + asin(sw / d);
        xn[3] = x[3] + u[1] * dt;
    }
};

#ifndef ISLS_RO_HOIST
#define ISLS_RO_HOIST 0   // reading the augmented-Lagrangian operands early: 0.59 -> 0.67 ms (register pressure), kept off
#endif
#ifndef ISLS_RO_WD
#define ISLS_RO_WD 2      // winner operand ring: 2 and 4 iterations ahead measure the same
#endif
#ifndef ISLS_RO_SW
#define ISLS_RO_SW 5
#endif
constexpr int kRolloutDepth = 2;   // steps of record elements in flight per lane (D = 2..5 run within 3 %: issue bound; 2 is leanest)
constexpr int kMaxSeg = 10;        // winner replay: at most this many segments

template <int NX, int NU>
struct RoLayout {
    // record (elements): K | xh | rx | wq | k | uh | ru | wr | dump
    static constexpr int O_K = 0, O_XH = O_K + NU * NX, O_RX = O_XH + NX, O_WQ = O_RX + NX, O_KK = O_WQ + NX,
                         O_UH = O_KK + NU, O_RU = O_UH + NU, O_WR = O_RU + NU, REC = O_WR + NU, O_DUMP = REC,
                         RECP = ((REC + 1) | 1);
    // slot (elements): aug[GL] | plain[GL] | model | 2 records | checkpoints [L+1][nseg][NX] (row L: dump for idle lanes)
    //                  | winner staging [nseg][WREC] + dump word
    // after the search the records and the checkpoints are dead: the winner's trajectory x [N][NX] | u [N][NU] is collected
    // there (the "stage") before it goes to HBM in one coalesced sweep
    static constexpr int MDL = NX * (NX + NU);                 // model words of the slot ([A B] of an LTI model)
    static constexpr int WREC = NU * NX + NU + NX + NU;        // K_t | k_t | xhat_t | uhat_t: what the winner replay reads per step
    __host__ __device__ static constexpr int ck_elems(int L, int nseg, int N, bool stage_on)
    {
        const int ckp = (L + 1) * nseg * NX + 1, stage = stage_on ? N * (NX + NU) - 2 * RECP + 1 : 0;
        return ckp > stage ? ckp : stage;
    }
    __host__ __device__ static constexpr int slot_elems(int L, int GL, int nseg, int N, bool stage_on)
    {
        return 2 * GL + MDL + 2 * RECP + ck_elems(L, nseg, N, stage_on) + nseg * WREC + 1;
    }
};

// register budget: two waves per SIMD where a batch of 4096 launches more waves than SIMDs (slots of >= 16 lanes), the whole
// file for the 8-lane slots (8 trajectories per wave) and for the 9-state models in 16-lane slots (4 per wave)
template <int NX, int GLMIN> constexpr int rollout_occupancy() { return (GLMIN >= 32 || (GLMIN >= 16 && NX < 9)) ? 2 : 1; }

template <typename T, int NX, int NU, int MODEL, int GLMIN>
__global__ __launch_bounds__(64, (rollout_occupancy<NX, GLMIN>())) void rollout_kernel(RoP<T> p)
{
    using LY = RoLayout<NX, NU>;
    constexpr int D = kRolloutDepth;
    constexpr int O_K = LY::O_K, O_XH = LY::O_XH, O_RX = LY::O_RX, O_WQ = LY::O_WQ, O_KK = LY::O_KK, O_UH = LY::O_UH,
                  O_RU = LY::O_RU, O_WR = LY::O_WR, REC = LY::REC, O_DUMP = LY::O_DUMP, RECP = LY::RECP;
    constexpr int JM = (REC + GLMIN - 1) / GLMIN;              // record elements per lane
    extern __shared__ __align__(16) unsigned char ro_smem[];
    T *lds = reinterpret_cast<T *>(ro_smem);

    const int L = p.L, N = p.N, NSEG = p.nseg, S = p.seg_len;
    const int GL = L > 8 ? L : 8, TPW = kWave / GL;
    const bool stage_on = p.stage_on != 0;
    const int SLOT = LY::slot_elems(L, GL, NSEG, N, stage_on);
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
    T *c_aug = slot, *c_pln = c_aug + GL, *mdl = c_pln + GL, *recs = mdl + LY::MDL, *ck = recs + 2 * RECP;
    T *stage = recs;                                           // winner trajectory, over the dead records + checkpoints
    const bool absolute = (p.flags & ISLS_RO_ABSOLUTE) != 0;
    const bool has_xh = !absolute && p.xhat != nullptr, has_uh = !absolute && p.uhat != nullptr;
    const bool has_wq = p.wq.p != nullptr, has_wr = p.wr.p != nullptr;

    // ---- per-lane load plan for the record elements e = c + GL*j : source word(s), step stride, LDS word ---
    // absent elements read K[b,0,0,0] with stride 0 (a valid word) and are zeroed when staged; staging is
    //   rec = ma * fma(-mb, b, a)   with 0/1 masks: exactly a - b (one rounding), a, or 0 -- two instructions per element
    // instead of a subtraction and four selects (the record of a non-finite K would be NaN rather than 0 in its absent
    // entries; every control is NaN then anyway)
    const T *pa[JM], *pb[JM];
    int stp[JM], dst[JM];
    T ma[JM], mb[JM];
    static_for<JM>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const int e = c < GL ? c + GL * j : REC;               // extra idle lanes stage nothing
        const T *a = nullptr, *bq = nullptr;
        int st = 0;
        if (e < O_XH) { a = p.K + bN * NU * NX + e; st = NU * NX; }
        else if (e < O_RX) { if (has_xh) { a = p.xhat + bN * NX + (e - O_XH); st = NX; } }
        else if (e < O_WQ) { if (has_wq) { a = p.zx + bN * NX + (e - O_RX); bq = p.lx + bN * NX + (e - O_RX); st = NX; } }
        else if (e < O_KK) { if (has_wq) { a = p.wq.at(bb, 0) + (e - O_WQ); st = (int)p.wq.st; } }
        else if (e < O_UH) { a = p.k + bN * NU + (e - O_KK); st = NU; }
        else if (e < O_RU) { if (has_uh) { a = p.uhat + bN * NU + (e - O_UH); st = NU; } }
        else if (e < O_WR) { if (has_wr) { a = p.zu + bN * NU + (e - O_RU); bq = p.lu + bN * NU + (e - O_RU); st = NU; } }
        else if (e < REC) { if (has_wr) { a = p.wr.at(bb, 0) + (e - O_WR); st = (int)p.wr.st; } }
        const bool has_a = a != nullptr, has_b = bq != nullptr;
        ma[j] = has_a ? T(1) : T(0);
        mb[j] = has_b ? T(1) : T(0);
        pa[j] = has_a ? a : p.K + bN * NU * NX;
        pb[j] = has_b ? bq : pa[j];
        stp[j] = has_a ? st : 0;
        dst[j] = (c < GL && e < REC) ? e : O_DUMP;
    });
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
    unsigned long long racc[6] = {0, 0, 0, 0, 0, 0};
#define RSTAMP_BEGIN unsigned long long tprev_ = __builtin_readcyclecounter();
#define RSTAMP(k) { unsigned long long t1_ = __builtin_readcyclecounter(); racc[k] += t1_ - tprev_; tprev_ = t1_; }
    const unsigned long long tstart_ = __builtin_readcyclecounter();
#else
#define RSTAMP_BEGIN
#define RSTAMP(k)
#endif
    // kernel-argument fields used inside the step lambdas are copied to locals first: a lambda that captures the
    // by-value argument struct by reference forces the whole struct into scratch memory
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
    // ================================ SEARCH ==========================================================
    const T alpha = absolute ? T(1) : p.alphas[c < L ? c : 0];
    T x[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = x0p ? x0p[j] : T(0);
    T cst = T(0), cu = T(0), ag = T(0);
    {
        // Ring of record elements D steps ahead (raw, unconditional loads; masks are applied when staging).  The
        // horizon is walked in groups of D steps with the ring index a compile-time constant; the last group is
        // padded with dead steps (t >= N: loads clamped, nothing accumulated or stored), so the body exists once,
        // stays free of closures (a step lambda keeps its captures in scratch memory, and a scratch access per
        // step drains the ring with s_waitcnt vmcnt(0)) and every VMEM instruction of the loop is unconditional.
        // running source pointers: the fetches walk t = 0, 1, 2, ... (each ring entry is refilled D steps ahead), so a
        // pointer advances by its stride after every fetch until it sits on step N-1 (the padding fetches repeat that step);
        // one 64-bit multiply-add per pointer instead of rebuilding base + t * stride for every load
        T ra[D][JM], rb[D][JM];
        const T *ca[JM], *cb[JM];
#pragma unroll
        for (int j = 0; j < JM; ++j) { ca[j] = pa[j]; cb[j] = pb[j]; }
        int tf = 0;                                            // step the pointers sit on
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int j = 0; j < JM; ++j) { ra[d][j] = *ca[j]; rb[d][j] = *cb[j]; }
            const int adv = tf < N - 1 ? 1 : 0;
            tf += adv;
#pragma unroll
            for (int j = 0; j < JM; ++j) { ca[j] += adv * stp[j]; cb[j] += adv * stp[j]; }
            __builtin_amdgcn_sched_barrier(0);                  // keep the issue order = consumption order (vmcnt is in-order)
        }
        T *ckc = ck + (c < L ? c : L) * NSEG * NX;             // this candidate's checkpoints (row L = dump)
        int next_ck = 0, seg = 0;
        for (int tb = 0; tb < N; tb += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int t = tb + d;
                const bool live = t < N;
                RSTAMP_BEGIN
                T *rec = recs + (t & 1) * RECP;
#pragma unroll
                for (int j = 0; j < JM; ++j)                   // unconditional ds_writes (dump word for surplus); z - lambda
                    rec[dst[j]] = ma[j] * fma(-mb[j], rb[d][j], ra[d][j]);
                slot_sync();                                   // record(t) visible to the slot
                RSTAMP(0)
                {                                              // refill this ring entry: step min(t + D, N-1), unconditional
#pragma unroll
                    for (int j = 0; j < JM; ++j) { ra[d][j] = *ca[j]; rb[d][j] = *cb[j]; }
                    const int adv = tf < N - 1 ? 1 : 0;
                    tf += adv;
#pragma unroll
                    for (int j = 0; j < JM; ++j) { ca[j] += adv * stp[j]; cb[j] += adv * stp[j]; }
                }
                RSTAMP(1)
                if (t == next_ck && live) {                    // uniform: state of every candidate at a segment start
#pragma unroll
                    for (int j = 0; j < NX; ++j) ckc[seg * NX + j] = x[j];
                    ++seg;
                    next_ck += S;
                }
                // augmented-Lagrangian operands of this step, read together with the gains: a second LDS round trip in the
                // middle of the step (the reads sat behind the has_wq / has_wr branches) cost ~100 cycles per step
                T al_ru[NU], al_wr[NU], al_rx[NX], al_wq[NX];
#if ISLS_RO_HOIST
#pragma unroll
                for (int r = 0; r < NU; ++r) { al_ru[r] = rec[O_RU + r]; al_wr[r] = rec[O_WR + r]; }
#endif
#if ISLS_RO_HOIST == 1
#pragma unroll
                for (int j = 0; j < NX; ++j) { al_rx[j] = rec[O_RX + j]; al_wq[j] = rec[O_WQ + j]; }
#endif
                // u = (x - xhat) K' + alpha k + uhat            (isls.py:328-329)
                T u[NU];
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    T acc = T(0);
#pragma unroll
                    for (int j = 0; j < NX; ++j) acc += (x[j] - rec[O_XH + j]) * rec[O_K + r * NX + j];
                    u[r] = (acc + alpha * rec[O_KK + r]) + rec[O_UH + r];
                }
                RSTAMP(2)
                T cst1 = cst, cu1 = cu, ag1 = ag;
                bool nz = live;                                // (x-z)'Q(x-z), skipped where Q_t == 0 (uniform test)
                if (use_mask) {
                    const int w = t >> 6;
                    const unsigned long long mm = w == 0 ? qm0 : (w == 1 ? qm1 : (w == 2 ? qm2 : qm3));
                    nz = live && ((mm >> (t & 63)) & 1ull) != 0;
                }
                if (kHasPH && phuber) {                        // sum_i cu_i u_i^2 + cx_i ph(x_i,px_i) (+ final term)
                    nz = false;
#pragma unroll
                    for (int j = 0; j < NX; ++j) cst1 += ph_cx[j] * (sqrt(x[j] * x[j] + ph_px[j] * ph_px[j]) - ph_px[j]);
                    if (t == N - 1) {
#pragma unroll
                        for (int j = 0; j < NX; ++j) cst1 += ph_cf[j] * (sqrt(x[j] * x[j] + ph_pf[j] * ph_pf[j]) - ph_pf[j]);
                    }
#pragma unroll
                    for (int r = 0; r < NU; ++r) cu1 += ph_cu[r] * (u[r] * u[r]);
                }
                if (nz) {
                    const int sq = seqp[t];
                    const T *Q = Qtab + (int64_t)sq * NX * NX, *z = ztab + (int64_t)sq * NX;
                    T dq[NX];
#pragma unroll
                    for (int j = 0; j < NX; ++j) dq[j] = x[j] - z[j];
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        T acc = T(0);
#pragma unroll
                        for (int j = 0; j < NX; ++j) acc += Q[i * NX + j] * dq[j];
                        cst1 += dq[i] * acc;
                    }
                }
                if (!(kHasPH && phuber)) {
#pragma unroll
                    for (int r = 0; r < NU; ++r) cu1 += u[r] * (ustd * u[r]);
                }
                if (has_wq) {
#pragma unroll
                    for (int j = 0; j < NX; ++j) {
#if ISLS_RO_HOIST != 1
                        al_rx[j] = rec[O_RX + j]; al_wq[j] = rec[O_WQ + j];
#endif
                        const T df = x[j] - al_rx[j]; ag1 += (df * df) * al_wq[j];
                    }
                }
                if (has_wr) {
#pragma unroll
                    for (int r = 0; r < NU; ++r) {
#if !ISLS_RO_HOIST
                        al_ru[r] = rec[O_RU + r]; al_wr[r] = rec[O_WR + r];
#endif
                        const T df = u[r] - al_ru[r]; ag1 += (df * df) * al_wr[r];
                    }
                }
                cst = live ? cst1 : cst;                       // dead (padding) steps leave the sums alone
                cu = live ? cu1 : cu;
                ag = live ? ag1 : ag;
                RSTAMP(3)
                T xn[NX];
                model.step(x, u, xn);                          // x = f(x, u)   (isls.py:332)
#pragma unroll
                for (int j = 0; j < NX; ++j) x[j] = xn[j];
                RSTAMP(4)
            }
        }
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

#ifdef ISLS_DIAG
    const unsigned long long targmin_ = __builtin_readcyclecounter();
    unsigned long long twplan_ = 0;
#endif
    // ================================ WINNER ==========================================================
    // lane c < NSEG replays steps [c*S, min((c+1)*S, N)) of candidate `ind` from its checkpoint and streams
    // x_t, u_t to HBM (or copies the kept nominal when the acceptance test failed)
    //
    // The replay's operands (K_t, k_t, xhat_t, uhat_t of NSEG different steps per iteration) are fetched by ALL lanes of the
    // slot, two iterations ahead, and handed over through LDS: fetched by the segment lanes themselves they were ~30
    // dependent scattered loads per step, one HBM round trip per step -- 45 % of the kernel's time (cycle stamps, B = 4096).
    {
        constexpr int WREC = LY::WREC, WJ = 8, WD = ISLS_RO_WD;         // WJ words per lane and iteration (NSEG * WREC <= WJ * GL), WD iterations
                                                               // in flight: an iteration is short (~500 cycles), HBM is ~3000 away
        T *wbuf = ck + LY::ck_elems(L, NSEG, N, stage_on);
        const int wtot = NSEG * WREC;
        const bool wlane = valid && c < NSEG;
        const int cs = c < NSEG ? c : 0;
        const T alpha_w = absolute ? T(1) : p.alphas[ind];
        const int t0 = cs * S, t1 = (t0 + S < N) ? t0 + S : N;
        T xw[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) xw[j] = ck[(ind * NSEG + cs) * NX + j];
        const T *wp[WJ];
        int wst[WJ], wmax[WJ], wdst[WJ];
        T wm[WJ];                                              // 0 for the words of an absent array (staged as zeros), else 1
        static_for<WJ>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int e = c + GL * j;
            const bool ok = c < GL && e < wtot;
            const int ee = ok ? e : 0;
            const int q = ee / WREC, w = ee - q * WREC;
            const int64_t o = bN + (int64_t)q * S;              // first step of segment q
            const T *ptr = p.K + bN * NU * NX;                 // absent arrays: a valid word, stride 0, masked to zero
            int st = 0;
            bool present = true;
            if (w < NU * NX) { ptr = p.K + o * NU * NX + w; st = NU * NX; }
            else if (w < NU * NX + NU) { ptr = p.k + o * NU + (w - NU * NX); st = NU; }
            else if (w < NU * NX + NU + NX) { if (has_xh) { ptr = p.xhat + o * NX + (w - NU * NX - NU); st = NX; } else present = false; }
            else if (has_uh) { ptr = p.uhat + o * NU + (w - NU * NX - NU - NX); st = NU; }
            else present = false;
            wp[j] = ptr; wst[j] = st;
            wm[j] = present ? T(1) : T(0);
            wmax[j] = N - 1 - q * S;                           // last iteration whose step is inside the horizon
            wdst[j] = ok ? e : wtot;                           // surplus lanes / elements: dump word
        });
#ifdef ISLS_DIAG
        twplan_ = __builtin_readcyclecounter();
#endif
        T wr[WD][WJ];
#pragma unroll
        for (int d = 0; d < WD; ++d) {
#pragma unroll
            for (int j = 0; j < WJ; ++j) wr[d][j] = wp[j][(int64_t)(d < wmax[j] ? d : wmax[j]) * wst[j]];
            __builtin_amdgcn_sched_barrier(0);
        }
        const T *wrec = wbuf + cs * WREC;
        T *xo = p.x_out + (bN + t0) * NX, *uo = p.u_out + (bN + t0) * NU;     // direct stores when the stage does not fit
        const int uoff = p.fa_zx ? N * NX : 0;
        for (int i0 = 0; i0 < S; i0 += WD) {
#pragma unroll
            for (int d = 0; d < WD; ++d) {
                const int i = i0 + d, t = t0 + i;
#pragma unroll
                for (int j = 0; j < WJ; ++j) wbuf[wdst[j]] = wm[j] * wr[d][j];
                slot_sync();
#pragma unroll
                for (int j = 0; j < WJ; ++j) wr[d][j] = wp[j][(int64_t)(i + WD < wmax[j] ? i + WD : wmax[j]) * wst[j]];
                if (wlane && i < S && t < t1) {
                    T u[NU], xh[NX], uh[NU];                   // unconditional reads: a select per word became a branch per word
#pragma unroll
                    for (int j = 0; j < NX; ++j) xh[j] = wrec[NU * NX + NU + j];
#pragma unroll
                    for (int r = 0; r < NU; ++r) uh[r] = wrec[NU * NX + NU + NX + r];
#pragma unroll
                    for (int r = 0; r < NU; ++r) {
                        T acc = T(0);
#pragma unroll
                        for (int j = 0; j < NX; ++j) acc += (xw[j] - xh[j]) * wrec[r * NX + j];
                        u[r] = (acc + alpha_w * wrec[NU * NX + r]) + uh[r];
                    }
                    // x_t, u_t go to the stage in LDS: a global store in this divergent block would make the wait for the
                    // prefetched operands conservative (it then also waits for the previous iteration's write acknowledgements)
                    if (stage_on) {
#pragma unroll
                        for (int j = 0; j < NX; ++j) stage[t * NX + j] = xw[j];
#pragma unroll
                        for (int r = 0; r < NU; ++r) stage[N * NX + t * NU + r] = u[r];
                    } else {                                   // long horizons: straight to HBM (and to the fused sweep's LDS area)
#pragma unroll
                        for (int j = 0; j < NX; ++j) xo[i * NX + j] = accept ? xw[j] : wrec[NU * NX + NU + j];
#pragma unroll
                        for (int r = 0; r < NU; ++r) uo[i * NU + r] = accept ? u[r] : wrec[NU * NX + NU + NX + r];
                        if (p.fa_on && p.fa_zx) {
#pragma unroll
                            for (int j = 0; j < NX; ++j) ck[t * NX + j] = xw[j];
                        }
                        if (p.fa_on && p.fa_zu) {
#pragma unroll
                            for (int r = 0; r < NU; ++r) ck[uoff + t * NU + r] = u[r];
                        }
                    }
                    T xn[NX];
                    model.step(xw, u, xn);
#pragma unroll
                    for (int j = 0; j < NX; ++j) xw[j] = xn[j];
                }
            }
        }
    }
#ifdef ISLS_DIAG
    const unsigned long long twloop_ = __builtin_readcyclecounter();
#endif
    // the winner's trajectory (or the kept nominal when the acceptance test failed: isls.py:365-369) leaves in one coalesced sweep
    slot_sync();
    if (stage_on && valid && c < GL) {
        T *xo = p.x_out + bN * NX, *uo = p.u_out + bN * NU;
        if (accept) {
            for (int e = c; e < N * NX; e += GL) xo[e] = stage[e];
            for (int e = c; e < N * NU; e += GL) uo[e] = stage[N * NX + e];
        } else {
            for (int e = c; e < N * NX; e += GL) xo[e] = p.xhat[bN * NX + e];
            for (int e = c; e < N * NU; e += GL) uo[e] = p.uhat[bN * NU + e];
        }
    }
#ifdef ISLS_DIAG
    const unsigned long long twinner_ = __builtin_readcyclecounter();
#endif
    if (p.fa_on) {
        // Fused ADMM update (isls_admm_update semantics, admm.py:43-85): the slot's lanes sweep the flat [N*d] blocks,
        // x / u from LDS, z / lambda / bounds coalesced from HBM -- independent iterations, no per-step round trip.
        slot_sync();
        T prim = T(0), dual = T(0);
        for (int blk = 0; blk < 2; ++blk) {
            const bool isx = blk == 0;
            T *zz = isx ? p.fa_zx : p.fa_zu, *ll = isx ? p.fa_lx : p.fa_lu;
            if (zz == nullptr) continue;                       // uniform
            const int d = isx ? NX : NU, cnt = N * d, proj = isx ? p.fa_proj_x : p.fa_proj_u;
            const T *src = stage_on ? (isx ? stage : stage + N * NX) : (isx ? ck : ck + (p.fa_zx ? N * NX : 0));
            const View<T> &lo = isx ? p.fa_xlo : p.fa_ulo, &hi = isx ? p.fa_xhi : p.fa_uhi;
            T p2 = T(0), d2 = T(0);
            if (valid && c < GL) {
                const int64_t o = bN * d;
                // chunks of SW elements per lane: all loads of a chunk are issued before its first store, so a lane pays one
                // HBM round trip per chunk instead of one per element (the stores to z / lambda kept the compiler from
                // hoisting the next element's loads)
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
                            lo_v[q] = lo.at(b, t)[i];
                            hi_v[q] = hi.at(b, t)[i];
                        } else {
                            lo_v[q] = hi_v[q] = T(0);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < SW; ++q) {
                        const int e = e0 + GL * q;
                        if (e < cnt) {
                            const T xv = src[e];
                            const T arg = (p.fa_relax * xv + (T(1) - p.fa_relax) * zp[q]) + lv[q];
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
        }
        if (valid && c == 0) {
            T *res = p.fa_res + (int64_t)b * 2;
            T *prev = p.fa_res_prev ? p.fa_res_prev + (int64_t)b * 2 : nullptr;
            if (p.fa_active && prev) {
                bool stop = false;
                if (prim < p.fa_tol_abs && dual < p.fa_tol_abs) stop = true;
                else {
                    const T pc = fabs(prev[0] - prim) / (prev[0] + T(1e-30));
                    const T dc = fabs(prev[1] - dual) / (prev[1] + T(1e-30));
                    stop = pc < p.fa_tol_rel && dc < p.fa_tol_rel;
                }
                if (stop) p.fa_active[b] = 0;
            }
            res[0] = prim; res[1] = dual;
            if (prev) { prev[0] = prim; prev[1] = dual; }
            if (p.fa_iters) p.fa_iters[b] += 1;
        }
    }
#ifdef ISLS_DIAG
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 700))
        printf("ro diag block %d: put+sync %llu fetch %llu u %llu cost %llu model %llu | search %llu argmin %llu winner %llu (plan done %llu, loop done %llu) sweep %llu cycles (N=%d nseg=%d)\n",
               blockIdx.x, racc[0], racc[1], racc[2], racc[3], racc[4], tsearch_ - tstart_, targmin_ - tsearch_, twinner_ - targmin_, twplan_ - targmin_, twloop_ - targmin_,
               __builtin_readcyclecounter() - twinner_, N, NSEG);
#endif
}

template <typename T>
int launch_rollout(const isls_rollout_args &a, hipStream_t s, const isls_admm_args *fused, bool *did_fuse)
{
    if (did_fuse) *did_fuse = false;
    if (a.B < 0 || a.N < 1 || a.L < 1 || a.L > 64) return ISLS_ERR_ARG;
    if (!a.model_par || !a.K || !a.k || !a.alphas || !a.x_out || !a.u_out) return ISLS_ERR_ARG;
    if (a.cost_model == ISLS_COST_VIA && (!a.Qtab || !a.ztab || !a.seq)) return ISLS_ERR_ARG;
    if (!(a.flags & ISLS_RO_ABSOLUTE) && (!a.xhat || !a.uhat)) return ISLS_ERR_ARG;
    if ((a.flags & ISLS_RO_ABSOLUTE) && !a.x0) return ISLS_ERR_ARG;
    if ((a.flags & ISLS_RO_ACCEPT_TEST) && !a.cost_cur) return ISLS_ERR_ARG;
    if (a.wq.p && (!a.zx || !a.lx)) return ISLS_ERR_ARG;
    if (a.wr.p && (!a.zu || !a.lu)) return ISLS_ERR_ARG;
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
    p.cost_model = a.cost_model; p.cpar = (const T *)a.cost_par;
    if (a.cost_model != ISLS_COST_VIA && (a.cost_model != ISLS_COST_PHUBER || a.model != ISLS_MODEL_TASSA || !a.cost_par))
        return ISLS_ERR_UNSUPPORTED;
    const int GL = a.L > 8 ? a.L : 8, TPW = kWave / GL;
    const int grid = (a.B + TPW - 1) / TPW;
    // winner replay geometry: NSEG segments of S steps, NSEG <= lanes of a slot
    int nseg = GL < kMaxSeg ? GL : kMaxSeg;
    if (nseg > a.N) nseg = a.N;
    bool stage_on = true;
    auto smem_bytes = [&](int ns) {
        const int slot = (a.n == 6 ? RoLayout<6, 3>::slot_elems(a.L, GL, ns, a.N, stage_on) : a.n == 2 ? RoLayout<2, 1>::slot_elems(a.L, GL, ns, a.N, stage_on)
                          : a.n == 4 ? RoLayout<4, 2>::slot_elems(a.L, GL, ns, a.N, stage_on) : RoLayout<9, 3>::slot_elems(a.L, GL, ns, a.N, stage_on));
        return (size_t)TPW * slot * sizeof(T);
    };
    // the winner's trajectory is collected in LDS when that keeps the workgroup under 28 KB (>= 5 per CU); longer horizons
    // store it step by step
    if (smem_bytes(1) > 28 * 1024) stage_on = false;
    while (nseg > 1 && smem_bytes(nseg) > 26 * 1024 + 512) --nseg;   // <= 26.5 KB per wavefront keeps 6 workgroups per CU
    while (nseg > 1 && nseg * (a.m * a.n + 2 * a.m + a.n) > 8 * GL) --nseg;    // winner staging: <= 8 words per lane and iteration
    p.seg_len = (a.N + nseg - 1) / nseg;
    p.nseg = (a.N + p.seg_len - 1) / p.seg_len;                // drop empty trailing segments
    p.stage_on = stage_on ? 1 : 0;
    p.fa_on = 0;
    p.fa_zx = p.fa_lx = p.fa_zu = p.fa_lu = p.fa_res = p.fa_res_prev = nullptr;
    p.fa_active = p.fa_iters = nullptr;
    p.fa_proj_x = p.fa_proj_u = 0;
    p.fa_relax = p.fa_tol_abs = p.fa_tol_rel = T(0);
    const size_t smem = smem_bytes(p.nseg);
    if (smem > 64 * 1024) return ISLS_ERR_UNSUPPORTED;
    // without the stage the fused ADMM sweep takes x_t, u_t through the checkpoint area: [N][n] + [N][m] words must fit
    if (fused && !stage_on && (int64_t)a.N * ((fused->zx ? a.n : 0) + (fused->zu ? a.m : 0)) > (int64_t)(a.L + 1) * p.nseg * a.n) fused = nullptr;
    if (fused) {                                               // validated by the caller (rollout_can_fuse_admm)
        if (did_fuse) *did_fuse = true;
        const isls_admm_args &f = *fused;
        p.fa_on = 1; p.fa_proj_x = f.proj_x; p.fa_proj_u = f.proj_u;
        p.fa_relax = (T)f.relax; p.fa_tol_abs = (T)f.tol_abs; p.fa_tol_rel = (T)f.tol_rel;
        p.fa_zx = (T *)f.zx; p.fa_lx = (T *)f.lx; p.fa_zu = (T *)f.zu; p.fa_lu = (T *)f.lu;
        p.fa_xlo = View<T>(f.x_lo); p.fa_xhi = View<T>(f.x_hi); p.fa_ulo = View<T>(f.u_lo); p.fa_uhi = View<T>(f.u_hi);
        p.fa_res = (T *)f.res; p.fa_res_prev = (T *)f.res_prev; p.fa_active = f.active; p.fa_iters = f.iters;
    }
#define LAUNCH_G(NX_, NU_, MODEL_, G_) \
    hipLaunchKernelGGL((rollout_kernel<T, NX_, NU_, MODEL_, G_>), dim3(grid), dim3(64), smem, s, p)
#define LAUNCH(NX_, NU_, MODEL_)                                     \
    {                                                                \
        if (GL >= 32) LAUNCH_G(NX_, NU_, MODEL_, 32);                \
        else if (GL >= 16) LAUNCH_G(NX_, NU_, MODEL_, 16);           \
        else LAUNCH_G(NX_, NU_, MODEL_, 8);                          \
    }
    if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_LTI) LAUNCH(4, 2, ISLS_MODEL_LTI)
    else if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_CAR) LAUNCH(4, 2, ISLS_MODEL_CAR)
    else if (a.n == 9 && a.m == 3 && a.model == ISLS_MODEL_LTI) LAUNCH(9, 3, ISLS_MODEL_LTI)
    else if (a.n == 9 && a.m == 3 && a.model == ISLS_MODEL_ARM3R) LAUNCH(9, 3, ISLS_MODEL_ARM3R)
    else if (a.n == 6 && a.m == 3 && a.model == ISLS_MODEL_LTI) LAUNCH(6, 3, ISLS_MODEL_LTI)
    else if (a.n == 2 && a.m == 1 && a.model == ISLS_MODEL_LTI) LAUNCH(2, 1, ISLS_MODEL_LTI)
    else if (a.n == 6 && a.m == 3 && a.model == ISLS_MODEL_DI) LAUNCH(6, 3, ISLS_MODEL_DI)
    else if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_DI) LAUNCH(4, 2, ISLS_MODEL_DI)
    else if (a.n == 2 && a.m == 1 && a.model == ISLS_MODEL_DI) LAUNCH(2, 1, ISLS_MODEL_DI)
    else if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_TASSA) LAUNCH(4, 2, ISLS_MODEL_TASSA)
    else return ISLS_ERR_UNSUPPORTED;
#undef LAUNCH
#undef LAUNCH_G
    return check_launch();
}
template int launch_rollout<double>(const isls_rollout_args &, hipStream_t, const isls_admm_args *, bool *);
template int launch_rollout<float>(const isls_rollout_args &, hipStream_t, const isls_admm_args *, bool *);

// ---------------------------------------------------------------------------------------------------------------------
// Monte-Carlo closed loop of a dense causal controller about a nominal (iSLSBase.get_trajectory_sls,
// isls/isls_base.py:28-42): one thread per initial state, the state history is the thread's own x_log row.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
struct DenseLoopP {
    int M, N;
    const T *par, *K, *k, *xhat, *uhat, *x0;
    T *x_log, *u_log;
};

template <typename T, int NX, int NU, int MODEL>
__global__ __launch_bounds__(64) void dense_closed_loop_kernel(DenseLoopP<T> p)
{
    extern __shared__ __align__(16) unsigned char dl_smem[];
    Model<T, NX, NU, MODEL> mdl;
    mdl.load(p.par, reinterpret_cast<T *>(dl_smem), threadIdx.x, 64);
    __syncthreads();
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= p.M) return;
    const int N = p.N;
    T *xs = p.x_log + (int64_t)s * N * NX, *us = p.u_log + (int64_t)s * N * NU;
    T x[NX], u[NU], xn[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = p.x0[(int64_t)s * NX + j];
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < NX; ++j) xs[i * NX + j] = x[j];
#pragma unroll
        for (int r = 0; r < NU; ++r) {
            const T *Kr = p.K + (int64_t)(i * NU + r) * N * NX;
            T acc = T(0);
            for (int j = 0; j < (i + 1) * NX; ++j) acc += (xs[j] - (p.xhat ? p.xhat[j] : T(0))) * Kr[j];
            u[r] = (acc + p.k[i * NU + r]) + (p.uhat ? p.uhat[i * NU + r] : T(0));
            us[i * NU + r] = u[r];
        }
        mdl.step(x, u, xn);
#pragma unroll
        for (int j = 0; j < NX; ++j) x[j] = xn[j];
    }
}

template <typename T>
int launch_dense_closed_loop(const isls_dense_loop_args &a, hipStream_t s)
{
    if (a.M < 0 || a.N < 1 || !a.model_par || !a.K || !a.k || !a.x0 || !a.x_log || !a.u_log) return ISLS_ERR_ARG;
    if (a.M == 0) return ISLS_OK;
    DenseLoopP<T> p;
    p.M = a.M; p.N = a.N;
    p.par = (const T *)a.model_par; p.K = (const T *)a.K; p.k = (const T *)a.k;
    p.xhat = (const T *)a.xhat; p.uhat = (const T *)a.uhat; p.x0 = (const T *)a.x0;
    p.x_log = (T *)a.x_log; p.u_log = (T *)a.u_log;
    const int grid = (a.M + 63) / 64;
#define LAUNCH(NX_, NU_, MODEL_)                                                                                          \
    hipLaunchKernelGGL((dense_closed_loop_kernel<T, NX_, NU_, MODEL_>), dim3(grid), dim3(64),                              \
                       sizeof(T) * (Model<T, NX_, NU_, MODEL_>::LDS_WORDS + 1), s, p)
    if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_LTI) LAUNCH(4, 2, ISLS_MODEL_LTI);
    else if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_CAR) LAUNCH(4, 2, ISLS_MODEL_CAR);
    else if (a.n == 9 && a.m == 3 && a.model == ISLS_MODEL_LTI) LAUNCH(9, 3, ISLS_MODEL_LTI);
    else if (a.n == 9 && a.m == 3 && a.model == ISLS_MODEL_ARM3R) LAUNCH(9, 3, ISLS_MODEL_ARM3R);
    else if (a.n == 6 && a.m == 3 && a.model == ISLS_MODEL_LTI) LAUNCH(6, 3, ISLS_MODEL_LTI);
    else if (a.n == 2 && a.m == 1 && a.model == ISLS_MODEL_LTI) LAUNCH(2, 1, ISLS_MODEL_LTI);
    else if (a.n == 6 && a.m == 3 && a.model == ISLS_MODEL_DI) LAUNCH(6, 3, ISLS_MODEL_DI);
    else if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_DI) LAUNCH(4, 2, ISLS_MODEL_DI);
    else if (a.n == 2 && a.m == 1 && a.model == ISLS_MODEL_DI) LAUNCH(2, 1, ISLS_MODEL_DI);
    else if (a.n == 4 && a.m == 2 && a.model == ISLS_MODEL_TASSA) LAUNCH(4, 2, ISLS_MODEL_TASSA);
    else return ISLS_ERR_UNSUPPORTED;
#undef LAUNCH
    return check_launch();
}
template int launch_dense_closed_loop<double>(const isls_dense_loop_args &, hipStream_t);
template int launch_dense_closed_loop<float>(const isls_dense_loop_args &, hipStream_t);

// The ADMM update can ride on the winner replay when it is the plain element-wise form (no set projections), works
// on the arrays this rollout writes, shares its active mask and no acceptance test can keep the old nominal.
bool rollout_can_fuse_admm(const isls_rollout_args &r, const isls_admm_args &a)
{
    if (r.flags & (ISLS_RO_ACCEPT_TEST | ISLS_RO_ABSOLUTE)) return false;
    if (a.B != r.B || a.N != r.N || a.n != r.n || a.m != r.m || !a.res) return false;
    if (a.xx != r.x_out || a.xu != r.u_out || a.active != r.active) return false;
    if ((a.zx && a.proj_x == ISLS_PROJ_SETS) || (a.zu && a.proj_u == ISLS_PROJ_SETS)) return false;
    if (a.zx && (!a.lx || (a.proj_x == ISLS_PROJ_BOX && (!a.x_lo.p || !a.x_hi.p)))) return false;
    if (a.zu && (!a.lu || (a.proj_u == ISLS_PROJ_BOX && (!a.u_lo.p || !a.u_hi.p)))) return false;
    return true;
}

}  // namespace isls
