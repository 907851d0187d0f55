// riccati.hip -- backward Riccati pass on gfx950: the gain pass (the feed-forward pass is riccati_ff.hip).
//
// Reference semantics: the K / V half of iSLS.backward_pass_DP (isls/isls.py:229-308) and of SLS.solve_dp
// (isls/sls.py:85-166).  See include/isls_hip.h for the exact formulas and array formats.
//
// Mapping: one 64-lane wavefront per workgroup, cut into TPW = 64/(n+m) slots of
// G = n+m lanes; slot s owns trajectory blockIdx.x*TPW + s.  Lane i of a slot owns ROW i of the
// stacked (n+m) x (n+m) matrix [Qxx Qxu; Qux Quu] = C + [A B]' V [A B]:
//     S_i  = sum_k [A B][k,i] * V[k,:]          (row i of [A B]'V,   n FMAs x n)
//     M_i  = sum_k S_i[k]   * [A B][k,:]        (row i of the stack, n FMAs x (n+m))
// so lanes i<n end up with a row of Qxx, lanes i>=n with a row of [Qux Quu].  V, [A B] and the small
// factors live in the slot's LDS record and are read back as broadcasts (all lanes of a slot read the
// same word) or as the lane's own column.
//
// The recursion is sequential in t, so HBM latency cannot be hidden by the dependent chain itself:
// every lane keeps a RING of D steps of operands in flight in registers (loads for step t-D are issued
// while step t is computed; the compiler's counted s_waitcnt vmcnt lets the oldest entry land while the
// younger ones are still travelling).
#include "isls_common.hpp"

// Ablation switch (wrong results by design; tools/ab_build.sh + tools/kbench.py): -DISLS_GAIN_EXP_NOFLUSH
// drops the stores of the record image and of K -- the "gain pass without its stores" figure.

namespace isls {

constexpr int kGainDepth = 2;   // steps of A,B,C in flight per lane (a step takes ~2 us, far more than an HBM round trip; deeper rings
                                // only cost registers: D = 4 spilled into AGPRs and ran 13 % slower)

// ================================================================================================
// Gain pass
// ================================================================================================
template <typename T>
struct GainP {
    int B, N, mode;
    View<T> A, Bm, Cxx, Cuu, Cux;
    T *K, *Quu, *fac, *Qux;
    T *rec;                        // nullable: packed step records [Phi | B | K | fac] for riccati_ffrec_kernel
    int32_t *status;
    const int32_t *active;
    // FF form only: the first feed-forward pass of the outer iteration rides on the gain pass (same recursion as
    // riccati_ffrec_kernel, sequential over the whole horizon)
    View<T> c0x, c0u, Qr, Rr;
    const T *xhat, *uhat, *zx, *lx, *zu, *lu;
    T *kff;
    int rev;                       // 1: the grid walks the trajectory blocks from the last to the first
    const T *lin_par;              // LIN form (isls_gain_args.lin_on): the model's parameters, batch stride lin_par_sb
    int64_t lin_par_sb;
};

// 1/sqrt(a) by v_rsq + two coupled Newton steps (g -> sqrt(a), h -> 1/(2 sqrt(a))): 9 instructions against the ~30 of
// sqrt() followed by a division (both carry range scaling the pivots of a cost Hessian never need); error 1-2 ulp.
// a <= 0 or NaN gives NaN / inf exactly like sqrt would: the caller flags the pivot.
__device__ __forceinline__ double rsqrt_nr(double a)
{
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    return h + h;
}
__device__ __forceinline__ float rsqrt_nr(float a)
{
    const float y = __builtin_amdgcn_rsqf(a);
    const float g = a * y, h = 0.5f * y;
    const float r = fmaf(-h, g, 0.5f);
    const float h2 = fmaf(h, r, h);
    return h2 + h2;
}
// Upper Cholesky with reciprocal pivots only (the factor's diagonal itself is never used: see chol_solve)
template <int M, typename T>
__device__ __forceinline__ bool chol_upper_rd(const T (&A)[M][M], T (&U)[M][M], T (&rd)[M])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        T ajj = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) ajj -= U[k][j] * U[k][j];
        ok = ok && (ajj > T(0));
        rd[j] = rsqrt_nr(ajj);
        U[j][j] = T(0);
#pragma unroll
        for (int c = j + 1; c < M; ++c) {
            T sacc = A[j][c];
#pragma unroll
            for (int k = 0; k < j; ++k) sacc -= U[k][j] * U[k][c];
            U[j][c] = sacc * rd[j];
        }
    }
    return ok;
}

// MODE: ISLS_SOLVE_CHOL / ISLS_SOLVE_INV at compile time.
// FF = true: the pass also runs the v / k recursion of the feed-forward pass (riccati_ffrec.hip: v = cx + K'cu + Phi'v,
// k = -Quu^-1 (cu + B'v), the same sums in the same order) for the linear terms the caller's ADMM state gives -- the first
// of the J feed-forward passes of an outer iteration rides on this pass instead of streaming the records once more.
// Needs the packed records and time-invariant Qr / Rr rows.
// REC: the packed records are written; ARR: the Quu / fac / Qux arrays are (the single-call form).
// Every memory instruction of the step loop is unconditional -- a conditional one makes the compiler's vmcnt bookkeeping
// fall back to vmcnt(0), which drains the stores in flight at every wait for the register ring.  Lanes without a
// trajectory of their own are therefore exact copies: surplus lanes of the wavefront repeat its last lane, and a slot whose
// trajectory is past the batch or inactive shadows the first valid trajectory of the wavefront (same loads, same
// arithmetic, same stores to the same addresses; only the record image keeps the slot's own place).
#ifndef ISLS_GAIN_OCC
#define ISLS_GAIN_OCC 1
#endif
// LIN = 1 (isls_gain_args.lin_on, ISLS_MODEL_DI): [A B] = [I aI b0 I; 0 I b1 I] is not loaded or staged at all.  Column i of it
// has at most two entries, c1 at row r1 and c2 at row r2 > r1, so  row i of [A B]'V = c1 V[r1,:] + c2 V[r2,:]  (two rows of V
// instead of all n: 12 of 36 multiply-adds and LDS words at n = 6),  M = S [A B]  takes one or two products per entry (12 of
// 54, none of the 54 words of [A B]),  and  Phi[:, i] = A[:, i] + B K[:, i]  one product per entry (6 of 18).  Every term of the
// dense form that is left out adds an exact zero, and the ones that stay are written in the dense form's order and source
// form.  fp64: bit-identical to the dense kernel (finite operands; tests compare K, records and k bit by bit).  fp32: equal to
// rounding only -- there the compiler packs some two-term sums of the DENSE kernel into unfused v_pk_mul / v_pk_add pairs.
// RL (isls_gain_args.lin_on): only the tail [K | fac | model words] of every record leaves for HBM, at its own dense stride
// rec_lean_stride -- the layout the model-structured feed-forward pass reads (the [Phi | B] blocks nobody would read are 54 of
// 81 words at n = 6, m = 3: 177 MB of stores per launch at the headline size).  The record IMAGE in LDS keeps its full form.
template <typename T, int NX, int NU, int D, int MODE, bool FF, bool REC, bool ARR, int LIN = 0, bool RL = (LIN == 1)>
__global__ __launch_bounds__(64, ISLS_GAIN_OCC) void riccati_gain_kernel(GainP<T> p)
{
    static_assert(LIN == 0 || NX == 2 * NU, "double integrator: n = 2 d, m = d");
    static_assert(!RL || (REC && !ARR), "lean records: record form without the Quu / fac / Qux arrays");
    constexpr int G = NX + NU, W = NX + NU, TPW = kWave / G;
    constexpr int V_OFF = 0, AB_OFF = V_OFF + NX * NX, Q_OFF = AB_OFF + NX * W, K_OFF = Q_OFF + NU * W;
    constexpr int DUMP_OFF = K_OFF + NU * NX;              // W words that absorb the LDS writes of lanes with nothing to publish
    // FF form: d[W] | v[NX] | cu[NU] | qu[NU] behind the dump words
    constexpr int D_OFF = DUMP_OFF + W, VV_OFF = D_OFF + W, CU_OFF = VV_OFF + NX, QU_OFF = CU_OFF + NU;
    constexpr int SLOT = FF ? ((QU_OFF + NU) | 1) : ((DUMP_OFF + W) | 1);   // odd stride: slots start on different banks
    constexpr int JA = (NX * NX + G - 1) / G, JB = (NX * NU + G - 1) / G, JQ = (NU * W + G - 1) / G;
    constexpr int RB = NX * NX, RK = RB + NX * NU, RFAC = RK + NU * NX, RW = rec_stride(NX, NU);   // packed record (padded stride), see riccati_ffrec.hip
    __shared__ T lds[TPW * SLOT];
    // Image of the step's packed records, [2][TPW + 1][RW] (double buffer): the lanes drop
    // their words of a step here (this is also where the V update reads K from) and read the finished image back lane-linearly
    // while the V update computes -- the wavefront's TPW records are one contiguous run of TPW*RW words, written with
    // ceil(TPW*RW/128) fully coalesced 16-byte stores instead of W scattered 8-byte stores per lane.  The K array leaves the
    // same way.  Two buffers: a step's image is still being read when the next step starts writing.
    constexpr int IMG = (TPW + 1) * RW;
    __shared__ __align__(16) T img[2 * IMG];

    const int lane = threadIdx.x;
    const int bx = p.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;   // trajectory block of this wavefront
    const bool inslot = lane / G < TPW;
    const int s = inslot ? lane / G : TPW - 1, i = inslot ? lane - (lane / G) * G : G - 1;   // surplus lanes repeat the last lane
    const int b = bx * TPW + s;
    const bool valid = b < p.B && (p.active == nullptr || p.active[b] != 0);
    const unsigned long long vmask = __ballot(valid);
    if (vmask == 0ull) return;                                 // nothing to do for this wavefront
    const int vlane = __builtin_ctzll(vmask);                  // a lane of the first valid slot
    const int bsh = __builtin_amdgcn_readlane(b, vlane), ssh = __builtin_amdgcn_readlane(s, vlane);
    const int N = p.N;
    const int bb = valid ? b : bsh;
    T *rec = lds + s * SLOT;
    T *Vs = rec + V_OFF, *ABs = rec + AB_OFF, *Qs = rec + Q_OFF;
    const bool xl = i < NX;                                   // lane owns a row of Qxx
    const int a_row = xl ? 0 : i - NX;                        // row of [Qux Quu] for u-lanes
    const int64_t bN = (int64_t)bb * N;
    // LDS destinations of everything a lane publishes, fixed for the whole horizon: lanes (or elements) with nothing to
    // publish point at the dump words, so no ds_write of the step loop sits behind an exec-mask branch
    int dA[JA], dB[JB];
#pragma unroll
    for (int j = 0; j < JA; ++j) { const int e = i + G * j; dA[j] = (e < NX * NX) ? AB_OFF + (e / NX) * W + (e % NX) : DUMP_OFF; }
#pragma unroll
    for (int j = 0; j < JB; ++j) { const int e = i + G * j; dB[j] = (e < NX * NU) ? AB_OFF + (e / NU) * W + NX + (e % NU) : DUMP_OFF; }
    const int qdst = !xl ? Q_OFF + a_row * W : DUMP_OFF;                     // row of [Qux Quu]
    const int vdst = xl ? V_OFF + i * NX : DUMP_OFF;                         // row i of V

    // ---- global load plan: per-lane bases (trajectory, own elements) fixed for the horizon; a step adds the uniform t * stride
    const bool has_cux = p.Cux.p != nullptr;
    // (when the slot's lanes tile an array exactly -- n = 6, m = 3: 36 = 4 x 9, 18 = 2 x 9 -- no element needs the clamp and the
    // lane's loads are ONE base plus compile-time offsets: one address register per array instead of one per load)
#ifndef ISLS_GAIN_EXACT
#define ISLS_GAIN_EXACT 1
#endif
    constexpr bool A_EXACT = ISLS_GAIN_EXACT && (NX * NX) % G == 0, B_EXACT = ISLS_GAIN_EXACT && (NX * NU) % G == 0;
    const T *pA[JA], *pB[JB];
#pragma unroll
    for (int j = 0; j < JA; ++j) { const int e = i + G * j; pA[j] = p.A.at(bb, 0) + (A_EXACT ? e : (e < NX * NX ? e : NX * NX - 1)); }
#pragma unroll
    for (int j = 0; j < JB; ++j) { const int e = i + G * j; pB[j] = p.Bm.at(bb, 0) + (B_EXACT ? e : (e < NX * NU ? e : NX * NU - 1)); }
    // row i of the cost Hessian stack: [Cxx[i,:]] or [Cux[a,:] Cuu[a,:]].  Raw, unconditional loads: x-lanes never use the
    // columns >= NX of crow; u-lanes without a Cux array read Cxx instead and the step multiplies that part by zero
    const T *pcl = xl ? p.Cxx.at(bb, 0) + i * NX : (has_cux ? p.Cux.at(bb, 0) + a_row * NX : p.Cxx.at(bb, 0));
    const int64_t cl_st = xl ? p.Cxx.st : (has_cux ? p.Cux.st : p.Cxx.st);
    const T *pcr = xl ? pcl : p.Cuu.at(bb, 0) + a_row * NU;
    const int64_t cr_st = xl ? cl_st : p.Cuu.st;
    const int64_t a_st = p.A.st, b_st = p.Bm.st;
    const T zmask = (xl || has_cux) ? T(1) : T(0);            // Cux absent -> 0
    const T xmask = xl ? T(1) : T(0);
    // LIN form: the lane's column of [A B] (see the kernel's head)
    T l_a = T(0), l_b0 = T(0), l_b1 = T(0), l_c1 = T(0), l_c2 = T(0), colc[NX];
    int l_r1 = 0, l_r2 = 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) colc[k] = T(0);
    if constexpr (LIN == 1) {
        const T *lp = p.lin_par + (int64_t)bb * p.lin_par_sb;
        l_a = lp[0]; l_b0 = lp[1]; l_b1 = lp[2];
        if (xl && i < NU) { l_c1 = T(1); l_r1 = i; l_c2 = T(0); l_r2 = i; }
        else if (xl) { l_c1 = l_a; l_r1 = i - NU; l_c2 = T(1); l_r2 = i; }
        else { l_c1 = l_b0; l_r1 = a_row; l_c2 = l_b1; l_r2 = NU + a_row; }
#pragma unroll
        for (int k = 0; k < NX; ++k) colc[k] = (k == l_r1) ? l_c1 : ((k == l_r2) ? l_c2 : T(0));
    }

    // ---- FF form: per-lane plan of the linear terms (lane i owns c_i, d_i = xhat_i - (z_i - lambda_i), v_i) ----------
    const bool ff_hasreg = FF && (xl ? p.Qr.p != nullptr : p.Rr.p != nullptr);
    const int ff_lim = xl ? NX : NU;
    const T *ff_c0 = nullptr, *ff_h = nullptr, *ff_z = nullptr, *ff_l = nullptr;
    int64_t ff_c0st = 0, ff_vst = 0, ff_hst = 0;
    bool ff_hash = false;
    T ff_row[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) ff_row[j] = T(0);
    if constexpr (FF) {
        ff_c0 = xl ? p.c0x.at(bb, 0) + i : p.c0u.at(bb, 0) + a_row;
        ff_c0st = xl ? p.c0x.st : p.c0u.st;
        const int dd = xl ? NX : NU;
        const int64_t ovec = bN * dd + (xl ? i : a_row);
        const T *ph = xl ? p.xhat : p.uhat;
        ff_hash = ff_hasreg && ph != nullptr;
        // lanes without a regularised block load c0 again (always a valid address) and drop the values
        ff_h = ff_hash ? ph + ovec : ff_c0;
        ff_z = ff_hasreg ? (xl ? p.zx : p.zu) + ovec : ff_c0;
        ff_l = ff_hasreg ? (xl ? p.lx : p.lu) + ovec : ff_c0;
        ff_vst = ff_hasreg ? dd : ff_c0st;
        ff_hst = ff_hash ? ff_vst : ff_c0st;
        if (ff_hasreg) {
            const T *prow = xl ? p.Qr.at(bb, 0) + i * NX : p.Rr.at(bb, 0) + a_row * NU;
#pragma unroll
            for (int j = 0; j < NX; ++j) ff_row[j] = (j < ff_lim) ? T(2) * prow[j < ff_lim ? j : ff_lim - 1] : T(0);   // 2 * row, exact
        }
    }
    const T ff_dmask = ff_hasreg ? T(1) : T(0), ff_hmask = ff_hash ? T(1) : T(0);
    const int ff_ddst = D_OFF + i;                                            // d_i
    const int ff_vdst = xl ? VV_OFF + i : DUMP_OFF;                           // v_i
    const int ff_cdst = !xl ? CU_OFF + a_row : DUMP_OFF;                      // cu_r
    const int ff_qdst = !xl ? QU_OFF + a_row : DUMP_OFF + 1;                  // qu_r
    const int ff_ku = xl ? i % NU : a_row;                                    // the entry of k_t this lane stores (x-lanes: a copy)
    const int ff_doff = D_OFF + (xl ? 0 : NX);
    // c_i = c0_i + 2 * (row of Qr / Rr) . d      (isls/sls.py:132-137; riccati_ffrec.hip reg_grad: c0 + 2 * sum, the factor
    // folded into the row -- a power of two, so every product and the sum round exactly as there)
    auto ff_grad = [&](T c0v) -> T {
        T sacc = T(0);
#pragma unroll
        for (int j = 0; j < NX; ++j) sacc += ff_row[j] * rec[ff_doff + j];   // j >= lim: zero row entry times a neighbour word
        return c0v + sacc;
    };

    // ---- terminal step: K[N-1] = 0 (isls.py:245), V = Cxx[N-1] (isls.py:251/257) -----------------
    if constexpr (FF) {                                        // v = cx[N-1], k[N-1] = 0
        // ff_grad multiplies words behind a u-lane's d block by zero row entries: they must be finite from the start
        for (int e = i; e < SLOT - D_OFF; e += G) rec[D_OFF + e] = T(0);
        slot_sync();
        const int64_t tl = N - 1;
        const T c0v = ff_c0[tl * ff_c0st], hv = ff_h[tl * ff_hst], zv = ff_z[tl * ff_vst], lv = ff_l[tl * ff_vst];
        rec[ff_ddst] = ff_hasreg ? (ff_hash ? hv : T(0)) - (zv - lv) : T(0);
        slot_sync();
        rec[ff_vdst] = ff_grad(c0v);
        if (valid && !xl) p.kff[(bN + tl) * NU + a_row] = T(0);
        slot_sync();
    }
    {
        T r[JA];
        coop_load<NX * NX, G>(p.Cxx.at(bb, N - 1), r, i, valid);
        coop_put<NX * NX, G>(Vs, r, i, true);
        if (valid) {
            const int64_t o = bN + (N - 1);
#pragma unroll
            for (int j = 0; j < (NU * NX + G - 1) / G; ++j) {
                const int e = i + G * j;
                if (e < NU * NX) { p.K[o * NU * NX + e] = T(0); if (p.Qux) p.Qux[o * NU * NX + e] = T(0); }
            }
#pragma unroll
            for (int j = 0; j < (NU * NU + G - 1) / G; ++j) {
                const int e = i + G * j;
                if (e < NU * NU) { if (p.Quu) p.Quu[o * NU * NU + e] = T(0); if (p.fac) p.fac[o * NU * NU + e] = T(0); }
            }
        }
    }

    // ---- register ring: operands of D steps in flight ------------------------------------------------
    struct Stage {
        T ra[JA], rb[JB], crow[W];
        T c0, hv, zv, lv;                                      // FF form
    };
#ifndef ISLS_GAIN_RING
#define ISLS_GAIN_RING 1                                        // one step ahead covers an HBM round trip (a step takes ~2 us); two cost 38 registers per
#endif                                                          // lane, which the compiler parks in AGPRs and copies back every step (212 -> 205 us)
    constexpr int RD = ISLS_GAIN_RING;                          // steps of operands in flight (<= D: the unrolled group)
    static_assert(RD >= 1 && RD <= D, "ring depth");
    Stage ring[RD];
    auto fetch = [&](int tq, Stage &g) {
        const int t = __builtin_amdgcn_readfirstlane(tq);      // uniform: the offsets below are scalar arithmetic
        const int64_t oa = (int64_t)t * a_st, ob = (int64_t)t * b_st;
        if constexpr (LIN == 0) {
#pragma unroll
            for (int j = 0; j < JA; ++j) g.ra[j] = ISLS_NT_GAIN_LD ? ld_stream(pA[j] + oa) : pA[j][oa];
#pragma unroll
            for (int j = 0; j < JB; ++j) g.rb[j] = ISLS_NT_GAIN_LD ? ld_stream(pB[j] + ob) : pB[j][ob];
        }
        const T *cl = pcl + (int64_t)t * cl_st, *cr = pcr + (int64_t)t * cr_st;
#pragma unroll
        for (int j = 0; j < NX; ++j) g.crow[j] = cl[j];
#pragma unroll
        for (int j = 0; j < NU; ++j) g.crow[NX + j] = cr[j];
        if constexpr (FF) {
            g.c0 = ff_c0[(int64_t)t * ff_c0st];
            g.hv = ff_h[(int64_t)t * ff_hst];
            g.zv = ff_z[(int64_t)t * ff_vst];
            g.lv = ff_l[(int64_t)t * ff_vst];
        }
    };
#pragma unroll
    for (int d = 0; d < RD; ++d) {
        fetch(N - 2 - d > 0 ? N - 2 - d : 0, ring[d]);          // unconditional (clamped): exact vmcnt bookkeeping
        __builtin_amdgcn_sched_barrier(0);                      // issue order = consumption order
    }
    bool pd_ok = true;
    // where a lane's W words of the packed record go: x-lane i -> column i of Phi (stride NX) then of K (stride NX);
    // u-lane r -> column r of B (stride NU) then row r of fac: 9 lanes x 9 words = the whole 81-word record at n=6, m=3
    const int r_b1 = s * RW + (xl ? i : RB + a_row), r_s1 = xl ? NX : NU;
    const int r_b2 = s * RW + (xl ? RK + i : RFAC + a_row * NU), r_s2 = xl ? NX : 1;
    // lane-linear plan of the image's way out: 16-byte pair q = lane + 64 j of the TPW*RW record words (the last pair again
    // for surplus lanes: same words to the same address), and pair / word e of the slots' K blocks for the K array
    typedef T V2 __attribute__((ext_vector_type(2)));
    constexpr int RSW = RL ? rec_lean_stride(NX, NU) : RW;     // words of a record that leave for HBM, = its stride there
    constexpr int RSRC = RL ? RK : 0;                          // the first of them in the image
    static_assert(RSRC % 2 == 0 && RSRC + RSW <= RW, "record tail: pair-aligned, inside the image");
    constexpr int NPAIR = TPW * RSW / 2, JP = (NPAIR + kWave - 1) / kWave;
    constexpr bool KPAIRS = (NU * NX) % 2 == 0 && RK % 2 == 0;
    constexpr int KU = KPAIRS ? NU * NX / 2 : NU * NX, JK = (TPW * KU + kWave - 1) / kWave;   // K pieces per slot / loads per lane
    int kso[JK];                                               // image word of the lane's K piece
    int64_t kgo[JK];                                           // its word in the K array at t = 0
#pragma unroll
    for (int j = 0; j < JK; ++j) {
        // pieces of slots without a trajectory go where the shadowed trajectory's go (same words); lanes past the last piece
        // repeat piece 0 of the first valid slot
        const int e = lane + kWave * j, in = e < TPW * KU;
        const int sl = in ? e / KU : ssh, pc = in ? e - (e / KU) * KU : 0;
        const int bk = bx * TPW + sl;
        const bool ok = bk < p.B && (p.active == nullptr || p.active[bk] != 0);
        kso[j] = sl * RW + RK + pc * (KPAIRS ? 2 : 1);
        kgo[j] = (int64_t)(ok ? bk : bsh) * N * (NU * NX) + pc * (KPAIRS ? 2 : 1);
    }
    int fw[JP], fd[JP];                                        // image word of the lane's j-th record pair; its word in the step's run
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        const int pq = lane + kWave * j;
        fd[j] = 2 * (pq < NPAIR ? pq : NPAIR - 1);
        fw[j] = (fd[j] / RSW) * RW + RSRC + fd[j] % RSW;       // (dense records: fw == fd)
    }
    T *const recg = REC ? p.rec + (int64_t)bx * N * (TPW * RSW) : nullptr;   // the wavefront's records, step 0

    // The image of a step is read back into fl / fk behind its sync (c) and leaves for HBM during the NEXT step, one store
    // between two blocks of that step's arithmetic: a wavefront waits while a store's data drains (the CU moves ~7-16 B per
    // clock), so stores issued back to back are paid in full, spread out they overlap with the arithmetic.
    V2 fl[JP], fk[JK];
    T fk1[JK];
    auto send = [&](int tq, int j) {                           // j-th of the JP + JK stores of step tq's image
        if (j < JP) {
            if constexpr (REC) {
                V2 *dr = reinterpret_cast<V2 *>(recg + (int64_t)tq * (TPW * RSW) + fd[j < JP ? j : 0]);
                if constexpr (ISLS_NT_GAIN_ST) st_stream(dr, fl[j < JP ? j : 0]);
                else *dr = fl[j < JP ? j : 0];
            }
        } else if (j < JP + JK) {
            const int jj = j - JP < JK ? (j - JP >= 0 ? j - JP : 0) : 0;
            T *dk = p.K + kgo[jj] + (int64_t)tq * (NU * NX);
            if constexpr (KPAIRS) {
                if constexpr (ISLS_NT_GAIN_ST) st_stream(reinterpret_cast<V2 *>(dk), fk[jj]);
                else *reinterpret_cast<V2 *>(dk) = fk[jj];
            } else {
                *dk = fk1[jj];
            }
        }
    };
    // the JP + JK stores take evenly spaced places among the 3 NX blocks of the step's three accumulation loops
    constexpr int NSEND = JP + JK, NPOS = 3 * NX;
    static_assert(NSEND <= NPOS, "at most one store per block");
    auto send_at = [&](int tq, auto POS) {
        constexpr int pos = decltype(POS)::value;
#ifndef ISLS_GAIN_EXP_NOFLUSH
        static_for<NSEND>([&](auto JJ) {
            constexpr int j = decltype(JJ)::value;
            if constexpr ((j * NPOS) / NSEND == pos) {
                __builtin_amdgcn_sched_barrier(0);
                send(tq, j);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
#endif
    };
    auto step = [&](int t, Stage &g, int q, auto FLUSH) {
        constexpr bool flush_prev = decltype(FLUSH)::value;
        T *imq = img + q * IMG;
        // stage [A_t B_t] into the record: row k = [A[k,:] B[k,:]]
        if constexpr (LIN == 0) {
#pragma unroll
            for (int j = 0; j < JA; ++j) rec[dA[j]] = g.ra[j];
#pragma unroll
            for (int j = 0; j < JB; ++j) rec[dB[j]] = g.rb[j];
        }
        T ff_c0v = T(0);
        if constexpr (FF) {
            rec[ff_ddst] = ff_dmask * (ff_hmask * g.hv - (g.zv - g.lv));
            ff_c0v = g.c0;
        }
        slot_sync();                                          // (a) ABs, Vs visible to the slot

        // Everything the step reads from the slot's V and [A B] goes into registers in ONE batch of LDS reads (a lone wavefront
        // pays the full LDS latency at every wait: read-a-little / compute-a-little costs that latency ~50 times per step)
        // V first; the rows of [A B] are requested one by one as the rows of V are used up, so they arrive while (1) computes
        // and both are never in registers in full.  AHEAD rows are in flight at a time: the whole of V (AHEAD = NX) costs
        // registers for nothing -- two rows ahead cover the LDS latency just as well (n = 6 alone: 182 -> 171 us; with the
        // feed-forward recursion inside 2 / 3 / 4 / 6 rows measure the same, 3 kept) and at n = 9 the whole batch would
        // spill into scratch memory.
#ifndef ISLS_GAIN_AHEAD_FF
#define ISLS_GAIN_AHEAD_FF 3
#endif
#ifndef ISLS_GAIN_AHEAD
#define ISLS_GAIN_AHEAD 2
#endif
        constexpr int AHEAD_ = (NX * NX + NX * W <= 100) ? (FF ? ISLS_GAIN_AHEAD_FF : ISLS_GAIN_AHEAD) : 2;
        constexpr int AHEAD = AHEAD_ < NX ? AHEAD_ : NX;
        T S[NX], colv[NX], Vr[NX][NX], Fr[NX][W];
        T ff_v[NX], ff_dr[NX];
        if constexpr (LIN == 1) {
            // (1) S = c1 V[r1,:] + c2 V[r2,:]: the dense sum's two non-zero terms in its order (r1 < r2, or one term and + 0),
            // written the way the dense loop writes them (fp64: fused multiply-adds in both kernels)
            T v1[NX], v2[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) { v1[j] = Vs[l_r1 * NX + j]; v2[j] = Vs[l_r2 * NX + j]; colv[j] = colc[j]; }
            if constexpr (FF) {
#pragma unroll
                for (int k = 0; k < NX; ++k) { ff_v[k] = rec[VV_OFF + k]; ff_dr[k] = rec[ff_doff + k]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<NX>([&](auto KK) {
                constexpr int k = decltype(KK)::value;
                T acc = T(0);
                acc += l_c1 * v1[k];
                acc += l_c2 * v2[k];
                S[k] = acc;
                if constexpr (flush_prev) send_at(t + 1, std::integral_constant<int, k>{});
            });
        } else {
#pragma unroll
        for (int k = 0; k < NX; ++k) colv[k] = ABs[k * W + i];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
#pragma unroll
            for (int j = 0; j < NX; ++j) Vr[k][j] = Vs[k * NX + j];
        }
        if constexpr (FF) {
#pragma unroll
            for (int k = 0; k < NX; ++k) { ff_v[k] = rec[VV_OFF + k]; ff_dr[k] = rec[ff_doff + k]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (1) S = row i of [A B]'V
#pragma unroll
        for (int j = 0; j < NX; ++j) S[j] = T(0);
        static_for<NX>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
#pragma unroll
            for (int j = 0; j < NX; ++j) S[j] += colv[k] * Vr[k][j];
            if constexpr (k + AHEAD < NX) {                    // the next row of V that is not in flight yet
#pragma unroll
                for (int j = 0; j < NX; ++j) Vr[k + AHEAD][j] = Vs[(k + AHEAD) * NX + j];
            }
            if constexpr (k - (NX - AHEAD) >= 0) {             // the first AHEAD rows of [A B] are in flight when (1) ends
                constexpr int kf = k - (NX - AHEAD);
#pragma unroll
                for (int c = 0; c < W; ++c) Fr[kf][c] = ABs[kf * W + c];
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (flush_prev) send_at(t + 1, std::integral_constant<int, k>{});
        });
        }
        // FF form: c_i, and the u-lanes' qu_r = cu_r + (B'v)_r from their column of B
        T ff_ci = T(0);
        if constexpr (FF) {
            T sacc = T(0);
#pragma unroll
            for (int j = 0; j < NX; ++j) sacc += ff_row[j] * ff_dr[j];   // j >= lim: zero row entry times a neighbour word
            ff_ci = ff_c0v + sacc;
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < NX; ++k) acc += colv[k] * ff_v[k];
            rec[ff_cdst] = ff_ci;
            rec[ff_qdst] = ff_ci + acc;
        }
        // (2) M = C_row + S [A B]      (Qxx = Cxx + (A'V)A etc., isls.py:288-290)
        T M[W];
        if constexpr (LIN == 1) {
            // column c of [A B]: c < d: e_c;  d <= c < n: a e_{c-d} + e_c;  n + r: b0 e_r + b1 e_{d+r}
            static_for<NX>([&](auto KK) {
                constexpr int k = decltype(KK)::value;
                if constexpr (k < NU) {
                    M[k] = S[k];
                    T acc = T(0);
                    acc += S[k] * l_b0;
                    acc += S[NU + k] * l_b1;
                    M[NX + k] = acc;
                } else {
                    // the dense sum rounds a S[k-d] and then adds S[k] (times an exact one): never one fused operation
#pragma clang fp contract(off)
                    const T t_a = S[k - NU] * l_a;
                    M[k] = t_a + S[k];
                }
                if constexpr (flush_prev) send_at(t + 1, std::integral_constant<int, NX + k>{});
            });
        } else {
#pragma unroll
        for (int c = 0; c < W; ++c) M[c] = T(0);
        static_for<NX>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
#pragma unroll
            for (int c = 0; c < W; ++c) M[c] += S[k] * Fr[k][c];
            if constexpr (k + AHEAD < NX) {
#pragma unroll
                for (int c = 0; c < W; ++c) Fr[k + AHEAD][c] = ABs[(k + AHEAD) * W + c];
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (flush_prev) send_at(t + 1, std::integral_constant<int, NX + k>{});
        });
        }
#pragma unroll
        for (int c = 0; c < W; ++c) M[c] = (c < NX) ? fma(g.crow[c], zmask, M[c]) : g.crow[c] + M[c];   // one rounding either way
        __builtin_amdgcn_sched_barrier(0);
        fetch(t - RD > 0 ? t - RD : 0, g);                     // refill this ring entry (clamped, unconditional): ~RD - 0.5 steps ahead
        // (3) u-lanes publish their row of [Qux Quu] (x-lanes write the dump words)
#pragma unroll
        for (int c = 0; c < W; ++c) rec[qdst + c] = M[c];
        slot_sync();                                          // (b)

        // (4) factor Quu (redundantly in every lane), solve for column i of K (u-lanes: a zero right-hand side, K column 0)
        T Quu[NU][NU], U[NU][NU], rd[NU], rhs[NU], Kc[NU], inv[NU][NU];
#pragma unroll
        for (int r = 0; r < NU; ++r) {
#pragma unroll
            for (int c = 0; c < NU; ++c) { Quu[r][c] = Qs[r * W + NX + c]; U[r][c] = T(0); inv[r][c] = T(0); }
            rhs[r] = Qs[r * W + (xl ? i : 0)] * xmask;         // column i of Qux
        }
        pd_ok = chol_upper_rd<NU>(Quu, U, rd) && pd_ok;
        if constexpr (MODE == ISLS_SOLVE_CHOL) {
            T x[NU];
            chol_solve<NU>(U, rd, rhs, x);                     // sol = -solve(Quu, Qux)  (isls.py:296)
#pragma unroll
            for (int r = 0; r < NU; ++r) Kc[r] = -x[r];
        } else {
#pragma unroll
            for (int c = 0; c < NU; ++c) {                     // Quu_inv column by column (sls.py:149)
                T e[NU], x[NU];
#pragma unroll
                for (int r = 0; r < NU; ++r) e[r] = (r == c) ? T(1) : T(0);
                chol_solve<NU>(U, rd, e, x);
#pragma unroll
                for (int r = 0; r < NU; ++r) inv[r][c] = x[r];
            }
#pragma unroll
            for (int r = 0; r < NU; ++r) {                     // Kt = -Quu_inv.dot(Qux)  (sls.py:150)
                T sacc = T(0);
#pragma unroll
                for (int c = 0; c < NU; ++c) sacc += inv[r][c] * rhs[c];
                Kc[r] = -sacc;
            }
        }
        const int64_t o = bN + t;
        // the lane's column of [Phi | B]: x-lane i -> Phi[:, i] = A[:, i] + B K[:, i]; u-lane r -> B[:, r] (its K column is zero)
#ifndef ISLS_GAIN_REREAD_B
#define ISLS_GAIN_REREAD_B 0
#endif
        // (-DISLS_GAIN_REREAD_B=1: B_t from the slot's LDS again here instead of registers kept since (2): 402 -> 394 registers,
        // no faster with the feed-forward pass inside (208-213 us both) and 6 us slower without (172 vs 166 us); capping the
        // kernel at 256 registers -- -DISLS_GAIN_OCC=2 -- spills 456 B into scratch and takes 385 us: its real pressure is ~370)
        T Bq[NX][NU];
        if constexpr (ISLS_GAIN_REREAD_B && LIN == 0) {
#pragma unroll
            for (int k = 0; k < NX; ++k)
#pragma unroll
                for (int r = 0; r < NU; ++r) Bq[k][r] = ABs[k * W + NX + r];
        }
        T phc[NX];
        static_for<NX>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            T ph = colv[k];
            if constexpr (LIN == 1) {
                ph += (k < NU ? l_b0 : l_b1) * Kc[k % NU];        // row k of B has one entry
            } else {
#pragma unroll
                for (int r = 0; r < NU; ++r) ph += (ISLS_GAIN_REREAD_B ? Bq[k][r] : Fr[k][NX + r]) * Kc[r];
            }
            phc[k] = ph;
            if constexpr (flush_prev) send_at(t + 1, std::integral_constant<int, 2 * NX + k>{});
        });
        // the cached factor's row a_row (identical values in every lane; the row is picked by selects); x-lanes carry their
        // column of K in the same registers
        T tail[NU];
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            T v = xl ? Kc[c] : ((MODE == ISLS_SOLVE_CHOL) ? (c == 0 ? rd[0] : U[0][c]) : inv[0][c]);
#pragma unroll
            for (int r = 1; r < NU; ++r) {
                const T vr = (MODE == ISLS_SOLVE_CHOL) ? ((c == r) ? rd[r] : (c > r ? U[r][c] : T(0))) : inv[r][c];
                v = (a_row == r) ? vr : v;
            }
            tail[c] = v;
        }
        if constexpr (FF) {
            T qu[NU], cuv[NU], kt[NU];
#pragma unroll
            for (int r = 0; r < NU; ++r) { cuv[r] = rec[CU_OFF + r]; qu[r] = rec[QU_OFF + r]; }
            if constexpr (MODE == ISLS_SOLVE_CHOL) {
                T x[NU];
                chol_solve<NU>(U, rd, qu, x);
#pragma unroll
                for (int r = 0; r < NU; ++r) kt[r] = -x[r];
            } else {
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    T a2 = T(0);
#pragma unroll
                    for (int c = 0; c < NU; ++c) a2 += inv[r][c] * qu[c];
                    kt[r] = -a2;
                }
            }
            T kv = kt[0];
#pragma unroll
            for (int r = 1; r < NU; ++r) kv = (ff_ku == r) ? kt[r] : kv;
            p.kff[o * NU + ff_ku] = kv;                        // every lane (x-lanes and shadows store copies)
            // v_i = (cx_i + (Phi'v)_i) + (K'cu)_i
            T acc = T(0), kcu = T(0);
#pragma unroll
            for (int k = 0; k < NX; ++k) acc += phc[k] * ff_v[k];
#pragma unroll
            for (int r = 0; r < NU; ++r) kcu += Kc[r] * cuv[r];
            rec[ff_vdst] = (ff_ci + acc) + kcu;                // read again only after the next (a)
        }
        // the lane's W words of the record image (unconditional: surplus lanes own the image's spare slot)
#pragma unroll
        for (int k = 0; k < NX; ++k) imq[r_b1 + k * r_s1] = phc[k];
#pragma unroll
        for (int c = 0; c < NU; ++c) imq[r_b2 + c * r_s2] = tail[c];
        if constexpr (REC && rec_model_words(NX, NU) == 6) {
            // the model words of the step behind fac (rec_model_words: the arm's A[6:8, 0:3], the car's six entries): lanes 0..5 of
            // a slot carry one word each, the others word 0 again
            const int je = i < 6 ? i : 0;
            int src = rec_model_src(NX, NU, 0);
#pragma unroll
            for (int e = 1; e < 6; ++e) src = (je == e) ? rec_model_src(NX, NU, e) : src;
            imq[s * RW + RFAC + NU * NU + je] = ABs[src];
        }
        if constexpr (ARR) {
            {
                // cooperative store of [Qux Quu] rows and the factor (skipped when the caller only wants K and the packed
                // records: every consumer of Qux / Quu / fac then reads the records instead)
#pragma unroll
                for (int j = 0; j < JQ; ++j) {
                    const int e = i + G * j;
                    if (e < NU * W) {
                        const int r = e / W, c = e % W;
                        const T v = Qs[e];
                        if (c < NX) p.Qux[(o * NU + r) * NX + c] = v;
                        else p.Quu[(o * NU + r) * NU + (c - NX)] = v;
                    }
                }
                if (!xl) {
#pragma unroll
                    for (int c = 0; c < NU; ++c) p.fac[(o * NU + a_row) * NU + c] = tail[c];
                }
            }
        }
        slot_sync();                                          // (c) the image (K among it) visible
        const T *Ks = imq + s * RW + RK;
        // the finished image goes back into registers lane-linearly (unconditional reads, issued ahead of the V update whose
        // arithmetic covers their latency) and leaves for HBM behind it
        if constexpr (REC) {
#pragma unroll
            for (int j = 0; j < JP; ++j) fl[j] = *reinterpret_cast<const V2 *>(imq + fw[j]);
        }
#pragma unroll
        for (int j = 0; j < JK; ++j) {
            if constexpr (KPAIRS) fk[j] = *reinterpret_cast<const V2 *>(imq + kso[j]);
            else fk1[j] = imq[kso[j]];
        }

        // (5) V row i = Qxx + (K'Quu)K + Qux'K + K'Qux   (isls.py:300 / sls.py:153); u-lanes run the same instructions
        // on their zero column and write the dump words
        {
            T Kr[NU][NX], Qxr[NU][NX];                         // K and Qux in one batch of reads
#pragma unroll
            for (int r = 0; r < NU; ++r) {
#pragma unroll
                for (int j = 0; j < NX; ++j) { Kr[r][j] = Ks[r * NX + j]; Qxr[r][j] = Qs[r * W + j]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            T Wr[NU];
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                T sacc = T(0);
#pragma unroll
                for (int r = 0; r < NU; ++r) sacc += Kc[r] * Quu[r][c];
                Wr[c] = sacc;
            }
#ifndef ISLS_GAIN_JOSEPH2
#define ISLS_GAIN_JOSEPH2 1
#endif
#if ISLS_GAIN_JOSEPH2
            // the same four terms with (K'Quu) K and Qux'K under one sum: ((K'Quu)_i. + Qux_.i) . K_.j + K_.i . Qux_.j -- the first
            // bracket is the residual of Quu K = -Qux, so the value keeps the form's insensitivity to the rounding of K (two
            // multiply-adds per (r, j) instead of three)
            T Gr[NU];
#pragma unroll
            for (int r = 0; r < NU; ++r) Gr[r] = Wr[r] + rhs[r];
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                T t12 = T(0), t3 = T(0);
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    t12 += Gr[r] * Kr[r][j];                   // (K'Quu + Qux') K
                    t3 += Kc[r] * Qxr[r][j];                   // K' Qux
                }
                rec[vdst + j] = (M[j] + t3) + t12;
            }
#else
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                T t1 = T(0), t2 = T(0), t3 = T(0);
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    const T kj = Kr[r][j];
                    t1 += Wr[r] * kj;                          // (K'Quu) K
                    t2 += rhs[r] * kj;                         // Qux' K
                    t3 += Kc[r] * Qxr[r][j];                   // K' Qux
                }
                const T vn = (MODE == ISLS_SOLVE_CHOL) ? ((M[j] + t1) + t2) + t3 : ((M[j] + t2) + t3) + t1;
                rec[vdst + j] = vn;
            }
#endif
        }
    };

    // full groups of D steps run branch-free (every VMEM op of the steady state is unconditional, so the
    // compiler's vmcnt bookkeeping is exact and D steps of loads really stay in flight); then the remainder
    static_assert(D == 2, "the record image is double-buffered by ring position");
    int tb = N - 2;
    int tlast = -1;
    if (tb >= 0) {                                             // first step: no image to send yet
        step(tb, ring[0], 0, std::false_type{});
        tlast = tb;
        --tb;
    }
    for (; tb - (D - 1) >= 0; tb -= D) {
        step(tb, ring[RD - 1], 1, std::true_type{});
        step(tb - 1, ring[0], 0, std::true_type{});
        tlast = tb - 1;
    }
    if (tb >= 0) {                                             // one more step (tb == 0)
        step(tb, ring[RD - 1], 1, std::true_type{});
        tlast = tb;
    }
#ifndef ISLS_GAIN_EXP_NOFLUSH
    if (tlast >= 0) {                                          // the last step's image
#pragma unroll
        for (int j = 0; j < JP + JK; ++j) send(tlast, j);
    }
#endif
    if (valid && i == 0 && !pd_ok && p.status) atomicOr(&p.status[b], ISLS_ST_NOT_PD);
}

// dimensions for which the pass with the feed-forward recursion inside keeps its operands in registers (at n = 9 it spills
// into scratch memory: the first feed-forward pass then stays a launch of its own)
constexpr bool gain_ff_dims(int n, int m) { return n * n + n * (n + m) <= 100; }

template <typename T>
int launch_gain(const isls_gain_args &a, hipStream_t s, const isls_ff_args *ff, bool *did_ff, bool require_ff)
{
    if (did_ff) *did_ff = false;
    if (a.B < 0 || a.N < 1 || !a.A.p || !a.Bm.p || !a.Cxx.p || !a.Cuu.p || !a.K) return ISLS_ERR_ARG;
    // Quu, fac, Qux: all three or none; none only with records (every consumer then reads those)
    if ((!a.Quu || !a.Qux || !a.fac) && (!a.rec || a.Quu || a.Qux || a.fac)) return ISLS_ERR_ARG;
    if (a.solve_mode != ISLS_SOLVE_CHOL && a.solve_mode != ISLS_SOLVE_INV) return ISLS_ERR_ARG;
    if (a.B == 0) return ISLS_OK;
    if (!dims_supported(a.n, a.m)) return require_ff ? ISLS_OK : launch_gain_generic<T>(a, s);   // generic.hip (no ff pass inside)
    GainP<T> p;
    p.B = a.B; p.N = a.N; p.mode = a.solve_mode;
    p.A = View<T>(a.A); p.Bm = View<T>(a.Bm); p.Cxx = View<T>(a.Cxx); p.Cuu = View<T>(a.Cuu); p.Cux = View<T>(a.Cux);
    p.K = (T *)a.K; p.Quu = (T *)a.Quu; p.fac = (T *)a.fac; p.Qux = (T *)a.Qux; p.rec = (T *)a.rec;
    p.status = a.status; p.active = a.active;
    p.xhat = p.uhat = p.zx = p.lx = p.zu = p.lu = nullptr; p.kff = nullptr;
    static const int rev_mode = [] { const char *e = getenv("ISLS_GAIN_REV"); return e ? atoi(e) : 0; }();   // EXPERIMENT
    p.rev = rev_mode ? 1 : 0;
    // model-structured form (isls_gain_args.lin_on): on the record forms the drivers use; the caller switches it off by not
    // giving the hint (isls.Engine: use_model_structure / ISLS_FF_LEAN=0, for the gain pass and its readers together)
    if (ff && ff->rec == a.rec && a.rec && (ff->lin_on != 0) != (a.lin_on != 0)) return ISLS_ERR_ARG;   // one layout for writer and reader
    // lin_on also selects the LEAN record layout, which the feed-forward passes of the same hint read: no silent fall-back
    bool lin_di = false, lean_arm = false, lean_car = false;
    if (a.lin_on) {
        if (!a.rec || a.Quu || a.fac || a.Qux) return ISLS_ERR_UNSUPPORTED;        // record form without the arrays only
        if (a.lin_model == ISLS_MODEL_DI) { if (a.n != 2 * a.m || !a.lin_par) return ISLS_ERR_ARG; lin_di = true; }
        else if (a.lin_model == ISLS_MODEL_ARM3R) { if (a.n != 9 || a.m != 3) return ISLS_ERR_ARG; lean_arm = true; }   // dense arithmetic, lean records
        else if (a.lin_model == ISLS_MODEL_CAR) { if (a.n != 4 || a.m != 2) return ISLS_ERR_ARG; lean_car = true; }     // likewise
        else return ISLS_ERR_UNSUPPORTED;
    }
    p.lin_par = (const T *)a.lin_par; p.lin_par_sb = a.lin_par_sb;
    // the first feed-forward pass rides along when it would run on this pass's records with time-invariant Qr / Rr rows
    const bool with_ff = ff && gain_ff_dims(a.n, a.m) && a.rec && ff->rec == a.rec && ff->k && ff->B == a.B && ff->N == a.N && ff->n == a.n && ff->m == a.m &&
                         ff->solve_mode == a.solve_mode && ff->active == a.active && ff->c0x.p && ff->c0u.p &&
                         (!ff->Qr.p || ff->Qr.st == 0) && (!ff->Rr.p || ff->Rr.st == 0) && (!ff->Qr.p || (ff->zx && ff->lx)) &&
                         (!ff->Rr.p || (ff->zu && ff->lu)) && !a.Qux && !ff->Qr_term && ff->_pad <= 1;
    if (require_ff && !with_ff) return ISLS_OK;
    if (with_ff) {
        p.c0x = View<T>(ff->c0x); p.c0u = View<T>(ff->c0u); p.Qr = View<T>(ff->Qr); p.Rr = View<T>(ff->Rr);
        p.xhat = (const T *)ff->xhat; p.uhat = (const T *)ff->uhat;
        p.zx = (const T *)ff->zx; p.lx = (const T *)ff->lx; p.zu = (const T *)ff->zu; p.lu = (const T *)ff->lu;
        p.kff = (T *)ff->k;
    }
#define LAUNCH_G(NX_, NU_, MODE_, FF_, REC_, ARR_) \
    hipLaunchKernelGGL((riccati_gain_kernel<T, NX_, NU_, kGainDepth, MODE_, FF_, REC_, ARR_>), dim3(grid), dim3(64), 0, s, p)
#define LAUNCH_GL(NX_, NU_, MODE_, FF_) \
    hipLaunchKernelGGL((riccati_gain_kernel<T, NX_, NU_, kGainDepth, MODE_, FF_, true, false, 1>), dim3(grid), dim3(64), 0, s, p)
#define LAUNCH_GC(NX_, NU_, MODE_, FF_) /* dense arithmetic, lean records (the car) */      \
    hipLaunchKernelGGL((riccati_gain_kernel<T, NX_, NU_, kGainDepth, MODE_, FF_, true, false, 0, true>), dim3(grid), dim3(64), 0, s, p)
#define LAUNCH_M(NX_, NU_, MODE_)                                                           \
    {                                                                                       \
        if (with_ff) {                                                                      \
            if constexpr (gain_ff_dims(NX_, NU_)) {                                         \
                if constexpr (NX_ == 4 && NU_ == 2) { if (lin_di) LAUNCH_GL(NX_, NU_, MODE_, true); else if (lean_car) LAUNCH_GC(NX_, NU_, MODE_, true); else LAUNCH_G(NX_, NU_, MODE_, true, true, false); } \
                else if constexpr (NX_ == 2 * NU_) { if (lin_di) LAUNCH_GL(NX_, NU_, MODE_, true); else LAUNCH_G(NX_, NU_, MODE_, true, true, false); } \
                else LAUNCH_G(NX_, NU_, MODE_, true, true, false);                          \
            }                                                                               \
        } else if (a.rec && !a.Qux) {                                                       \
            if constexpr (NX_ == 4 && NU_ == 2) { if (lin_di) LAUNCH_GL(NX_, NU_, MODE_, false); else if (lean_car) LAUNCH_GC(NX_, NU_, MODE_, false); else LAUNCH_G(NX_, NU_, MODE_, false, true, false); } \
            else if constexpr (NX_ == 2 * NU_) { if (lin_di) LAUNCH_GL(NX_, NU_, MODE_, false); else LAUNCH_G(NX_, NU_, MODE_, false, true, false); } \
            else if constexpr (NX_ == 9 && NU_ == 3) {                                      \
                if (lean_arm) hipLaunchKernelGGL((riccati_gain_kernel<T, NX_, NU_, kGainDepth, MODE_, false, true, false, 0, true>), dim3(grid), dim3(64), 0, s, p); \
                else LAUNCH_G(NX_, NU_, MODE_, false, true, false);                         \
            } else LAUNCH_G(NX_, NU_, MODE_, false, true, false);                           \
        }                                                                                   \
        else if (a.rec) LAUNCH_G(NX_, NU_, MODE_, false, true, true);                       \
        else LAUNCH_G(NX_, NU_, MODE_, false, false, true);                                 \
    }
#define CALL(NX_, NU_)                                                                      \
    {                                                                                       \
        constexpr int TPW = kWave / (NX_ + NU_);                                            \
        const int grid = (a.B + TPW - 1) / TPW;                                             \
        if (a.solve_mode == ISLS_SOLVE_CHOL) LAUNCH_M(NX_, NU_, ISLS_SOLVE_CHOL)            \
        else LAUNCH_M(NX_, NU_, ISLS_SOLVE_INV)                                             \
    }
    ISLS_DISPATCH_DIMS(a.n, a.m, CALL)
#undef CALL
#undef LAUNCH_M
#undef LAUNCH_GC
#undef LAUNCH_GL
#undef LAUNCH_G
    if (did_ff) *did_ff = with_ff;
    return check_launch();
}
template int launch_gain<double>(const isls_gain_args &, hipStream_t, const isls_ff_args *, bool *, bool);
template int launch_gain<float>(const isls_gain_args &, hipStream_t, const isls_ff_args *, bool *, bool);


}  // namespace isls
