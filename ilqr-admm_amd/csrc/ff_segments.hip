// ff_segments.hip -- operators and stitching of the time-parallel feed-forward pass (isls_ffseg, include/isls_hip.h).
//
// The v/k recursion of iSLS.backward_pass_DP (isls/isls.py:285-302) / SLS.solve_dp_ff (isls/sls.py:168-202) is
// affine in v:   qx = cx + A'v, qu = cu + B'v, k = -Quu^{-1} qu, v' = qx + K'qu + K'Quu k + Qux'k
// so with cx = cu = 0 it is a linear map  v' = Phi_t v,  k = Gamma_t v  that depends on the gain pass only.
//   ff_prepare_kernel  pushes the n unit vectors through that homogeneous recursion over every segment but the
//                      last (one lane per unit vector; the step's matrices are staged once per (trajectory, segment)
//                      slot in LDS and read as broadcasts):  G_t = Gamma_t Phi_{t+1}..Phi_{t1-1}, Psi_s = Phi_{t0}..Phi_{t1-1}.
//   ff_stitch_kernel   after the segments of riccati_ff_kernel ran from v_in = 0: v_in(s-1) = v0(t0_s) + Psi_s v_in(s),
//                      then k_t += G_t v_in(seg(t)) for every step outside the last segment (HBM-bound stream over G).
#include "isls_common.hpp"

namespace isls {

constexpr int kPrepDepth = 2;     // steps of operands in flight per lane (ff_prepare_kernel)
constexpr int kMaxFfSeg = 16;     // segments the stitch kernel has LDS for
constexpr int kStitchTraj = 4;    // trajectories per stitch workgroup

int ff_segments(int N, int nseg_req, int *seg_len)
{
    const int steps = N - 1;                                   // recursion steps t = N-2 .. 0
    if (steps < 1 || nseg_req < 2) {
        if (seg_len) *seg_len = steps > 0 ? steps : 1;
        return 1;
    }
    if (nseg_req > kMaxFfSeg) nseg_req = kMaxFfSeg;
    if (nseg_req > steps) nseg_req = steps;
    const int len = (steps + nseg_req - 1) / nseg_req;
    if (seg_len) *seg_len = len;
    return (steps + len - 1) / len;
}

template <typename T>
struct FfPrepP {
    int B, N, mode, tpw, nseg, seg_len;
    View<T> A, Bm;
    const T *K, *Quu, *fac, *Qux;
    const T *rec;                  // packed step records of the gain pass (nullable): the operators come from those instead
    T *G, *Psi;
    const int32_t *active;
};

template <typename T, int NX, int NU, int D>
__global__ __launch_bounds__(64) void ff_prepare_kernel(FfPrepP<T> p)
{
    constexpr int W = NX + NU, GP = NX, MAXTPW = kWave / GP;
    // slot record (elements): AB[NX][W] | K[NU][NX] | Qux[NU][NX] | Quu[NU][NU] | fac[NU][NU] | dump
    constexpr int AB_OFF = 0, K_OFF = AB_OFF + NX * W, QUX_OFF = K_OFF + NU * NX, QUU_OFF = QUX_OFF + NU * NX,
                  FAC_OFF = QUU_OFF + NU * NU, DUMP_OFF = FAC_OFF + NU * NU;
    constexpr int RECP = ((DUMP_OFF + 1) | 1);
    constexpr int JU = (NU * NU + GP - 1) / GP;                // A, B, K, Qux split evenly: NX, NU, NU, NU per lane
    __shared__ T lds[MAXTPW * 2 * RECP];

    const int TPW = p.tpw, N = p.N, SL = p.seg_len, NS1 = p.nseg - 1;   // NS1 segments carry a correction
    const int lane = threadIdx.x;
    const int s = (lane / GP < TPW) ? lane / GP : TPW - 1;     // surplus lanes ride along in the last slot (stage nothing)
    const int j = lane - s * GP;                               // unit vector / column handled by this lane
    const bool extra = j >= GP;
    const int64_t Q = (int64_t)p.B * NS1;
    const int64_t q0 = (int64_t)blockIdx.x * TPW, q = q0 + s;
    const bool inrange = q < Q && !extra;
    const int64_t qq = q < Q ? q : q0;                         // idle slots shadow the block's first slot (loads only)
    const int b = (int)(qq / NS1), sg = (int)(qq - (int64_t)b * NS1);
    const int b0 = (int)(q0 / NS1);
    const bool valid = inrange && (p.active == nullptr || p.active[b] != 0);
    const int t_first = sg * SL + SL - 1;                      // steps t_first, t_first-1, .. sg*SL
    const int jl = extra ? 0 : j;
    T *recs = lds + s * 2 * RECP;

    // ---- load plan: uniform bases that move one step back per iteration + per-lane 32-bit offsets ----------
    const T *bA = p.A.at(b0, 0), *bB = p.Bm.at(b0, 0);
    const T *bK = p.K + (int64_t)b0 * N * NU * NX, *bQ = p.Qux + (int64_t)b0 * N * NU * NX;
    const T *bU = p.Quu + (int64_t)b0 * N * NU * NU, *bF = p.fac + (int64_t)b0 * N * NU * NU;
    const int db = b - b0;
    const uint32_t oA = (uint32_t)(db * p.A.sb + (int64_t)t_first * p.A.st) + jl;
    const uint32_t oB = (uint32_t)(db * p.Bm.sb + (int64_t)t_first * p.Bm.st) + jl;
    const uint32_t oK = (uint32_t)((db * N + t_first) * NU * NX) + jl;
    const uint32_t oU = (uint32_t)((db * N + t_first) * NU * NU);
    int dB[NU], dU[JU];
    uint32_t eU[JU];
#pragma unroll
    for (int a = 0; a < NU; ++a) { const int e = jl + GP * a; dB[a] = extra ? DUMP_OFF : AB_OFF + (e / NU) * W + NX + (e % NU); }
#pragma unroll
    for (int a = 0; a < JU; ++a) {
        const int e = jl + GP * a;
        dU[a] = (!extra && e < NU * NU) ? e : -1;
        eU[a] = oU + (uint32_t)(e < NU * NU ? e : NU * NU - 1);
    }
    const int dstA = extra ? DUMP_OFF : AB_OFF + jl, dstK = extra ? DUMP_OFF : K_OFF + jl, dstQ = extra ? DUMP_OFF : QUX_OFF + jl;
    const int strideA = extra ? 0 : W, strideK = extra ? 0 : GP;

    T ra[D][NX], rb[D][NU], rk[D][NU], rq[D][NU], ru[D][JU], rf[D][JU];
    const int64_t stA = p.A.st, stB = p.Bm.st;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int it = d < SL ? d : SL - 1;
        const T *a = bA - it * stA, *bm = bB - it * stB;
        const T *kk = bK - (int64_t)it * (NU * NX), *qx = bQ - (int64_t)it * (NU * NX);
        const T *uu = bU - (int64_t)it * (NU * NU), *ff = bF - (int64_t)it * (NU * NU);
#pragma unroll
        for (int e = 0; e < NX; ++e) ra[d][e] = a[oA + GP * e];
#pragma unroll
        for (int e = 0; e < NU; ++e) { rb[d][e] = bm[oB + GP * e]; rk[d][e] = kk[oK + GP * e]; rq[d][e] = qx[oK + GP * e]; }
#pragma unroll
        for (int e = 0; e < JU; ++e) { ru[d][e] = uu[eU[e]]; rf[d][e] = ff[eU[e]]; }
    }

    T psi[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) psi[i] = (i == jl) ? T(1) : T(0);
    T *gout = p.G + (((int64_t)b * N + t_first) * NU) * NX + jl;
    const int mode = p.mode;

    // groups of D steps with the ring index a compile-time constant; the last group is padded with dead steps
    // (loads clamped, nothing stored, psi kept), so every VMEM instruction of the loop is unconditional
    for (int itb = 0; itb < SL; itb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int it = itb + d;
            const bool live = it < SL;
            T *rec = recs + (it & 1) * RECP;
#pragma unroll
            for (int e = 0; e < NX; ++e) rec[dstA + strideA * e] = ra[d][e];          // A[e][jl]
#pragma unroll
            for (int e = 0; e < NU; ++e) { rec[dB[e]] = rb[d][e]; rec[dstK + strideK * e] = rk[d][e]; rec[dstQ + strideK * e] = rq[d][e]; }
#pragma unroll
            for (int e = 0; e < JU; ++e) { rec[dU[e] >= 0 ? QUU_OFF + dU[e] : DUMP_OFF] = ru[d][e]; rec[dU[e] >= 0 ? FAC_OFF + dU[e] : DUMP_OFF] = rf[d][e]; }
            slot_sync();
            {   // refill this ring entry (clamped, unconditional)
                const int itn = it + D < SL ? it + D : SL - 1;
                const T *a = bA - itn * stA, *bm = bB - itn * stB;
                const T *kk = bK - (int64_t)itn * (NU * NX), *qx = bQ - (int64_t)itn * (NU * NX);
                const T *uu = bU - (int64_t)itn * (NU * NU), *ff = bF - (int64_t)itn * (NU * NU);
#pragma unroll
                for (int e = 0; e < NX; ++e) ra[d][e] = a[oA + GP * e];
#pragma unroll
                for (int e = 0; e < NU; ++e) { rb[d][e] = bm[oB + GP * e]; rk[d][e] = kk[oK + GP * e]; rq[d][e] = qx[oK + GP * e]; }
#pragma unroll
                for (int e = 0; e < JU; ++e) { ru[d][e] = uu[eU[e]]; rf[d][e] = ff[eU[e]]; }
            }
            // homogeneous step on this lane's vector (every operand is an LDS broadcast)
            T qx[NX], qu[NU], kap[NU], y[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < NX; ++k) acc += rec[AB_OFF + k * W + i] * psi[k];
                qx[i] = acc;
            }
#pragma unroll
            for (int r = 0; r < NU; ++r) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < NX; ++k) acc += rec[AB_OFF + k * W + NX + r] * psi[k];
                qu[r] = acc;
            }
            if (mode == ISLS_SOLVE_CHOL) {
                T U[NU][NU], rd[NU], x[NU];
#pragma unroll
                for (int r = 0; r < NU; ++r) {
#pragma unroll
                    for (int c = 0; c < NU; ++c) U[r][c] = rec[FAC_OFF + r * NU + c];
                    rd[r] = U[r][r];
                }
                chol_solve<NU>(U, rd, qu, x);
#pragma unroll
                for (int r = 0; r < NU; ++r) kap[r] = -x[r];
            } else {
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    T acc = T(0);
#pragma unroll
                    for (int c = 0; c < NU; ++c) acc += rec[FAC_OFF + r * NU + c] * qu[c];
                    kap[r] = -acc;
                }
            }
#pragma unroll
            for (int r = 0; r < NU; ++r) {                     // y = qu + Quu kappa   (K'qu + K'Quu k = K'y)
                T acc = qu[r];
#pragma unroll
                for (int c = 0; c < NU; ++c) acc += rec[QUU_OFF + r * NU + c] * kap[c];
                y[r] = acc;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                T acc = qx[i];
#pragma unroll
                for (int r = 0; r < NU; ++r) acc += rec[K_OFF + r * NX + i] * y[r];
#pragma unroll
                for (int r = 0; r < NU; ++r) acc += rec[QUX_OFF + r * NX + i] * kap[r];
                psi[i] = live ? acc : psi[i];
            }
            if (valid && live) {
#pragma unroll
                for (int r = 0; r < NU; ++r) gout[r * NX] = kap[r];            // G_t[:, j]
            }
            gout -= NU * NX;
        }
    }
    if (valid && sg >= 1) {
        T *po = p.Psi + (((int64_t)b * p.nseg + sg) * NX) * NX + jl;
#pragma unroll
        for (int i = 0; i < NX; ++i) po[i * NX] = psi[i];                      // Psi_s[:, j]
    }
}

// The same operators from the packed step records [A+BK | B | K | fac] of the gain pass (isls_gain_args.rec): the
// homogeneous recursion is  psi' = (A + B K)' psi,  kappa = -Quu^-1 B' psi  -- 63 multiply-adds and one 81-word record per
// step against 126 and six arrays (108 words) in the array form above; same slots (trajectory, segment), one lane per unit
// vector.  Records are blocked by wavefront of the gain pass: [b / TR][t][b % TR][RW], TR = 64 / (n + m).
template <typename T, int NX, int NU, int D>
__global__ __launch_bounds__(64) void ff_prepare_rec_kernel(FfPrepP<T> p)
{
    constexpr int GP = NX, MAXTPW = kWave / GP, TR = kWave / (NX + NU);
    constexpr int PHI_OFF = 0, B_OFF = NX * NX, FAC_OFF = B_OFF + NX * NU + NU * NX, RW = FAC_OFF + NU * NU, DUMP_OFF = RW;
    constexpr int RS = rec_stride(NX, NU);                     // words between the records of consecutive trajectories
    constexpr int RECP = ((DUMP_OFF + 1) | 1), JR = (RW + GP - 1) / GP;
    __shared__ T lds[MAXTPW * 2 * RECP];

    const int TPW = p.tpw, N = p.N, SL = p.seg_len, NS1 = p.nseg - 1;
    const int lane = threadIdx.x;
    const int s = (lane / GP < TPW) ? lane / GP : TPW - 1;     // surplus lanes ride along in the last slot (stage nothing)
    const int j = lane - s * GP;
    const bool extra = j >= GP;
    const int64_t Q = (int64_t)p.B * NS1;
    const int64_t q0 = (int64_t)blockIdx.x * TPW, q = q0 + s;
    const bool inrange = q < Q && !extra;
    const int64_t qq = q < Q ? q : q0;                         // idle slots shadow the block's first slot (loads only)
    const int b = (int)(qq / NS1), sg = (int)(qq - (int64_t)b * NS1);
    const bool valid = inrange && (p.active == nullptr || p.active[b] != 0);
    const int t_first = sg * SL + SL - 1;                      // steps t_first, t_first-1, .. sg*SL
    const int jl = extra ? 0 : j;
    T *recs = lds + s * 2 * RECP;
    const T *base = p.rec + (((int64_t)(b / TR) * N + t_first) * TR + (b % TR)) * RS;    // record (b, t_first); t-1 is TR*RS below
    int off[JR], dst[JR];
#pragma unroll
    for (int a = 0; a < JR; ++a) {
        const int e = jl + GP * a;
        off[a] = e < RW ? e : RW - 1;
        dst[a] = (!extra && e < RW) ? e : DUMP_OFF;
    }
    T rr[D][JR];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int it = d < SL ? d : SL - 1;
#pragma unroll
        for (int a = 0; a < JR; ++a) rr[d][a] = base[off[a] - (int64_t)it * (TR * RS)];
    }
    T psi[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) psi[i] = (i == jl) ? T(1) : T(0);
    T *gout = p.G + (((int64_t)b * N + t_first) * NU) * NX + jl;
    const int mode = p.mode;
    for (int itb = 0; itb < SL; itb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int it = itb + d;
            const bool live = it < SL;
            T *rec = recs + (it & 1) * RECP;
#pragma unroll
            for (int a = 0; a < JR; ++a) rec[dst[a]] = rr[d][a];
            slot_sync();
            {
                const int itn = it + D < SL ? it + D : SL - 1;  // refill (clamped, unconditional)
#pragma unroll
                for (int a = 0; a < JR; ++a) rr[d][a] = base[off[a] - (int64_t)itn * (TR * RS)];
            }
            T pn[NX], qu[NU], kap[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < NX; ++k) acc += rec[PHI_OFF + k * NX + i] * psi[k];
                pn[i] = acc;
            }
#pragma unroll
            for (int r = 0; r < NU; ++r) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < NX; ++k) acc += rec[B_OFF + k * NU + r] * psi[k];
                qu[r] = acc;
            }
            if (mode == ISLS_SOLVE_CHOL) {
                T U[NU][NU], rd[NU], x[NU];
#pragma unroll
                for (int r = 0; r < NU; ++r) {
#pragma unroll
                    for (int c = 0; c < NU; ++c) U[r][c] = rec[FAC_OFF + r * NU + c];
                    rd[r] = U[r][r];
                }
                chol_solve<NU>(U, rd, qu, x);
#pragma unroll
                for (int r = 0; r < NU; ++r) kap[r] = -x[r];
            } else {
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    T acc = T(0);
#pragma unroll
                    for (int c = 0; c < NU; ++c) acc += rec[FAC_OFF + r * NU + c] * qu[c];
                    kap[r] = -acc;
                }
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) psi[i] = live ? pn[i] : psi[i];
            if (valid && live) {
#pragma unroll
                for (int r = 0; r < NU; ++r) gout[r * NX] = kap[r];            // G_t[:, j]
            }
            gout -= NU * NX;
        }
    }
    if (valid && sg >= 1) {
        T *po = p.Psi + (((int64_t)b * p.nseg + sg) * NX) * NX + jl;
#pragma unroll
        for (int i = 0; i < NX; ++i) po[i * NX] = psi[i];                      // Psi_s[:, j]
    }
}

template <typename T, int NX, int NU>
__global__ __launch_bounds__(256) void ff_stitch_kernel(int B, int N, int nseg, int SL, const T *__restrict__ G,
                                                        const T *__restrict__ Psi, const T *__restrict__ vseg,
                                                        T *__restrict__ k, const int32_t *__restrict__ active)
{
    constexpr int TB = kStitchTraj;
    __shared__ T vin[TB][kMaxFfSeg][NX];                       // vin[.][s] = v entering segment s (s <= nseg-2)
    // feedback column blockIdx.y (isls_ff_args.ncol): its own segment-start values and k; G and Psi are shared by the columns
    vseg += (int64_t)blockIdx.y * B * nseg * NX;
    k += (int64_t)blockIdx.y * B * N * NU;
    const int tid = threadIdx.x;
    const int tr = tid / NX, i = tid - tr * NX;
    const int b = blockIdx.x * TB + tr;
    const bool row = tr < TB && b < B;
    // ---- true segment inputs: the last segment is exact, the others chain through Psi -------------------
    if (row) vin[tr][nseg - 2][i] = vseg[((int64_t)b * nseg + nseg - 1) * NX + i];
    __syncthreads();
    for (int s = nseg - 2; s >= 1; --s) {
        if (row) {
            const T *ps = Psi + (((int64_t)b * nseg + s) * NX + i) * NX;
            T acc = vseg[((int64_t)b * nseg + s) * NX + i];
#pragma unroll
            for (int j = 0; j < NX; ++j) acc += ps[j] * vin[tr][s][j];
            vin[tr][s - 1][i] = acc;
        }
        __syncthreads();
    }
    // ---- k_t += G_t v_in(seg(t)) for the steps outside the last segment ------------------------------
    const int TC = (nseg - 1) * SL, per = TC * NU;
    for (int e = tid; e < TB * per; e += 256) {
        const int tr2 = e / per, rem = e - tr2 * per;
        const int t = rem / NU;
        const int b2 = blockIdx.x * TB + tr2;
        if (b2 >= B || (active != nullptr && active[b2] == 0)) continue;
        const int64_t o = (int64_t)b2 * N * NU + rem;          // (b2, t, r) of k ; G row follows at o*NX
        const T *g = G + o * NX, *vv = vin[tr2][t / SL];
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < NX; ++j) acc += g[j] * vv[j];
        k[o] += acc;
    }
}

static bool seg_ok(const isls_ffseg &sg, int N)
{
    if (sg.nseg < 2 || sg.nseg > kMaxFfSeg || sg.seg_len < 1 || !sg.G || !sg.Psi || !sg.v) return false;
    const int steps = N - 1;
    return (int64_t)sg.nseg * sg.seg_len >= steps && (int64_t)(sg.nseg - 1) * sg.seg_len < steps;
}

bool ff_seg_enabled(const isls_ffseg &sg) { return sg.nseg >= 2 && sg.G != nullptr; }

template <typename T>
int launch_ff_prepare(const isls_ff_prepare_args &a, hipStream_t s)
{
    if (a.B < 0 || a.N < 1) return ISLS_ERR_ARG;
    if (!a.rec && (!a.A.p || !a.Bm.p || !a.K || !a.Quu || !a.fac || !a.Qux)) return ISLS_ERR_ARG;
    if (a.solve_mode != ISLS_SOLVE_CHOL && a.solve_mode != ISLS_SOLVE_INV) return ISLS_ERR_ARG;
    if (!ff_seg_enabled(a.seg)) return ISLS_OK;                // sequential recursion: nothing to prepare
    if (!seg_ok(a.seg, a.N)) return ISLS_ERR_ARG;
    if (a.rec) {                                               // operators from the packed records of the gain pass
        if (a.B == 0) return ISLS_OK;
        FfPrepP<T> pr = {};
        pr.B = a.B; pr.N = a.N; pr.mode = a.solve_mode; pr.nseg = a.seg.nseg; pr.seg_len = a.seg.seg_len;
        pr.rec = (const T *)a.rec; pr.G = (T *)a.seg.G; pr.Psi = (T *)a.seg.Psi; pr.active = a.active;
        const int64_t Qr = (int64_t)a.B * (a.seg.nseg - 1);
#define CALLR(NX_, NU_)                                                                                      \
    {                                                                                                        \
        pr.tpw = kWave / NX_;                                                                                \
        const int grid = (int)((Qr + pr.tpw - 1) / pr.tpw);                                                  \
        hipLaunchKernelGGL((ff_prepare_rec_kernel<T, NX_, NU_, kPrepDepth>), dim3(grid), dim3(64), 0, s, pr); \
    }
        ISLS_DISPATCH_DIMS(a.n, a.m, CALLR)
#undef CALLR
        return check_launch();
    }
    if ((int64_t)a.N * a.n * a.n * 64 >= ((int64_t)1 << 31) || a.A.sb * 64 + a.A.st * a.N >= ((int64_t)1 << 31) ||
        a.Bm.sb * 64 + a.Bm.st * a.N >= ((int64_t)1 << 31))
        return ISLS_ERR_UNSUPPORTED;
    if (a.B == 0) return ISLS_OK;
    FfPrepP<T> p;
    p.B = a.B; p.N = a.N; p.mode = a.solve_mode; p.nseg = a.seg.nseg; p.seg_len = a.seg.seg_len;
    p.A = View<T>(a.A); p.Bm = View<T>(a.Bm);
    p.K = (const T *)a.K; p.Quu = (const T *)a.Quu; p.fac = (const T *)a.fac; p.Qux = (const T *)a.Qux; p.rec = nullptr;
    p.G = (T *)a.seg.G; p.Psi = (T *)a.seg.Psi; p.active = a.active;
    const int64_t Q = (int64_t)a.B * (a.seg.nseg - 1);
#define CALL(NX_, NU_)                                                                                    \
    {                                                                                                     \
        p.tpw = kWave / NX_;                                                                              \
        const int grid = (int)((Q + p.tpw - 1) / p.tpw);                                                  \
        hipLaunchKernelGGL((ff_prepare_kernel<T, NX_, NU_, kPrepDepth>), dim3(grid), dim3(64), 0, s, p);   \
    }
    ISLS_DISPATCH_DIMS(a.n, a.m, CALL)
#undef CALL
    return check_launch();
}
template int launch_ff_prepare<double>(const isls_ff_prepare_args &, hipStream_t);
template int launch_ff_prepare<float>(const isls_ff_prepare_args &, hipStream_t);

template <typename T>
int launch_ff_stitch(const isls_ff_args &a, hipStream_t s)
{
    if (!seg_ok(a.seg, a.N)) return ISLS_ERR_ARG;
    const int grid = (a.B + kStitchTraj - 1) / kStitchTraj;
#define CALL(NX_, NU_)                                                                                         \
    hipLaunchKernelGGL((ff_stitch_kernel<T, NX_, NU_>), dim3(grid, a._pad > 1 ? a._pad : 1), dim3(256), 0, s, a.B, a.N, a.seg.nseg, \
                       a.seg.seg_len, (const T *)a.seg.G, (const T *)a.seg.Psi, (const T *)a.seg.v, (T *)a.k, a.active);
    ISLS_DISPATCH_DIMS(a.n, a.m, CALL)
#undef CALL
    return check_launch();
}
template int launch_ff_stitch<double>(const isls_ff_args &, hipStream_t);
template int launch_ff_stitch<float>(const isls_ff_args &, hipStream_t);

}  // namespace isls
