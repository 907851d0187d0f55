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
};

template <typename T, int NX, int NU, int D>
__global__ __launch_bounds__(64) void riccati_gain_kernel(GainP<T> p)
{
    constexpr int G = NX + NU, W = NX + NU, TPW = kWave / G;
    constexpr int V_OFF = 0, AB_OFF = V_OFF + NX * NX, Q_OFF = AB_OFF + NX * W, K_OFF = Q_OFF + NU * W;
    constexpr int DUMP_OFF = K_OFF + NU * NX;              // W words that absorb the LDS writes of lanes with nothing to publish
    constexpr int SLOT = ((DUMP_OFF + W) | 1);             // odd stride: slots start on different banks
    constexpr int JA = (NX * NX + G - 1) / G, JB = (NX * NU + G - 1) / G, JQ = (NU * W + G - 1) / G;
    constexpr int RB = NX * NX, RK = RB + NX * NU, RFAC = RK + NU * NX, RW = rec_stride(NX, NU);   // packed record (padded stride), see riccati_ffrec.hip
    __shared__ T lds[TPW * SLOT];

    const int lane = threadIdx.x;
    const int s = lane / G, i = lane - s * G;
    const int b = blockIdx.x * TPW + s;
    const bool inslot = s < TPW;
    const bool valid = inslot && b < p.B && (p.active == nullptr || p.active[b] != 0);
    const int N = p.N;
    const int bb = valid ? b : 0;
    T *rec = lds + (inslot ? s : TPW - 1) * SLOT;
    T *Vs = rec + V_OFF, *ABs = rec + AB_OFF, *Qs = rec + Q_OFF, *Ks = rec + K_OFF;
    const bool xl = i < NX;                                   // lane owns a row of Qxx
    const int a_row = xl ? 0 : i - NX;                        // row of [Qux Quu] for u-lanes
    const int64_t bN = (int64_t)bb * N;
    // LDS destinations of everything a lane publishes, fixed for the whole horizon: lanes (or elements) with nothing to
    // publish point at the dump words, so no ds_write of the step loop sits behind an exec-mask branch
    int dA[JA], dB[JB];
#pragma unroll
    for (int j = 0; j < JA; ++j) { const int e = i + G * j; dA[j] = (valid && e < NX * NX) ? AB_OFF + (e / NX) * W + (e % NX) : DUMP_OFF; }
#pragma unroll
    for (int j = 0; j < JB; ++j) { const int e = i + G * j; dB[j] = (valid && e < NX * NU) ? AB_OFF + (e / NU) * W + NX + (e % NU) : DUMP_OFF; }
    const int qdst = (!xl && valid) ? Q_OFF + a_row * W : DUMP_OFF;          // row of [Qux Quu]
    const int kdst = (xl && valid) ? K_OFF + i : DUMP_OFF, kstr = (xl && valid) ? NX : 0;   // column i of K
    const int vdst = (xl && valid) ? V_OFF + i * NX : DUMP_OFF;              // row i of V

    // ---- terminal step: K[N-1] = 0 (isls.py:245), V = Cxx[N-1] (isls.py:251/257) -----------------
    {
        T r[JA];
        coop_load<NX * NX, G>(p.Cxx.at(bb, N - 1), r, i, valid);
        coop_put<NX * NX, G>(Vs, r, i, valid);
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
    const bool has_cux = p.Cux.p != nullptr;
    struct Stage {
        T ra[JA], rb[JB], crow[W];
    };
    Stage ring[D];
    auto fetch = [&](int t, Stage &g) {
        coop_load<NX * NX, G>(p.A.at(bb, t), g.ra, i, valid);
        coop_load<NX * NU, G>(p.Bm.at(bb, t), g.rb, i, valid);
        // row i of the cost Hessian stack: [Cxx[i,:]] or [Cux[a,:] Cuu[a,:]]
        // Raw, unconditional loads through per-lane pointers (no branch, no arithmetic on the results here).
        // x-lanes: columns >= NX of crow are never used; u-lanes without a Cux array read Cxx instead and
        // the step zeroes that part when it consumes the row.
        const T *cl = xl ? p.Cxx.at(bb, t) + i * NX : (has_cux ? p.Cux.at(bb, t) + a_row * NX : p.Cxx.at(bb, t));
        const T *cr = xl ? cl : p.Cuu.at(bb, t) + a_row * NU;
#pragma unroll
        for (int j = 0; j < NX; ++j) g.crow[j] = cl[j];
#pragma unroll
        for (int j = 0; j < NU; ++j) g.crow[NX + j] = cr[j];
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        fetch(N - 2 - d > 0 ? N - 2 - d : 0, ring[d]);          // unconditional (clamped): exact vmcnt bookkeeping
    bool pd_ok = true;

    auto step = [&](int t, Stage &g) {
        // stage [A_t B_t] into the record: row k = [A[k,:] B[k,:]]
#pragma unroll
        for (int j = 0; j < JA; ++j) rec[dA[j]] = g.ra[j];
#pragma unroll
        for (int j = 0; j < JB; ++j) rec[dB[j]] = g.rb[j];
        T c_now[W];
#pragma unroll
        for (int j = 0; j < W; ++j) c_now[j] = (j < NX && !xl && !has_cux) ? T(0) : g.crow[j];   // Cux absent -> 0
        slot_sync();                                          // (a) ABs, Vs visible to the slot
        fetch(t - D > 0 ? t - D : 0, g);                       // refill this ring entry (clamped, unconditional)

        // (1) S = row i of [A B]'V
        T S[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) S[j] = T(0);
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const T col = ABs[k * W + i];
#pragma unroll
            for (int j = 0; j < NX; ++j) S[j] += col * Vs[k * NX + j];
        }
        // (2) M = C_row + S [A B]      (Qxx = Cxx + (A'V)A etc., isls.py:288-290)
        T M[W];
#pragma unroll
        for (int c = 0; c < W; ++c) M[c] = T(0);
#pragma unroll
        for (int k = 0; k < NX; ++k) {
#pragma unroll
            for (int c = 0; c < W; ++c) M[c] += S[k] * ABs[k * W + c];
        }
#pragma unroll
        for (int c = 0; c < W; ++c) M[c] = c_now[c] + M[c];
        // (3) u-lanes publish their row of [Qux Quu] (x-lanes write the dump words)
#pragma unroll
        for (int c = 0; c < W; ++c) rec[qdst + c] = M[c];
        slot_sync();                                          // (b)

        // (4) factor Quu (redundantly in every lane), solve for column i of K
        T Quu[NU][NU], U[NU][NU], rd[NU], rhs[NU], Kc[NU], inv[NU][NU];
#pragma unroll
        for (int r = 0; r < NU; ++r) {
#pragma unroll
            for (int c = 0; c < NU; ++c) { Quu[r][c] = Qs[r * W + NX + c]; U[r][c] = T(0); inv[r][c] = T(0); }
            rhs[r] = Qs[r * W + (xl ? i : 0)];                 // column i of Qux
        }
        pd_ok = chol_upper<NU>(Quu, U, rd) && pd_ok;
        if (p.mode == ISLS_SOLVE_CHOL) {
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
        // packed records are blocked by wavefront, [block][t][slot][RW]: the TPW trajectories of a workgroup write (and the
        // feed-forward pass reads) one contiguous burst per step
        const int64_t orec = ((int64_t)blockIdx.x * N + t) * TPW + s;
#pragma unroll
        for (int r = 0; r < NU; ++r) rec[kdst + r * kstr] = Kc[r];
        if (xl && valid) {
#pragma unroll
            for (int r = 0; r < NU; ++r) p.K[(o * NU + r) * NX + i] = Kc[r];
        }
        if (valid && p.Qux) {
            // cooperative store of [Qux Quu] rows (skipped when the caller only wants K, fac and the packed records: every
            // consumer of Qux / Quu then reads the records instead)
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
        }
        if (valid) {
            // factor: row i written by lane i < NU (the values are identical in every lane; the row is picked by selects)
            if (i < NU) {
#pragma unroll
                for (int c = 0; c < NU; ++c) {
                    T v = T(0);
#pragma unroll
                    for (int r = 0; r < NU; ++r) {
                        const T vr = (p.mode == ISLS_SOLVE_CHOL) ? ((c == r) ? rd[r] : (c > r ? U[r][c] : T(0))) : inv[r][c];
                        v = (i == r) ? vr : v;
                    }
                    if (p.fac) p.fac[(o * NU + i) * NU + c] = v;
                    if (p.rec) p.rec[orec * RW + RFAC + i * NU + c] = v;
                }
            }
            // packed record of the feed-forward pass (riccati_ffrec.hip): closed-loop matrix Phi = A + B K (lane i owns
            // column i, like its column of K), then B, K as they are
            if (p.rec) {
                T *ro = p.rec + orec * RW;
                if (xl) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) {
                        T ph = ABs[k * W + i];
#pragma unroll
                        for (int r = 0; r < NU; ++r) ph += ABs[k * W + NX + r] * Kc[r];
                        ro[k * NX + i] = ph;
                    }
#pragma unroll
                    for (int r = 0; r < NU; ++r) ro[RK + r * NX + i] = Kc[r];
                }
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                    const int e = i + G * j;
                    if (e < NX * NU) ro[RB + e] = ABs[(e / NU) * W + NX + (e % NU)];
                }
            }
        }
        slot_sync();                                          // (c) Ks visible

        // (5) V row i = Qxx + (K'Quu)K + Qux'K + K'Qux   (isls.py:300 / sls.py:153); u-lanes run the same instructions
        // on their clamped column and write the dump words
        {
            T Wr[NU];
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                T sacc = T(0);
#pragma unroll
                for (int r = 0; r < NU; ++r) sacc += Kc[r] * Quu[r][c];
                Wr[c] = sacc;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                T t1 = T(0), t2 = T(0), t3 = T(0);
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    const T kj = Ks[r * NX + j];
                    t1 += Wr[r] * kj;                          // (K'Quu) K
                    t2 += rhs[r] * kj;                         // Qux' K
                    t3 += Kc[r] * Qs[r * W + j];               // K' Qux
                }
                const T vn = (p.mode == ISLS_SOLVE_CHOL) ? ((M[j] + t1) + t2) + t3 : ((M[j] + t2) + t3) + t1;
                rec[vdst + j] = vn;
            }
        }
    };

    // full groups of D steps run branch-free (every VMEM op of the steady state is unconditional, so the
    // compiler's vmcnt bookkeeping is exact and D steps of loads really stay in flight); then the remainder
    int tb = N - 2;
    for (; tb - (D - 1) >= 0; tb -= D) {
#pragma unroll
        for (int d = 0; d < D; ++d) step(tb - d, ring[d]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (tb - d >= 0) step(tb - d, ring[d]);
    if (valid && i == 0 && !pd_ok && p.status) atomicOr(&p.status[b], ISLS_ST_NOT_PD);
}

template <typename T>
int launch_gain(const isls_gain_args &a, hipStream_t s)
{
    if (a.B < 0 || a.N < 1 || !a.A.p || !a.Bm.p || !a.Cxx.p || !a.Cuu.p || !a.K) return ISLS_ERR_ARG;
    // Quu, fac, Qux: all three or none; none only with records (every consumer then reads those)
    if ((!a.Quu || !a.Qux || !a.fac) && (!a.rec || a.Quu || a.Qux || a.fac)) return ISLS_ERR_ARG;
    if (a.solve_mode != ISLS_SOLVE_CHOL && a.solve_mode != ISLS_SOLVE_INV) return ISLS_ERR_ARG;
    if (a.B == 0) return ISLS_OK;
    GainP<T> p;
    p.B = a.B; p.N = a.N; p.mode = a.solve_mode;
    p.A = View<T>(a.A); p.Bm = View<T>(a.Bm); p.Cxx = View<T>(a.Cxx); p.Cuu = View<T>(a.Cuu); p.Cux = View<T>(a.Cux);
    p.K = (T *)a.K; p.Quu = (T *)a.Quu; p.fac = (T *)a.fac; p.Qux = (T *)a.Qux; p.rec = (T *)a.rec;
    p.status = a.status; p.active = a.active;
#define CALL(NX_, NU_)                                                                                     \
    {                                                                                                      \
        constexpr int TPW = kWave / (NX_ + NU_);                                                           \
        const int grid = (a.B + TPW - 1) / TPW;                                                            \
        hipLaunchKernelGGL((riccati_gain_kernel<T, NX_, NU_, kGainDepth>), dim3(grid), dim3(64), 0, s, p);  \
    }
    ISLS_DISPATCH_DIMS(a.n, a.m, CALL)
#undef CALL
    return check_launch();
}
template int launch_gain<double>(const isls_gain_args &, hipStream_t);
template int launch_gain<float>(const isls_gain_args &, hipStream_t);


}  // namespace isls
