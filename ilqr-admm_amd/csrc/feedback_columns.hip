// feedback_columns.hip -- x-step roll-out and z/dual bookkeeping of the feedback columns [d, phi] of
// iSLS.isls_admm (isls/isls.py:503-712) in DP form; see isls_columns_args in include/isls_hip.h.
//
// Work shape: B problems x C = 1 + dim columns (C <= 4), N dependent steps of an n x (n + m) mat-vec each.  The
// columns of one problem share A_t, B_t, K_t, so they sit in neighbouring lanes (column index fastest) and the
// per-step operands are fetched once per cache line for the whole group.  The pass runs once per ADMM iteration
// next to C feed-forward passes and one line-search rollout and is a small fraction of them.
#include "isls_common.hpp"

namespace isls {

template <typename T>
struct ColP {
    int B, N, C;
    View<T> A, Bm, Cuu, c0u, Rr;
    const T *K, *k, *zu, *lu;
    T *dx, *du;
    const int32_t *active;
};

// One wavefront per workgroup: PB = 4 problems x C columns (lane = problem slot * C + column; the other lanes only help
// fetching).  The pass is bound by the latency of N dependent steps, each needing [K_t | A_t | B_t] of its problems from
// HBM: the operands are fetched cooperatively (consecutive lanes, consecutive words, wave-uniform base addresses) into a
// register ring kColDepth steps ahead and handed over through LDS, where the C lanes of a problem read them back as
// broadcasts.  Few problems per wavefront keep the ring small and put >= B / 4 wavefronts on the chip.  (Measured, B = 1024,
// n = 9, C = 4: a lane-private gather of the operands 296 us, 16 problems per wavefront one step ahead 352 us.)
constexpr int kColDepth = 4;

template <typename T, int NX, int NU, int C>
__global__ __launch_bounds__(64) void columns_rollout_kernel(ColP<T> p)
{
    constexpr int PB = 4, D = kColDepth;
    constexpr int OK_ = 0, OA = NU * NX, OB = OA + NX * NX, W = OB + NX * NU;       // words per problem and step
    constexpr int WP = W | 1;                                                         // odd stride: conflict-free broadcasts
    constexpr int CH = (W + 63) / 64;                                                 // 64-word chunks per problem
    __shared__ T lds[PB * WP];
    const int lane = threadIdx.x;
    const int slot = lane / C, c = lane - slot * C;
    const int b0 = blockIdx.x * PB;
    const int b = b0 + slot;
    const bool live = slot < PB && b < p.B && (!p.active || p.active[b]);
    const int N = p.N;
    // staging plan of this lane, the same for every problem: word w = lane + 64 i of [K_t | A_t | B_t]
    int woff[CH], kind[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int w = lane + 64 * i < W ? lane + 64 * i : W - 1;                      // clamped: surplus lanes re-read the last word
        kind[i] = w < OA ? 0 : (w < OB ? 1 : 2);
        woff[i] = w - (w < OA ? 0 : (w < OB ? OA : OB));
    }
    T stage[D][PB][CH], kst[D][NU];
    const int64_t col = (int64_t)c * p.B + (live ? b : 0);
    T *dx = p.dx + col * N * NX, *du = p.du + col * N * NU;
    const T *kc = p.k + col * N * NU;
    const T *rec = lds + (slot < PB ? slot : 0) * WP;
    T x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = (c >= 1 && i == c - 1) ? T(1) : T(0);

    // every load below is unconditional: steps beyond N-2 re-read step N-2, dead problem slots re-read a valid problem
#define COL_FETCH(d_, t_)                                                                                              \
    {                                                                                                                   \
        const int tf = (t_) < N - 1 ? (t_) : N - 2;                                                                    \
        _Pragma("unroll") for (int q = 0; q < PB; ++q)                                                                 \
        {                                                                                                               \
            const int bq = b0 + q < p.B ? b0 + q : p.B - 1;                                                            \
            const T *bk = p.K + ((int64_t)bq * N + tf) * NU * NX, *ba = p.A.at(bq, tf), *bb = p.Bm.at(bq, tf);         \
            _Pragma("unroll") for (int i = 0; i < CH; ++i)                                                             \
                stage[d_][q][i] = (kind[i] == 0 ? bk : (kind[i] == 1 ? ba : bb))[woff[i]];                             \
        }                                                                                                               \
        _Pragma("unroll") for (int r = 0; r < NU; ++r) kst[d_][r] = kc[tf * NU + r];                                   \
    }
#define COL_STEP(d_)                                                                                                   \
    {                                                                                                                   \
        const int t = t0 + d_;                                                                                         \
        const bool on = t < N - 1;                                                                                     \
        slot_sync();                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < PB; ++q)                                                                 \
            _Pragma("unroll") for (int i = 0; i < CH; ++i)                                                             \
                if (lane + 64 * i < W) lds[q * WP + lane + 64 * i] = stage[d_][q][i];                                  \
        slot_sync();                                                                                                    \
        T u[NU], xn[NX];                                                                                                \
        _Pragma("unroll") for (int r = 0; r < NU; ++r) u[r] = kst[d_][r];                                              \
        COL_FETCH(d_, t + D)                                                                                           \
        _Pragma("unroll") for (int r = 0; r < NU; ++r)                                                                 \
        {                                                                                                               \
            T acc = u[r];                                                                                               \
            _Pragma("unroll") for (int j = 0; j < NX; ++j) acc += rec[OK_ + r * NX + j] * x[j];                        \
            u[r] = acc;                                                                                                 \
        }                                                                                                               \
        _Pragma("unroll") for (int a = 0; a < NX; ++a)                                                                 \
        {                                                                                                               \
            T acc = T(0);                                                                                               \
            _Pragma("unroll") for (int j = 0; j < NX; ++j) acc += rec[OA + a * NX + j] * x[j];                         \
            _Pragma("unroll") for (int r = 0; r < NU; ++r) acc += rec[OB + a * NU + r] * u[r];                         \
            xn[a] = acc;                                                                                                \
        }                                                                                                               \
        if (live && on) {                                                                                               \
            _Pragma("unroll") for (int r = 0; r < NU; ++r) du[t * NU + r] = u[r];                                      \
            _Pragma("unroll") for (int i = 0; i < NX; ++i) dx[t * NX + i] = x[i];                                      \
        }                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < NX; ++i) x[i] = on ? xn[i] : x[i];                                       \
    }
    COL_FETCH(0, 0)
    COL_FETCH(1, 1)
    COL_FETCH(2, 2)
    COL_FETCH(3, 3)
    static_assert(D == 4, "the step loop below is unrolled by hand for a ring of four");
    for (int t0 = 0; t0 < N - 1; t0 += D) {
        COL_STEP(0)
        COL_STEP(1)
        COL_STEP(2)
        COL_STEP(3)
    }
#undef COL_STEP
#undef COL_FETCH
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NX; ++i) dx[(N - 1) * NX + i] = x[i];
    // last control: minimiser of its own cost term  (R + Rr) du = R ud + Rr u_reg  (isls.py:560-571, last block row)
    T g[NU], H[NU][NU], U[NU][NU], rd[NU], y[NU];
    const T *Cl = p.Cuu.at(b, N - 1);
#pragma unroll
    for (int r = 0; r < NU; ++r) {
        g[r] = c == 0 ? -p.c0u.at(b, N - 1)[r] : T(0);
#pragma unroll
        for (int j = 0; j < NU; ++j) H[r][j] = Cl[r * NU + j];
    }
    if (p.Rr.p) {
        const T *Rl = p.Rr.at(b, N - 1);
        const T *z = p.zu + col * N * NU + (N - 1) * NU, *l = p.lu + col * N * NU + (N - 1) * NU;
#pragma unroll
        for (int r = 0; r < NU; ++r)
#pragma unroll
            for (int j = 0; j < NU; ++j) g[r] += T(2) * Rl[r * NU + j] * (z[j] - l[j]);
    }
    chol_upper<NU>(H, U, rd);
    chol_solve<NU>(U, rd, g, y);
#pragma unroll
    for (int r = 0; r < NU; ++r) du[(N - 1) * NU + r] = y[r];
}

// Row form of the same roll-out: lane = (problem, column, row of [K_t; A_t B_t]).  The column form above gives a lane the
// whole n x (n + m) mat-vec of a step (135 dependent multiply-adds at n = 9, every operand a broadcast LDS read), B C lanes
// in all -- 64 wavefronts for 1024 arms; here a lane owns ONE row: u-lane r forms u_r = k_r + K[r,:] x, x-lane a forms
// x'_a = A[a,:] x + B[a,:] u, each from its own row, which it fetches itself (a contiguous run per lane, D steps ahead),
// with x and u handed around through LDS twice per step.  n + m multiply-adds per lane and step, C (n + m) lanes per
// problem (one problem per wavefront at n = 9, C = 4: as many wavefronts as problems).  The sums run in the order of the
// column form: bit-identical results.
template <typename T, int NX, int NU, int C>
__global__ __launch_bounds__(64) void columns_rollout_rows_kernel(ColP<T> p)
{
    constexpr int W = NX + NU, GC = C * W, PB = kWave / GC, D = kColDepth;
    static_assert(PB >= 1, "a problem's lanes fit one wavefront");
    // per (problem slot, column): x[2][NX] (double buffer: x' is written while x is still being read) | u[NU]
    constexpr int SL = 2 * NX + NU;
    __shared__ T lds[(PB * C + 1) * SL];
    const int lane = threadIdx.x;
    const bool inslot = lane / GC < PB;
    const int slot = inslot ? lane / GC : PB - 1;
    const int li = inslot ? lane - slot * GC : GC - 1;         // surplus lanes repeat the last lane of the last slot
    const int c = li / W, i = li - c * W;
    const bool xl = i < NX;
    const int r = xl ? 0 : i - NX;
    const int b = blockIdx.x * PB + slot;
    const bool valid = b < p.B && (!p.active || p.active[b]);
    const unsigned long long vm = __ballot(valid);
    if (vm == 0ull) return;
    const int bsh = __builtin_amdgcn_readlane(b, __builtin_ctzll(vm));
    const int bb = valid ? b : bsh;                            // a slot without a problem repeats the first valid one (loads only)
    const int N = p.N;
    const int64_t col = (int64_t)c * p.B + bb;
    T *dx = p.dx + col * N * NX, *du = p.du + col * N * NU;
    // the lane's row: x-lane a -> A_t[a,:] (NX words) and B_t[a,:] (NU words); u-lane r -> K_t[r,:] and k_t[r] (then padding)
    const T *pa = xl ? p.A.at(bb, 0) + i * NX : p.K + (int64_t)bb * N * NU * NX + r * NX;
    const int64_t sa = xl ? p.A.st : NU * NX;
    const T *pbv = xl ? p.Bm.at(bb, 0) + i * NU : p.k + col * N * NU + r;
    const int64_t sbv = xl ? p.Bm.st : NU;
    const int nbv = xl ? NU : 1;                               // words of the second run this lane may read
    T *st = lds + (slot * C + c) * SL;
    T *const dump = lds + (PB * C) * SL;                       // two words for the publishes a lane must not make
    T ra[D][NX], rb[D][NU];
    auto fetch = [&](int tq, T (&fa)[NX], T (&fb)[NU]) {
        const int t = __builtin_amdgcn_readfirstlane(tq < N - 1 ? tq : N - 2);
        const T *qa = pa + (int64_t)t * sa, *qb = pbv + (int64_t)t * sbv;
#pragma unroll
        for (int j = 0; j < NX; ++j) fa[j] = qa[j];
#pragma unroll
        for (int q = 0; q < NU; ++q) fb[q] = qb[q < nbv ? q : nbv - 1];
    };
    // x_0: zero for column 0, the unit vector e_{c-1} for column c >= 1
    (xl ? st + i : dump)[0] = (xl && c >= 1 && i == c - 1) ? T(1) : T(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        fetch(d, ra[d], rb[d]);
        __builtin_amdgcn_sched_barrier(0);                     // issue order = consumption order
    }
    slot_sync();
    // every lane stores one word per step: x-lane a the component x_t[a], u-lane r the control u_t[r] (slots without a
    // problem store the shadowed problem's words again); dead steps past N-2 store into row N-1, which is rewritten below
    T *const po = xl ? dx + i : du + r;
    const int ps = xl ? NX : NU;
    T *const udst = xl ? dump : st + 2 * NX + r;
    const int xoff = xl ? i : 0;
    for (int t0 = 0; t0 < N - 1; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = t0 + d;
            const bool on = t < N - 1;                         // uniform
            const T *xc = st + (t & 1) * NX;
            T x[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) x[j] = xc[j];
            const T xown = xc[xoff];
            T acc = xl ? T(0) : rb[d][0];                      // u-lane: k_r first, then K[r,:] x (the column form's order)
#pragma unroll
            for (int j = 0; j < NX; ++j) acc += ra[d][j] * x[j];
            T bv[NU];
#pragma unroll
            for (int q = 0; q < NU; ++q) bv[q] = xl ? rb[d][q] : T(0);
            fetch(t + D, ra[d], rb[d]);
            udst[0] = acc;                                     // u-lanes publish u_r (x-lanes: a dump word)
            slot_sync();
            T u[NU];
#pragma unroll
            for (int q = 0; q < NU; ++q) u[q] = st[2 * NX + q];
            T accx = acc;
#pragma unroll
            for (int q = 0; q < NU; ++q) accx += bv[q] * u[q]; // x-lanes: + B[a,:] u ; u-lanes: + 0
            po[(int64_t)(on ? t : N - 1) * ps] = xl ? xown : acc;
            // x' into the other buffer (a dead step carries x over); u-lanes write the second dump word
            (xl ? st + ((t + 1) & 1) * NX + i : dump + 1)[0] = on ? accx : xown;
            slot_sync();
        }
    }
    if (!valid) return;
    // x_{N-1}: after the loop the current buffer index is that of the first step not executed
    const int tend = ((N - 1 + D - 1) / D) * D;                // steps the loop walked (dead ones included)
    const T *xf = st + (tend & 1) * NX;
    if (xl) dx[(N - 1) * NX + i] = xf[i];
    if (i != 0) return;
    // last control: minimiser of its own cost term  (R + Rr) du = R ud + Rr u_reg  (isls.py:560-571, last block row)
    T g[NU], H[NU][NU], U[NU][NU], rd[NU], y[NU];
    const T *Cl = p.Cuu.at(b, N - 1);
#pragma unroll
    for (int q = 0; q < NU; ++q) {
        g[q] = c == 0 ? -p.c0u.at(b, N - 1)[q] : T(0);
#pragma unroll
        for (int j = 0; j < NU; ++j) H[q][j] = Cl[q * NU + j];
    }
    if (p.Rr.p) {
        const T *Rl = p.Rr.at(b, N - 1);
        const T *z = p.zu + col * N * NU + (N - 1) * NU, *l = p.lu + col * N * NU + (N - 1) * NU;
#pragma unroll
        for (int q = 0; q < NU; ++q)
#pragma unroll
            for (int j = 0; j < NU; ++j) g[q] += T(2) * Rl[q * NU + j] * (z[j] - l[j]);
    }
    chol_upper<NU>(H, U, rd);
    chol_solve<NU>(U, rd, g, y);
#pragma unroll
    for (int q = 0; q < NU; ++q) du[(N - 1) * NU + q] = y[q];
}

template <typename T>
int launch_columns_rollout(const isls_columns_args &a, hipStream_t s)
{
    if (a.B < 0 || a.N < 2 || a.C < 2 || a.C > 1 + a.n || a.C > ISLS_MAX_ROW_DIM) return ISLS_ERR_ARG;
    if (!a.A.p || !a.Bm.p || !a.Cuu.p || !a.c0u.p || !a.K || !a.k || !a.dx || !a.du) return ISLS_ERR_ARG;
    if (a.Rr.p && (!a.zu || !a.lu)) return ISLS_ERR_ARG;
    if (a.B == 0) return ISLS_OK;
    ColP<T> p;
    p.B = a.B; p.N = a.N; p.C = a.C;
    p.A = View<T>(a.A); p.Bm = View<T>(a.Bm); p.Cuu = View<T>(a.Cuu); p.c0u = View<T>(a.c0u); p.Rr = View<T>(a.Rr);
    p.K = (const T *)a.K; p.k = (const T *)a.k; p.zu = (const T *)a.zu; p.lu = (const T *)a.lu;
    p.dx = (T *)a.dx; p.du = (T *)a.du; p.active = a.active;
    const int pb = 4, blocks = (a.B + pb - 1) / pb;
    static const bool rows_on = [] { const char *e = getenv("ISLS_COL_ROWS"); return !e || atoi(e) != 0; }();
#define LAUNCH_C(NX_, NU_, C_)                                                                                         \
    {                                                                                                                  \
        if constexpr (C_ * (NX_ + NU_) <= 64) {                                                                        \
            if (rows_on) {                                                                                             \
                constexpr int PBR = 64 / (C_ * (NX_ + NU_));                                                           \
                hipLaunchKernelGGL((columns_rollout_rows_kernel<T, NX_, NU_, C_>), dim3((a.B + PBR - 1) / PBR), dim3(64), 0, s, p); \
            } else hipLaunchKernelGGL((columns_rollout_kernel<T, NX_, NU_, C_>), dim3(blocks), dim3(64), 0, s, p);     \
        } else hipLaunchKernelGGL((columns_rollout_kernel<T, NX_, NU_, C_>), dim3(blocks), dim3(64), 0, s, p);         \
    }
#define CALL(NX_, NU_)                                                                                               \
    if (a.C == 2) LAUNCH_C(NX_, NU_, 2)                                                                               \
    else if (a.C == 3) LAUNCH_C(NX_, NU_, 3)                                                                          \
    else LAUNCH_C(NX_, NU_, 4)
    ISLS_DISPATCH_DIMS(a.n, a.m, CALL)
#undef CALL
#undef LAUNCH_C
    return check_launch();
}
template int launch_columns_rollout<double>(const isls_columns_args &, hipStream_t);
template int launch_columns_rollout<float>(const isls_columns_args &, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------------
// z / dual step around the row projection (isls.py:626-665): one workgroup per problem, one thread per (step, column).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kColMaxD = 9;
constexpr int kColThreads = 256;

template <typename T>
struct ColAdmmP {
    int B, N, C, phase;
    T relax, tol_abs, tol_rel;
    int d[2];
    const T *x[2];
    T *z[2], *l[2], *zprev[2], *work[2];
    const T *nom[2];
    View<T> W[2];
    T *res, *res_prev;
    int32_t *active, *iters;
};

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *sh)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    T tot = T(0);
    for (int i = 0; i < kColThreads / 64; ++i) tot += sh[i];
    return tot;
}

template <typename T>
__global__ __launch_bounds__(kColThreads) void columns_admm_kernel(ColAdmmP<T> p)
{
    __shared__ T sh[kColThreads / 64];
    const int b = blockIdx.x;
    if (p.active && !p.active[b]) return;                      // uniform over the workgroup
    const int N = p.N, C = p.C;
    T prim = T(0), dual = T(0);
    for (int blk = 0; blk < 2; ++blk) {
        const int d = p.d[blk];
        if (d == 0) continue;
        T accp = T(0), accd = T(0);
        for (int it = threadIdx.x; it < N * C; it += kColThreads) {
            const int t = it / C, c = it - t * C;
            const int64_t e0 = (((int64_t)c * p.B + b) * N + t) * d;
            const int64_t w0 = ((int64_t)b * N + t) * d;         // row (t, i) of problem b is w0 + i
            const bool shift = c == 0 && p.nom[blk] != nullptr;
            if (p.phase == 0) {
                for (int i = 0; i < d; ++i) {
                    const T zi = p.z[blk][e0 + i];
                    T v = p.relax * p.x[blk][e0 + i] + (T(1) - p.relax) * zi + p.l[blk][e0 + i];
                    if (shift) v += p.nom[blk][w0 + i];
                    p.work[blk][(w0 + i) * C + c] = v;
                    p.zprev[blk][e0 + i] = zi;
                }
            } else {
                T r[kColMaxD], dz[kColMaxD];
                for (int i = 0; i < d; ++i) {
                    T zn = p.work[blk][(w0 + i) * C + c];
                    if (shift) zn -= p.nom[blk][w0 + i];
                    r[i] = p.x[blk][e0 + i] - zn;
                    dz[i] = zn - p.zprev[blk][e0 + i];
                    p.l[blk][e0 + i] += r[i];
                    p.z[blk][e0 + i] = zn;
                }
                const T *W = p.W[blk].at(b, t);
                for (int i = 0; i < d; ++i) {
                    T wr = T(0), wz = T(0);
                    for (int j = 0; j < d; ++j) {
                        wr += W[i * d + j] * r[j];
                        wz += W[i * d + j] * dz[j];
                    }
                    accp += wr * wr;
                    accd += wz * wz;
                }
            }
        }
        if (p.phase == 1) {
            prim += sqrt(block_sum(accp, sh));
            dual += sqrt(block_sum(accd, sh));
        }
    }
    if (p.phase == 1 && threadIdx.x == 0) {
        const T pp = p.res_prev[2 * b], dp = p.res_prev[2 * b + 1];
        p.res[2 * b] = prim;
        p.res[2 * b + 1] = dual;
        p.res_prev[2 * b] = prim;
        p.res_prev[2 * b + 1] = dual;
        if (p.iters) p.iters[b] += 1;
        if (p.active) {
            bool stop = prim < p.tol_abs && dual < p.tol_abs;
            if (!stop) {
                const T pc = fabs(pp - prim) / (pp + T(1e-30)), dc = fabs(dp - dual) / (dp + T(1e-30));
                stop = pc < p.tol_rel && dc < p.tol_rel;
            }
            if (stop) p.active[b] = 0;
        }
    }
}

template <typename T>
int launch_columns_admm(const isls_columns_admm_args &a, hipStream_t s)
{
    if (a.B < 0 || a.N < 1 || a.C < 1 || a.C > ISLS_MAX_ROW_DIM || (a.phase != 0 && a.phase != 1)) return ISLS_ERR_ARG;
    if (a.n < 1 || a.n > kColMaxD || a.m < 1 || a.m > kColMaxD) return ISLS_ERR_UNSUPPORTED;
    if (!a.xx && !a.xu) return ISLS_ERR_ARG;
    if (a.xx && (!a.zx || !a.lx || !a.zx_prev || !a.x_work || !a.Qr.p)) return ISLS_ERR_ARG;
    if (a.xu && (!a.zu || !a.lu || !a.zu_prev || !a.u_work || !a.Rr.p)) return ISLS_ERR_ARG;
    if (!a.res || !a.res_prev) return ISLS_ERR_ARG;
    if (a.B == 0) return ISLS_OK;
    ColAdmmP<T> p;
    p.B = a.B; p.N = a.N; p.C = a.C; p.phase = a.phase;
    p.relax = (T)a.relax; p.tol_abs = (T)a.tol_abs; p.tol_rel = (T)a.tol_rel;
    p.d[0] = a.xx ? a.n : 0; p.d[1] = a.xu ? a.m : 0;
    p.x[0] = (const T *)a.xx; p.x[1] = (const T *)a.xu;
    p.z[0] = (T *)a.zx; p.z[1] = (T *)a.zu; p.l[0] = (T *)a.lx; p.l[1] = (T *)a.lu;
    p.zprev[0] = (T *)a.zx_prev; p.zprev[1] = (T *)a.zu_prev;
    p.work[0] = (T *)a.x_work; p.work[1] = (T *)a.u_work;
    p.nom[0] = (const T *)a.x_nom; p.nom[1] = (const T *)a.u_nom;
    p.W[0] = View<T>(a.Qr); p.W[1] = View<T>(a.Rr);
    p.res = (T *)a.res; p.res_prev = (T *)a.res_prev; p.active = a.active; p.iters = a.iters;
    hipLaunchKernelGGL((columns_admm_kernel<T>), dim3(a.B), dim3(kColThreads), 0, s, p);
    return check_launch();
}
template int launch_columns_admm<double>(const isls_columns_admm_args &, hipStream_t);
template int launch_columns_admm<float>(const isls_columns_admm_args &, hipStream_t);

// ------------------------------------------------------------------------------------------------
// Behind the line search on column 0 (isls/isls.py:593-606): du[0] <- alpha* du[0], dx[0] <- x_noms[ind] - x_nom for the active
// problems (flat sweep, one wavefront per problem).
template <typename T>
__global__ __launch_bounds__(64) void columns_step_kernel(int N, int n, int m, const T *alphas, const int32_t *best, const T *x_out,
                                                          const T *xhat, T *dx0, T *du0, const int32_t *active)
{
    const int b = blockIdx.x;
    if (active && active[b] == 0) return;
    const T step = alphas[best[b]];
    const int64_t ox = (int64_t)b * N * n, ou = (int64_t)b * N * m;
    for (int e = threadIdx.x; e < N * m; e += kWave) du0[ou + e] *= step;
    for (int e = threadIdx.x; e < N * n; e += kWave) dx0[ox + e] = x_out[ox + e] - xhat[ox + e];
}

// flag <- 1 when a problem of the batch is still active (flag zeroed by the caller of this kernel)
__global__ __launch_bounds__(256) void any_active_kernel(int B, const int32_t *active, int32_t *flag)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = b < B && active[b] != 0;
    if (__ballot(on) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

template <typename T>
int launch_columns_iteration(const isls_columns_iteration_args &a, hipStream_t s)
{
    const isls_columns_args &c = a.cols;
    const int C = c.C, B = c.B, N = c.N, n = c.n, m = c.m;
    if (B < 0 || C < 1) return ISLS_ERR_ARG;
    if (B == 0) return ISLS_OK;
    int rc;
    // ---- x-step: C feed-forward passes, then the column rollout ---------------------------------------------------------
    isls_ff_args f = a.ff;
    f._pad = C;
    rc = (C > 1 && a.ff._pad != 1) ? launch_ff<T>(f, s) : ISLS_ERR_UNSUPPORTED;   // ff._pad == 1: one pass per column asked for
    if (rc == ISLS_ERR_UNSUPPORTED) {                          // one pass per column
        if (C > 1 && (!a.zero_x || !a.zero_u)) return ISLS_ERR_ARG;
        for (int col = 0; col < C; ++col) {
            isls_ff_args g = a.ff;
            g._pad = 0;
            const int64_t ox = (int64_t)col * B * N * n * (int64_t)sizeof(T), ou = (int64_t)col * B * N * m * (int64_t)sizeof(T);
            auto off = [](const void *p, int64_t o) -> const void * { return p ? (const char *)p + o : nullptr; };
            g.zx = off(a.ff.zx, ox); g.lx = off(a.ff.lx, ox); g.zu = off(a.ff.zu, ou); g.lu = off(a.ff.lu, ou);
            g.k = (char *)a.ff.k + ou;
            if (col > 0) { g.c0x = isls_view{a.zero_x, 0, 0}; g.c0u = isls_view{a.zero_u, 0, 0}; }
            if ((rc = launch_ff<T>(g, s)) != ISLS_OK) return rc;
        }
    } else if (rc != ISLS_OK) {
        return rc;
    }
    if ((rc = launch_columns_rollout<T>(c, s)) != ISLS_OK) return rc;
    // ---- line search on column 0 and its step ---------------------------------------------------------------------------
    if (a.ls.L > 0) {
        if (!a.ls.best || !a.ls.alphas || !a.ls.x_out || !a.ls.xhat) return ISLS_ERR_ARG;
        if ((rc = launch_rollout<T>(a.ls, s)) != ISLS_OK) return rc;
        hipLaunchKernelGGL((columns_step_kernel<T>), dim3(B), dim3(64), 0, s, N, n, m, (const T *)a.ls.alphas, (const int32_t *)a.ls.best,
                           (const T *)a.ls.x_out, (const T *)a.ls.xhat, (T *)c.dx, (T *)c.du, a.ls.active);
        if ((rc = check_launch()) != ISLS_OK) return rc;
    }
    // ---- z-step ---------------------------------------------------------------------------------------------------------
    if (a.admm.xx || a.admm.xu) {
        isls_columns_admm_args z = a.admm;
        z.phase = 0;
        if ((rc = launch_columns_admm<T>(z, s)) != ISLS_OK) return rc;
        if (a.admm.xx && a.proj_x && (rc = launch_project<T>(*a.proj_x, s)) != ISLS_OK) return rc;
        if (a.admm.xu && a.proj_u && (rc = launch_project<T>(*a.proj_u, s)) != ISLS_OK) return rc;
        z.phase = 1;
        if ((rc = launch_columns_admm<T>(z, s)) != ISLS_OK) return rc;
        if (a.log && hipMemcpyAsync(a.log, a.admm.res, sizeof(T) * (size_t)B * 2, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return ISLS_ERR_LAUNCH;
    }
    if (a.any_active) {
        if (!a.admm.active) return ISLS_ERR_ARG;
        if (hipMemsetAsync(a.any_active, 0, sizeof(int32_t), s) != hipSuccess) return ISLS_ERR_LAUNCH;
        hipLaunchKernelGGL(any_active_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, (const int32_t *)a.admm.active, a.any_active);
        if ((rc = check_launch()) != ISLS_OK) return rc;
    }
    return ISLS_OK;
}
template int launch_columns_iteration<double>(const isls_columns_iteration_args &, hipStream_t);
template int launch_columns_iteration<float>(const isls_columns_iteration_args &, hipStream_t);

}  // namespace isls
