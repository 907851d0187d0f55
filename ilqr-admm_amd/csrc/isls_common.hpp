// isls_common.hpp -- shared device helpers for the gfx950 kernels of libisls_hip.so.
//
// Execution model used by every time-recursive kernel here (Riccati gain / feed-forward, rollout):
//   * a workgroup is ONE 64-lane wavefront (blockDim.x == 64);
//   * the wavefront is cut into TPW "slots" of G lanes; one slot owns one trajectory for the whole
//     horizon (the recursion is sequential in t, parallel over trajectories and over matrix rows /
//     line-search candidates inside the slot);
//   * per-step operands are fetched from HBM by the slot's lanes with cooperative contiguous loads
//     one step AHEAD of their use (register staging), written to the slot's LDS record, and then
//     consumed from LDS as broadcast (all lanes of a slot read the same word) or column reads.
// Nothing in here assumes a dispatch order or an XCD placement.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/isls_hip.h"

namespace isls {

constexpr int kWave = 64;

template <typename T>
struct View {
    const T *p;
    int64_t sb, st;
    __host__ __device__ View() : p(nullptr), sb(0), st(0) {}
    __host__ __device__ explicit View(const isls_view &v) : p(static_cast<const T *>(v.p)), sb(v.sb), st(v.st) {}
    __device__ __forceinline__ const T *at(int b, int t) const { return p + (int64_t)b * sb + (int64_t)t * st; }
};

// Compile-time loop: body(integral_constant<int, j>) for j = 0..N-1.  `#pragma unroll` is only a hint and the
// inliner/unroller gives up on some lambda bodies; a runtime index into a small per-lane array then sends the
// array to scratch memory.  static_for makes every index a constant expression.
template <int N, typename F, int... Js>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Js...>)
{
    (f(std::integral_constant<int, Js>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl<N>(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// Hand-off through LDS between the lanes of ONE wavefront (every recursive kernel here runs
// single-wave workgroups).  DS instructions of a wave execute in issue order, so a ds_write followed by a
// ds_read needs no s_barrier and, crucially, no `s_waitcnt vmcnt(0)`: __syncthreads() would drain the
// global loads that are deliberately kept in flight several steps ahead.  This only pins the compiler's
// ordering of memory operations (wavefront-scope fences and a wave barrier emit no instructions).
// Streaming accesses (the `nt` bit of a gfx950 global load / store): data a launch touches once -- the packed records of a
// feed-forward pass, the arrays a pass writes for a later launch -- should not push the re-used lines (the vectors a
// recursion walks 8 bytes at a time, the gains the winner replay reads again) out of the L2.  Measured per site
// (tools/kbench.py): the switches default to what paid.
template <typename V>
__device__ __forceinline__ V ld_stream(const V *p) { return __builtin_nontemporal_load(p); }
template <typename V>
__device__ __forceinline__ void st_stream(V *p, V v) { __builtin_nontemporal_store(v, p); }
#ifndef ISLS_NT_FFREC
#define ISLS_NT_FFREC 1
#endif
#ifndef ISLS_NT_GAIN_LD
#define ISLS_NT_GAIN_LD 0
#endif
#ifndef ISLS_NT_GAIN_ST
#define ISLS_NT_GAIN_ST 1
#endif
#ifndef ISLS_NT_RO_LD
#define ISLS_NT_RO_LD 0
#endif

__device__ __forceinline__ void slot_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cooperative load of CNT contiguous elements by the G lanes of a slot: lane i gets src[i + G*j].
// The loads are UNCONDITIONAL (index clamped into the segment; callers pass a valid trajectory's pointer
// for idle lanes) and their results are not touched here: a predicate would cost a branch per load and
// any arithmetic on the result would force an s_waitcnt right behind the load, defeating the prefetch
// ring.  Surplus elements are simply never written to LDS (coop_put checks the range).
template <int CNT, int G, typename T>
__device__ __forceinline__ void coop_load(const T *src, T (&r)[(CNT + G - 1) / G], int i, bool /*ok*/)
{
#pragma unroll
    for (int j = 0; j < (CNT + G - 1) / G; ++j) {
        const int e = i + G * j;
        r[j] = src[e < CNT ? e : CNT - 1];
    }
}
template <int CNT, int G, typename T>
__device__ __forceinline__ void coop_put(T *dst, const T (&r)[(CNT + G - 1) / G], int i, bool ok)
{
#pragma unroll
    for (int j = 0; j < (CNT + G - 1) / G; ++j) {
        const int e = i + G * j;
        if (ok && e < CNT) dst[e] = r[j];
    }
}

// Upper Cholesky Quu = U'U in registers (unblocked, column order of LAPACK dpotf2 'U').
// U keeps the strict upper part; rd[j] = 1 / U_jj.  Returns false if a pivot is not positive.
template <int M, typename T>
__device__ __forceinline__ bool chol_upper(const T (&A)[M][M], T (&U)[M][M], T (&rd)[M])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        T ajj = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) ajj -= U[k][j] * U[k][j];
        ok = ok && (ajj > T(0));
        const T d = sqrt(ajj);
        U[j][j] = d;
        rd[j] = T(1) / d;
#pragma unroll
        for (int c = j + 1; c < M; ++c) {
            T s = A[j][c];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= U[k][j] * U[k][c];
            U[j][c] = s * rd[j];
        }
    }
    return ok;
}
// x = (U'U)^{-1} b with the reciprocal diagonal rd.
template <int M, typename T>
__device__ __forceinline__ void chol_solve(const T (&U)[M][M], const T (&rd)[M], const T (&b)[M], T (&x)[M])
{
    T y[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        T s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= U[k][i] * y[k];
        y[i] = s * rd[i];
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        T s = y[i];
#pragma unroll
        for (int k = i + 1; k < M; ++k) s -= U[i][k] * x[k];
        x[i] = s * rd[i];
    }
}

// 64-lane butterfly sum (every lane ends with the total).
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T w = __shfl_xor(v, o, 64);
        v = (w > v) ? w : v;
    }
    return v;
}

// numpy's float `%`: fmod followed by a sign fix (result takes the sign of the divisor).
__device__ __forceinline__ double py_mod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0.0 && ((b < 0.0) != (r < 0.0))) r += b;
    return r;
}
__device__ __forceinline__ float py_mod(float a, float b)
{
    float r = fmodf(a, b);
    if (r != 0.0f && ((b < 0.0f) != (r < 0.0f))) r += b;
    return r;
}

// sin and cos of one argument for the forward models and their linearisations.
// fp64: the library's sincos is ~100 instructions on its fast path (a double-double Cody-Waite reduction), and the arm's line
// search takes three per candidate and step -- 30 k of the 53 k vector instructions of a wavefront (profiles/r03_sq_counters_config3).
// For |a| <= 1e5: k = rint(a 2/pi), r = (a - k P1) - k P2 with two fused multiply-adds (P1 + P2 = pi/2 to 107 bits; |k| < 2^16, so
// the part of pi/2 left out contributes < 1e-27), the fdlibm minimax polynomials on [-pi/4, pi/4], quadrant by selects: ~45
// instructions.  Worst ABSOLUTE error against a 200-bit reference over 62 000 points (random up to 1e5, and at / next to /
// within 1e-9 of the first 2000 multiples of pi/2): 1.6e-16 (the host libm: 5.6e-17) -- the results enter sums of order one
// (end-effector positions, headings), where that is below a unit in the last place.  Larger, infinite or NaN arguments take the
// library path (a branch no lane of these workloads takes).
__device__ __forceinline__ void sin_cos(double a, double &s, double &c)
{
    if (!(fabs(a) <= 1.0e5)) { sincos(a, &s, &c); return; }
    constexpr double TWO_OVER_PI = 0x1.45f306dc9c883p-1, P1 = 0x1.921fb54442d18p+0, P2 = 0x1.1a62633145c07p-54;
    constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double k = rint(a * TWO_OVER_PI);
    double r = fma(-k, P1, a);
    r = fma(-k, P2, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    const double sn = fma(z * r, fma(z, ps, S1), r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    const double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int n = (int)k;
    const double ss = (n & 1) ? cs : sn, cc = (n & 1) ? sn : cs;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}
__device__ __forceinline__ void sin_cos(float a, float &s, float &c) { sincosf(a, &s, &c); }

// Words between the packed step records [A+BK | B | K | fac] of consecutive trajectories of a wavefront (isls_gain_args.rec):
// the record padded to an even word count, so that the feed-forward pass fetches records as aligned 16-byte pairs that
// never straddle two trajectories.
// rec_model_words: words behind fac that describe the step's linearisation for the model-structured feed-forward form
// (isls_ff_args.lin_on).  The pair (9, 3) carries the six words A[6:8, 0:3] -- the Jacobian rows of ISLS_MODEL_ARM3R, from which
// its A and B follow -- so that pass reads one contiguous tail [K | fac | J] instead of gathering J from the A array (scattered
// 8-byte loads at a 64.8 KB stride: the structured form then ran no faster than the dense one).  Other pairs: none.
// The pair (4, 2) carries the six entries of ISLS_MODEL_CAR's linearisation that are not 0, 1 or dt: A[0,2], A[1,2], A[0,3],
// A[1,3], A[2,3], B[2,0] (for the other models of that pair they are copied all the same and nobody reads them).
__host__ __device__ constexpr int rec_model_words(int n, int m) { return ((n == 9 && m == 3) || (n == 4 && m == 2)) ? 6 : 0; }
// word e of the model words within one staged step [A_t B_t] (row stride n + m)
__host__ __device__ constexpr int rec_model_src(int n, int m, int e)
{
    return (n == 9) ? (6 + e / 3) * (n + m) + e % 3
                    : (e == 0 ? 0 * 6 + 2 : e == 1 ? 1 * 6 + 2 : e == 2 ? 0 * 6 + 3 : e == 3 ? 1 * 6 + 3 : e == 4 ? 2 * 6 + 3 : 2 * 6 + 4);
}
__host__ __device__ constexpr int rec_stride(int n, int m) { return (n * n + 2 * n * m + m * m + rec_model_words(n, m) + 1) & ~1; }
// The lean layout (isls_gain_args.lin_on / isls_ff_args.lin_on): only the tail [K | fac | model words] of a record, at this stride
// (the same buffer: isls_ff_record_elems covers the dense layout, the lean one uses a prefix of it).
__host__ __device__ constexpr int rec_lean_stride(int n, int m) { return (m * n + m * m + rec_model_words(n, m) + 1) & ~1; }

// Launch wrappers implemented one per .hip file; each returns ISLS_OK / ISLS_ERR_*.
// ff != nullptr: the pass may also run the first feed-forward pass (same records, time-invariant Qr / Rr); *did_ff tells
// require_ff: launch nothing unless the feed-forward pass can ride along (*did_ff stays false)
template <typename T>
int launch_gain(const isls_gain_args &a, hipStream_t s, const isls_ff_args *ff = nullptr, bool *did_ff = nullptr, bool require_ff = false);
template <typename T> int launch_ff(const isls_ff_args &a, hipStream_t s);
template <typename T> int launch_ff_record(const isls_ff_args &a, hipStream_t s);
template <typename T> int launch_ff_prepare(const isls_ff_prepare_args &a, hipStream_t s);
template <typename T> int launch_ff_stitch(const isls_ff_args &a, hipStream_t s);
bool ff_seg_enabled(const isls_ffseg &sg);
int ff_segments(int N, int nseg_req, int *seg_len);
template <typename T>
// last = false (fused form only): further ADMM iterations of the same outer iteration follow, the x-step of a trajectory that
// goes on need not be written out
int launch_rollout(const isls_rollout_args &a, hipStream_t s, const isls_admm_args *fused = nullptr, bool *did_fuse = nullptr, bool last = true);
bool rollout_can_fuse_admm(const isls_rollout_args &r, const isls_admm_args &a);
template <typename T> int launch_admm(const isls_admm_args &a, hipStream_t s);
template <typename T> int launch_project(const isls_project_args &a, hipStream_t s);
template <typename T> int launch_sls_admm(const isls_sls_admm_args &a, hipStream_t s);
template <typename T>
int launch_sls_closed_loop(int M, int N, int n, int m, const void *A, const void *B, const void *K, const void *k,
                           const void *x0, void *x_log, void *u_log, hipStream_t s);
template <typename T> int launch_dense_closed_loop(const isls_dense_loop_args &a, hipStream_t s);
template <typename T> int launch_columns_rollout(const isls_columns_args &a, hipStream_t s);
template <typename T> int launch_columns_admm(const isls_columns_admm_args &a, hipStream_t s);
template <typename T> int launch_columns_iteration(const isls_columns_iteration_args &a, hipStream_t s);
template <typename T> int launch_expand(const isls_expand_args &a, hipStream_t s);
template <typename T> int launch_linearize(const isls_linearize_args &a, hipStream_t s);
template <typename T> int launch_accept(const isls_accept_args &a, hipStream_t s);
template <typename T> int launch_advance(const isls_advance_args &a, hipStream_t s);
// any (n <= 16, m <= 8) without an instantiation of the fast kernels: generic.hip (array form, one trajectory per wavefront)
bool dims_generic(int n, int m);
template <typename T> int launch_gain_generic(const isls_gain_args &a, hipStream_t s);
template <typename T> int launch_ff_generic(const isls_ff_args &a, hipStream_t s);
template <typename T> int launch_rollout_generic(const isls_rollout_args &a, hipStream_t s);
template <typename T> int launch_reduce(int32_t B, const void *cost, const void *res, const int32_t *active,
                                        const int32_t *status, void *out5, hipStream_t s, int row = -1, int rows = 0);
template <typename T> int launch_outer_begin(int32_t B, int32_t N, int32_t n, int32_t m, int32_t *admm_active,
                                             const int32_t *outer_active, void *lx, void *lu, void *res_prev,
                                             int32_t *iters, hipStream_t s);

// Trajectories per wavefront for the slot kernels (env override for tuning experiments).
inline int pick_tpw(int B, int max_tpw, const char *env)
{
    int tpw = max_tpw;       // measured on MI355X (B=4096): fuller wavefronts win, the per-step latency does not shrink with fewer slots
    (void)B;
    if (const char *e = getenv(env)) tpw = atoi(e);
    if (tpw < 1) tpw = 1;
    return tpw > max_tpw ? max_tpw : tpw;
}

inline int check_launch()
{
    return hipGetLastError() == hipSuccess ? ISLS_OK : ISLS_ERR_LAUNCH;
}

// Supported (n, m) pairs: the reference notebooks' systems and every get_double_integrator_AB(nb_dim <= 3,
// nb_deriv <= 3) system (isls/utils.py:266-276: n = nb_dim * nb_deriv, m = nb_dim).  The kernels are templates over the
// dimensions (rows live in registers); a further pair is one line here, one in rollout.hip and one in the Makefile.
#define ISLS_FOR_EACH_DIMS(X) X(6, 3) X(2, 1) X(4, 2) X(9, 3) X(3, 1) X(6, 2) X(2, 2) X(3, 3)
#define ISLS_DISPATCH_DIMS(n, m, CALL)                  \
    {                                                   \
        const int n_ = (n), m_ = (m);                   \
        if (n_ == 6 && m_ == 3) { CALL(6, 3); }         \
        else if (n_ == 2 && m_ == 1) { CALL(2, 1); }    \
        else if (n_ == 4 && m_ == 2) { CALL(4, 2); }    \
        else if (n_ == 9 && m_ == 3) { CALL(9, 3); }    \
        else if (n_ == 3 && m_ == 1) { CALL(3, 1); }    \
        else if (n_ == 6 && m_ == 2) { CALL(6, 2); }    \
        else if (n_ == 2 && m_ == 2) { CALL(2, 2); }    \
        else if (n_ == 3 && m_ == 3) { CALL(3, 3); }    \
        else return ISLS_ERR_UNSUPPORTED;               \
    }
inline bool dims_supported(int n, int m)
{
#define ISLS_DIMS_TEST_(NX_, NU_) if (n == NX_ && m == NU_) return true;
    ISLS_FOR_EACH_DIMS(ISLS_DIMS_TEST_)
#undef ISLS_DIMS_TEST_
    return false;
}

}  // namespace isls
