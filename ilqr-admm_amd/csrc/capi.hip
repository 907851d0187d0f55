// capi.hip -- extern "C" surface of libisls_hip.so (declared in include/isls_hip.h) and the
// stream-ordered driver of one outer DP-form iLQR-ADMM iteration.
#include <new>
#include <vector>

#include "isls_common.hpp"

namespace isls {

// ---- optional per-kernel-family timing with HIP events on the launch stream -----------------------
// (bench.py reads these: average launch duration of the dominant kernel for the roofline line).  The context is
// caller-owned (isls_timing_create / isls_timing_destroy) and reaches the driver through isls_outer_args.timing: the
// library itself keeps no state.  One context belongs to one host thread / stream at a time.
struct Timing {
    bool paused = false;           // events are recorded only while !paused (bench.py samples every k-th step)
    static constexpr int kKinds = 5;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[kKinds];
    size_t used[kKinds] = {0, 0, 0, 0, 0};
};

struct ScopedTimer {
    hipStream_t s;
    hipEvent_t stop = nullptr;
    ScopedTimer(Timing *tm, int kind, hipStream_t s_) : s(s_)
    {
        if (!tm || tm->paused) return;
        auto &pool = tm->ev[kind];
        size_t &u = tm->used[kind];
        if (u == pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess) return;
            if (hipEventCreate(&b) != hipSuccess) { hipEventDestroy(a); return; }
            pool.emplace_back(a, b);
        }
        hipEventRecord(pool[u].first, s);
        stop = pool[u].second;
        ++u;
    }
    ~ScopedTimer()
    {
        if (stop) hipEventRecord(stop, s);
    }
};

template <typename T>
static int outer_iteration(const isls_outer_args &a, hipStream_t s)
{
    const isls_admm_args &ad = a.admm;
    Timing *const tmg = static_cast<Timing *>(a.timing);
    int rc = ISLS_OK;
    if (!a.begin_done &&
        (rc = launch_outer_begin<T>(ad.B, ad.N, ad.n, ad.m, ad.active, a.outer_active, ad.lx, ad.lu, ad.res_prev, ad.iters, s)) != ISLS_OK)
        return rc;
    bool ff_done = false;                                      // the gain pass ran the first feed-forward pass as well
    // the gain pass writes the records in the layout its hint selects and the feed-forward passes read them in theirs
    if (a.J > 0 && a.gain.rec && a.ff.rec == a.gain.rec && (a.gain.lin_on != 0) != (a.ff.lin_on != 0)) return ISLS_ERR_ARG;
    if (a.ff.lin_on && ff_seg_enabled(a.ff.seg)) return ISLS_ERR_UNSUPPORTED;   // the segment operators need the dense records
    if (!a.skip_gain) {
        {
            ScopedTimer tm(tmg, 0, s);
            static const bool fuse_ff = [] { const char *e = getenv("ISLS_GAIN_FF"); return !e || atoi(e) != 0; }();
            // The gain pass with the recursion inside holds 370-400 registers: one wavefront per SIMD.  While its wavefronts fit
            // the chip's SIMDs that costs nothing; beyond (B > 7168 at n = 6, m = 3 on 256 CUs) the surplus runs as a second round
            // and the launch doubles (123 -> 251 us from B = 4096 to 8192), where the pass without the recursion (<= 256 registers,
            // two wavefronts per SIMD) and a feed-forward launch of its own take 148 + 82 us.
            static const int64_t simds = [] {
                int dev = 0; hipDeviceProp_t pr;
                if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return (int64_t)1024;
                return (int64_t)pr.multiProcessorCount * 4;
            }();
            const int lanes = a.gain.n + a.gain.m;
            const int64_t waves = lanes > 0 && lanes <= kWave ? (a.gain.B + kWave / lanes - 1) / (kWave / lanes) : 0;
            const bool fuse_now = fuse_ff && a.J > 0 && waves <= simds;
            if ((rc = launch_gain<T>(a.gain, s, fuse_now ? &a.ff : nullptr, &ff_done)) != ISLS_OK) return rc;
        }
        if (ff_seg_enabled(a.ff.seg)) {                        // operators of the time-parallel feed-forward pass
            const isls_ff_args &f = a.ff;
            isls_ff_prepare_args pr = {};
            pr.B = f.B; pr.N = f.N; pr.n = f.n; pr.m = f.m; pr.solve_mode = f.solve_mode;
            pr.A = f.A; pr.Bm = f.Bm; pr.K = f.K; pr.Quu = f.Quu; pr.fac = f.fac; pr.Qux = f.Qux;
            pr.active = f.active; pr.seg = f.seg; pr.rec = f.rec;
            ScopedTimer tm(tmg, 4, s);
            if ((rc = launch_ff_prepare<T>(pr, s)) != ISLS_OK) return rc;
        }
    }
    // element-wise ADMM updates ride on the winner replay of the rollout (one launch and one pass over x, u less)
    const bool fuse = rollout_can_fuse_admm(a.ro, a.admm);
    bool fused = false;
    for (int j = 0; j < a.J; ++j) {
        if (!(j == 0 && ff_done)) {
            ScopedTimer tm(tmg, 1, s);
            if ((rc = launch_ff<T>(a.ff, s)) != ISLS_OK) return rc;
        }
        {
            ScopedTimer tm(tmg, 2, s);
            if ((rc = launch_rollout<T>(a.ro, s, fuse ? &a.admm : nullptr, &fused, j == a.J - 1 || a.log != nullptr)) != ISLS_OK) return rc;
        }
        if (!fused) {
            ScopedTimer tm(tmg, 3, s);
            if ((rc = launch_admm<T>(a.admm, s)) != ISLS_OK) return rc;
        }
        if (a.log) {
            if (hipMemcpyAsync((T *)a.log + (size_t)j * ad.B * 2, ad.res, sizeof(T) * (size_t)ad.B * 2,
                               hipMemcpyDeviceToDevice, s) != hipSuccess)
                return ISLS_ERR_LAUNCH;
        }
    }
    return ISLS_OK;
}

}  // namespace isls

using namespace isls;

#define ISLS_API extern "C" __attribute__((visibility("default")))

#define DEFINE_ENTRY(name, args_t, launcher, kind)                                        \
    ISLS_API int isls_##name##_f64(const args_t *a, void *stream)                          \
    {                                                                                      \
        if (!a) return ISLS_ERR_ARG;                                                       \
        if (a->B == 0) return ISLS_OK; /* empty batch: nothing to check, nothing to do */  \
        return launcher<double>(*a, (hipStream_t)stream);                                  \
    }                                                                                      \
    ISLS_API int isls_##name##_f32(const args_t *a, void *stream)                          \
    {                                                                                      \
        if (!a) return ISLS_ERR_ARG;                                                       \
        if (a->B == 0) return ISLS_OK;                                                     \
        return launcher<float>(*a, (hipStream_t)stream);                                   \
    }

DEFINE_ENTRY(riccati_gain, isls_gain_args, launch_gain, 0)
DEFINE_ENTRY(riccati_ff, isls_ff_args, launch_ff, 1)
DEFINE_ENTRY(riccati_ff_prepare, isls_ff_prepare_args, launch_ff_prepare, 4)
DEFINE_ENTRY(rollout_ls, isls_rollout_args, launch_rollout, 2)
DEFINE_ENTRY(admm_update, isls_admm_args, launch_admm, 3)

ISLS_API int isls_riccati_gain_ff_f64(const isls_gain_args *g, const isls_ff_args *ff, void *stream)
{
    if (!g || !ff) return ISLS_ERR_ARG;
    if (g->B == 0) return ISLS_OK;
    bool done = false;
    const int rc = launch_gain<double>(*g, (hipStream_t)stream, ff, &done, /*require_ff=*/true);
    return rc != ISLS_OK ? rc : (done ? ISLS_OK : ISLS_ERR_UNSUPPORTED);
}
ISLS_API int isls_riccati_gain_ff_f32(const isls_gain_args *g, const isls_ff_args *ff, void *stream)
{
    if (!g || !ff) return ISLS_ERR_ARG;
    if (g->B == 0) return ISLS_OK;
    bool done = false;
    const int rc = launch_gain<float>(*g, (hipStream_t)stream, ff, &done, /*require_ff=*/true);
    return rc != ISLS_OK ? rc : (done ? ISLS_OK : ISLS_ERR_UNSUPPORTED);
}

ISLS_API int isls_expand_quadratic_f64(const isls_expand_args *a, void *stream)
{
    if (a && a->B == 0) return ISLS_OK;
    return a ? launch_expand<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_expand_quadratic_f32(const isls_expand_args *a, void *stream)
{
    if (a && a->B == 0) return ISLS_OK;
    return a ? launch_expand<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_linearize_f64(const isls_linearize_args *a, void *stream)
{
    if (a && a->B == 0) return ISLS_OK;
    return a ? launch_linearize<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_linearize_f32(const isls_linearize_args *a, void *stream)
{
    if (a && a->B == 0) return ISLS_OK;
    return a ? launch_linearize<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_project_rows_f64(const isls_project_args *a, void *stream)
{
    return a ? launch_project<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_project_rows_f32(const isls_project_args *a, void *stream)
{
    return a ? launch_project<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_sls_admm_f64(const isls_sls_admm_args *a, void *stream)
{
    return a ? launch_sls_admm<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_sls_admm_f32(const isls_sls_admm_args *a, void *stream)
{
    return a ? launch_sls_admm<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_sls_closed_loop_f64(int32_t M, int32_t N, int32_t n, int32_t m, const void *A, const void *B, const void *K,
                                      const void *k, const void *x0, void *x_log, void *u_log, void *stream)
{
    return launch_sls_closed_loop<double>(M, N, n, m, A, B, K, k, x0, x_log, u_log, (hipStream_t)stream);
}
ISLS_API int isls_sls_closed_loop_f32(int32_t M, int32_t N, int32_t n, int32_t m, const void *A, const void *B, const void *K,
                                      const void *k, const void *x0, void *x_log, void *u_log, void *stream)
{
    return launch_sls_closed_loop<float>(M, N, n, m, A, B, K, k, x0, x_log, u_log, (hipStream_t)stream);
}
ISLS_API int isls_dense_closed_loop_f64(const isls_dense_loop_args *a, void *stream)
{
    return a ? launch_dense_closed_loop<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_dense_closed_loop_f32(const isls_dense_loop_args *a, void *stream)
{
    return a ? launch_dense_closed_loop<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_columns_rollout_f64(const isls_columns_args *a, void *stream)
{
    return a ? launch_columns_rollout<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_columns_rollout_f32(const isls_columns_args *a, void *stream)
{
    return a ? launch_columns_rollout<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_columns_admm_f64(const isls_columns_admm_args *a, void *stream)
{
    return a ? launch_columns_admm<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_columns_admm_f32(const isls_columns_admm_args *a, void *stream)
{
    return a ? launch_columns_admm<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_columns_iteration_f64(const isls_columns_iteration_args *a, void *stream)
{
    return a ? launch_columns_iteration<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_columns_iteration_f32(const isls_columns_iteration_args *a, void *stream)
{
    return a ? launch_columns_iteration<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_accept_step_f64(const isls_accept_args *a, void *stream)
{
    if (a && a->B == 0) return ISLS_OK;
    return a ? launch_accept<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_accept_step_f32(const isls_accept_args *a, void *stream)
{
    if (a && a->B == 0) return ISLS_OK;
    return a ? launch_accept<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_outer_advance_f64(const isls_advance_args *a, void *stream)
{
    return a ? launch_advance<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_outer_advance_f32(const isls_advance_args *a, void *stream)
{
    return a ? launch_advance<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_reduce_convergence_f64(int32_t B, const void *cost, const void *res, const int32_t *active,
                                         const int32_t *status, void *out5, void *stream)
{
    return launch_reduce<double>(B, cost, res, active, status, out5, (hipStream_t)stream);
}
ISLS_API int isls_reduce_convergence_f32(int32_t B, const void *cost, const void *res, const int32_t *active,
                                         const int32_t *status, void *out5, void *stream)
{
    return launch_reduce<float>(B, cost, res, active, status, out5, (hipStream_t)stream);
}
ISLS_API int isls_reduce_convergence_table_f64(int32_t B, const void *cost, const void *res, const int32_t *active,
                                               const int32_t *status, void *table, int32_t rank, int32_t world, void *stream)
{
    if (rank < 0) return ISLS_ERR_ARG;
    return launch_reduce<double>(B, cost, res, active, status, table, (hipStream_t)stream, rank, world);
}
ISLS_API int isls_reduce_convergence_table_f32(int32_t B, const void *cost, const void *res, const int32_t *active,
                                               const int32_t *status, void *table, int32_t rank, int32_t world, void *stream)
{
    if (rank < 0) return ISLS_ERR_ARG;
    return launch_reduce<float>(B, cost, res, active, status, table, (hipStream_t)stream, rank, world);
}
ISLS_API int isls_ilqr_admm_outer_f64(const isls_outer_args *a, void *stream)
{
    return a ? outer_iteration<double>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}
ISLS_API int isls_ilqr_admm_outer_f32(const isls_outer_args *a, void *stream)
{
    return a ? outer_iteration<float>(*a, (hipStream_t)stream) : ISLS_ERR_ARG;
}

ISLS_API int32_t isls_ff_segments(int32_t N, int32_t nseg_requested, int32_t *seg_len)
{
    int len = 0;
    const int n = ff_segments(N, nseg_requested, &len);
    if (seg_len) *seg_len = len;
    return n;
}

ISLS_API int64_t isls_ff_record_elems(int32_t B, int32_t N, int32_t n, int32_t m)
{
    if (B < 0 || N < 1 || n < 1 || m < 1 || n + m > kWave) return 0;
    const int64_t tpw = kWave / (n + m), rw = rec_stride(n, m);
    return ((B + tpw - 1) / tpw) * tpw * N * rw;
}

ISLS_API int isls_version(void) { return ISLS_VERSION; }

ISLS_API int32_t isls_dims_supported(int32_t n, int32_t m) { return dims_supported(n, m) ? 1 : 0; }

ISLS_API int32_t isls_dims_generic(int32_t n, int32_t m) { return dims_generic(n, m) ? 1 : 0; }

ISLS_API const char *isls_error_string(int code)
{
    switch (code) {
        case ISLS_OK: return "ok";
        case ISLS_ERR_ARG: return "bad argument (null pointer or dimension)";
        case ISLS_ERR_UNSUPPORTED: return "unsupported (n,m) pair, model, projection or L";
        case ISLS_ERR_LAUNCH: return "HIP launch failed";
        default: return "unknown";
    }
}

ISLS_API void *isls_timing_create(void) { return new (std::nothrow) Timing(); }

ISLS_API void isls_timing_destroy(void *h)
{
    Timing *tm = static_cast<Timing *>(h);
    if (!tm) return;
    for (int k = 0; k < Timing::kKinds; ++k)
        for (auto &pr : tm->ev[k]) {
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
    delete tm;
}

// Start a new measurement window (the recorded events of the previous one are reused).
ISLS_API int isls_timing_reset(void *h)
{
    Timing *tm = static_cast<Timing *>(h);
    if (!tm) return ISLS_ERR_ARG;
    tm->paused = false;
    for (int k = 0; k < Timing::kKinds; ++k) tm->used[k] = 0;
    return ISLS_OK;
}

// Suspend / resume event recording inside a measurement window without resetting it (an event pair per launch
// costs a few microseconds of queue bubbles; sampling every k-th step keeps the timed region representative).
ISLS_API int isls_timing_pause(void *h, int paused)
{
    Timing *tm = static_cast<Timing *>(h);
    if (!tm) return ISLS_ERR_ARG;
    tm->paused = paused != 0;
    return ISLS_OK;
}

// Sum of the event-bracketed durations of kernel family `kind` since isls_timing_reset(); *count = number of
// launches.  Synchronises on the recorded events (call it outside the timed region).
ISLS_API double isls_timing_read_ms(void *h, int kind, int *count)
{
    Timing *tm = static_cast<Timing *>(h);
    if (!tm || kind < 0 || kind >= Timing::kKinds) return -1.0;
    double total = 0.0;
    const size_t n = tm->used[kind];
    for (size_t i = 0; i < n; ++i) {
        float ms = 0.f;
        hipEventSynchronize(tm->ev[kind][i].second);
        if (hipEventElapsedTime(&ms, tm->ev[kind][i].first, tm->ev[kind][i].second) == hipSuccess) total += ms;
    }
    if (count) *count = (int)n;
    return total;
}
