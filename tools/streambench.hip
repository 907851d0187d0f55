// Micro-benchmark (not part of the product): HBM throughput of the solver's access pattern -- tens of thousands of
// concurrent per-trajectory streams that each advance by a small chunk per step -- against the burst size.
//   hipcc --offload-arch=gfx950 -O3 -o streambench streambench.hip
// A wavefront owns SLOTS trajectories x ARRAYS streams.  Per step and stream it reads one chunk of CH doubles
// (lanes of a slot read consecutive words).  GROUP steps are fetched together (one burst of GROUP*CH doubles per
// stream), then `work` dependent FMAs per step imitate the recursion the real kernels run between fetches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int GROUP, int ARRAYS, int PER>
__global__ __launch_bounds__(64) void k(const double *__restrict__ base, double *out, int steps, int ch, int slots, int glanes,
                                        long stream_stride, int work, int ntraj)
{
    const int lane = threadIdx.x;
    const int g = lane / glanes, i = lane - g * glanes;
    const bool on = g < slots;
    long traj = (long)blockIdx.x * slots + (on ? g : 0);
    if (traj >= ntraj) traj = ntraj - 1;                        // surplus slots of the last block shadow a valid stream
    double acc = 0.0;
    double buf[2][GROUP * ARRAYS * PER];
    auto fetch = [&](int t0, double *dst) {
#pragma unroll
        for (int a = 0; a < ARRAYS; ++a) {
            const double *s = base + (traj * ARRAYS + a) * stream_stride + (long)t0 * ch;
#pragma unroll
            for (int u = 0; u < GROUP; ++u)
#pragma unroll
                for (int w = 0; w < PER; ++w) {
                    const int e = i + glanes * w;
                    dst[(a * GROUP + u) * PER + w] = s[u * ch + (e < ch ? e : ch - 1)];
                }
        }
    };
    fetch(0, buf[0]);
    for (int t0 = 0; t0 < steps; t0 += 2 * GROUP) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tn = t0 + (half + 1) * GROUP;
            fetch(tn < steps ? tn : 0, buf[half ^ 1]);
#pragma unroll
            for (int u = 0; u < GROUP; ++u) {
                double x = 0.0;
#pragma unroll
                for (int a = 0; a < ARRAYS; ++a)
#pragma unroll
                    for (int w = 0; w < PER; ++w) x += buf[half][(a * GROUP + u) * PER + w];
                for (int q = 0; q < work; ++q) x = __builtin_fma(x, 1.0000001, 1e-9);   // the recursion between fetches
                acc += x;
            }
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int GROUP, int ARRAYS, int PER>
static void run(int traj, int steps, int work)
{
    const int glanes = 9, slots = 7, ch = 9 * PER;
    const long stream_stride = (long)steps * ch;
    const size_t words = (size_t)traj * ARRAYS * stream_stride;
    double *base, *out;
    hipMalloc(&base, words * 8); hipMemset(base, 0, words * 8);
    const int blocks = (traj + slots - 1) / slots;
    hipMalloc(&out, (size_t)blocks * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<GROUP, ARRAYS, PER><<<blocks, 64>>>(base, out, steps, ch, slots, glanes, stream_stride, work, traj);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<GROUP, ARRAYS, PER><<<blocks, 64>>>(base, out, steps, ch, slots, glanes, stream_stride, work, traj);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("traj %5d arrays %2d chunk %3d B group %2d (burst %4d B) work %4d: %.3f ms  %.2f TB/s  (%d waves)\n", traj, ARRAYS,
           ch * 8, GROUP, GROUP * ch * 8, work, ms, words * 8.0 / ms * 1e-9, blocks);
    hipFree(base); hipFree(out);
}

int main(int argc, char **argv)
{
    const int steps = 96;
    for (int traj : {4096, 8192}) {
        for (int work : {0, 200}) {
            run<1, 14, 1>(traj, steps, work);     // 14 streams of 72 B per step (what riccati_ff does today)
            run<2, 14, 1>(traj, steps, work);
            run<4, 14, 1>(traj, steps, work);
            run<1, 7, 2>(traj, steps, work);      // 7 streams of 144 B
            run<2, 7, 2>(traj, steps, work);
            run<4, 7, 2>(traj, steps, work);
            run<1, 3, 4>(traj, steps, work);      // 3 streams of 288 B
            run<2, 3, 4>(traj, steps, work);
            run<4, 3, 4>(traj, steps, work);
            run<1, 1, 14>(traj, steps, work);     // everything packed: 1 stream of 1008 B per step
            run<2, 1, 14>(traj, steps, work);
        }
    }
    return 0;
}
