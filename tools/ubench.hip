// Micro-benchmarks used to calibrate the kernel design (not part of the product): cost of fp64 FMAs issued by
// one or two wavefronts per SIMD, dependent chains, LDS-broadcast-fed FMAs.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K> __device__ __forceinline__ double rowbcast(double v)   // lane K of each 16-lane row -> whole row
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, unsigned long long *cyc)
{
    __shared__ double lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 8 independent chains, 8 FMAs per iteration
            a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
            a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
        } else if (MODE == 1) {  // one dependent chain of 8
            a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
            a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
        } else if (MODE == 2) {  // 8 FMAs fed by LDS broadcast reads (address depends on it to stop hoisting)
            const double *p = lds + ((it & 7) * 8);
            a0 = __builtin_fma(a0, p[0], c); a1 = __builtin_fma(a1, p[1], c); a2 = __builtin_fma(a2, p[2], c); a3 = __builtin_fma(a3, p[3], c);
            a4 = __builtin_fma(a4, p[4], c); a5 = __builtin_fma(a5, p[5], c); a6 = __builtin_fma(a6, p[6], c); a7 = __builtin_fma(a7, p[7], c);
        } else if (MODE == 4) {  // 6x6 mat-vec recursion, x exchanged inside each 16-lane row by DPP row_newbcast
            double acc = c;
            acc = __builtin_fma(a1, rowbcast<0>(a0), acc); acc = __builtin_fma(a2, rowbcast<1>(a0), acc);
            acc = __builtin_fma(a3, rowbcast<2>(a0), acc); acc = __builtin_fma(a4, rowbcast<3>(a0), acc);
            acc = __builtin_fma(a5, rowbcast<4>(a0), acc); acc = __builtin_fma(a6, rowbcast<5>(a0), acc);
            a0 = acc * 1e-3;
        } else if (MODE == 5) {  // the same recursion with the exchange through LDS (write, wave fence, 6 broadcast reads)
            lds[threadIdx.x] = a0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double *p = lds + (threadIdx.x & 48);
            double acc = c;
            acc = __builtin_fma(a1, p[0], acc); acc = __builtin_fma(a2, p[1], acc); acc = __builtin_fma(a3, p[2], acc);
            acc = __builtin_fma(a4, p[3], acc); acc = __builtin_fma(a5, p[4], acc); acc = __builtin_fma(a6, p[5], acc);
            a0 = acc * 1e-3;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == 6) {  // 8 independent 32-bit integer VALU ops
            int *q = reinterpret_cast<int *>(&a0);
            (void)q;
            unsigned u0 = __double2loint(a0), u1 = __double2loint(a1), u2 = __double2loint(a2), u3 = __double2loint(a3);
            unsigned u4 = __double2hiint(a0), u5 = __double2hiint(a1), u6 = __double2hiint(a2), u7 = __double2hiint(a3);
            u0 = u0 * 3u + it; u1 = u1 * 5u + it; u2 = u2 * 7u + it; u3 = u3 * 9u + it;
            u4 = (u4 ^ it) + 1u; u5 = (u5 ^ it) + 3u; u6 = (u6 ^ it) + 5u; u7 = (u7 ^ it) + 7u;
            a0 = __hiloint2double(u4, u0); a1 = __hiloint2double(u5, u1); a2 = __hiloint2double(u6, u2); a3 = __hiloint2double(u7, u3);
        } else if (MODE == 3) {  // fp32: 8 independent FMAs
            float f0 = (float)a0, f1 = (float)a1;
            (void)f0; (void)f1;
            a0 = (double)__builtin_fmaf((float)a0, 1.0000001f, 1e-9f); a1 = (double)__builtin_fmaf((float)a1, 1.0000001f, 1e-9f);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks, int iters)
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, sizeof(double) * blocks * 64); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s blocks %5d: %.3f ms, s_memtime cycles/iter %.1f, ns/iter %.2f (8 ops per iter)\n", name, blocks, ms,
           (double)h / iters, ms * 1e6 / iters);
    hipFree(out); hipFree(cyc);
}

int main()
{
    const int it = 200000;
    for (int blocks : {256, 1024, 2048, 4096}) {
        run<0>("f64 8 indep FMA", blocks, it);
        run<1>("f64 8 dependent FMA", blocks, it);
        run<2>("f64 8 FMA + LDS bcast", blocks, it);
        run<4>("matvec6 chain, DPP", blocks, it);
        run<5>("matvec6 chain, LDS", blocks, it);
        run<6>("int32 8 ops(16 instr)", blocks, it);
    }
    return 0;
}
