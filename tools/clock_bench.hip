// clock_bench.hip -- what does the shader clock do under an FP64-heavy kernel?
// Each block times an FP64 FMA loop with s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz): ticks ratio = MHz while the kernel ran.
// build: hipcc -O3 --offload-arch=gfx950 tools/clock_bench.hip -o tools/clock_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) fma_loop(double* out, unsigned long long* clk, int iters, int use_f64)
    {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it)
        {
        if (use_f64)
            {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                {
                a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                }
            }
        else
            {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                {
                f0 = __builtin_fmaf(f0, 1.0000001f, 1e-9f); f1 = __builtin_fmaf(f1, 1.0000001f, 1e-9f);
                f2 = __builtin_fmaf(f2, 1.0000001f, 1e-9f); f3 = __builtin_fmaf(f3, 1.0000001f, 1e-9f);
                }
            }
        }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3;
    if (threadIdx.x == 0)
        {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        }
    }

int main()
    {
    const int nblk = 256 * 4;
    double* d; unsigned long long* dc;
    (void)hipMalloc(&d, sizeof(double) * nblk * 256);
    (void)hipMalloc(&dc, sizeof(unsigned long long) * 2 * nblk);
    std::vector<unsigned long long> h(2 * nblk);
    for (int f64 = 1; f64 >= 0; --f64)
        for (int iters : {200, 2000, 20000})
            {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(fma_loop, dim3(nblk), dim3(256), 0, 0, d, dc, iters, f64);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(fma_loop, dim3(nblk), dim3(256), 0, 0, d, dc, iters, f64);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
            double mhz = 0;
            for (int b = 0; b < nblk; ++b) mhz += (double)h[2 * b] / ((double)h[2 * b + 1] / 100.0);
            mhz /= nblk;
            // wave-level FMA instructions per SIMD: 4 waves x iters x 64
            const double inst = 4.0 * iters * 64;
            printf("%s iters %6d: %.3f ms, s_memtime/s_memrealtime -> %.0f MHz; %.2f ns per wave FMA per SIMD = %.2f cycles at that clock\n",
                   f64 ? "fp64" : "fp32", iters, ms, mhz, ms * 1e6 / inst, ms * 1e6 / inst * mhz * 1e-3);
            }
    return 0;
    }
