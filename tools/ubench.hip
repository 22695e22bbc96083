// ubench.hip -- per-instruction VALU issue cost on gfx950 for the ops the pair
// kernels use (fp64 fma/mul/add, v_rcp_f64, v_cmp_f64, v_cndmask_b32, cvt),
// at 1..4 waves per SIMD, plus the accuracy of v_rcp_f64 / v_rsq_f64.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template<int OP> __global__ void k(double* out, int iters, double seed)
    {
    double a0 = seed + threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    int i0 = threadIdx.x, i1 = i0 + 1;
    for (int it = 0; it < iters; ++it)
        {
        if (OP == 0) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));) }
        if (OP == 1) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
        if (OP == 2) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 3) { REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 4) { REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc\n v_cmp_lt_f64 vcc, %2, %0\n v_cndmask_b32 %3, %3, %1, vcc\n v_cmp_lt_f64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc\n v_cmp_lt_f64 vcc, %2, %0\n v_cndmask_b32 %3, %3, %1, vcc" : "+v"(a0), "+v"(i0), "+v"(a1), "+v"(i1) :: "vcc");) }
        if (OP == 5) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(i0), "+v"(i1) :: "vcc");) }
        if (OP == 6) { float f0 = (float)a0, f1 = (float)a1; REP8(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0" : "+v"(f0), "+v"(f1));) a0 += f0; a1 += f1; }
        if (OP == 7) { REP8(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 8) { float f0, f1, f2, f3; REP8(asm volatile("v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1\n v_cvt_f32_f64 %6, %2\n v_cvt_f32_f64 %7, %3\n v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3));) }
        if (OP == 9) { float f0 = (float)a0, f1 = (float)a1; REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1" : "+v"(f0), "+v"(f1));) a0 += f0; a1 += f1; }
        }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1;
    }

__global__ void acc(const double* x, double* r1, double* r2, int n)
    {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        {
        r1[i] = __builtin_amdgcn_rcp(x[i]);
        r2[i] = __builtin_amdgcn_rsq(x[i]);
        }
    }

template<int OP> void run(const char* name, double* d, int waves_per_simd)
    {
    const int iters = 2000;
    const int nblk = 256 * waves_per_simd; // 256 CUs x (4 waves = 1 per SIMD) per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<nblk, 256>>>(d, 10, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<nblk, 256>>>(d, iters, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions per wave = iters*64; per SIMD: waves_per_simd waves
    const double inst_per_simd = (double)iters * 64 * waves_per_simd;
    printf("%-22s waves/SIMD=%d  %.3f ms  ns/inst/SIMD=%.3f  (cycles @2.1GHz: %.2f)\n", name, waves_per_simd, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.1);
    }

int main()
    {
    double* d; hipMalloc(&d, sizeof(double) * 256 * 256 * 8);
    for (int w : {1, 2, 4})
        {
        run<0>("v_fma_f64", d, w); run<1>("v_mul_f64", d, w); run<2>("v_add_f64", d, w); run<3>("v_rcp_f64", d, w);
        run<7>("v_rsq_f64", d, w); run<4>("v_cmp_f64+cndmask (x2)", d, w); run<5>("v_cndmask_b32", d, w);
        run<6>("v_fma_f32", d, w); run<8>("v_cvt f64<->f32", d, w); run<9>("v_rcp_f32", d, w);
        }
    // accuracy
    const int n = 1 << 20;
    std::vector<double> x(n), r1(n), r2(n);
    for (int i = 0; i < n; ++i) x[i] = 0.5 + 9.5 * (double)((i * 2654435761u) & 0xffffff) / 16777216.0 * (1 + i * 1e-7);
    double *dx, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    acc<<<n / 256, 256>>>(dx, d1, d2, n);
    hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i)
        {
        e1 = fmax(e1, fabs(r1[i] * x[i] - 1.0));
        e2 = fmax(e2, fabs(r2[i] * sqrt(x[i]) - 1.0));
        }
    printf("v_rcp_f64 max rel err = %.3e (2^%.1f)   v_rsq_f64 max rel err = %.3e (2^%.1f)\n", e1, log2(e1), e2, log2(e2));
    return 0;
    }
