// lds_bench.hip -- cost of the tile kernel's LDS gather pattern (3 x ds_read_b64 per
// neighbor from SoA x|y|z) under different slot orderings. Modes:
//   0 random slots            1 diagonal: slot % 32 == (lane + step) % 32
//   2 78 % diagonal + 22 % random misfits     3 broadcast (all lanes one slot)
//   4 consecutive (lane)      5 sorted-like: slot = base(step) + small random spread
//   6 diagonal mod 16         7-10: diagonal mod 16 with 16 / 10 / 5 / 2 % random misfits
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_bench.hip -o tools/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

constexpr int CAP = 1536;
constexpr int K = 24; // chunks of 8 entries per lane

__global__ void __launch_bounds__(256, 4) gather(const uint4* __restrict__ chunks, double* out, int reps)
    {
    __shared__ double s[3 * CAP];
    for (int i = threadIdx.x; i < 3 * CAP; i += 256)
        s[i] = (double)i;
    __syncthreads();
    const char* bx = reinterpret_cast<const char*>(s);
    const uint4* c = chunks + (size_t)(threadIdx.x >> 6) * K * 64 + (threadIdx.x & 63);
    double a0 = 0, a1 = 0, a2 = 0;
    for (int r = 0; r < reps; ++r)
        #pragma unroll 2
        for (int k = 0; k < K; ++k)
            {
            const uint4 u = c[k * 64];
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
            double x[8], y[8], z[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                {
                const uint32_t off = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
                x[e] = *reinterpret_cast<const double*>(bx + off);
                y[e] = *reinterpret_cast<const double*>(bx + off + CAP * 8);
                z[e] = *reinterpret_cast<const double*>(bx + off + CAP * 16);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; e += 2)
                {
                a0 += x[e] * x[e + 1];
                a1 += y[e] * y[e + 1];
                a2 += z[e] * z[e + 1];
                }
            __builtin_amdgcn_sched_barrier(0);
            }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2;
    }

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd()
    {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 20);
    }

int main()
    {
    const int nblocks = 256 * 4 * 8;
    const int reps = 8;
    double* d_out;
    hipMalloc(&d_out, sizeof(double) * nblocks * 256);
    uint4* d_chunks;
    hipMalloc(&d_chunks, sizeof(uint4) * 4 * K * 64);
    for (int mode = 0; mode <= 10; ++mode)
        {
        std::vector<uint16_t> h(4 * K * 64 * 8);
        for (int wave = 0; wave < 4; ++wave)
            for (int k = 0; k < K; ++k)
                {
                for (int e = 0; e < 8; ++e)
                    {
                    const int step = k * 8 + e;
                    const uint32_t base = rnd() % (CAP - 128);
                    for (int lane = 0; lane < 64; ++lane)
                        {
                        uint32_t slot;
                        const uint32_t diag = (rnd() % (CAP / 32)) * 32 + ((lane + step) & 31);
                        switch (mode)
                            {
                        case 0: slot = rnd() % CAP; break;
                        case 1: slot = diag; break;
                        case 2: slot = (rnd() % 100 < 22) ? rnd() % CAP : diag; break;
                        case 3: slot = base; break;
                        case 4: slot = lane; break;
                        case 6: slot = (rnd() % (CAP / 16)) * 16 + ((lane + step) & 15); break;
                        case 7: slot = (rnd() % 100 < 16) ? rnd() % CAP : (rnd() % (CAP / 16)) * 16 + ((lane + step) & 15); break;
                        case 8: slot = (rnd() % 100 < 10) ? rnd() % CAP : (rnd() % (CAP / 16)) * 16 + ((lane + step) & 15); break;
                        case 9: slot = (rnd() % 100 < 5) ? rnd() % CAP : (rnd() % (CAP / 16)) * 16 + ((lane + step) & 15); break;
                        case 10: slot = (rnd() % 100 < 2) ? rnd() % CAP : (rnd() % (CAP / 16)) * 16 + ((lane + step) & 15); break;
                        default: slot = base + rnd() % 96; break;
                            }
                        h[(((size_t)wave * K + k) * 64 + lane) * 8 + e] = (uint16_t)(slot * 8);
                        }
                    }
                }
        hipMemcpy(d_chunks, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(gather, dim3(nblocks), dim3(256), 0, 0, d_chunks, d_out, reps);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(gather, dim3(nblocks), dim3(256), 0, 0, d_chunks, d_out, reps);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // gathers (wave-level ds_read_b64) per CU: nblocks/256 blocks per CU x 4 waves x reps x K x 8 x 3
        const double reads_per_cu = (double)nblocks / 256 * 4 * reps * K * 8 * 3;
        printf("mode %d: %.3f ms, %.2f cycles per 64-lane ds_read_b64 per CU (2.4 GHz)\n", mode, ms, ms * 1e-3 * 2.4e9 / reads_per_cu);
        }
    return 0;
    }
