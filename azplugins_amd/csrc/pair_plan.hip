// pair_plan.hip -- builds the tile plan (pair_plan.hpp) from a HOOMD-format
// neighbor list, on the GPU, and owns its device workspace.
//
// Two passes over the list, both one workgroup per tile:
//   count: LDS hash-set of the tile's neighbor indices -> number of staged
//          particles (a tile whose neighbor set exceeds 4095 particles -- e.g.
//          unsorted particle order -- invalidates the plan) and per-slice chunk
//          counts
//   fill : same hash-set, compacted and bitonic-sorted in LDS -> stage_idx; then
//          every row entry is translated (binary search in LDS) to a u16 byte
//          offset and written as 16-byte chunks in the force kernel's own
//          lane order (coalesced 1 KiB stores).
// Host-side exclusive scans sit between the passes (plan build is a sync point,
// like HOOMD's own neighbor-list overflow check).
#include <algorithm>
#include <vector>

#include "azp_device.hpp"
#include "pair_plan.hpp"

namespace azp
{
struct PlanKArgs
    {
    const double* pos;
    const uint32_t* n_neigh;
    const uint32_t* nlist;
    const uint64_t* head_list;
    uint32_t* tile_nstage;
    const uint64_t* tile_head;
    uint32_t* stage_idx;
    uint32_t* slice_K;
    const uint64_t* slice_head;
    uint4* cnl;
    uint32_t* flags;
    BoxDev box;
    uint32_t N;
    };

__device__ __forceinline__ uint32_t plan_hash(uint32_t j) { return (j * 2654435761u) >> 19; } // 13 bits


template<int TPP, bool FILL> __global__ void __launch_bounds__(256) plan_build_kernel(const PlanKArgs a)
    {
    constexpr int TB = 256 / TPP; // particles per tile
    constexpr int PW = 64 / TPP;  // particles per wave (slice)
    __shared__ uint32_t table[PLAN_HASH_CAP];
    __shared__ uint32_t list[FILL ? 4096 : 1];
    __shared__ uint32_t s_n, s_overflow;

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const uint32_t tile = blockIdx.x;
    const uint32_t first = tile * TB;
    const uint32_t count = min((uint32_t)TB, a.N - first);

    for (uint32_t t = tid; t < PLAN_HASH_CAP; t += 256)
        table[t] = PLAN_EMPTY;
    if (tid == 0) { s_n = 0; s_overflow = 0; }
    __syncthreads();

    // ---- hash-set of all neighbor indices of the tile ----
    for (uint32_t p = wave; p < count; p += 4)
        {
        const uint32_t i = first + p;
        const uint32_t n = a.n_neigh[i];
        const uint32_t* row = a.nlist + a.head_list[i];
        for (uint32_t k = lane; k < n; k += 64)
            {
            const uint32_t j = row[k];
            uint32_t h = plan_hash(j);
            bool done = false;
            for (uint32_t probe = 0; probe < PLAN_HASH_CAP && !done; ++probe)
                {
                const uint32_t old = atomicCAS(&table[h], PLAN_EMPTY, j);
                if (old == PLAN_EMPTY || old == j)
                    done = true;
                else
                    h = (h + 1) & (PLAN_HASH_CAP - 1);
                }
            if (!done)
                s_overflow = 1;
            }
        }
    __syncthreads();

    if (!FILL)
        {
        // ---- count uniques ----
        uint32_t mine = 0;
        for (uint32_t t = tid; t < PLAN_HASH_CAP; t += 256)
            mine += (table[t] != PLAN_EMPTY);
        atomicAdd(&s_n, mine);
        // per-slice chunk count: K = ceil(max row length / (8 * TPP))
        uint32_t nrow = 0;
        if (lane < PW && wave * PW + lane < count)
            nrow = a.n_neigh[first + wave * PW + lane];
        for (int off = 32; off > 0; off >>= 1)
            nrow = max(nrow, (uint32_t)__shfl_xor((int)nrow, off, 64));
        if (lane == 0)
            a.slice_K[tile * 4 + wave] = (nrow + 8 * TPP - 1) / (8 * TPP);
        __syncthreads();
        if (tid == 0)
            {
            a.tile_nstage[tile] = s_n;
            if (s_overflow || s_n > PLAN_MAX_STAGE)
                atomicOr(&a.flags[1], 1u);
            atomicMax(&a.flags[2], s_n);
            }
        return;
        }

    // ---- FILL: compact, sort, publish the stage list ----
    for (uint32_t t = tid; t < 4096; t += 256)
        list[t] = PLAN_EMPTY;
    __syncthreads();
    for (uint32_t t = tid; t < PLAN_HASH_CAP; t += 256)
        {
        const uint32_t j = table[t];
        if (j != PLAN_EMPTY)
            {
            const uint32_t slot = atomicAdd(&s_n, 1u);
            if (slot < 4096)
                list[slot] = j;
            }
        }
    __syncthreads();
    const uint32_t n_stage = min(s_n, (uint32_t)PLAN_MAX_STAGE);
    uint32_t npow = 64;
    while (npow < n_stage) npow <<= 1;
    // bitonic sort of list[0..npow) (padding = 0xFFFFFFFF sorts to the end)
    for (uint32_t size = 2; size <= npow; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1)
            {
            for (uint32_t t = tid; t < (npow >> 1); t += 256)
                {
                const uint32_t lo = 2 * t - (t & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const uint32_t x = list[lo], y = list[hi];
                if ((x > y) == up)
                    {
                    list[lo] = y;
                    list[hi] = x;
                    }
                }
            __syncthreads();
            }
    uint32_t* stage = a.stage_idx + a.tile_head[tile];
    for (uint32_t t = tid; t < n_stage; t += 256)
        stage[t] = list[t];

    // ---- translate rows into the force kernel's chunk order ----
    const uint32_t pl = lane / TPP, sub = lane % TPP;
    const uint32_t slice = tile * 4 + wave;
    const uint32_t K = a.slice_K[slice];
    uint4* out = a.cnl + (a.slice_head[slice] * 64ull);
    const uint32_t pidx = wave * PW + pl;
    uint32_t n = 0;
    const uint32_t* row = a.nlist;
    if (pidx < count)
        {
        n = a.n_neigh[first + pidx];
        row = a.nlist + a.head_list[first + pidx];
        }
    for (uint32_t kk = 0; kk < K; ++kk)
        {
        const uint32_t base = (kk * TPP + sub) * 8;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            {
            const uint32_t k = base + e;
            uint32_t off = 0;
            if (k < n)
                {
                const uint32_t j = row[k];
                uint32_t lo = 0, hi = n_stage;
                while (lo < hi)
                    {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (list[mid] < j) lo = mid + 1; else hi = mid;
                    }
                off = (lo + 1) * 8;
                }
            w[e >> 1] |= off << (16 * (e & 1));
            }
        out[(uint64_t)kk * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }

template<class T> static hipError_t ensure(T*& ptr, size_t& cap, size_t need)
    {
    if (need <= cap && ptr)
        return hipSuccess;
    if (ptr)
        {
        hipError_t e = hipFree(ptr);
        if (e != hipSuccess) return e;
        ptr = nullptr;
        }
    const size_t newcap = need + need / 8 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), newcap * sizeof(T));
    cap = (e == hipSuccess) ? newcap : 0;
    return e;
    }

static void plan_free(PairPlan& p)
    {
    if (p.d_tile_nstage) (void)hipFree(p.d_tile_nstage);
    if (p.d_tile_head) (void)hipFree(p.d_tile_head);
    if (p.d_stage_idx) (void)hipFree(p.d_stage_idx);
    if (p.d_slice_K) (void)hipFree(p.d_slice_K);
    if (p.d_slice_head) (void)hipFree(p.d_slice_head);
    if (p.d_cnl) (void)hipFree(p.d_cnl);
    if (p.d_flags) (void)hipFree(p.d_flags);
    p = PairPlan();
    }

template<int TPP> static void launch_plan_kernel(bool fill, const PlanKArgs& k, uint32_t n_tiles, hipStream_t s)
    {
    if (fill)
        hipLaunchKernelGGL((plan_build_kernel<TPP, true>), dim3(n_tiles), dim3(256), 0, s, k);
    else
        hipLaunchKernelGGL((plan_build_kernel<TPP, false>), dim3(n_tiles), dim3(256), 0, s, k);
    }

static void launch_plan(uint32_t tpp, bool fill, const PlanKArgs& k, uint32_t n_tiles, hipStream_t s)
    {
    if (tpp == 4) launch_plan_kernel<4>(fill, k, n_tiles, s);
    else if (tpp == 2) launch_plan_kernel<2>(fill, k, n_tiles, s);
    else launch_plan_kernel<1>(fill, k, n_tiles, s);
    }

#define AZP_HIP_TRY(expr)                      \
    do                                         \
        {                                      \
        hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) return (int)e_;  \
        } while (0)

static int plan_build_tpp(PairPlan& p, const azp_pair_args& args, uint32_t tpp, hipStream_t s)
    {
    p.valid = false;
    p.invalid_reason = 0;
    p.tpp = tpp;
    p.tile = 256 / tpp;
    p.n_tiles = (args.N + p.tile - 1) / p.tile;
    p.n_slices = p.n_tiles * 4;

    size_t cap_heads_t = p.d_tile_head ? p.cap_tiles : 0, cap_heads_s = p.d_slice_head ? p.cap_slices : 0;
    AZP_HIP_TRY(ensure(p.d_tile_nstage, p.cap_tiles, p.n_tiles));
    AZP_HIP_TRY(ensure(p.d_tile_head, cap_heads_t, p.n_tiles));
    AZP_HIP_TRY(ensure(p.d_slice_K, p.cap_slices, p.n_slices));
    AZP_HIP_TRY(ensure(p.d_slice_head, cap_heads_s, p.n_slices));
    size_t cap_flags = p.d_flags ? 4 : 0;
    AZP_HIP_TRY(ensure(p.d_flags, cap_flags, 4));
    AZP_HIP_TRY(hipMemsetAsync(p.d_flags, 0, 4 * sizeof(uint32_t), s));

    PlanKArgs k;
    k.pos = args.d_pos;
    k.n_neigh = args.d_n_neigh;
    k.nlist = args.d_nlist;
    k.head_list = args.d_head_list;
    k.tile_nstage = p.d_tile_nstage;
    k.tile_head = p.d_tile_head;
    k.stage_idx = nullptr;
    k.slice_K = p.d_slice_K;
    k.slice_head = p.d_slice_head;
    k.cnl = nullptr;
    k.flags = p.d_flags;
    k.box = make_box_dev(args.box);
    k.N = args.N;
    launch_plan(tpp, false, k, p.n_tiles, s);
    AZP_HIP_TRY(hipGetLastError());

    std::vector<uint32_t> h_nstage(p.n_tiles), h_K(p.n_slices);
    uint32_t h_flags[4];
    AZP_HIP_TRY(hipMemcpyAsync(h_nstage.data(), p.d_tile_nstage, sizeof(uint32_t) * p.n_tiles, hipMemcpyDeviceToHost, s));
    AZP_HIP_TRY(hipMemcpyAsync(h_K.data(), p.d_slice_K, sizeof(uint32_t) * p.n_slices, hipMemcpyDeviceToHost, s));
    AZP_HIP_TRY(hipMemcpyAsync(h_flags, p.d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, s));
    AZP_HIP_TRY(hipStreamSynchronize(s));
    p.max_stage = h_flags[2];
    if (h_flags[1])
        {
        p.invalid_reason = 2;
        return AZP_SUCCESS;
        }
    std::vector<uint64_t> h_thead(p.n_tiles), h_shead(p.n_slices);
    uint64_t acc = 0;
    for (uint32_t t = 0; t < p.n_tiles; ++t) { h_thead[t] = acc; acc += h_nstage[t]; }
    p.total_stage = acc;
    acc = 0;
    for (uint32_t t = 0; t < p.n_slices; ++t) { h_shead[t] = acc; acc += h_K[t]; }
    p.total_chunks = acc;
    p.cap = p.max_stage + 1 <= 1024 ? 1024 : (p.max_stage + 1 <= 2048 ? 2048 : 4096);

    AZP_HIP_TRY(ensure(p.d_stage_idx, p.cap_stage, (size_t)std::max<uint64_t>(p.total_stage, 1)));
    AZP_HIP_TRY(ensure(p.d_cnl, p.cap_cnl, (size_t)std::max<uint64_t>(p.total_chunks * 64, 1)));
    AZP_HIP_TRY(hipMemcpyAsync(p.d_tile_head, h_thead.data(), sizeof(uint64_t) * p.n_tiles, hipMemcpyHostToDevice, s));
    AZP_HIP_TRY(hipMemcpyAsync(p.d_slice_head, h_shead.data(), sizeof(uint64_t) * p.n_slices, hipMemcpyHostToDevice, s));
    k.stage_idx = p.d_stage_idx;
    k.cnl = p.d_cnl;
    launch_plan(tpp, true, k, p.n_tiles, s);
    AZP_HIP_TRY(hipGetLastError());
    // the host vectors must outlive the async copies
    AZP_HIP_TRY(hipStreamSynchronize(s));
    p.valid = true;
    return AZP_SUCCESS;
    }

int plan_build(PairPlan& p, const azp_pair_args& args, hipStream_t s)
    {
    p.valid = false;
    p.invalid_reason = 0;
    p.N = args.N;
    p.n_max = args.n_max;
    p.nlist_ptr = args.d_nlist;
    p.head_ptr = args.d_head_list;
    p.size_nlist = args.size_nlist;
    ++p.builds;
    if (args.N == 0)
        return AZP_SUCCESS;
    // Measured on MI355X (PerturbedLJ, <n> = 136): 256-particle tiles (one lane per
    // particle) beat 128 and 64 -- fewer staged loads per particle, less row padding.
    // Fall back to smaller tiles only when a tile's neighbor set overflows the LDS
    // budget (4095 staged particles), or for very long rows.
    uint32_t tpp = args.threads_per_particle;
    const bool fixed = (tpp == 1 || tpp == 2 || tpp == 4);
    if (!fixed)
        {
        const double mean = (args.size_nlist && args.N) ? (double)args.size_nlist / args.N : 64.0;
        tpp = mean >= 1024.0 ? 4 : (mean >= 512.0 ? 2 : 1);
        }
    for (;;)
        {
        const int rc = plan_build_tpp(p, args, tpp, s);
        if (rc != AZP_SUCCESS || p.valid || fixed || tpp == 4 || p.invalid_reason != 2)
            return rc;
        tpp *= 2; // halve the tile and retry
        }
    }

} // namespace azp

extern "C" int azp_pair_plan_create(azp_pair_plan** out)
    {
    if (!out)
        return AZP_ERROR_INVALID_ARGUMENT;
    *out = reinterpret_cast<azp_pair_plan*>(new azp::PairPlan());
    return AZP_SUCCESS;
    }

extern "C" void azp_pair_plan_destroy(azp_pair_plan* plan)
    {
    if (!plan)
        return;
    azp::PairPlan* p = reinterpret_cast<azp::PairPlan*>(plan);
    azp::plan_free(*p);
    delete p;
    }

extern "C" int azp_pair_plan_build(azp_pair_plan* plan, const azp_pair_args* args, void* stream)
    {
    if (!plan || !args || !args->d_pos || !args->d_n_neigh || !args->d_nlist || !args->d_head_list)
        return AZP_ERROR_INVALID_ARGUMENT;
    return azp::plan_build(*reinterpret_cast<azp::PairPlan*>(plan), *args, static_cast<hipStream_t>(stream));
    }

extern "C" int azp_pair_plan_query(const azp_pair_plan* plan, azp_pair_plan_info* info)
    {
    if (!plan || !info)
        return AZP_ERROR_INVALID_ARGUMENT;
    const azp::PairPlan* p = reinterpret_cast<const azp::PairPlan*>(plan);
    info->valid = p->valid ? 1 : 0;
    info->invalid_reason = p->invalid_reason;
    info->threads_per_particle = p->tpp;
    info->tile_size = p->tile;
    info->lds_slots = p->cap;
    info->n_tiles = p->n_tiles;
    info->max_stage = p->max_stage;
    info->total_stage = p->total_stage;
    info->compiled_bytes = p->total_chunks * 64ull * 16ull;
    info->builds = p->builds;
    return AZP_SUCCESS;
    }
