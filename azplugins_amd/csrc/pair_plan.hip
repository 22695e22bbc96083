// pair_plan.hip -- builds the tile plan (pair_plan.hpp) from a HOOMD-format
// neighbor list, on the GPU, and owns its device workspace.
//
// Two passes over the list, both one workgroup per tile:
//   chunks: per-slice chunk counts from the row lengths (tiny kernel), host scan
//   build : one workgroup per tile. LDS hash-set of the tile's neighbor indices,
//           compacted and bitonic-sorted in LDS -> stage_idx (a tile whose neighbor
//           set exceeds PLAN_MAX_STAGE particles -- e.g. unsorted particle order --
//           invalidates the plan); the staged positions are loaded into LDS; every
//           row entry is translated (hash lookup) to a u16 byte offset, classified
//           core / near / buffer shell 0 .. PLAN_SHELLS - 1, placed bank-aware inside
//           its class, and the row is written as 16-byte chunks in the force
//           kernel's own lane order; per slice the chunk counts up to the end of the
//           in-range entries and of every shell are kept for the displacement bound.
// The host-side scan makes plan build a sync point, like HOOMD's own
// neighbor-list overflow check.
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
    const double* rcutsq;
    const double* rinnersq; // optional (may be null): "core" class radius^2 per type pair
    uint32_t* slice_Kend;   // PLAN_SHELLS + 1 per slice, zeroed before the build kernel
    uint32_t* slice_Kphase; // [2][n_slices]: core chunks (zeroed before the build kernel), sure chunks (set to ~0 before it)
    uint32_t n_slices;
    double r_list_max;      // caller's hint (r_cut_max + 2 r_buff), 0 = unknown
    double r_list_estimate; // used when r_list_max is unknown: an estimate of r_cut_max + r_buff (0: one shell holds
                            // the whole buffer)
    uint32_t bank_order;    // bank-aware row order (build option)
    uint32_t* tile_nstage;
    uint64_t* tile_head;
    uint32_t* stage_idx;
    uint32_t* slice_K;
    const uint64_t* slice_head;
    uint4* cnl;
    uint32_t* flags;
    BoxDev box;
    uint32_t N;
    uint32_t ntypes;
    uint32_t stage_stride; // stage_idx entries reserved per tile
    };

template<uint32_t HC> __device__ __forceinline__ uint32_t plan_hash(uint32_t j)
    {
    return (j * 2654435761u) >> (HC == 8192 ? 19 : 20); // 13 or 12 bits
    }

// Insert j into the open-addressing hash set; returns false if the table is full.
// Neighboring particles list mostly the same neighbors, so ~96 % of the inserts
// find their key already present: probe with a plain read first and fall back to
// the (slower) atomic only on an empty slot.
template<uint32_t HC> __device__ __forceinline__ bool plan_insert(uint32_t* table, uint32_t j)
    {
    uint32_t h = plan_hash<HC>(j);
    for (uint32_t probe = 0; probe < HC; ++probe)
        {
        uint32_t cur = table[h];
        if (cur == PLAN_EMPTY)
            cur = atomicCAS(&table[h], PLAN_EMPTY, j);
        if (cur == PLAN_EMPTY || cur == j)
            return true;
        h = (h + 1) & (HC - 1);
        }
    return false;
    }

// Position of key j in the table (j is known to be present).
template<uint32_t HC> __device__ __forceinline__ uint32_t plan_find(const uint32_t* table, uint32_t j)
    {
    uint32_t h = plan_hash<HC>(j);
    while (table[h] != j)
        h = (h + 1) & (HC - 1);
    return h;
    }

// Chunk count per slice: K = ceil(max row length in the slice / (8 * TPP)).
template<int TPP> __global__ void __launch_bounds__(256) plan_chunks_kernel(const PlanKArgs a)
    {
    constexpr int PW = 64 / TPP;
    const uint32_t slice = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t i = slice * PW + lane;
    uint32_t nrow = (lane < PW && i < a.N) ? a.n_neigh[i] : 0u;
    for (int off = 32; off > 0; off >>= 1)
        nrow = max(nrow, (uint32_t)__shfl_xor((int)nrow, off, 64));
    if (lane == 0)
        {
        a.slice_K[slice] = (nrow + 8 * TPP - 1) / (8 * TPP);
        if (nrow + 8 * TPP > PLAN_ROWBUF)
            atomicOr(&a.flags[1], 1u); // row too long for the builder's row buffer
        }
    }

// One workgroup per tile: dedup the tile's neighbor indices (LDS hash set), sort
// them (bitonic, LDS) into the stage list, then compile every row.
//
// Compiled row order: entries that are inside the cutoff at build time come
// first ("near"), the Verlet-buffer entries after them ("far"), padding last; the
// original order is kept inside each part. The force kernel tests each half chunk
// exactly and skips the arithmetic when no lane of the wave has a pair in range --
// which is what the tail of every row looks like -- so the ordering is purely a
// performance hint.
#ifndef PLAN_BANK_ORDER
#define PLAN_BANK_ORDER 1 // bank-aware ordering of the compiled rows (performance hint only)
#endif
constexpr int PLAN_BUILD_THREADS = 512; // 8 waves per tile: more independent dependency chains in flight
constexpr int PLAN_BUILD_WAVES = PLAN_BUILD_THREADS / 64;

template<int TPP, uint32_t HC>
__global__ void __launch_bounds__(PLAN_BUILD_THREADS) plan_build_kernel(const PlanKArgs a)
    {
    constexpr int TB = 256 / TPP; // particles per tile
    constexpr int PW = 64 / TPP;  // particles per slice (one force-kernel wave)
    constexpr int ITERS = PLAN_ROWBUF / 64;
    constexpr int NT = PLAN_BUILD_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    // keys and slots in separate arrays: (key, slot) pairs would save one dependent
    // LDS read per lookup but cost 8 KiB more, which halves the occupancy once a
    // tile stages more than 1,536 particles (a liquid, as opposed to the lattice)
    uint32_t* table = reinterpret_cast<uint32_t*>(s_raw);          // HC keys
    uint16_t* slot_of = reinterpret_cast<uint16_t*>(table + HC);   // HC
    uint16_t* w_rowbuf = slot_of + HC;                             // PLAN_BUILD_WAVES x PLAN_ROWBUF
    // list[4096] (sort scratch) and the staged positions (stage_stride slots) share
    // one region: the list is dead once slots and positions are published
    uint32_t* list = reinterpret_cast<uint32_t*>(w_rowbuf + PLAN_BUILD_WAVES * PLAN_ROWBUF);
    // staged positions relative to the tile's reference particle, as float4
    // (x, y, z, type): they only feed the near/far ordering hint, never a force
    float4* s_p = reinterpret_cast<float4*>(list);
    __shared__ uint32_t s_n, s_overflow;
    __shared__ float s_shell_w;   // shell width w = r_buff / PLAN_SHELLS (0: no hint, no shells)
    __shared__ float s_sure_rsq;  // one particle type: entries certainly closer than this form class sure (0: no such class)
    __shared__ float s_rcutsq[64], s_rinnersq[64], s_rcut[64]; // up to 8 types cached; more types read the global tables
    // bank-aware row ordering (TPP == 1): per wave 16 bank counters and one misfit counter
    // per class, and the list of unclaimed positions
    __shared__ uint32_t s_bank_cnt[PLAN_BUILD_WAVES][PLAN_CLASSES * 17];
    __shared__ uint16_t s_holes[PLAN_BUILD_WAVES][PLAN_ROWBUF];

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const uint32_t tile = blockIdx.x;
    const uint32_t first = tile * TB;
    const uint32_t count = min((uint32_t)TB, a.N - first);
    const bool rc_cached = a.ntypes <= 8;

    for (uint32_t t = tid; t < HC; t += NT)
        table[t] = PLAN_EMPTY;
    for (uint32_t t = tid; t < 4096; t += NT)
        list[t] = PLAN_EMPTY;
    if (tid == 0)
        {
        s_n = 0; s_overflow = 0;
        // the Verlet buffer r_cut <= r < r_cut + r_buff is cut into PLAN_SHELLS shells of width
        // w = r_buff / PLAN_SHELLS, r_buff = (r_list_max - r_cut_max) / 2; published to the host
        // through flags[3]
        double rc_max_sq = 0.0;
        for (uint32_t t = 0; t < a.ntypes * a.ntypes; ++t)
            rc_max_sq = fmax(rc_max_sq, a.rcutsq[t]);
        const double w = (a.r_list_max > 0.0) ? 0.5 * (a.r_list_max - sqrt(rc_max_sq)) / PLAN_SHELLS
                                              : (a.r_list_estimate > 0.0 ? (a.r_list_estimate - sqrt(rc_max_sq)) / PLAN_SHELLS : 0.0);
        s_shell_w = (w > 0.0) ? (float)w : 0.f;
        // class sure (pair_plan.hpp): closer than r_cut - r_buff, taken 2e-4 short for the single-precision separation;
        // class core: what the force kernel may rely on is the inner radius less 1e-4
        const float r_sure = (a.ntypes == 1 && w > 0.0) ? (float)(sqrt(rc_max_sq) - PLAN_SHELLS * w) - 2e-4f : 0.f;
        s_sure_rsq = r_sure > 0.f ? r_sure * r_sure : 0.f;
        if (blockIdx.x == 0)
            {
            a.flags[3] = (uint32_t)__float_as_int(s_shell_w);
            a.flags[0] = (uint32_t)__float_as_int(r_sure > 0.f ? r_sure + 1e-4f : 0.f);
            const float rin = (a.ntypes == 1 && a.rinnersq && a.rinnersq[0] > 0.0) ? (float)sqrt(a.rinnersq[0]) : 0.f;
            a.flags[7] = (uint32_t)__float_as_int(rin > 0.f ? rin - 1e-4f : 0.f);
            }
        }
    if (rc_cached && tid < a.ntypes * a.ntypes)
        {
        s_rcutsq[tid] = (float)a.rcutsq[tid];
        s_rinnersq[tid] = a.rinnersq ? (float)a.rinnersq[tid] : 0.f;
        }
    // single precision is enough for the classification (the in-range / buffer-shell split keeps a 1e-4 safety margin)
    const float bLx = (float)a.box.Lx, bLy = (float)a.box.Ly, bLz = (float)a.box.Lz;
    const float bLxi = (float)a.box.Lxinv, bLyi = (float)a.box.Lyinv, bLzi = (float)a.box.Lzinv;
    __syncthreads();

    if (rc_cached && tid < a.ntypes * a.ntypes)
        s_rcut[tid] = sqrtf(fmaxf(s_rcutsq[tid], 0.f));
    // (published by the barriers of the hash-set phase, long before the rows are compiled)
    // ---- hash-set of all neighbor indices of the tile: every wave takes rows
    // round-robin and issues all of a row's index loads before probing ----
    for (uint32_t p = wave; p < count; p += PLAN_BUILD_WAVES)
        {
        const uint32_t i = first + p;
        const uint32_t n = a.n_neigh[i];
        const uint32_t* row = a.nlist + a.head_list[i];
        uint32_t jj[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
            {
            const uint32_t k = (uint32_t)it * 64u + lane;
            jj[it] = (k < n) ? row[k] : PLAN_EMPTY;
            }
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
            if (jj[it] != PLAN_EMPTY && !plan_insert<HC>(table, jj[it]))
                s_overflow = 1;
        }
    __syncthreads();

#if defined(PLAN_ABLATE) && PLAN_ABLATE == 1
    return;
#endif
    // ---- compact + sort ----
    for (uint32_t t = tid; t < HC; t += NT)
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
    const bool bad = s_overflow || s_n > PLAN_MAX_STAGE || s_n > a.stage_stride || s_n > HC * 5 / 8;
    const uint32_t n_stage = bad ? 0u : s_n;
    if (tid == 0)
        {
        a.tile_nstage[tile] = n_stage;
        a.tile_head[tile] = (uint64_t)tile * a.stage_stride;
        if (bad)
            atomicOr(&a.flags[1], 1u);
        atomicMax(&a.flags[2], s_n);
        }
    if (bad)
        return; // the plan is invalid; nothing downstream will read this tile
    uint32_t npow = 64;
    while (npow < n_stage) npow <<= 1;
    // bitonic sort of list[0..npow) (padding = 0xFFFFFFFF sorts to the end). A wave's
    // compare-exchange pairs t in [64 w, 64 w + 63] (+ multiples of NT) touch only
    // the 128-element block 128 w .. 128 w + 127 whenever stride <= 64, so those
    // stages need no workgroup barrier -- only the stride >= 128 stages do.
    auto cmpswap = [&](uint32_t t, uint32_t size, uint32_t stride)
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
        };
    for (uint32_t size = 2; size <= npow; size <<= 1)
        {
        uint32_t stride = size >> 1;
        for (; stride >= 128; stride >>= 1)
            {
            for (uint32_t t = tid; t < (npow >> 1); t += NT)
                cmpswap(t, size, stride);
            __syncthreads();
            }
        for (; stride > 0; stride >>= 1)
            {
            for (uint32_t t = tid; t < (npow >> 1); t += NT)
                cmpswap(t, size, stride);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // same-wave LDS ordering
            }
        __syncthreads();
        }
    // publish; remember each key's slot; stage positions relative to the tile's
    // reference particle for the near/far classification. The positions overwrite
    // the sort scratch, so every thread first pulls its keys into registers.
    uint32_t* stage = a.stage_idx + (uint64_t)tile * a.stage_stride;
    const double3 c = load_scalar3_of4(a.pos, first);
    constexpr int SMAX = (PLAN_MAX_STAGE + NT) / NT;
    float4 pp4[SMAX];
#pragma unroll
    for (int it = 0; it < SMAX; ++it)
        {
        const uint32_t t = (uint32_t)it * NT + tid;
        pp4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < n_stage)
            {
            const uint32_t j = list[t];
            stage[t] = j;
            slot_of[plan_find<HC>(table, j)] = (uint16_t)t;
            const double4 pj = load_scalar4(a.pos, j);
            double dx = pj.x - c.x, dy = pj.y - c.y, dz = pj.z - c.z;
            min_image(a.box, dx, dy, dz);
            pp4[it] = make_float4((float)dx, (float)dy, (float)dz, __int_as_float(type_from_w(pj.w)));
            }
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SMAX; ++it)
        {
        const uint32_t t = (uint32_t)it * NT + tid;
        if (t < n_stage)
            s_p[t] = pp4[it];
        }
    __syncthreads();

#if defined(PLAN_ABLATE) && PLAN_ABLATE == 2
    return;
#endif
    // ---- compile rows, one row per wave at a time (no block barriers from here on).
    // Stable near/far partition with wave ballots: every lane classifies up to
    // ITERS entries (k = lane, lane + 64, ...), the ballots give each entry its rank
    // inside its part, and the u16 offsets land in the wave's row buffer.
    uint16_t* rowbuf = w_rowbuf + wave * PLAN_ROWBUF;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    float rsq_listed_max = 0.f; // largest listed separation^2 seen by this lane (flags[4]: a hint for callers
                                // that do not know r_cut + r_buff, e.g. the cache behind the HOOMD-signature entry)
    for (uint32_t p = wave; p < (uint32_t)TB; p += PLAN_BUILD_WAVES)
        {
        const uint32_t fwave = p / PW, pl = p % PW; // force-kernel wave (slice) and particle inside it
        const uint32_t slice = tile * 4 + fwave;
        const uint32_t K = a.slice_K[slice];
        uint4* out = a.cnl + (a.slice_head[slice] * 64ull);
        const uint32_t row_cap = K * TPP * 8u; // entries the slice's rectangle gives each particle
        uint32_t n = 0;
        const uint32_t* row = a.nlist;
        float xi = 0.f, yi = 0.f, zi = 0.f;
        uint32_t trow = 0;
        if (p < count)
            {
            const uint32_t i = first + p;
            n = a.n_neigh[i];
            row = a.nlist + a.head_list[i];
            const double4 pp = load_scalar4(a.pos, i);
            double dxi = pp.x - c.x, dyi = pp.y - c.y, dzi = pp.z - c.z;
            min_image(a.box, dxi, dyi, dzi);
            xi = (float)dxi; yi = (float)dyi; zi = (float)dzi;
            trow = (uint32_t)type_from_w(pp.w) * a.ntypes;
            }
        // all index loads of the row in flight together
        uint32_t jj[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
            {
            const uint32_t k = (uint32_t)it * 64u + lane;
            jj[it] = (k < n) ? row[k] : PLAN_EMPTY;
            }
        const uint32_t iters = (n + 63u) >> 6;
        // pass A: translate + classify. Classes: 0 core (inside the evaluator's inner
        // radius hint, e.g. the WCA core of PerturbedLJ), 1 near (inside the cutoff
        // now), 2 + s: buffer shell s (r_cut + s w <= r, s < PLAN_SHELLS). Rows are
        // written core | near | shell 0 | shell 1 | ... The core / near split is an
        // ordering hint; the near / shell splits let the force kernel stop early when
        // the caller bounds the displacement since this build, so they must be
        // CONSERVATIVE: the single-precision separation (relative error < 1e-5) has to
        // clear a boundary by a factor 1 + 5e-5 before an entry counts as outside it.
        uint32_t cnt_cls[PLAN_CLASSES]; // entries per class (wave-uniform)
#pragma unroll
        for (uint32_t cidx = 0; cidx < PLAN_CLASSES; ++cidx)
            cnt_cls[cidx] = 0;
        const float shell_winv = (s_shell_w > 0.f) ? 1.0f / s_shell_w : 0.f;
        uint32_t enc[ITERS]; // (offset << 4) | class, 0 = no entry
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
            {
            enc[it] = 0;
            if ((uint32_t)it < iters)
                {
                uint32_t cls = PLAN_CLASSES; // no entry
                if (jj[it] != PLAN_EMPTY)
                    {
                    const uint32_t sidx = slot_of[plan_find<HC>(table, jj[it])];
                    const float4 q = s_p[sidx];
                    float dx = xi - q.x, dy = yi - q.y, dz = zi - q.z;
                    if (!a.box.triclinic)
                        {
                        if (a.box.pz) dz = __builtin_fmaf(-bLz, rintf(dz * bLzi), dz);
                        if (a.box.py) dy = __builtin_fmaf(-bLy, rintf(dy * bLyi), dy);
                        if (a.box.px) dx = __builtin_fmaf(-bLx, rintf(dx * bLxi), dx);
                        }
                    else
                        {
                        double ddx = dx, ddy = dy, ddz = dz;
                        min_image(a.box, ddx, ddy, ddz);
                        dx = (float)ddx; dy = (float)ddy; dz = (float)ddz;
                        }
                    const float rsq = dx * dx + dy * dy + dz * dz;
                    rsq_listed_max = fmaxf(rsq_listed_max, rsq);
                    const uint32_t tp = trow + (uint32_t)__float_as_int(q.w);
                    const float rcsq = rc_cached ? s_rcutsq[tp] : (float)a.rcutsq[tp];
                    const bool in = !(rsq >= rcsq * 1.0001f); // "inside, or too close to call"
                    if (in)
                        cls = (rsq < (rc_cached ? s_rinnersq[tp] : (a.rinnersq ? (float)a.rinnersq[tp] : 0.f)))
                                  ? PLAN_CLS_CORE : (rsq < s_sure_rsq ? PLAN_CLS_SURE : PLAN_CLS_NEAR);
                    else
                        {
                        // shell s <=> the separation is certainly >= r_cut + s w: the single-precision
                        // r is shortened by 5e-5 (its own error is < 1e-5) before it is binned
                        const float rc = rc_cached ? s_rcut[tp] : sqrtf(fmaxf(rcsq, 0.f));
                        const float sh = floorf((sqrtf(rsq) * 0.99995f - rc) * shell_winv);
                        cls = PLAN_CLS_SHELL0 + (uint32_t)fminf(fmaxf(sh, 0.f), (float)(PLAN_SHELLS - 1)); // NaN (w = 0) -> shell 0
                        }
                    enc[it] = (((sidx + 1u) * 8u) << 4) | cls;
                    }
#pragma unroll
                for (uint32_t cidx = 0; cidx < PLAN_CLASSES; ++cidx)
                    cnt_cls[cidx] += (uint32_t)__popcll(__ballot(cls == cidx));
                }
            }
        uint32_t seg[PLAN_CLASSES + 1]; // class cidx occupies row positions [seg[cidx], seg[cidx + 1])
        seg[0] = 0;
#pragma unroll
        for (uint32_t cidx = 0; cidx < PLAN_CLASSES; ++cidx)
            seg[cidx + 1] = seg[cidx] + cnt_cls[cidx];
        if (lane == 0 && p < count)
            {
            // chunks a force-kernel wave must process to cover every in-range entry [0] / every
            // entry up to the end of shell s [1 + s], over the rows of its slice
#pragma unroll
            for (uint32_t sh = 0; sh <= PLAN_SHELLS; ++sh)
                atomicMax(&a.slice_Kend[(PLAN_SHELLS + 1) * slice + sh], (seg[PLAN_CLS_SHELL0 + sh] + 8u * TPP - 1u) / (8u * TPP));
            // row phases of the force kernel: chunks that cover the core entries, chunks made of core / sure entries alone
            atomicMax(&a.slice_Kphase[slice], (seg[PLAN_CLS_SURE] + 8u * TPP - 1u) / (8u * TPP));
            atomicMin(&a.slice_Kphase[a.n_slices + slice], seg[PLAN_CLS_NEAR] / (8u * TPP));
            }
        if (TPP == 1 && PLAN_BANK_ORDER && a.bank_order)
            {
            // pass B (bank-aware): lane l of the force kernel reads its row entry q at
            // step q, together with the other 63 rows of the slice. A 64-lane
            // ds_read_b64 is served 16 lanes at a time from 32 banks, so it is
            // conflict-free when the 16 lanes of a group touch 16 different 8-byte
            // bank pairs (slot mod 16), or the same slot. Give position q of row l
            // the "home" bank (l + q) mod 16: within a group all homes differ. An
            // entry goes to the next free home position of its bank inside its class;
            // the ~16 % that do not fit (banks are not evenly used by one row) fill
            // the positions left over. Measured on the gather pattern alone
            // (tools/lds_bench.hip): 10.7 -> 7.4 LDS cycles per wave read.
            uint32_t* cnt = s_bank_cnt[wave];
            uint16_t* holes = s_holes[wave];
            for (uint32_t t = lane; t < PLAN_CLASSES * 17; t += 64)
                cnt[t] = 0;
            for (uint32_t t = lane; t < row_cap; t += 64)
                rowbuf[t] = (t < n) ? (uint16_t)0xffffu : (uint16_t)0; // unclaimed | padding
            __builtin_amdgcn_wave_barrier();
            uint32_t misfit = 0; // bit it: entry it found no home position
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                {
                if ((uint32_t)it < iters && enc[it] != 0)
                    {
                    const uint32_t cls = enc[it] & 15u;
                    const uint32_t off = enc[it] >> 4, bank = (off >> 3) & 15u;
                    const uint32_t r = atomicAdd(&cnt[cls * 16u + bank], 1u);
                    uint32_t bs = 0, es = 0;
#pragma unroll
                    for (uint32_t cidx = 0; cidx < PLAN_CLASSES; ++cidx)
                        if (cls == cidx)
                            {
                            bs = seg[cidx];
                            es = seg[cidx + 1];
                            }
                    const uint32_t q = bs + ((bank - pl - bs) & 15u) + 16u * r;
                    if (q < es)
                        rowbuf[q] = (uint16_t)off;
                    else
                        misfit |= 1u << it;
                    }
                }
            __builtin_amdgcn_wave_barrier();
            // unclaimed positions, in order (class by class); holes_lt[cidx] = holes before class cidx
            uint32_t n_holes = 0;
            uint32_t holes_lt[PLAN_CLASSES];
#pragma unroll
            for (uint32_t cidx = 0; cidx < PLAN_CLASSES; ++cidx)
                holes_lt[cidx] = 0;
            for (uint32_t t0 = 0; t0 < n; t0 += 64)
                {
                const uint32_t t = t0 + lane;
                const bool hole = (t < n) && rowbuf[t] == 0xffffu;
                const uint64_t m = __ballot(hole);
                if (hole)
                    holes[n_holes + (uint32_t)__popcll(m & lt_mask)] = (uint16_t)t;
                n_holes += (uint32_t)__popcll(m);
#pragma unroll
                for (uint32_t cidx = 1; cidx < PLAN_CLASSES; ++cidx)
                    holes_lt[cidx] += (uint32_t)__popcll(__ballot(hole && t < seg[cidx]));
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                {
                if (misfit & (1u << it))
                    {
                    const uint32_t cls = enc[it] & 15u;
                    uint32_t before = 0;
#pragma unroll
                    for (uint32_t cidx = 0; cidx < PLAN_CLASSES; ++cidx)
                        if (cls == cidx)
                            before = holes_lt[cidx];
                    const uint32_t m = atomicAdd(&cnt[PLAN_CLASSES * 16u + cls], 1u) + before;
                    rowbuf[holes[m]] = (uint16_t)(enc[it] >> 4);
                    }
                }
            __builtin_amdgcn_wave_barrier();
            }
        else
            {
            // pass B: stable partition into the row buffer, class by class
            uint32_t base[PLAN_CLASSES];
#pragma unroll
            for (uint32_t cidx = 0; cidx < PLAN_CLASSES; ++cidx)
                base[cidx] = seg[cidx];
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                {
                if ((uint32_t)it < iters)
                    {
                    const bool valid = enc[it] != 0;
                    const uint32_t cls = enc[it] & 15u;
                    uint32_t posn = 0;
#pragma unroll
                    for (uint32_t cidx = 0; cidx < PLAN_CLASSES; ++cidx)
                        {
                        const uint64_t m = __ballot(valid && cls == cidx);
                        if (cls == cidx)
                            posn = base[cidx] + (uint32_t)__popcll(m & lt_mask);
                        base[cidx] += (uint32_t)__popcll(m);
                        }
                    if (valid)
                        rowbuf[posn] = (uint16_t)(enc[it] >> 4);
                    }
                }
            // pad to the slice's rectangle with the dummy slot
            for (uint32_t t = n + lane; t < row_cap; t += 64)
                rowbuf[t] = 0;
            }
#if defined(PLAN_ABLATE) && PLAN_ABLATE == 3
        continue;
#endif
        __builtin_amdgcn_wave_barrier();
        // 16-byte chunks in the force kernel's (iteration, lane) order
        const uint4* rb4 = reinterpret_cast<const uint4*>(rowbuf);
        for (uint32_t cidx = lane; cidx < K * TPP; cidx += 64)
            out[(uint64_t)(cidx / TPP) * 64 + pl * TPP + (cidx % TPP)] = rb4[cidx];
        __builtin_amdgcn_wave_barrier();
        }
    for (int off = 32; off > 0; off >>= 1)
        rsq_listed_max = fmaxf(rsq_listed_max, __shfl_xor(rsq_listed_max, off, 64));
    if (lane == 0)
        atomicMax(&a.flags[4], (uint32_t)__float_as_int(rsq_listed_max)); // non-negative floats order like their bits
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

void plan_free(PairPlan& p)
    {
    if (p.d_tile_nstage) (void)hipFree(p.d_tile_nstage);
    if (p.d_tile_head) (void)hipFree(p.d_tile_head);
    if (p.d_stage_idx) (void)hipFree(p.d_stage_idx);
    if (p.d_slice_K) (void)hipFree(p.d_slice_K);
    if (p.d_slice_Kend) (void)hipFree(p.d_slice_Kend);
    if (p.d_slice_Kphase) (void)hipFree(p.d_slice_Kphase);
    if (p.d_tile_ids) (void)hipFree(p.d_tile_ids);
    if (p.d_slice_head) (void)hipFree(p.d_slice_head);
    if (p.d_cnl) (void)hipFree(p.d_cnl);
    if (p.d_flags) (void)hipFree(p.d_flags);
    if (p.d_raw) (void)hipFree(p.d_raw);
    if (p.d_perm) (void)hipFree(p.d_perm);
    p = PairPlan();
    }

static size_t plan_lds_bytes(uint32_t hc, uint32_t stride)
    {
    const size_t shared_region = std::max<size_t>(4096 * 4, (size_t)stride * 16);
    return (size_t)hc * 4 + (size_t)hc * 2 + (size_t)PLAN_BUILD_WAVES * PLAN_ROWBUF * 2 + shared_region;
    }

template<int TPP, uint32_t HC> static hipError_t launch_plan_build(const PlanKArgs& k, uint32_t n_tiles, hipStream_t s)
    {
    const size_t lds = plan_lds_bytes(HC, k.stage_stride);
    auto kern = plan_build_kernel<TPP, HC>;
    if (lds + 16 * 1024 > 64 * 1024) // + the kernel's static LDS (bank counters, hole lists: ~14.5 KiB)
        {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        }
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(PLAN_BUILD_THREADS), lds, s, k);
    return hipGetLastError();
    }

template<int TPP> static hipError_t launch_plan_kernels(int which, const PlanKArgs& k, uint32_t n_tiles, hipStream_t s)
    {
    if (which == 0)
        hipLaunchKernelGGL((plan_chunks_kernel<TPP>), dim3(n_tiles), dim3(256), 0, s, k);
    else
        {
        // a 4096-entry hash set holds every admissible stage set (<= 2,559 keys, load factor
        // <= 0.625: ~1.8 probes per lookup) in half the LDS of an 8192-entry one, so two build
        // workgroups share a CU up to ~2,300 staged particles per tile
        return launch_plan_build<TPP, 4096>(k, n_tiles, s);
        }
    return hipGetLastError();
    }

static hipError_t launch_plan(uint32_t tpp, int which, const PlanKArgs& k, uint32_t n_tiles, hipStream_t s)
    {
    if (tpp == 4) return launch_plan_kernels<4>(which, k, n_tiles, s);
    if (tpp == 2) return launch_plan_kernels<2>(which, k, n_tiles, s);
    return launch_plan_kernels<1>(which, k, n_tiles, s);
    }

#define AZP_HIP_TRY(expr)                      \
    do                                         \
        {                                      \
        hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) return (int)e_;  \
        } while (0)

// Build with a given tile size. Sequence: chunk counts (tiny kernel) -> host scan
// of the slice heads (one small D2H/H2D round trip) -> one kernel per tile that
// dedups, sorts and compiles. Stage lists live at a fixed stride per tile
// (stage_stride entries), sized from the previous build and grown on overflow.
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
    AZP_HIP_TRY(ensure(p.d_slice_Kend, p.cap_kend, (PLAN_SHELLS + 1) * (size_t)p.n_slices));
    AZP_HIP_TRY(ensure(p.d_slice_head, cap_heads_s, p.n_slices));
    AZP_HIP_TRY(ensure(p.d_slice_Kphase, p.cap_kphase, 2 * (size_t)p.n_slices));
    size_t cap_flags = p.d_flags ? 16 : 0; // (16 words: pair_plan_cells.hip shares the buffer)
    AZP_HIP_TRY(ensure(p.d_flags, cap_flags, 16));
    AZP_HIP_TRY(hipMemsetAsync(p.d_flags, 0, 8 * sizeof(uint32_t), s));

    PlanKArgs k;
    k.pos = args.d_pos;
    k.n_neigh = args.d_n_neigh;
    k.nlist = args.d_nlist;
    k.head_list = args.d_head_list;
    k.rcutsq = args.d_rcutsq;
    k.rinnersq = args.d_rinnersq;
    k.slice_Kend = p.d_slice_Kend;
    k.slice_Kphase = p.d_slice_Kphase;
    k.n_slices = p.n_slices;
    // buffer shells: r_buff = (r_list_max - r_cut_max) / 2 is not known here, so the caller's
    // r_list_max hint and the largest cutoff are used when given
    k.r_list_max = args.r_list_max;
    k.r_list_estimate = p.shell_hint_r_list;
    k.bank_order = p.bank_order ? 1u : 0u;
    k.tile_nstage = p.d_tile_nstage;
    k.tile_head = p.d_tile_head;
    k.stage_idx = nullptr;
    k.slice_K = p.d_slice_K;
    k.slice_head = p.d_slice_head;
    k.cnl = nullptr;
    k.flags = p.d_flags;
    k.box = make_box_dev(args.box);
    k.N = args.N;
    k.ntypes = args.ntypes;
    k.stage_stride = 0;
    AZP_HIP_TRY(launch_plan(tpp, 0, k, p.n_tiles, s));

    std::vector<uint32_t> h_K(p.n_slices);
    uint32_t h_flags[8];
    AZP_HIP_TRY(hipMemcpyAsync(h_K.data(), p.d_slice_K, sizeof(uint32_t) * p.n_slices, hipMemcpyDeviceToHost, s));
    AZP_HIP_TRY(hipMemcpyAsync(h_flags, p.d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, s));
    AZP_HIP_TRY(hipStreamSynchronize(s));
    if (h_flags[1])
        {
        p.invalid_reason = 2;
        return AZP_SUCCESS;
        }
    std::vector<uint64_t> h_shead(p.n_slices);
    uint64_t acc = 0;
    for (uint32_t t = 0; t < p.n_slices; ++t) { h_shead[t] = acc; acc += h_K[t]; }
    p.total_chunks = acc;
    AZP_HIP_TRY(ensure(p.d_cnl, p.cap_cnl, (size_t)std::max<uint64_t>(p.total_chunks * 64, 1)));
    AZP_HIP_TRY(hipMemcpyAsync(p.d_slice_head, h_shead.data(), sizeof(uint64_t) * p.n_slices, hipMemcpyHostToDevice, s));

    // stage stride: last build's maximum + 25 % (first build: full budget if it fits
    // in 256 MiB, else a first guess), rounded to 64; retried larger on overflow
    uint32_t stride = p.stage_stride_hint;
    if (stride == 0)
        stride = ((uint64_t)p.n_tiles * (PLAN_MAX_STAGE + 1) * 4 <= (256ull << 20)) ? PLAN_MAX_STAGE + 1 : 1536;
    for (;;)
        {
        stride = std::min<uint32_t>((stride + 63u) & ~63u, PLAN_MAX_STAGE + 1);
        AZP_HIP_TRY(ensure(p.d_stage_idx, p.cap_stage, (size_t)p.n_tiles * stride));
        AZP_HIP_TRY(hipMemsetAsync(p.d_flags, 0, 8 * sizeof(uint32_t), s));
        AZP_HIP_TRY(hipMemsetAsync(p.d_slice_Kend, 0, (PLAN_SHELLS + 1) * sizeof(uint32_t) * p.n_slices, s));
        AZP_HIP_TRY(hipMemsetAsync(p.d_slice_Kphase, 0, sizeof(uint32_t) * p.n_slices, s));
        AZP_HIP_TRY(hipMemsetAsync(p.d_slice_Kphase + p.n_slices, 0xff, sizeof(uint32_t) * p.n_slices, s));
        k.stage_idx = p.d_stage_idx;
        k.cnl = p.d_cnl;
        k.stage_stride = stride;
        AZP_HIP_TRY(launch_plan(tpp, 1, k, p.n_tiles, s));
        AZP_HIP_TRY(hipMemcpyAsync(h_flags, p.d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, s));
        p.h_tile_nstage.resize(p.n_tiles); // per-tile stage counts ride on the same synchronisation
        AZP_HIP_TRY(hipMemcpyAsync(p.h_tile_nstage.data(), p.d_tile_nstage, sizeof(uint32_t) * p.n_tiles, hipMemcpyDeviceToHost, s));
        AZP_HIP_TRY(hipStreamSynchronize(s));
        p.max_stage = h_flags[2];
        if (!h_flags[1])
            break;
        if (p.max_stage > PLAN_MAX_STAGE || stride >= PLAN_MAX_STAGE + 1)
            {
            p.invalid_reason = 2;
            return AZP_SUCCESS;
            }
        stride = p.max_stage + p.max_stage / 8 + 64; // the stride was the problem: grow and redo
        }
    // next build: a little head room only -- the stride also sizes the builder's LDS region for
    // the staged positions (16 B per slot), and 25 % of it cost the builder half its occupancy
    // once a tile staged ~2,000 particles; an overflow is retried with a larger stride anyway
    p.stage_stride_hint = p.max_stage + p.max_stage / 16 + 32;
    p.total_stage = (uint64_t)p.n_tiles * stride;
    {
    float fm;
    __builtin_memcpy(&fm, &h_flags[3], sizeof(fm));
    p.shell_width = fm;
    __builtin_memcpy(&fm, &h_flags[4], sizeof(fm));
    p.max_listed_r = std::sqrt(fm);
    __builtin_memcpy(&p.sure_r, &h_flags[0], sizeof(float));
    __builtin_memcpy(&p.core_r, &h_flags[7], sizeof(float));
    }
    p.cap = plan_cap_for(p.max_stage);
    p.valid = true;
    return AZP_SUCCESS;
    }

int plan_build(PairPlan& p, const azp_pair_args& args, hipStream_t s)
    {
    p.valid = false;
    p.invalid_reason = 0;
    p.from_cells = false;
    p.balanced = false;
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
    if (!plan || !args || !args->d_pos || !args->d_n_neigh || !args->d_nlist || !args->d_head_list || !args->d_rcutsq
        || args->ntypes == 0)
        return AZP_ERROR_INVALID_ARGUMENT;
    return azp::plan_build(*reinterpret_cast<azp::PairPlan*>(plan), *args, static_cast<hipStream_t>(stream));
    }

extern "C" int azp_pair_plan_set_bank_order(azp_pair_plan* plan, int enabled)
    {
    if (!plan)
        return AZP_ERROR_INVALID_ARGUMENT;
    reinterpret_cast<azp::PairPlan*>(plan)->bank_order = enabled != 0;
    return AZP_SUCCESS;
    }

extern "C" int azp_pair_plan_set_balance(azp_pair_plan* plan, int enabled)
    {
    if (!plan)
        return AZP_ERROR_INVALID_ARGUMENT;
    reinterpret_cast<azp::PairPlan*>(plan)->balance = enabled != 0;
    return AZP_SUCCESS;
    }

extern "C" int azp_pair_plan_tile_stage(const azp_pair_plan* plan, uint32_t* out, uint32_t n)
    {
    if (!plan || !out)
        return AZP_ERROR_INVALID_ARGUMENT;
    const azp::PairPlan* p = reinterpret_cast<const azp::PairPlan*>(plan);
    const uint32_t m = std::min<uint32_t>(n, (uint32_t)p->h_tile_nstage.size());
    for (uint32_t t = 0; t < m; ++t)
        out[t] = p->h_tile_nstage[t];
    return (int)m;
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
    info->max_row = p->max_row;
    info->from_cells = p->from_cells ? 1 : 0;
    info->balanced = p->balanced ? 1 : 0;
    info->row_capacity = p->row_cap;
    info->list_id = reinterpret_cast<uint64_t>(p->nlist_ptr);
    info->head_id = reinterpret_cast<uint64_t>(p->head_ptr);
    info->core_radius = p->core_r;
    info->sure_radius = p->sure_r;
    info->max_member_cells = p->max_member_cells;
    return AZP_SUCCESS;
    }

extern "C" int azp_pair_plan_phase_chunks(const azp_pair_plan* plan, float out[3])
    {
    if (!plan || !out)
        return AZP_ERROR_INVALID_ARGUMENT;
    const azp::PairPlan* p = reinterpret_cast<const azp::PairPlan*>(plan);
    out[0] = out[1] = out[2] = 0.f;
    if (!p->valid || !p->d_slice_Kphase || !p->n_slices)
        return AZP_SUCCESS;
    std::vector<uint32_t> h(3 * (size_t)p->n_slices);
    AZP_HIP_TRY(hipMemcpy(h.data(), p->d_slice_Kphase, 2 * sizeof(uint32_t) * p->n_slices, hipMemcpyDeviceToHost));
    AZP_HIP_TRY(hipMemcpy(h.data() + 2 * (size_t)p->n_slices, p->d_slice_K, sizeof(uint32_t) * p->n_slices, hipMemcpyDeviceToHost));
    double s0 = 0, s1 = 0, s2 = 0;
    for (uint32_t t = 0; t < p->n_slices; ++t)
        {
        const uint32_t K = h[2 * (size_t)p->n_slices + t], k0 = std::min(h[t], K), k1 = std::min(std::max(h[p->n_slices + t], k0), K);
        s0 += k0; s1 += k1; s2 += K;
        }
    out[0] = (float)(s0 / p->n_slices);
    out[1] = (float)(s1 / p->n_slices);
    out[2] = (float)(s2 / p->n_slices);
    return AZP_SUCCESS;
    }
