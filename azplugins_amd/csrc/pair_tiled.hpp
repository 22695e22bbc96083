// pair_tiled.hpp -- tile-staged pair-force kernel (uses a PairPlan).
//
// One workgroup (4 waves) per tile of 256/TPP consecutive particles:
//   1. stage: the tile's sorted unique neighbor set is loaded once (index list
//      coalesced, positions semi-coalesced because the list is sorted) into LDS
//      as SoA x | y | z (| type), shifted to the periodic image nearest the
//      tile's reference particle. Slot 0 is a dummy parked at 1e30 for padding.
//   2. loop: each lane reads one 16-byte chunk = 8 u16 byte offsets per
//      iteration (a wave reads 1 KiB contiguous), then for each of the 8
//      neighbors three LDS gathers and the evaluator, software-pipelined in
//      batches of 4 pairs. No global gathers, no minimum image, no row-length
//      test (rows are padded with the dummy). Rows are ordered core | near |
//      buffer shell A | B (pair_plan.hip): a batch with no pair in range is
//      skipped after the separations, a batch with no pair in the evaluator's
//      core uses the cheaper tail form (PerturbedLJ), and the row ends before
//      the buffer entries when the caller bounds the displacement since the
//      plan was built (skip_level).
//   3. DPP butterfly over the TPP lanes, lane 0 stores force (and virial).
// Same evaluators, same outputs as pair_kernel.hpp; results agree with it to
// rounding (the periodic shift is applied to r_j instead of to r_i - r_j).
#pragma once

#include <cstdlib>
#include <type_traits>

#include "pair_kernel.hpp"
#include "pair_plan.hpp"

#ifndef AZP_TILED_WAVES_PER_SIMD
#define AZP_TILED_WAVES_PER_SIMD 4 // register budget: <= 128 VGPRs => 4 workgroups (of 4 waves) per CU
#endif

namespace azp
{
#ifdef AZP_TIMELINE
// debug build only (tools/timeline.py): per (tile, wave) realtime stamps (100 MHz):
// kernel entry, tile staged, loop done, and the hardware id of the wave
extern __device__ unsigned long long g_timeline[8 * 4 * 16384];
#endif

// Words a speculatively launched tile kernel reads from device memory (pair_auto.hpp): written by the
// list-check kernel that runs right before it on the same stream.
struct TileDyn
    {
    uint32_t n_shells;   // buffer shells to walk for the displacement measured by the check
    uint32_t stale;      // != 0: the plan does not describe the list any more -- leave at once (the call is repeated)
    double bound;        // that displacement (< 0: unknown)
    };

struct TiledKArgs
    {
    PairKArgs p;
    const uint32_t* tile_nstage;
    const uint64_t* tile_head;
    const uint32_t* stage_idx;
    const uint32_t* slice_K;
    const uint32_t* slice_Kend;  // PLAN_SHELLS + 1 per slice: chunks covering the in-range entries [0] / the entries up to
                                 // the end of buffer shell s [1 + s]
    uint32_t n_shells;           // buffer shells this launch has to walk: 0 = none (positions as at plan build) ...
                                 // PLAN_SHELLS = whole rows
    const uint64_t* slice_head;
    const uint4* cnl;
    const uint8_t* perm;         // balanced plans (one lane per particle): lane -> member of the tile; NULL = identity
    // Row phases (evaluators with a split form, one type pair): per slice the chunk count that covers every
    // entry of class "core" [0] and the chunk count up to which every lane holds only entries of the classes
    // core / sure [1] (pair_plan.hpp). With a displacement bound from the caller the chunks beyond [0] cannot
    // hold a pair inside the evaluator's core, and the chunks [0] .. [1] hold only pairs that are certainly
    // inside the cutoff: their tests are dropped (decided per wave from the bound and the radii below; exact).
    const uint32_t* slice_Kcore;  // per slice; NULL: no phases
    const uint32_t* slice_Ksure;
    double bound;                 // the caller's displacement bound, < 0: unknown
    float core_r, sure_r;         // class radii at build time, margins included; 0: class not built
    // Local displacement bound (azp_pair_args.d_displacement): per-particle upper bounds on the distance moved since the
    // plan was built. A tile then walks the shells ITS members and staged neighbors can have crossed -- an entry of shell s
    // was at least r_cut + s w away, and the two particles of a pair have closed in by at most the sum of their own
    // displacements <= 2 x the largest one in the tile -- instead of the shells the fastest particle of the whole system
    // dictates. bound_extra is added to every entry (a plan compiled later than the positions the displacements refer to).
    const uint32_t* tile_ids;     // NULL: workgroup b computes tile first / TB + b; else tile_ids[b], b < n_tile_ids
    uint32_t n_tile_ids;
    const TileDyn* dyn;           // NULL: n_shells / bound above are final
    // the same idea with the raw words of the neighbor list's distance check (azp_pair_args.d_stale_flag,
    // d_displacement_sq_bits): leave when *dflag != 0, bound = sqrt(double(*dbits)) + bound_extra
    const uint32_t* dflag;
    const unsigned long long* dbits;
    const float* disp;            // n_max entries; NULL: the global bound above
    double shell_w;               // shell width of the plan
    double shell_winv;            // 1 / shell_w (0: no shells)
    double bound_extra;
    };

// wave-wide maximum of a non-negative float (all lanes get it)
__device__ __forceinline__ float wave_max_nonneg(float v)
    {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
    }

// Shells a tile has to walk for the displacement bound b of its own particles (device form of plan_shells_for)
__device__ __forceinline__ uint32_t tile_shells_for(double b, double shell_winv)
    {
    if (!(b >= 0.0))
        return PLAN_SHELLS; // NaN
    if (b == 0.0)
        return 0u;
    if (!(shell_winv > 0.0))
        return PLAN_SHELLS;
    const double n = ceil(2.0 * b * (1.0 + 1e-9) * shell_winv); // (1e-9 covers the rounded reciprocal)
    return n >= (double)PLAN_SHELLS ? PLAN_SHELLS : (uint32_t)n;
    }

// Stride (in slots) between the x, y and z arrays in LDS. With a stride of CAP the compiler
// fuses the x and y gathers of a pair into one ds_read2st64_b64, which runs at half the LDS
// rate of ds_read_b64 (MI355X_MICROARCH.md, LDS table); an odd stride (CAP + 1) keeps them two
// ds_read_b64: whole rows 0.143 -> 0.135 ms, MD cycle mean 0.127 -> 0.124 ms (the kernel is
// as much LDS- as VALU-bound).
#ifndef AZP_TILE_SOA_PAD
#define AZP_TILE_SOA_PAD 1
#endif
#define AZP_TILE_STRIDE(CAP) ((CAP) + AZP_TILE_SOA_PAD)

#ifndef AZP_TILE_BATCH
#define AZP_TILE_BATCH 4 // pairs per register batch: 4 = half a chunk, 8 = a whole chunk
#endif

// A batch of gathered neighbor data (NB = 4: half a chunk, NB = 8: a whole chunk).
struct TileBatch
    {
    double x[AZP_TILE_BATCH], y[AZP_TILE_BATCH], z[AZP_TILE_BATCH];
    uint32_t off[AZP_TILE_BATCH];
    };

// the local-displacement fields of the kernel arguments (pair_tiled.hpp and xtiled.hpp launchers)
inline void fill_local_bound(TiledKArgs& k, const PairPlan& plan, const azp_pair_args& args)
    {
    // per-particle displacements count only together with a (global) bound: has_displacement_bound says the caller
    // tracks displacements since the plan build at all; displacement_bound_extra covers a plan built later than the
    // reference positions of d_displacement
    const bool local = args.d_displacement && args.has_displacement_bound && args.displacement_bound >= 0.0;
    k.disp = (local && tuning().local_bound != 0) ? args.d_displacement : nullptr;
    k.dflag = (args.d_stale_flag && args.d_displacement_sq_bits) ? args.d_stale_flag : nullptr;
    k.dbits = k.dflag ? args.d_displacement_sq_bits : nullptr;
    if (k.dflag)
        k.disp = nullptr; // (the per-particle displacements of a check whose result is not known yet are not either)
    k.shell_w = plan.shell_width;
    k.shell_winv = plan.shell_width > 0.0 ? 1.0 / plan.shell_width : 0.0;
    k.bound_extra = args.displacement_bound_extra > 0.0 ? args.displacement_bound_extra : 0.0;
    }

// phase 1: issue the LDS gathers of a batch (half H of the chunk when NB = 4)
template<int CAP, int H> __device__ __forceinline__ void tile_gather(TileBatch& b, const uint4& u, const char* bx)
    {
    constexpr int NB = AZP_TILE_BATCH;
    if (NB == 4)
        {
        const uint32_t w0 = H ? u.z : u.x, w1 = H ? u.w : u.y;
        b.off[0] = w0 & 0xffffu;
        b.off[1] = w0 >> 16;
        b.off[2] = w1 & 0xffffu;
        b.off[3] = w1 >> 16;
        }
    else
        {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < NB; ++e)
            b.off[e] = (e & 1) ? (w[(e >> 1) & 3] >> 16) : (w[(e >> 1) & 3] & 0xffffu);
        }
#pragma unroll
    for (int e = 0; e < NB; ++e)
        {
#if defined(AZP_ABLATE) && (AZP_ABLATE == 2)
        b.x[e] = (double)b.off[e]; b.y[e] = 1.0; b.z[e] = 2.0; // ablation 2: no LDS gathers
#else
        b.x[e] = *reinterpret_cast<const double*>(bx + b.off[e]);
        b.y[e] = *reinterpret_cast<const double*>(bx + b.off[e] + AZP_TILE_STRIDE(CAP) * 8);
        b.z[e] = *reinterpret_cast<const double*>(bx + b.off[e] + AZP_TILE_STRIDE(CAP) * 16);
#endif
        }
    }

// phase 2: arithmetic on a gathered half chunk. The separations of the 4 pairs
// are formed first; if no lane of the wave has any of them inside the (largest)
// cutoff, the evaluator work is skipped -- exact, and the common case at the far
// end of the near-first ordered rows.
// MODE (split evaluators only; everything else runs MODE 0): 0 = every test (pairs may be in the evaluator's
// core, in its tail, or out of range), 1 = "sure": every pair of the batch is inside the cutoff and outside the
// core -- no test at all, 2 = "tail": no pair is inside the core; cutoff test and the skip of batches without a
// pair in range as in mode 0.
template<class E, int CAP, bool VIRIAL, bool SINGLE, bool XPLOR, bool WRAP, int MODE = 0>
__device__ __forceinline__ void tile_compute(const TileBatch& b, const TiledKArgs& a, const char* bt,
                                             const typename E::Coeff* __restrict__ s_coeff,
                                             const double* __restrict__ s_ronsq, const typename E::Coeff& c0, double ronsq0,
                                             double rcutsq_max, const double3& pi, int typei, double& fx, double& fy,
                                             double& fz, double& pe, double (&v)[6], uint32_t& n_core, uint32_t& n_in, double (&es)[2])
    {
    typedef typename E::Coeff Coeff;
    constexpr int NB = AZP_TILE_BATCH;
    constexpr bool SPLIT = SINGLE && !XPLOR && E::kSplitEnergy; // energy offsets counted, see EvalPLJ::eval_split
    double dx[NB], dy[NB], dz[NB], rsq[NB];
    bool any_in = false;
#pragma unroll
    for (int e = 0; e < NB; ++e)
        {
        dx[e] = pi.x - b.x[e]; dy[e] = pi.y - b.y[e]; dz[e] = pi.z - b.z[e];
        if (WRAP)
            min_image(a.p.box, dx[e], dy[e], dz[e]);
        rsq[e] = __builtin_fma(dz[e], dz[e], __builtin_fma(dy[e], dy[e], dx[e] * dx[e]));
        if (WRAP)
            rsq[e] = (b.off[e] == 0) ? 1.0e60 : rsq[e]; // the minimum image would fold the padding slot back into the box
                                                        // (1e60: out of range, and a product of four stays finite for rcp4)
        if (MODE != 1)
            any_in = any_in || (rsq[e] < rcutsq_max);
        }
#if defined(AZP_ABLATE) && (AZP_ABLATE == 3)
    fx += rsq[0] + rsq[1] + rsq[2] + rsq[3]; // ablation 3: gathers + separations only
    return;
#endif
    if (MODE != 1 && !__any(any_in))
        return;
#ifdef AZP_TILE_LANE_MASK
    // experiment: lanes without a pair in range in this batch sit the evaluator out (EXEC
    // mask): the same issue cycles, fewer active FP64 lanes (the kernel runs into the power
    // limit)
    if (!any_in)
        return;
#endif
    double fd[NB]; // force / r of the batch (SPLIT: filled by one of two forms of the evaluator)
    if constexpr (SPLIT)
        {
        // one v_rcp_f64 for the four pairs of the batch (EvalPLJ::rcp4)
        double x[NB];
#ifndef AZP_NO_RCP4
        if constexpr (NB == 4)
            E::rcp4(rsq, x);
        else
#endif
            {
#pragma unroll
            for (int e = 0; e < NB; ++e)
                x[e] = fast_rcp1(rsq[e]);
            }
        // rows list the pairs inside the evaluator's core first (plan hint), so beyond
        // the first chunks no lane of the wave has one and the cheaper tail-only form
        // applies to the whole batch (exact: tested on the actual separations). Both
        // branches only produce fd[]; the accumulation below is shared, so the force
        // accumulators are not live-out of either branch (no register copies at the join).
        bool any_core = false;
        if (MODE == 0)
            {
#pragma unroll
            for (int e = 0; e < NB; ++e)
                any_core = any_core || E::in_core(c0, rsq[e]);
            }
        if (MODE == 1)
            {
#pragma unroll
            for (int e = 0; e < NB; ++e)
                E::eval_split_sure(c0, x[e], fd[e], es[0], es[1]);
            n_in += NB; // every pair of the batch is in range (finish_split ignores the count when the shift is zero)
            }
        else if (MODE == 2 || !__any(any_core))
            {
            const bool count_in = c0.tail_add != 0.0;
#pragma unroll
            for (int e = 0; e < NB; ++e)
                E::eval_split_tail(c0, rsq[e], x[e], fd[e], es[0], es[1], n_in, count_in);
            }
        else
            {
#pragma unroll
            for (int e = 0; e < NB; ++e)
                E::eval_split(c0, rsq[e], x[e], fd[e], pe, n_core, n_in);
            }
        }
#pragma unroll
    for (int e = 0; e < NB; ++e)
        {
        double force_divr, pair_eng = 0.0;
        if constexpr (SPLIT)
            force_divr = fd[e];
        else if (SINGLE)
            {
            const bool evaluated = E::eval(c0, rsq[e], force_divr, pair_eng);
            if (XPLOR && evaluated)
                apply_xplor(rsq[e], ronsq0, c0.rcutsq, force_divr, pair_eng);
            }
        else
            {
            const int typej = *reinterpret_cast<const int*>(bt + (b.off[e] >> 1));
            const uint32_t tp = (uint32_t)typei * a.p.ntypes + (uint32_t)typej;
            const Coeff cc = s_coeff[tp];
            const bool evaluated = E::eval(cc, rsq[e], force_divr, pair_eng);
            if (XPLOR && evaluated)
                apply_xplor(rsq[e], s_ronsq[tp], cc.rcutsq, force_divr, pair_eng);
            }
        fx = __builtin_fma(dx[e], force_divr, fx);
        fy = __builtin_fma(dy[e], force_divr, fy);
        fz = __builtin_fma(dz[e], force_divr, fz);
        if constexpr (!SPLIT)
            pe += pair_eng;
        if (VIRIAL)
            {
            const double fxx = force_divr * dx[e], fyy = force_divr * dy[e];
            v[0] = __builtin_fma(fxx, dx[e], v[0]);
            v[1] = __builtin_fma(fxx, dy[e], v[1]);
            v[2] = __builtin_fma(fxx, dz[e], v[2]);
            v[3] = __builtin_fma(fyy, dy[e], v[3]);
            v[4] = __builtin_fma(fyy, dz[e], v[4]);
            v[5] = __builtin_fma(force_divr * dz[e], dz[e], v[5]);
            }
        }
    }

// Inner loop over this lane's chunks, software-pipelined at half-chunk
// granularity: while the arithmetic of one half (4 pairs) runs, the 12 LDS gathers
// of the next half and the index load of the next chunk are in flight. WRAP = true
// re-applies the minimum image to every pair (tiles that are wide compared with the
// box, triclinic boxes, or no r_list_max hint); WRAP = false trusts the staged
// image (the common case).
template<class E, int TPP, int CAP, bool VIRIAL, bool SINGLE, bool XPLOR, bool WRAP>
__device__ __forceinline__ void tiled_loop(const TiledKArgs& a, const char* bx, const char* bt,
                                           const typename E::Coeff* __restrict__ s_coeff,
                                           const double* __restrict__ s_ronsq, const typename E::Coeff& c0, double ronsq0,
                                           double rcutsq_max, const char* __restrict__ slice_base, uint32_t lane_off, uint32_t K,
                                           uint32_t K1, uint32_t K2, double3 pi,
                                           int typei, double& fx, double& fy, double& fz, double& pe, double (&v)[6], uint32_t& n_core,
                                           uint32_t& n_in, double (&es)[2])
    {
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    // chunk kk of this lane: uniform base + kk KiB (scalar) + 16 lane (one VGPR)
#if defined(AZP_ABLATE) && (AZP_ABLATE == 4)
    // ablation 4: HALF the index stream (8 bytes per lane and chunk, used twice): same LDS gathers, same arithmetic
    auto chunk_at = [&](uint32_t kk) -> uint4
        {
        const uint2 h = *reinterpret_cast<const uint2*>(slice_base + (uint64_t)kk * 512u + (lane_off >> 1));
        return make_uint4(h.x, h.y, h.x, h.y);
        };
#else
    auto chunk_at = [&](uint32_t kk) -> uint4
        { return *reinterpret_cast<const uint4*>(slice_base + (uint64_t)kk * 1024u + lane_off); };
#endif
#if AZP_TILE_BATCH == 4
    // (prefetching the chunk indices two iterations ahead instead of one costs 9 more
    // spilled registers and measures 3 % slower)
    uint4 u = (K > 0) ? chunk_at(0) : zero4;
    uint4 un = (K > 1) ? chunk_at(1) : u;
    TileBatch A, B;
    tile_gather<CAP, 0>(A, u, bx);
    uint32_t kk = 0;
    // the three phases of a row (K1 = K2 = K: one phase, every test) share the pipeline state
    auto run = [&](auto mode, const uint32_t kend)
        {
        constexpr int MODE = decltype(mode)::value;
        for (; kk < kend; ++kk)
            {
            // indices two chunks ahead (consumed 1.5 iterations from now: one iteration
            // does not always cover an HBM round trip); past the end the last chunk is
            // reloaded (an unconditional 16-byte load: a predicated one is split into
            // four 4-byte loads)
            const uint4 un2 = chunk_at((kk + 2 < K) ? kk + 2 : K - 1);
            tile_gather<CAP, 1>(B, u, bx);
            __builtin_amdgcn_sched_barrier(0);
            tile_compute<E, CAP, VIRIAL, SINGLE, XPLOR, WRAP, MODE>(A, a, bt, s_coeff, s_ronsq, c0, ronsq0, rcutsq_max, pi, typei, fx, fy, fz, pe, v, n_core, n_in, es);
            __builtin_amdgcn_sched_barrier(0);
            tile_gather<CAP, 0>(A, un, bx); // when kk + 1 == K: gathered, never used
            __builtin_amdgcn_sched_barrier(0);
            tile_compute<E, CAP, VIRIAL, SINGLE, XPLOR, WRAP, MODE>(B, a, bt, s_coeff, s_ronsq, c0, ronsq0, rcutsq_max, pi, typei, fx, fy, fz, pe, v, n_core, n_in, es);
            __builtin_amdgcn_sched_barrier(0);
            u = un;
            un = un2;
            }
        };
    run(std::integral_constant<int, 0>(), K1);
    if constexpr (SINGLE && !XPLOR && E::kSplitEnergy)
        {
        run(std::integral_constant<int, 1>(), K2);
        run(std::integral_constant<int, 2>(), K);
        }
#else
    // whole-chunk batches: chunk k+1's 24 gathers fly while chunk k is evaluated
    uint4 u0 = (K > 0) ? chunk_at(0) : zero4;
    uint4 u1 = (K > 1) ? chunk_at(1) : zero4;
    TileBatch A, B;
    tile_gather<CAP, 0>(A, u0, bx);
    for (uint32_t kk = 0; kk < K; kk += 2)
        {
        const uint4 u2 = (kk + 2 < K) ? chunk_at(kk + 2) : zero4;
        const uint4 u3 = (kk + 3 < K) ? chunk_at(kk + 3) : zero4;
        tile_gather<CAP, 0>(B, u1, bx);
        __builtin_amdgcn_sched_barrier(0);
        tile_compute<E, CAP, VIRIAL, SINGLE, XPLOR, WRAP>(A, a, bt, s_coeff, s_ronsq, c0, ronsq0, rcutsq_max, pi, typei, fx, fy, fz, pe, v, n_core, n_in, es);
        __builtin_amdgcn_sched_barrier(0);
        tile_gather<CAP, 0>(A, u2, bx);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < K)
            tile_compute<E, CAP, VIRIAL, SINGLE, XPLOR, WRAP>(B, a, bt, s_coeff, s_ronsq, c0, ronsq0, rcutsq_max, pi, typei, fx, fy, fz, pe, v, n_core, n_in, es);
        __builtin_amdgcn_sched_barrier(0);
        u1 = u3;
        }
#endif
    }

template<class E, int TPP, int CAP, bool VIRIAL, bool SINGLE, bool XPLOR>
__global__ void __launch_bounds__(256, E::kTileWaves) pair_forces_tiled_kernel(const TiledKArgs a, const typename E::Params* __restrict__ params)
    {
    typedef typename E::Coeff Coeff;
    constexpr int TB = 256 / TPP;
    constexpr int PW = 64 / TPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double* s_x = reinterpret_cast<double*>(s_raw);
    double* s_y = s_x + AZP_TILE_STRIDE(CAP);
    double* s_z = s_y + AZP_TILE_STRIDE(CAP);
    int* s_t = reinterpret_cast<int*>(s_z + AZP_TILE_STRIDE(CAP)); // CAP ints, only when !SINGLE
    Coeff* s_coeff = reinterpret_cast<Coeff*>(s_raw + (size_t)AZP_TILE_STRIDE(CAP) * 24 + (SINGLE ? 0 : (size_t)CAP * 4 + 8));
    double* s_ronsq = reinterpret_cast<double*>(s_coeff + (SINGLE ? 0 : a.p.ntypes * a.p.ntypes));

    const uint32_t tid = threadIdx.x;
#ifndef AZP_NO_STAGE_PRIO
    // A new tile's waves share their SIMDs with three older waves that are deep in
    // the pair loop; raise the priority while staging so the short prologue is not
    // starved (tools/timeline.py: the staging took 17 of a tile's 41 us).
    __builtin_amdgcn_s_setprio(3);
#endif
#ifdef AZP_TIMELINE
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tl_c0 = __builtin_amdgcn_s_memtime(); // shader clock
    unsigned long long tl_tb = 0, tl_tc = 0, tl_td = 0, tl_ta = 0;
#endif
    // a.p.first / a.p.end are tile-aligned outwards by the launcher
    uint32_t tile = a.p.first / TB + xcd_remap(blockIdx.x, a.p.nblocks_padded);
    if (a.tile_ids)
        {
        const uint32_t b = xcd_remap(blockIdx.x, a.p.nblocks_padded);
        if (b >= a.n_tile_ids)
            return;
        tile = a.tile_ids[b];
        }
    const uint32_t first = tile * TB;
    if (first >= a.p.end)
        return;
    if (a.dyn && a.dyn->stale) // speculative launch on a plan that turned out stale (uniform: whole grid leaves)
        return;
    if (a.dflag && *a.dflag)   // ... or whose list has to be rebuilt first (the distance check said so)
        return;

    Coeff c0;
    double ronsq0 = 0.0;
    double rcutsq_max = 0.0; // largest cutoff^2 over the type pairs: bound for the skip test
    if (SINGLE)
        {
        c0 = to_uniform(prepare_coeff<E>(a.p, params, 0)); // same for every lane: keep it in SGPRs
        if (XPLOR)
            ronsq0 = a.p.ronsq[0];
        rcutsq_max = c0.rcutsq; // the evaluator's own (effective) cutoff: one compare serves both tests
        }
    else
        {
        const uint32_t ntp = a.p.ntypes * a.p.ntypes;
        for (uint32_t t = tid; t < ntp; t += 256)
            {
            s_coeff[t] = prepare_coeff<E>(a.p, params, t);
            s_ronsq[t] = XPLOR ? a.p.ronsq[t] : 0.0;
            }
        for (uint32_t t = 0; t < ntp; ++t)
            rcutsq_max = fmax(rcutsq_max, a.p.rcutsq[t]);
        }

#ifdef AZP_TIMELINE
    tl_ta = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- stage: positions shifted to the periodic image nearest the tile's
    // reference particle c ----
    const uint32_t n_stage = a.tile_nstage[tile];
    const uint32_t* __restrict__ stage = a.stage_idx + a.tile_head[tile];
    const double3 c = load_scalar3_of4(a.p.pos, first);
    if (tid == 0)
        {
        s_x[0] = PLAN_FAR; s_y[0] = PLAN_FAR; s_z[0] = PLAN_FAR;
        if (!SINGLE) s_t[0] = 0;
        }
    // this lane's own particle: its load is issued first and consumed after the staging
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63; // wave id: scalar
    const uint32_t pl = lane / TPP;
    const uint32_t idx = first + ((TPP == 1 && a.perm) ? (uint32_t)a.perm[(uint64_t)tile * 256u + tid] : wave * PW + pl);
    const bool active = idx < a.p.end;
    const double4 own = load_scalar4(a.p.pos, active ? idx : first);
    // All loads of the staging are issued before the first result is used: the index
    // loads of every round first, then every position load (two dependent HBM round
    // trips per tile instead of two per 256 staged particles -- the staging took 17 of
    // a tile's 41 us when it was written as a plain loop, tools/timeline.py).
    // Without the r_list_max hint (callers that only have HOOMD's pair_args_t) the tile is
    // "compact" when every staged image and every member lies within L / 4 of the reference
    // particle on each periodic axis: then any two of them are closer than L / 2 per axis, so
    // the staged image of a neighbor IS its minimum image for every member.
    const bool no_hint = !(a.p.r_list_max > 0.0);
    bool far_from_c = false;
    float dmax = a.disp ? a.disp[active ? idx : first] : 0.f; // largest displacement among what this lane stages, and its own
    {
    constexpr int ROUNDS = (CAP + 255) / 256; // n_stage < CAP
    uint32_t sj[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        {
        const uint32_t sidx = (uint32_t)r * 256u + tid;
        sj[r] = (sidx < n_stage) ? stage[sidx] : first; // out-of-range lanes load the reference particle (unused)
        }
#ifdef AZP_TIMELINE
    __builtin_amdgcn_s_waitcnt(0); // all counters
    tl_tb = __builtin_amdgcn_s_memrealtime();
#endif
    double sxv[ROUNDS], syv[ROUNDS], szv[ROUNDS];
    int stv[ROUNDS];
    if (a.disp)
        {
        // (lanes beyond the staged set read the reference particle's entry: a member of the tile anyway)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            dmax = fmaxf(dmax, a.disp[sj[r]]);
        }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        {
        if (SINGLE)
            {
            const double3 pj = load_scalar3_of4(a.p.pos, sj[r]);
            sxv[r] = pj.x; syv[r] = pj.y; szv[r] = pj.z;
            }
        else
            {
            const double4 pj = load_scalar4(a.p.pos, sj[r]);
            sxv[r] = pj.x; syv[r] = pj.y; szv[r] = pj.z;
            stv[r] = type_from_w(pj.w);
            }
        }
#ifdef AZP_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);
    tl_tc = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        {
        const uint32_t sidx = (uint32_t)r * 256u + tid;
        if (sidx < n_stage)
            {
            double x = sxv[r], y = syv[r], z = szv[r];
            if (!a.p.box.triclinic)
                {
                if (a.p.box.px) x = __builtin_fma(-a.p.box.Lx, rint((x - c.x) * a.p.box.Lxinv), x);
                if (a.p.box.py) y = __builtin_fma(-a.p.box.Ly, rint((y - c.y) * a.p.box.Lyinv), y);
                if (a.p.box.pz) z = __builtin_fma(-a.p.box.Lz, rint((z - c.z) * a.p.box.Lzinv), z);
                if (no_hint)
                    far_from_c = far_from_c || (a.p.box.px && fabs(x - c.x) >= 0.25 * a.p.box.Lx)
                                 || (a.p.box.py && fabs(y - c.y) >= 0.25 * a.p.box.Ly)
                                 || (a.p.box.pz && fabs(z - c.z) >= 0.25 * a.p.box.Lz);
                }
            s_x[sidx + 1] = x; s_y[sidx + 1] = y; s_z[sidx + 1] = z;
            if (!SINGLE)
                s_t[sidx + 1] = stv[r];
            }
        }
    }

    // ---- this lane's particle, in the same image frame ----
    double3 pi = make_double3(0.0, 0.0, 0.0);
    int typei = 0;
    // Fast path condition, per tile: every member is closer to c than L/2 minus
    // the largest possible pair separation, so the staged image of each listed
    // neighbor IS its minimum image. Otherwise every pair is re-imaged.
    bool lane_wide = far_from_c || a.p.box.triclinic;
    if (active)
        {
        const double4 p = own;
        double x = p.x, y = p.y, z = p.z;
        if (!a.p.box.triclinic)
            {
            if (a.p.box.px) x = __builtin_fma(-a.p.box.Lx, rint((x - c.x) * a.p.box.Lxinv), x);
            if (a.p.box.py) y = __builtin_fma(-a.p.box.Ly, rint((y - c.y) * a.p.box.Lyinv), y);
            if (a.p.box.pz) z = __builtin_fma(-a.p.box.Lz, rint((z - c.z) * a.p.box.Lzinv), z);
            // (without the hint r_reach = L / 4: the member itself has to be within L / 4 of c)
            const double rx = no_hint ? 0.25 * a.p.box.Lx : a.p.r_list_max, ry = no_hint ? 0.25 * a.p.box.Ly : a.p.r_list_max,
                         rz = no_hint ? 0.25 * a.p.box.Lz : a.p.r_list_max;
            lane_wide = lane_wide || (a.p.box.px && fabs(x - c.x) + rx >= 0.5 * a.p.box.Lx)
                        || (a.p.box.py && fabs(y - c.y) + ry >= 0.5 * a.p.box.Ly)
                        || (a.p.box.pz && fabs(z - c.z) + rz >= 0.5 * a.p.box.Lz);
            }
        pi = make_double3(x, y, z);
        typei = type_from_w(p.w);
        }
#ifdef AZP_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);
    tl_td = __builtin_amdgcn_s_memrealtime();
#endif
    __shared__ float s_dmax[4];
    if (a.disp)
        {
        dmax = wave_max_nonneg(dmax);
        if (lane == 0)
            s_dmax[wave] = dmax;
        }
    const bool wide = __syncthreads_or(lane_wide); // also publishes the staged tile
#ifndef AZP_NO_STAGE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef AZP_TIMELINE
    const unsigned long long tl_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    // the displacement bound of this tile and the shells it has to walk
    double bound = a.dyn ? a.dyn->bound : a.bound;
    uint32_t n_shells = a.dyn ? min(a.dyn->n_shells, PLAN_SHELLS) : a.n_shells;
    if (a.dbits)
        {
        bound = to_uniform(sqrt(__longlong_as_double((long long)*a.dbits)) + a.bound_extra);
        n_shells = tile_shells_for(bound, a.shell_winv);
        if (!(bound >= 0.0) || !(bound < 1.0e300))
            bound = -1.0;
        }
    if (a.disp)
        {
        const float d4 = fmaxf(fmaxf(s_dmax[0], s_dmax[1]), fmaxf(s_dmax[2], s_dmax[3]));
        bound = to_uniform((double)d4 + a.bound_extra); // NaN / inf displacements: whole rows, every test
        n_shells = tile_shells_for(bound, a.shell_winv);
        if (!(bound >= 0.0) || !(bound < 1.0e300))
            bound = -1.0;
        }

    const uint32_t slice = tile * 4 + wave;
    // scalar trip count (the loop counter and the chunk address stay in SGPRs). With a
    // displacement bound from the caller the row ends early: entries that were at
    // least 2 x bound outside the cutoff when the plan was built cannot be in range.
    const uint32_t K = to_uniform(n_shells >= PLAN_SHELLS ? a.slice_K[slice] : a.slice_Kend[(PLAN_SHELLS + 1) * slice + n_shells]);
    // wave-uniform slice base (SGPRs) + lane: the loads use scalar-base addressing
    const uint64_t slice_head = to_uniform(a.slice_head[slice]);
    const char* __restrict__ slice_base = reinterpret_cast<const char*>(a.cnl + slice_head * 64ull);
    const uint32_t lane_off = lane * 16u;

    double fx = 0.0, fy = 0.0, fz = 0.0, pe = 0.0;
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    uint32_t n_core = 0, n_in = 0;
    double es[2] = {0.0, 0.0}; // tail-path energy sums (EvalPLJ::eval_split_tail)
    const char* bx = reinterpret_cast<const char*>(s_x);
    const char* bt = reinterpret_cast<const char*>(s_t);
    // row phases: [0, K1) every test, [K1, K2) no test, [K2, K) no core test (see TiledKArgs)
    uint32_t K1 = K, K2 = K;
    if constexpr (SINGLE && !XPLOR && E::kSplitEnergy)
        {
        if (a.slice_Kcore && bound >= 0.0 && a.core_r > 0.f)
            {
            const double reach = 2.0 * bound;
            const bool core_ok = E::core_radius(c0) + reach <= (double)a.core_r; // false for NaN (no interaction: c0.rcutsq < 0)
            const bool sure_ok = core_ok && a.sure_r > 0.f && (double)a.sure_r + reach <= sqrt(c0.rcutsq);
            if (core_ok)
                {
                K1 = to_uniform(min(a.slice_Kcore[slice], K));
                K2 = sure_ok ? to_uniform(min(max(a.slice_Ksure[slice], K1), K)) : K1;
                }
            }
        }
#ifdef AZP_DEBUG_PHASES
    if (tile == 7 && lane == 0)
        printf("tile %u wave %u: K1 %u K2 %u K %u bound %g (global %g) core_r %g sure_r %g n_shells %u (global %u) wide %d\n", tile, wave, K1, K2, K, bound, a.bound, (double)a.core_r, (double)a.sure_r, n_shells, a.n_shells, (int)wide);
#endif
    if (wide)
        tiled_loop<E, TPP, CAP, VIRIAL, SINGLE, XPLOR, true>(a, bx, bt, s_coeff, s_ronsq, c0, ronsq0, rcutsq_max, slice_base, lane_off, K, K1, K2, pi,
                                                            typei, fx, fy, fz, pe, v, n_core, n_in, es);
    else
        tiled_loop<E, TPP, CAP, VIRIAL, SINGLE, XPLOR, false>(a, bx, bt, s_coeff, s_ronsq, c0, ronsq0, rcutsq_max, slice_base, lane_off, K, K1, K2, pi,
                                                             typei, fx, fy, fz, pe, v, n_core, n_in, es);
    if constexpr (SINGLE && !XPLOR && E::kSplitEnergy)
        pe = E::finish_split(c0, pe, es[0], es[1], n_core, n_in);

#ifdef AZP_TIMELINE
    if (lane == 0 && tile < 16384)
        {
        unsigned long long* t = g_timeline + (tile * 4 + wave) * 8;
        t[4] = tl_ta; t[5] = tl_tb; t[6] = tl_tc; t[7] = tl_td;
        if (wave == 3)
            t[4] = __builtin_amdgcn_s_memtime() - tl_c0; // wave 3 reports shader-clock ticks over its lifetime
        t[0] = tl_t0; t[1] = tl_t1; t[2] = __builtin_amdgcn_s_memrealtime();
        t[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11)) << 32)
               | (unsigned)__builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
        }
#endif
    fx = group_sum<TPP>(fx);
    fy = group_sum<TPP>(fy);
    fz = group_sum<TPP>(fz);
    pe = group_sum<TPP>(pe);
    if (VIRIAL)
        {
#pragma unroll
        for (int cidx = 0; cidx < 6; ++cidx)
            v[cidx] = group_sum<TPP>(v[cidx]);
        }
    if (active && (lane % TPP) == 0)
        {
        store_scalar4(a.p.force, idx, fx, fy, fz, 0.5 * pe);
        if (VIRIAL)
            {
#pragma unroll
            for (int cidx = 0; cidx < 6; ++cidx)
                a.p.virial[(uint64_t)cidx * a.p.virial_pitch + idx] = 0.5 * v[cidx];
            }
        }
    }

template<class E, int TPP, int CAP, bool VIRIAL, bool SINGLE, bool XPLOR>
int launch_tiled_instance2(const PairPlan& plan, const azp_pair_args& args, const typename E::Params* d_params,
                          hipStream_t stream, const TileDyn* dyn, const uint32_t* tile_ids = nullptr, uint32_t n_tile_ids = 0)
    {
    TiledKArgs k = {};
    k.p = make_pair_kargs(args);
    k.tile_nstage = plan.d_tile_nstage;
    k.tile_head = plan.d_tile_head;
    k.stage_idx = plan.d_stage_idx;
    k.perm = plan.balanced ? plan.d_perm : nullptr;
    k.slice_K = plan.d_slice_K;
    k.slice_Kend = plan.d_slice_Kend;
    k.n_shells = plan_shells_for(plan, args);
    const bool phases = tuning().row_phases != 0; // (azp_tuning_set: A/B measurements)
    k.slice_Kcore = phases ? plan.d_slice_Kphase : nullptr;
    k.slice_Ksure = (phases && plan.d_slice_Kphase) ? plan.d_slice_Kphase + plan.n_slices : nullptr;
    k.bound = (args.has_displacement_bound && args.displacement_bound >= 0.0) ? args.displacement_bound : -1.0;
    k.core_r = plan.core_r;
    k.sure_r = plan.sure_r;
    fill_local_bound(k, plan, args);
    k.dyn = dyn;
    k.slice_head = plan.d_slice_head;
    k.cnl = plan.d_cnl;
    // sub-range launches are rounded outwards to whole tiles (a tile computed by
    // two launches of one step gets the later launch's values: stream order)
    const uint32_t tb = plan.tile;
    const uint32_t t0 = k.p.first / tb, t1 = (k.p.end + tb - 1) / tb;
    k.p.first = t0 * tb;
    k.p.end = (t1 * tb < args.N) ? t1 * tb : args.N;
    k.tile_ids = tile_ids;
    k.n_tile_ids = n_tile_ids;
    const uint32_t nblocks = ((tile_ids ? n_tile_ids : t1 - t0) + 7u) & ~7u;
    k.p.nblocks_padded = nblocks;
    size_t lds = (size_t)AZP_TILE_STRIDE(CAP) * 24 + (SINGLE ? 0 : (size_t)CAP * 4 + 8);
    if (!SINGLE)
        lds += (sizeof(typename E::Coeff) + sizeof(double)) * (size_t)args.ntypes * args.ntypes;
    if (lds > 160 * 1024)
        return AZP_ERROR_TOO_MANY_TYPES;
    auto kern = pair_forces_tiled_kernel<E, TPP, CAP, VIRIAL, SINGLE, XPLOR>;
    if (lds > 64 * 1024)
        {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return (int)e;
        }
    LaunchInfo& li = last_launch();
    li.block_size = 256; li.tpp = TPP; li.grid = nblocks; li.lds_bytes = (uint32_t)lds;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, stream, k, d_params);
    return (int)hipGetLastError();
    }

template<class E, int TPP, int CAP, bool VIRIAL, bool SINGLE>
int launch_tiled_instance(const PairPlan& plan, const azp_pair_args& args, const typename E::Params* d_params,
                          hipStream_t stream, const TileDyn* dyn, const uint32_t* tile_ids = nullptr, uint32_t n_tile_ids = 0)
    {
    if (args.shift_mode == AZP_SHIFT_XPLOR)
        return launch_tiled_instance2<E, TPP, CAP, VIRIAL, SINGLE, true>(plan, args, d_params, stream, dyn, tile_ids, n_tile_ids);
    return launch_tiled_instance2<E, TPP, CAP, VIRIAL, SINGLE, false>(plan, args, d_params, stream, dyn, tile_ids, n_tile_ids);
    }

// tile numbers of the two groups of a split launch (see PairPlan::h_tile_ids), uploaded once per build
inline int plan_split_tiles(const PairPlan& plan, hipStream_t s)
    {
    if (plan.tile_ids_build == plan.builds && plan.d_tile_ids)
        return AZP_SUCCESS;
    const uint32_t n = plan.n_tiles;
    plan.h_tile_ids.resize(n);
    uint32_t small = 0;
    for (uint32_t t = 0; t < n; ++t)
        if (plan.h_tile_nstage[t] + 1u <= 1664u)
            plan.h_tile_ids[small++] = t;
    uint32_t large = small;
    for (uint32_t t = 0; t < n; ++t)
        if (plan.h_tile_nstage[t] + 1u > 1664u)
            plan.h_tile_ids[large++] = t;
    if (plan.cap_tile_ids < n)
        {
        if (plan.d_tile_ids) (void)hipFree(plan.d_tile_ids);
        plan.d_tile_ids = nullptr;
        plan.cap_tile_ids = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&plan.d_tile_ids), sizeof(uint32_t) * (size_t)(n + 64));
        if (e != hipSuccess)
            return (int)e;
        plan.cap_tile_ids = n + 64;
        }
    hipError_t e = hipMemcpyAsync(plan.d_tile_ids, plan.h_tile_ids.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, s);
    if (e != hipSuccess)
        return (int)e;
    plan.n_small_tiles = small;
    plan.tile_ids_build = plan.builds;
    return AZP_SUCCESS;
    }

template<class E, int TPP, bool VIRIAL, bool SINGLE>
int launch_tiled_cap(const PairPlan& plan, const azp_pair_args& args, const typename E::Params* d_params, hipStream_t s, const TileDyn* dyn)
    {
    // LDS variant: from the tiles this launch covers (all of them unless a range is given)
    uint32_t cap = plan.cap;
    if (args.range_count != 0 && !plan.h_tile_nstage.empty())
        {
        const uint32_t tb = plan.tile;
        const uint32_t end = (args.range_first + args.range_count < args.N) ? args.range_first + args.range_count : args.N;
        const uint32_t t0 = args.range_first / tb, t1 = (end + tb - 1) / tb;
        uint32_t most = 0;
        for (uint32_t t = t0; t < t1 && t < plan.n_tiles; ++t)
            most = plan.h_tile_nstage[t] > most ? plan.h_tile_nstage[t] : most;
        cap = plan_cap_for(most);
        }
    if (TPP == 1 && args.range_count == 0 && cap > 1664u && tuning().split_tiles && plan.h_tile_nstage.size() == plan.n_tiles)
        {
        // a liquid: the tiles that fit the 1,664-slot variant (four workgroups per CU) first, then the rest
        const int rc = plan_split_tiles(plan, s);
        if (rc != AZP_SUCCESS)
            return rc;
        const uint32_t n_small = plan.n_small_tiles, n_large = plan.n_tiles - n_small;
        if (n_small * 2u >= plan.n_tiles && n_large > 0)
            {
            const int r1 = launch_tiled_instance<E, TPP, 1664, VIRIAL, SINGLE>(plan, args, d_params, s, dyn, plan.d_tile_ids, n_small);
            if (r1 != AZP_SUCCESS)
                return r1;
            if (cap == 2048u)
                return launch_tiled_instance<E, TPP, 2048, VIRIAL, SINGLE>(plan, args, d_params, s, dyn, plan.d_tile_ids + n_small, n_large);
            return launch_tiled_instance<E, TPP, 2560, VIRIAL, SINGLE>(plan, args, d_params, s, dyn, plan.d_tile_ids + n_small, n_large);
            }
        }
    switch (cap)
        {
    case 1024: return launch_tiled_instance<E, TPP, 1024, VIRIAL, SINGLE>(plan, args, d_params, s, dyn);
    case 1536: return launch_tiled_instance<E, TPP, 1536, VIRIAL, SINGLE>(plan, args, d_params, s, dyn);
    case 1664: return launch_tiled_instance<E, TPP, 1664, VIRIAL, SINGLE>(plan, args, d_params, s, dyn);
    case 2048: return launch_tiled_instance<E, TPP, 2048, VIRIAL, SINGLE>(plan, args, d_params, s, dyn);
    case 2560: return launch_tiled_instance<E, TPP, 2560, VIRIAL, SINGLE>(plan, args, d_params, s, dyn);
    default: return AZP_ERROR_INVALID_ARGUMENT;
        }
    }

template<class E, bool VIRIAL, bool SINGLE>
int launch_tiled_tpp(const PairPlan& plan, const azp_pair_args& args, const typename E::Params* d_params, hipStream_t s, const TileDyn* dyn)
    {
    switch (plan.tpp)
        {
    case 1: return launch_tiled_cap<E, 1, VIRIAL, SINGLE>(plan, args, d_params, s, dyn);
    case 2: return launch_tiled_cap<E, 2, VIRIAL, SINGLE>(plan, args, d_params, s, dyn);
    case 4: return launch_tiled_cap<E, 4, VIRIAL, SINGLE>(plan, args, d_params, s, dyn);
    default: return AZP_ERROR_INVALID_ARGUMENT;
        }
    }

// Entry: use the plan when it is valid for these arguments, else the generic kernel.
template<class E>
int launch_pair_planned(azp_pair_plan* plan_, const azp_pair_args* args, const typename E::Params* d_params, void* stream,
                        const TileDyn* dyn = nullptr)
    {
    if (!plan_)
        return AZP_ERROR_INVALID_ARGUMENT;
    const PairPlan& plan = *reinterpret_cast<const PairPlan*>(plan_);
    const int bad = validate_pair_args(args, d_params);
    if (bad < 0) return bad;
    if (bad > 0) return AZP_SUCCESS;
    // a plan compiled from a different list is a caller bug, not a fallback case
    if (plan.builds == 0 || plan.N != args->N || plan.nlist_ptr != args->d_nlist || plan.head_ptr != args->d_head_list)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (!plan.valid) // (a plan compiled from the cell list has no HOOMD-format list to fall back to)
        return plan.from_cells ? AZP_ERROR_INVALID_ARGUMENT : launch_pair<E>(args, d_params, stream);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool single = (args->ntypes == 1);
    if (args->compute_virial)
        return single ? launch_tiled_tpp<E, true, true>(plan, *args, d_params, s, dyn)
                      : launch_tiled_tpp<E, true, false>(plan, *args, d_params, s, dyn);
    return single ? launch_tiled_tpp<E, false, true>(plan, *args, d_params, s, dyn)
                  : launch_tiled_tpp<E, false, false>(plan, *args, d_params, s, dyn);
    }

} // namespace azp
