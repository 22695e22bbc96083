// xtiled.hpp -- tile-staged kernel skeleton for the pair potentials that need more than
// the separation of a pair: the DPD thermostat (velocities + tags, src/
// PotentialPairDPDThermoGPUKernel.cu.inc:21-24) and TwoPatchMorse (orientations, force +
// torque, src/AnisoPotentialPairGPUKernel.cu.inc:21-25). Same plan, same compiled rows as
// pair_tiled.hpp; per staged particle the LDS additionally holds what the potential reads
// of a neighbor, loaded (or derived) ONCE per tile instead of once per pair:
//
//   DPD       x | y | z | vx | vy | vz | tag           52 B per slot
//   TwoPatch  x | y | z | nx | ny | nz                 48 B per slot, n = rotate(q_j, x^):
//             the 32-byte quaternion gather and the rotation leave the pair loop
//
// Rows are ordered in-range first (pair_plan.hip), so the expensive per-pair block
// (Philox4x32-10, exponentials) runs on dense waves: with the generic kernel a row position
// holds an in-range pair in ~35 % (DPD, <n> = 34.5, 12.6 in range) of the lanes, here the
// first ~2 chunks are in range in every lane and the rest of the row is skipped by a
// wave-uniform test (or never walked: displacement bound, as in pair_tiled.hpp).
//
// One lane per particle (plans with threads_per_particle = 1); a policy class X supplies
// the payload and the arithmetic.
#pragma once

#include <type_traits>

#include "pair_tiled.hpp"

namespace azp
{
// LDS bytes of the staged tile: positions, the policy's extra doubles, tags, types
template<class X, bool SINGLE> __host__ __device__ constexpr size_t xtiled_lds_slots(size_t cap)
    {
    return cap * (24 + 8 * (size_t)X::kExtra + (X::kTag ? 4 : 0) + (SINGLE ? 0 : 4)); // cap is even: a multiple of 8
    }

template<class X, int CAP, bool VIRIAL, bool SINGLE>
__global__ void __launch_bounds__(256, X::kMinWaves) xtiled_kernel(const TiledKArgs a, const typename X::KExtra x, const typename X::Params* __restrict__ params)
    {
    typedef typename X::Coeff Coeff;
    constexpr int NE = X::kExtra; // extra doubles per slot
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double* s_pos = reinterpret_cast<double*>(s_raw);                   // 3 x CAP
    double* s_ext = s_pos + 3 * CAP;                                    // NE x CAP
    uint32_t* s_tag = reinterpret_cast<uint32_t*>(s_ext + NE * CAP);    // CAP (X::kTag)
    int* s_type = reinterpret_cast<int*>(s_tag + (X::kTag ? CAP : 0));  // CAP (!SINGLE)
    Coeff* s_coeff = reinterpret_cast<Coeff*>(s_raw + xtiled_lds_slots<X, SINGLE>(CAP));

    const uint32_t tid = threadIdx.x;
    const uint32_t tile = a.p.first / 256u + xcd_remap(blockIdx.x, a.p.nblocks_padded);
    const uint32_t first = tile * 256u;
    if (first >= a.p.end)
        return;
    if (a.dyn && a.dyn->stale) // speculative launch on a stale plan (pair_auto.hpp)
        return;
    if (a.dflag && *a.dflag)   // ... or on a list the distance check wants rebuilt (azp_pair_args.d_stale_flag)
        return;

    Coeff c0;
    if (SINGLE)
        c0 = to_uniform(X::prepare(params[0], a.p.rcutsq[0], x, a.p.shift_mode));
    else
        {
        const uint32_t ntp = a.p.ntypes * a.p.ntypes;
        for (uint32_t t = tid; t < ntp; t += 256)
            s_coeff[t] = X::prepare(params[t], a.p.rcutsq[t], x, a.p.shift_mode);
        }

    // ---- stage ----
    const uint32_t n_stage = a.tile_nstage[tile];
    const uint32_t* __restrict__ stage = a.stage_idx + a.tile_head[tile];
    const double3 c = load_scalar3_of4(a.p.pos, first);
    if (tid == 0)
        {
        s_pos[0] = PLAN_FAR; s_pos[CAP] = PLAN_FAR; s_pos[2 * CAP] = PLAN_FAR;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            s_ext[e * CAP] = 0.0;
        if (X::kTag) s_tag[0] = 0;
        if (!SINGLE) s_type[0] = 0;
        }
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63;
    const uint32_t idx = first + (a.perm ? (uint32_t)a.perm[(uint64_t)tile * 256u + tid] : wave * 64u + lane);
    const bool active = idx < a.p.end;
    const bool no_hint = !(a.p.r_list_max > 0.0);
    bool lane_wide = a.p.box.triclinic;
    float dmax = a.disp ? a.disp[active ? idx : first] : 0.f; // local displacement bound (pair_tiled.hpp)
    // all loads of the staging are issued before the first result is used: the index loads of
    // every round, then every position / payload load (two dependent round trips per tile)
    __builtin_amdgcn_s_setprio(3); // a new tile shares its SIMDs with waves deep in the pair loop
    {
    constexpr int ROUNDS = (CAP + 255) / 256;
    uint32_t sj[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        {
        const uint32_t sidx = (uint32_t)r * 256u + tid;
        sj[r] = (sidx < n_stage) ? stage[sidx] : first;
        }
    double4 pj[ROUNDS];
    double ext[ROUNDS][NE];
    uint32_t tagj[ROUNDS];
    if (a.disp)
        {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            dmax = fmaxf(dmax, a.disp[sj[r]]);
        }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        {
        pj[r] = load_scalar4(a.p.pos, sj[r]);
        tagj[r] = 0;
        X::load_extra(x, sj[r], ext[r], tagj[r]);
        }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        {
        const uint32_t sidx = (uint32_t)r * 256u + tid;
        if (sidx < n_stage)
            {
            double px = pj[r].x, py = pj[r].y, pz = pj[r].z;
            if (!a.p.box.triclinic)
                {
                if (a.p.box.px) px = __builtin_fma(-a.p.box.Lx, rint((px - c.x) * a.p.box.Lxinv), px);
                if (a.p.box.py) py = __builtin_fma(-a.p.box.Ly, rint((py - c.y) * a.p.box.Lyinv), py);
                if (a.p.box.pz) pz = __builtin_fma(-a.p.box.Lz, rint((pz - c.z) * a.p.box.Lzinv), pz);
                if (no_hint)
                    lane_wide = lane_wide || (a.p.box.px && fabs(px - c.x) >= 0.25 * a.p.box.Lx)
                                || (a.p.box.py && fabs(py - c.y) >= 0.25 * a.p.box.Ly) || (a.p.box.pz && fabs(pz - c.z) >= 0.25 * a.p.box.Lz);
                }
            s_pos[sidx + 1] = px; s_pos[CAP + sidx + 1] = py; s_pos[2 * CAP + sidx + 1] = pz;
#pragma unroll
            for (int e = 0; e < NE; ++e)
                s_ext[e * CAP + sidx + 1] = ext[r][e];
            if (X::kTag) s_tag[sidx + 1] = tagj[r];
            if (!SINGLE) s_type[sidx + 1] = type_from_w(pj[r].w);
            }
        }
    }

    // ---- this lane's particle, in the same image frame ----
    double3 pi = make_double3(0.0, 0.0, 0.0);
    int typei = 0;
    typename X::Own own;
    X::load_own(x, active ? idx : first, own);
    if (active)
        {
        const double4 p = load_scalar4(a.p.pos, idx);
        double px = p.x, py = p.y, pz = p.z;
        if (!a.p.box.triclinic)
            {
            if (a.p.box.px) px = __builtin_fma(-a.p.box.Lx, rint((px - c.x) * a.p.box.Lxinv), px);
            if (a.p.box.py) py = __builtin_fma(-a.p.box.Ly, rint((py - c.y) * a.p.box.Lyinv), py);
            if (a.p.box.pz) pz = __builtin_fma(-a.p.box.Lz, rint((pz - c.z) * a.p.box.Lzinv), pz);
            const double rx = no_hint ? 0.25 * a.p.box.Lx : a.p.r_list_max, ry = no_hint ? 0.25 * a.p.box.Ly : a.p.r_list_max,
                         rz = no_hint ? 0.25 * a.p.box.Lz : a.p.r_list_max;
            lane_wide = lane_wide || (a.p.box.px && fabs(px - c.x) + rx >= 0.5 * a.p.box.Lx) || (a.p.box.py && fabs(py - c.y) + ry >= 0.5 * a.p.box.Ly)
                        || (a.p.box.pz && fabs(pz - c.z) + rz >= 0.5 * a.p.box.Lz);
            }
        pi = make_double3(px, py, pz);
        typei = type_from_w(p.w);
        }
    __shared__ float s_dmax[4];
    if (a.disp)
        {
        dmax = wave_max_nonneg(dmax);
        if (lane == 0)
            s_dmax[wave] = dmax;
        }
    const bool wide = __syncthreads_or(lane_wide); // also publishes the staged tile
    __builtin_amdgcn_s_setprio(0);
    uint32_t n_shells = a.dyn ? min(a.dyn->n_shells, PLAN_SHELLS) : a.n_shells;
    if (a.dbits)
        n_shells = tile_shells_for(to_uniform(sqrt(__longlong_as_double((long long)*a.dbits)) + a.bound_extra), a.shell_winv);
    if (a.disp)
        n_shells = tile_shells_for(to_uniform((double)fmaxf(fmaxf(s_dmax[0], s_dmax[1]), fmaxf(s_dmax[2], s_dmax[3])) + a.bound_extra), a.shell_winv);

    const uint32_t slice = tile * 4 + wave;
    const uint32_t K = to_uniform(n_shells >= PLAN_SHELLS ? a.slice_K[slice] : a.slice_Kend[(PLAN_SHELLS + 1) * slice + n_shells]);
    const uint64_t slice_head = to_uniform(a.slice_head[slice]);
    const uint4* __restrict__ rows = a.cnl + slice_head * 64ull + lane;

    typename X::Acc acc;
    X::zero(acc);
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double rcutsq_max = 0.0;
    if (SINGLE)
        rcutsq_max = c0.rcutsq;
    else
        for (uint32_t t = 0; t < a.p.ntypes * a.p.ntypes; ++t)
            rcutsq_max = fmax(rcutsq_max, a.p.rcutsq[t]);

#if defined(AZP_XTILED_ABLATE) && AZP_XTILED_ABLATE == 1
    const uint32_t Kloop = 0; // ablation: staging only
#else
    const uint32_t Kloop = K;
#endif
    auto walk = [&](auto wide_tag)
        {
        constexpr bool WIDE = decltype(wide_tag)::value;
        uint4 u = (Kloop > 0) ? rows[0] : make_uint4(0, 0, 0, 0);
        for (uint32_t kk = 0; kk < Kloop; ++kk)
            {
            const uint4 un = rows[(uint64_t)((kk + 1 < Kloop) ? kk + 1 : kk) * 64u]; // next chunk in flight
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
            uint32_t slot[8];
            double dx[8], dy[8], dz[8], rsq[8];
            bool any_in = false;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                {
                slot[e] = ((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu)) >> 3; // byte offset / 8
                dx[e] = pi.x - s_pos[slot[e]];
                dy[e] = pi.y - s_pos[CAP + slot[e]];
                dz[e] = pi.z - s_pos[2 * CAP + slot[e]];
                if (WIDE)
                    min_image(a.p.box, dx[e], dy[e], dz[e]);
                rsq[e] = __builtin_fma(dz[e], dz[e], __builtin_fma(dy[e], dy[e], dx[e] * dx[e]));
                if (WIDE)
                    rsq[e] = (slot[e] == 0) ? 1.0e60 : rsq[e]; // the minimum image would fold the padding slot back into the box
                any_in = any_in || !(rsq[e] > rcutsq_max);
                }
            if (__any(any_in))
                {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    {
                    Coeff cc;
                    if (SINGLE)
                        cc = c0;
                    else
                        cc = s_coeff[(uint32_t)typei * a.p.ntypes + (uint32_t)s_type[slot[e]]];
                    if (X::in_range(cc, rsq[e]))
                        {
                        double ext[NE];
#pragma unroll
                        for (int q = 0; q < NE; ++q)
                            ext[q] = s_ext[q * CAP + slot[e]];
                        const uint32_t tagj = X::kTag ? s_tag[slot[e]] : 0u;
                        X::template pair<VIRIAL>(cc, x, own, dx[e], dy[e], dz[e], rsq[e], ext, tagj, acc, v);
                        }
                    }
                }
            u = un;
            }
        };
    if (wide)
        walk(std::true_type());
    else
        walk(std::false_type());
    if (active)
        {
        X::store(acc, a.p, x, idx);
        if (VIRIAL)
            {
#pragma unroll
            for (int cidx = 0; cidx < 6; ++cidx)
                a.p.virial[(uint64_t)cidx * a.p.virial_pitch + idx] = 0.5 * v[cidx];
            }
        }
    }

template<class X, int CAP, bool VIRIAL, bool SINGLE>
int launch_xtiled_instance(const PairPlan& plan, const azp_pair_args& args, const typename X::KExtra& x, const typename X::Params* d_params,
                           hipStream_t stream, const TileDyn* dyn)
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
    k.slice_Kcore = k.slice_Ksure = nullptr; // (row phases: pair_tiled.hpp only)
    k.bound = -1.0;
    k.core_r = k.sure_r = 0.f;
    fill_local_bound(k, plan, args);
    k.dyn = dyn;
    k.slice_head = plan.d_slice_head;
    k.cnl = plan.d_cnl;
    const uint32_t t0 = k.p.first / 256u, t1 = (k.p.end + 255u) / 256u;
    k.p.first = t0 * 256u;
    k.p.end = (t1 * 256u < args.N) ? t1 * 256u : args.N;
    const uint32_t nblocks = (t1 - t0 + 7u) & ~7u;
    k.p.nblocks_padded = nblocks;
    size_t lds = xtiled_lds_slots<X, SINGLE>(CAP);
    if (!SINGLE)
        lds += sizeof(typename X::Coeff) * (size_t)args.ntypes * args.ntypes;
    if (lds > 160 * 1024)
        return AZP_ERROR_TOO_MANY_TYPES;
    auto kern = xtiled_kernel<X, CAP, VIRIAL, SINGLE>;
    if (lds > 64 * 1024)
        {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return (int)e;
        }
    LaunchInfo& li = last_launch();
    li.block_size = 256; li.tpp = 1; li.grid = nblocks; li.lds_bytes = (uint32_t)lds;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, stream, k, x, d_params);
    return (int)hipGetLastError();
    }

// true when the plan can drive the xtiled kernel for these arguments
inline bool xtiled_usable(const PairPlan& plan, const azp_pair_args& args)
    {
    return plan.valid && plan.tpp == 1 && plan.builds != 0 && plan.N == args.N && plan.nlist_ptr == args.d_nlist && plan.head_ptr == args.d_head_list;
    }

template<class X, bool VIRIAL, bool SINGLE>
int launch_xtiled_cap(const PairPlan& plan, const azp_pair_args& args, const typename X::KExtra& x, const typename X::Params* d_params, hipStream_t s,
                      const TileDyn* dyn)
    {
    uint32_t cap = plan.cap;
    if (args.range_count != 0 && !plan.h_tile_nstage.empty())
        {
        const uint32_t end = (args.range_first + args.range_count < args.N) ? args.range_first + args.range_count : args.N;
        const uint32_t t0 = args.range_first / 256u, t1 = (end + 255u) / 256u;
        uint32_t most = 0;
        for (uint32_t t = t0; t < t1 && t < plan.n_tiles; ++t)
            most = plan.h_tile_nstage[t] > most ? plan.h_tile_nstage[t] : most;
        cap = plan_cap_for(most);
        }
    switch (cap)
        {
    case 1024: return launch_xtiled_instance<X, 1024, VIRIAL, SINGLE>(plan, args, x, d_params, s, dyn);
    case 1536: return launch_xtiled_instance<X, 1536, VIRIAL, SINGLE>(plan, args, x, d_params, s, dyn);
    case 1664:
    case 2048: return launch_xtiled_instance<X, 2048, VIRIAL, SINGLE>(plan, args, x, d_params, s, dyn);
    case 2560: return launch_xtiled_instance<X, 2560, VIRIAL, SINGLE>(plan, args, x, d_params, s, dyn);
    default: return AZP_ERROR_INVALID_ARGUMENT;
        }
    }

template<class X>
int launch_xtiled(const PairPlan& plan, const azp_pair_args& args, const typename X::KExtra& x, const typename X::Params* d_params, hipStream_t s,
                  const TileDyn* dyn = nullptr)
    {
    const bool single = (args.ntypes == 1);
    if (args.compute_virial)
        return single ? launch_xtiled_cap<X, true, true>(plan, args, x, d_params, s, dyn) : launch_xtiled_cap<X, true, false>(plan, args, x, d_params, s, dyn);
    return single ? launch_xtiled_cap<X, false, true>(plan, args, x, d_params, s, dyn) : launch_xtiled_cap<X, false, false>(plan, args, x, d_params, s, dyn);
    }
} // namespace azp
