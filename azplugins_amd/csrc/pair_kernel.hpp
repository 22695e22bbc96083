// pair_kernel.hpp -- the neighbor-list pair-force kernel for isotropic
// evaluators (replaces HOOMD's gpu_compute_pair_forces<E>, requested by the
// reference at src/PotentialPairGPUKernel.cu.inc:25-28).
//
// Mapping (gfx950, wave64):
//   * TPP consecutive lanes cooperate on one particle; a wave covers 64/TPP
//     consecutive particles, so the neighbor-index rows a wave streams are
//     adjacent in memory and every fetched 128-B line is fully consumed.
//   * lanes stride the row: lane s handles entries s, s+TPP, ... ; the next
//     index is prefetched one iteration ahead.
//   * neighbor positions are gathered as 16-B loads and rely on L1 / the
//     XCD's L2 (block -> particle range mapping is XCD-aware).
//   * per-type-pair coefficients: registers when ntypes == 1, LDS otherwise.
//   * interior waves (every particle farther than r_list_max from all periodic
//     faces) skip the minimum-image arithmetic; the choice is wave-uniform.
//   * FP64 accumulate, DPP butterfly reduction over the TPP lanes, lane 0
//     writes force (fx, fy, fz, e) with two 16-B stores.
#pragma once

#include "evaluators.hpp"
#include "pair_kernel_host.hpp"

namespace azp
{
struct PairKArgs
    {
    double* force;
    double* virial;
    uint64_t virial_pitch;
    const double* pos;
    const uint32_t* n_neigh;
    const uint32_t* nlist;
    const uint64_t* head_list;
    const double* rcutsq;
    const double* ronsq;
    BoxDev box;
    double r_list_max;
    uint32_t N;
    uint32_t ntypes;
    uint32_t shift_mode;
    uint32_t nblocks_padded; // grid size, multiple of 8
    uint32_t first;          // particles [first, end) are computed by this launch
    uint32_t end;
    };

template<class E> __device__ __forceinline__ typename E::Coeff
prepare_coeff(const PairKArgs& a, const typename E::Params* params, uint32_t tp)
    {
    const double rcutsq = a.rcutsq[tp];
    bool energy_shift = (a.shift_mode == AZP_SHIFT_SHIFT);
    if (a.shift_mode == AZP_SHIFT_XPLOR && a.ronsq[tp] > rcutsq)
        energy_shift = true;
    return E::prepare(params[tp], rcutsq, energy_shift);
    }

template<class E, int TPP, bool VIRIAL, bool SINGLE, bool XPLOR, bool WRAP>
__device__ __forceinline__ void pair_loop(const PairKArgs& a, const typename E::Coeff* __restrict__ s_coeff,
                                          const double* __restrict__ s_ronsq, const typename E::Coeff& c0,
                                          double ronsq0, uint32_t sub, uint32_t n, uint64_t head, double3 pi,
                                          int typei, double& fx, double& fy, double& fz, double& pe, double (&v)[6])
    {
    const uint32_t* __restrict__ row = a.nlist + head;
    uint32_t k = sub;
    uint32_t j = (k < n) ? row[k] : 0u;
    while (k < n)
        {
        const uint32_t kn = k + TPP;
        const uint32_t jn = (kn < n) ? row[kn] : 0u; // prefetch next index
        double dx, dy, dz;
        int typej = 0;
        if (SINGLE)
            {
            const double3 pj = load_scalar3_of4(a.pos, j);
            dx = pi.x - pj.x; dy = pi.y - pj.y; dz = pi.z - pj.z;
            }
        else
            {
            const double4 pj = load_scalar4(a.pos, j);
            dx = pi.x - pj.x; dy = pi.y - pj.y; dz = pi.z - pj.z;
            typej = type_from_w(pj.w);
            }
        if (WRAP)
            min_image(a.box, dx, dy, dz);
        const double rsq = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));

        double force_divr, pair_eng;
        bool evaluated;
        if (SINGLE)
            {
            evaluated = E::eval(c0, rsq, force_divr, pair_eng);
            if (XPLOR && evaluated)
                apply_xplor(rsq, ronsq0, c0.rcutsq, force_divr, pair_eng);
            }
        else
            {
            const uint32_t tp = (uint32_t)typei * a.ntypes + (uint32_t)typej;
            const typename E::Coeff c = s_coeff[tp];
            evaluated = E::eval(c, rsq, force_divr, pair_eng);
            if (XPLOR && evaluated)
                apply_xplor(rsq, s_ronsq[tp], c.rcutsq, force_divr, pair_eng);
            }
        // E::eval returns force_divr = pair_eng = 0 when not evaluated
        fx = __builtin_fma(dx, force_divr, fx);
        fy = __builtin_fma(dy, force_divr, fy);
        fz = __builtin_fma(dz, force_divr, fz);
        pe += pair_eng;
        if (VIRIAL)
            {
            const double fxx = force_divr * dx, fyy = force_divr * dy;
            v[0] = __builtin_fma(fxx, dx, v[0]);
            v[1] = __builtin_fma(fxx, dy, v[1]);
            v[2] = __builtin_fma(fxx, dz, v[2]);
            v[3] = __builtin_fma(fyy, dy, v[3]);
            v[4] = __builtin_fma(fyy, dz, v[4]);
            v[5] = __builtin_fma(force_divr * dz, dz, v[5]);
            }
        k = kn;
        j = jn;
        }
    }

template<class E, int TPP, bool VIRIAL, bool SINGLE, bool XPLOR>
__global__ void __launch_bounds__(256) pair_forces_kernel(const PairKArgs a, const typename E::Params* __restrict__ params)
    {
    typedef typename E::Coeff Coeff;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    Coeff* s_coeff = reinterpret_cast<Coeff*>(s_raw);
    double* s_ronsq = reinterpret_cast<double*>(s_raw + sizeof(Coeff) * (SINGLE ? 0 : a.ntypes * a.ntypes));

    Coeff c0;
    double ronsq0 = 0.0;
    if (SINGLE)
        {
        c0 = prepare_coeff<E>(a, params, 0);
        if (XPLOR)
            ronsq0 = a.ronsq[0];
        }
    else
        {
        const uint32_t ntp = a.ntypes * a.ntypes;
        for (uint32_t t = threadIdx.x; t < ntp; t += blockDim.x)
            {
            s_coeff[t] = prepare_coeff<E>(a, params, t);
            s_ronsq[t] = XPLOR ? a.ronsq[t] : 0.0;
            }
        __syncthreads();
        }

    const uint32_t block = xcd_remap(blockIdx.x, a.nblocks_padded);
    const uint32_t groups_per_block = blockDim.x / TPP;
    const uint32_t idx = a.first + block * groups_per_block + threadIdx.x / TPP;
    const uint32_t sub = threadIdx.x % TPP;
    const bool active = idx < a.end;

    uint32_t n = 0;
    uint64_t head = 0;
    double3 pi = make_double3(0.0, 0.0, 0.0);
    int typei = 0;
    if (active)
        {
        n = a.n_neigh[idx];
        head = a.head_list[idx];
        const double4 p = load_scalar4(a.pos, idx);
        pi = make_double3(p.x, p.y, p.z);
        typei = type_from_w(p.w);
        }

    double fx = 0.0, fy = 0.0, fz = 0.0, pe = 0.0;
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    // wave-uniform choice: can this wave skip the minimum image?
    bool wrap = true;
    if (a.r_list_max > 0.0 && !a.box.triclinic)
        {
        const bool interior = !active || is_interior(a.box, pi.x, pi.y, pi.z, a.r_list_max);
        wrap = !__all(interior);
        }
    if (wrap)
        pair_loop<E, TPP, VIRIAL, SINGLE, XPLOR, true>(a, s_coeff, s_ronsq, c0, ronsq0, sub, n, head, pi, typei, fx, fy, fz, pe, v);
    else
        pair_loop<E, TPP, VIRIAL, SINGLE, XPLOR, false>(a, s_coeff, s_ronsq, c0, ronsq0, sub, n, head, pi, typei, fx, fy, fz, pe, v);

    fx = group_sum<TPP>(fx);
    fy = group_sum<TPP>(fy);
    fz = group_sum<TPP>(fz);
    pe = group_sum<TPP>(pe);
    if (VIRIAL)
        {
#pragma unroll
        for (int c = 0; c < 6; ++c)
            v[c] = group_sum<TPP>(v[c]);
        }
    if (active && sub == 0)
        {
        store_scalar4(a.force, idx, fx, fy, fz, 0.5 * pe);
        if (VIRIAL)
            {
#pragma unroll
            for (int c = 0; c < 6; ++c)
                a.virial[(uint64_t)c * a.virial_pitch + idx] = 0.5 * v[c];
            }
        }
    }

// ---------------------------------------------------------------------------
// host-side driver
// ---------------------------------------------------------------------------

inline uint32_t choose_tpp(const azp_pair_args& args)
    {
    uint32_t tpp = args.threads_per_particle;
    if (tpp == 0)
        {
        // mean row length decides: rows shorter than ~2*TPP waste lanes in the tail
        double mean = (args.size_nlist && args.N) ? (double)args.size_nlist / (double)args.N : 64.0;
        if (mean >= 96.0) tpp = 8;
        else if (mean >= 40.0) tpp = 4;
        else if (mean >= 16.0) tpp = 2;
        else tpp = 1;
        }
    return tpp;
    }

template<class E, int TPP, bool VIRIAL, bool SINGLE, bool XPLOR>
int launch_pair_instance2(const azp_pair_args& args, const PairKArgs& k, const typename E::Params* d_params,
                          uint32_t block_size, hipStream_t stream)
    {
    PairKArgs ka = k;
    const uint32_t groups_per_block = block_size / TPP;
    uint32_t nblocks = (ka.end - ka.first + groups_per_block - 1) / groups_per_block;
    nblocks = (nblocks + 7u) & ~7u;
    ka.nblocks_padded = nblocks;
    size_t lds = 0;
    if (!SINGLE)
        lds = (sizeof(typename E::Coeff) + sizeof(double)) * (size_t)args.ntypes * args.ntypes;
    if (lds > 160 * 1024)
        return AZP_ERROR_TOO_MANY_TYPES;
    auto kern = pair_forces_kernel<E, TPP, VIRIAL, SINGLE, XPLOR>;
    if (lds > 64 * 1024)
        {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return (int)e;
        }
    LaunchInfo& li = last_launch();
    li.block_size = block_size; li.tpp = TPP; li.grid = nblocks; li.lds_bytes = (uint32_t)lds;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(block_size), lds, stream, ka, d_params);
    return (int)hipGetLastError();
    }

template<class E, int TPP, bool VIRIAL, bool SINGLE>
int launch_pair_instance(const azp_pair_args& args, const PairKArgs& k, const typename E::Params* d_params,
                         uint32_t block_size, hipStream_t stream)
    {
    if (args.shift_mode == AZP_SHIFT_XPLOR)
        return launch_pair_instance2<E, TPP, VIRIAL, SINGLE, true>(args, k, d_params, block_size, stream);
    return launch_pair_instance2<E, TPP, VIRIAL, SINGLE, false>(args, k, d_params, block_size, stream);
    }

template<class E, bool VIRIAL, bool SINGLE>
int launch_pair_tpp(const azp_pair_args& args, const PairKArgs& k, const typename E::Params* d_params, uint32_t tpp,
                    uint32_t block_size, hipStream_t stream)
    {
    switch (tpp)
        {
    case 1: return launch_pair_instance<E, 1, VIRIAL, SINGLE>(args, k, d_params, block_size, stream);
    case 2: return launch_pair_instance<E, 2, VIRIAL, SINGLE>(args, k, d_params, block_size, stream);
    case 4: return launch_pair_instance<E, 4, VIRIAL, SINGLE>(args, k, d_params, block_size, stream);
    case 8: return launch_pair_instance<E, 8, VIRIAL, SINGLE>(args, k, d_params, block_size, stream);
    case 16: return launch_pair_instance<E, 16, VIRIAL, SINGLE>(args, k, d_params, block_size, stream);
    case 32: return launch_pair_instance<E, 32, VIRIAL, SINGLE>(args, k, d_params, block_size, stream);
    default: return AZP_ERROR_INVALID_ARGUMENT;
        }
    }

inline int validate_pair_args(const azp_pair_args* args, const void* d_params)
    {
    if (!args || !d_params) return AZP_ERROR_INVALID_ARGUMENT;
    if (args->N == 0) return 1; // nothing to do (caller returns success)
    if (!args->d_force || !args->d_pos || !args->d_n_neigh || !args->d_nlist || !args->d_head_list || !args->d_rcutsq)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (args->ntypes == 0 || args->shift_mode > AZP_SHIFT_XPLOR) return AZP_ERROR_INVALID_ARGUMENT;
    if (args->shift_mode == AZP_SHIFT_XPLOR && !args->d_ronsq) return AZP_ERROR_INVALID_ARGUMENT;
    if (args->compute_virial && (!args->d_virial || args->virial_pitch < args->N)) return AZP_ERROR_INVALID_ARGUMENT;
    if (args->n_max < args->N) return AZP_ERROR_INVALID_ARGUMENT;
    if (args->block_size && (args->block_size % 64 || args->block_size > 256)) return AZP_ERROR_INVALID_ARGUMENT;
    if (args->range_count && (uint64_t)args->range_first + args->range_count > args->N) return AZP_ERROR_INVALID_ARGUMENT;
    return 0;
    }

inline PairKArgs make_pair_kargs(const azp_pair_args& args)
    {
    PairKArgs k;
    k.force = args.d_force;
    k.virial = args.d_virial;
    k.virial_pitch = args.virial_pitch;
    k.pos = args.d_pos;
    k.n_neigh = args.d_n_neigh;
    k.nlist = args.d_nlist;
    k.head_list = args.d_head_list;
    k.rcutsq = args.d_rcutsq;
    k.ronsq = args.d_ronsq;
    k.box = make_box_dev(args.box);
    k.r_list_max = args.r_list_max;
    k.N = args.N;
    k.ntypes = args.ntypes;
    k.shift_mode = args.shift_mode;
    k.nblocks_padded = 0;
    k.first = args.range_count ? args.range_first : 0u;
    k.end = args.range_count ? args.range_first + args.range_count : args.N;
    return k;
    }

template<class E> int launch_pair(const azp_pair_args* args, const typename E::Params* d_params, void* stream)
    {
    const int bad = validate_pair_args(args, d_params);
    if (bad < 0) return bad;
    if (bad > 0) return AZP_SUCCESS;
    const PairKArgs k = make_pair_kargs(*args);
    const uint32_t tpp = choose_tpp(*args);
    const uint32_t bs = args->block_size ? args->block_size : 256u;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool single = (args->ntypes == 1);
    if (args->compute_virial)
        return single ? launch_pair_tpp<E, true, true>(*args, k, d_params, tpp, bs, s)
                      : launch_pair_tpp<E, true, false>(*args, k, d_params, tpp, bs, s);
    return single ? launch_pair_tpp<E, false, true>(*args, k, d_params, tpp, bs, s)
                  : launch_pair_tpp<E, false, false>(*args, k, d_params, tpp, bs, s);
    }

} // namespace azp
