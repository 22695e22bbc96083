// dpd_forces.hip -- DPD thermostat pair force (conservative + drag + random),
// generalized weight (1 - r/rc)^s. Replaces HOOMD's
// gpu_compute_dpd_forces<DPDPairEvaluatorGeneralWeight>, requested by the
// reference at src/PotentialPairDPDThermoGPUKernel.cu.inc:21-24; per-pair
// arithmetic restated from src/DPDPairEvaluatorGeneralWeight.h:198-255.
//
// Same lane mapping as pair_kernel.hpp (TPP lanes per particle, DPP reduce).
// Extra per-neighbor gathers: velocity (24 B) and tag (4 B). One Philox4x32-10
// block per in-range pair, computed in registers, keyed by
// (seed, timestep, min tag, max tag) so both owners of a pair draw the same
// number (cross-rank consistency relies on this: :213-231).
#include "pair_auto.hpp"
#include "xtiled.hpp"

namespace azp
{
struct DPDCoeff
    {
    double rcutsq, A, gamma, half_s, rcut, rcutinv, noise; // noise = rsqrt(dt / (6 kT gamma))
    };

struct DPDKArgs
    {
    PairKArgs p;
    const double* vel;
    const uint32_t* tag;
    uint64_t timestep;
    double deltaT;
    double T;
    uint32_t seed;
    uint32_t _pad;
    };

__device__ __forceinline__ DPDCoeff dpd_prepare(const azp_dpd_params& p, double rcutsq, double deltaT, double T)
    {
    DPDCoeff c;
    c.rcutsq = rcutsq;
    c.A = p.A;
    c.gamma = p.gamma;
    c.half_s = 0.5 * p.s;
    c.rcutinv = 1.0 / sqrt(rcutsq);
    c.rcut = 1.0 / c.rcutinv;
    // fast::rsqrt(m_deltaT / (m_T * gamma * 6)) (:246); T == 0 -> rsqrt(inf) = 0
    c.noise = 1.0 / sqrt(deltaT / (T * p.gamma * 6.0));
    return c;
    }

// (1 - r/rc)^(s/2): the common exponents avoid the generic pow()
__device__ __forceinline__ double weight_pow(double x, double half_s)
    {
    if (half_s == 1.0) return x;
    if (half_s == 0.5) return fast_sqrt(x);
    if (half_s == 0.25) return fast_sqrt(fast_sqrt(x));
    return pow(x, half_s);
    }

template<int TPP, bool VIRIAL, bool SINGLE, bool WRAP>
__device__ __forceinline__ void dpd_loop(const DPDKArgs& a, const DPDCoeff* __restrict__ s_coeff, const DPDCoeff& c0,
                                         uint32_t sub, uint32_t n, uint64_t head, double3 pi, double3 vi, int typei,
                                         uint32_t tagi, double& fx, double& fy, double& fz, double& pe, double (&v)[6])
    {
    const uint32_t* __restrict__ row = a.p.nlist + head;
    uint32_t k = sub;
    uint32_t j = (k < n) ? row[k] : 0u;
    while (k < n)
        {
        const uint32_t kn = k + TPP;
        const uint32_t jn = (kn < n) ? row[kn] : 0u;
        const double4 pj = load_scalar4(a.p.pos, j);
        double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        if (WRAP)
            min_image(a.p.box, dx, dy, dz);
        const double rsq = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
        DPDCoeff c;
        if (SINGLE)
            c = c0;
        else
            c = s_coeff[(uint32_t)typei * a.p.ntypes + (uint32_t)type_from_w(pj.w)];
        if (rsq < c.rcutsq)
            {
            const double3 vj = load_scalar3_of4(a.vel, j);
            const uint32_t tagj = a.tag[j];
            const double rdotv = dx * (vi.x - vj.x) + dy * (vi.y - vj.y) + dz * (vi.z - vj.z);
            const double alpha = dpd_alpha((uint16_t)a.seed, tagi, tagj, a.timestep);
            const double rinv = fast_rsqrt(rsq);
            const double r = rsq * rinv;
            const double force_divr_cons = c.A * (rinv - c.rcutinv);
            const double wR = weight_pow(1.0 - r * c.rcutinv, c.half_s) * rinv;
            double force_divr = force_divr_cons - c.gamma * wR * wR * rdotv;
            force_divr += c.noise * wR * alpha;
            const double pair_eng = c.A * (c.rcut - r) - 0.5 * c.A * c.rcutinv * (c.rcutsq - rsq);
            fx = __builtin_fma(dx, force_divr, fx);
            fy = __builtin_fma(dy, force_divr, fy);
            fz = __builtin_fma(dz, force_divr, fz);
            pe += pair_eng;
            if (VIRIAL)
                {
                // virial from the conservative part only (:193-194)
                const double fxx = force_divr_cons * dx, fyy = force_divr_cons * dy;
                v[0] = __builtin_fma(fxx, dx, v[0]);
                v[1] = __builtin_fma(fxx, dy, v[1]);
                v[2] = __builtin_fma(fxx, dz, v[2]);
                v[3] = __builtin_fma(fyy, dy, v[3]);
                v[4] = __builtin_fma(fyy, dz, v[4]);
                v[5] = __builtin_fma(force_divr_cons * dz, dz, v[5]);
                }
            }
        k = kn;
        j = jn;
        }
    }

template<int TPP, bool VIRIAL, bool SINGLE>
__global__ void __launch_bounds__(256) dpd_forces_kernel(const DPDKArgs a, const azp_dpd_params* __restrict__ params)
    {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    DPDCoeff* s_coeff = reinterpret_cast<DPDCoeff*>(s_raw);
    DPDCoeff c0;
    if (SINGLE)
        c0 = dpd_prepare(params[0], a.p.rcutsq[0], a.deltaT, a.T);
    else
        {
        const uint32_t ntp = a.p.ntypes * a.p.ntypes;
        for (uint32_t t = threadIdx.x; t < ntp; t += blockDim.x)
            s_coeff[t] = dpd_prepare(params[t], a.p.rcutsq[t], a.deltaT, a.T);
        __syncthreads();
        }

    const uint32_t block = xcd_remap(blockIdx.x, a.p.nblocks_padded);
    const uint32_t idx = a.p.first + block * (blockDim.x / TPP) + threadIdx.x / TPP;
    const uint32_t sub = threadIdx.x % TPP;
    const bool active = idx < a.p.end;

    uint32_t n = 0, tagi = 0;
    uint64_t head = 0;
    double3 pi = make_double3(0.0, 0.0, 0.0), vi = make_double3(0.0, 0.0, 0.0);
    int typei = 0;
    if (active)
        {
        n = a.p.n_neigh[idx];
        head = a.p.head_list[idx];
        const double4 p = load_scalar4(a.p.pos, idx);
        pi = make_double3(p.x, p.y, p.z);
        typei = type_from_w(p.w);
        vi = load_scalar3_of4(a.vel, idx);
        tagi = a.tag[idx];
        }
    double fx = 0.0, fy = 0.0, fz = 0.0, pe = 0.0;
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    bool wrap = true;
    if (a.p.r_list_max > 0.0 && !a.p.box.triclinic)
        {
        const bool interior = !active || is_interior(a.p.box, pi.x, pi.y, pi.z, a.p.r_list_max);
        wrap = !__all(interior);
        }
    if (wrap)
        dpd_loop<TPP, VIRIAL, SINGLE, true>(a, s_coeff, c0, sub, n, head, pi, vi, typei, tagi, fx, fy, fz, pe, v);
    else
        dpd_loop<TPP, VIRIAL, SINGLE, false>(a, s_coeff, c0, sub, n, head, pi, vi, typei, tagi, fx, fy, fz, pe, v);

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
        store_scalar4(a.p.force, idx, fx, fy, fz, 0.5 * pe);
        if (VIRIAL)
            {
#pragma unroll
            for (int c = 0; c < 6; ++c)
                a.p.virial[(uint64_t)c * a.p.virial_pitch + idx] = 0.5 * v[c];
            }
        }
    }

template<int TPP, bool VIRIAL, bool SINGLE>
static int launch_dpd_instance(const azp_dpd_args& args, DPDKArgs k, const azp_dpd_params* d_params, uint32_t bs,
                               hipStream_t stream)
    {
    const uint32_t groups = bs / TPP;
    uint32_t nblocks = (k.p.end - k.p.first + groups - 1) / groups;
    nblocks = (nblocks + 7u) & ~7u;
    k.p.nblocks_padded = nblocks;
    size_t lds = SINGLE ? 0 : sizeof(DPDCoeff) * (size_t)args.pair.ntypes * args.pair.ntypes;
    if (lds > 160 * 1024)
        return AZP_ERROR_TOO_MANY_TYPES;
    auto kern = dpd_forces_kernel<TPP, VIRIAL, SINGLE>;
    if (lds > 64 * 1024)
        {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return (int)e;
        }
    LaunchInfo& li = last_launch();
    li.block_size = bs; li.tpp = TPP; li.grid = nblocks; li.lds_bytes = (uint32_t)lds;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(bs), lds, stream, k, d_params);
    return (int)hipGetLastError();
    }

template<bool VIRIAL, bool SINGLE>
static int launch_dpd_tpp(const azp_dpd_args& args, const DPDKArgs& k, const azp_dpd_params* d_params, uint32_t tpp,
                          uint32_t bs, hipStream_t stream)
    {
    switch (tpp)
        {
    case 1: return launch_dpd_instance<1, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 2: return launch_dpd_instance<2, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 4: return launch_dpd_instance<4, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 8: return launch_dpd_instance<8, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 16: return launch_dpd_instance<16, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 32: return launch_dpd_instance<32, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    default: return AZP_ERROR_INVALID_ARGUMENT;
        }
    }

// ---- tile-staged form (xtiled.hpp): positions, velocities and tags of a tile's neighbors
// are staged once in LDS; same per-pair arithmetic as dpd_loop above ----
struct XDPD
    {
    typedef azp_dpd_params Params;
    typedef DPDCoeff Coeff;
    struct KExtra
        {
        const double* vel;
        const uint32_t* tag;
        uint64_t timestep;
        double deltaT, T;
        uint32_t seed, _pad;
        };
    static constexpr int kExtra = 3; // vx, vy, vz
    static constexpr bool kTag = true;
    static constexpr int kMinWaves = 2; // 52 B per slot: two workgroups per CU at 1,536 slots (80 KiB of LDS each)
    struct Own
        {
        double3 v;
        uint32_t tag;
        };
    struct Acc
        {
        double fx, fy, fz, pe;
        };
    static __device__ __forceinline__ Coeff prepare(const Params& p, double rcutsq, const KExtra& x, uint32_t)
        {
        return dpd_prepare(p, rcutsq, x.deltaT, x.T);
        }
    static __device__ __forceinline__ void load_extra(const KExtra& x, uint32_t j, double (&e)[3], uint32_t& tag)
        {
        const double3 vj = load_scalar3_of4(x.vel, j);
        e[0] = vj.x; e[1] = vj.y; e[2] = vj.z;
        tag = x.tag[j];
        }
    static __device__ __forceinline__ void load_own(const KExtra& x, uint32_t idx, Own& o)
        {
        o.v = load_scalar3_of4(x.vel, idx);
        o.tag = x.tag[idx];
        }
    static __device__ __forceinline__ void zero(Acc& a) { a.fx = a.fy = a.fz = a.pe = 0.0; }
    static __device__ __forceinline__ bool in_range(const Coeff& c, double rsq) { return rsq < c.rcutsq; }
    template<bool VIRIAL>
    static __device__ __forceinline__ void pair(const Coeff& c, const KExtra& x, const Own& o, double dx, double dy, double dz, double rsq,
                                                const double (&vj)[3], uint32_t tagj, Acc& a, double (&v)[6])
        {
        const double rdotv = dx * (o.v.x - vj[0]) + dy * (o.v.y - vj[1]) + dz * (o.v.z - vj[2]);
        const double alpha = dpd_alpha((uint16_t)x.seed, o.tag, tagj, x.timestep);
        const double rinv = fast_rsqrt(rsq);
        const double r = rsq * rinv;
        const double force_divr_cons = c.A * (rinv - c.rcutinv);
        const double wR = weight_pow(1.0 - r * c.rcutinv, c.half_s) * rinv;
        double force_divr = force_divr_cons - c.gamma * wR * wR * rdotv;
        force_divr += c.noise * wR * alpha;
        a.fx = __builtin_fma(dx, force_divr, a.fx);
        a.fy = __builtin_fma(dy, force_divr, a.fy);
        a.fz = __builtin_fma(dz, force_divr, a.fz);
        a.pe += c.A * (c.rcut - r) - 0.5 * c.A * c.rcutinv * (c.rcutsq - rsq);
        if (VIRIAL)
            {
            const double fxx = force_divr_cons * dx, fyy = force_divr_cons * dy;
            v[0] = __builtin_fma(fxx, dx, v[0]);
            v[1] = __builtin_fma(fxx, dy, v[1]);
            v[2] = __builtin_fma(fxx, dz, v[2]);
            v[3] = __builtin_fma(fyy, dy, v[3]);
            v[4] = __builtin_fma(fyy, dz, v[4]);
            v[5] = __builtin_fma(force_divr_cons * dz, dz, v[5]);
            }
        }
    static __device__ __forceinline__ void store(const Acc& a, const PairKArgs& p, const KExtra&, uint32_t idx)
        {
        store_scalar4(p.force, idx, a.fx, a.fy, a.fz, 0.5 * a.pe);
        }
    };
} // namespace azp

static int dpd_generic(const azp_dpd_args* args, const azp_dpd_params* d_params, void* stream);

static int azp_dpd_validate(const azp_dpd_args* args, const azp_dpd_params* d_params)
    {
    if (!args)
        return AZP_ERROR_INVALID_ARGUMENT;
    const int bad = azp::validate_pair_args(&args->pair, d_params);
    if (bad != 0)
        return bad;
    if (!args->d_vel || !args->d_tag || args->pair.shift_mode != AZP_SHIFT_NONE)
        return AZP_ERROR_INVALID_ARGUMENT; // DPD accepts mode "none" only (src/pair.py:215)
    return 0;
    }

extern "C" int azp_dpd_forces_planned_general_weight(azp_pair_plan* plan_, const azp_dpd_args* args, const azp_dpd_params* d_params,
                                                     void* stream)
    {
    using namespace azp;
    if (!plan_)
        return AZP_ERROR_INVALID_ARGUMENT;
    const int bad = azp_dpd_validate(args, d_params);
    if (bad < 0) return bad;
    if (bad > 0) return AZP_SUCCESS;
    const PairPlan& plan = *reinterpret_cast<const PairPlan*>(plan_);
    if (plan.builds == 0 || plan.N != args->pair.N || plan.nlist_ptr != args->pair.d_nlist || plan.head_ptr != args->pair.d_head_list)
        return AZP_ERROR_INVALID_ARGUMENT; // a plan compiled from a different list is a caller bug
    if (!xtiled_usable(plan, args->pair))
        return plan.from_cells ? AZP_ERROR_INVALID_ARGUMENT : dpd_generic(args, d_params, stream);
    XDPD::KExtra x;
    x.vel = args->d_vel; x.tag = args->d_tag; x.timestep = args->timestep; x.deltaT = args->deltaT; x.T = args->T;
    x.seed = args->seed; x._pad = 0;
    return launch_xtiled<XDPD>(plan, args->pair, x, d_params, static_cast<hipStream_t>(stream));
    }

// what gpu_compute_dpd_forces<E> forwards to (src/PotentialPairDPDThermoGPUKernel.cu.inc:21-24): the tile-staged kernel
// from libazp's own plan cache (pair_auto.hpp) unless the caller asks for the generic kernel
extern "C" int azp_dpd_forces_general_weight(const azp_dpd_args* args, const azp_dpd_params* d_params, void* stream)
    {
    using namespace azp;
    const int bad = azp_dpd_validate(args, d_params);
    if (bad < 0) return bad;
    if (bad > 0) return AZP_SUCCESS;
    if (!auto_plan_wanted(args->pair))
        return dpd_generic(args, d_params, stream);
    XDPD::KExtra x;
    x.vel = args->d_vel; x.tag = args->d_tag; x.timestep = args->timestep; x.deltaT = args->deltaT; x.T = args->T;
    x.seed = args->seed; x._pad = 0;
    return auto_plan_run(
        args->pair, true, static_cast<hipStream_t>(stream),
        [&](const AutoLaunch& l)
            {
            if (!xtiled_usable(*l.plan, *l.args))
                return dpd_generic(args, d_params, stream);
            return launch_xtiled<XDPD>(*l.plan, *l.args, x, d_params, static_cast<hipStream_t>(stream), l.dyn);
            },
        [&]() { return dpd_generic(args, d_params, stream); });
    }

static int dpd_generic(const azp_dpd_args* args, const azp_dpd_params* d_params, void* stream)
    {
    using namespace azp;
    DPDKArgs k;
    k.p = make_pair_kargs(args->pair);
    k.vel = args->d_vel;
    k.tag = args->d_tag;
    k.timestep = args->timestep;
    k.deltaT = args->deltaT;
    k.T = args->T;
    k.seed = args->seed;
    k._pad = 0;
    const uint32_t tpp = choose_tpp(args->pair);
    const uint32_t bs = args->pair.block_size ? args->pair.block_size : 256u;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool single = (args->pair.ntypes == 1);
    if (args->pair.compute_virial)
        return single ? launch_dpd_tpp<true, true>(*args, k, d_params, tpp, bs, s)
                      : launch_dpd_tpp<true, false>(*args, k, d_params, tpp, bs, s);
    return single ? launch_dpd_tpp<false, true>(*args, k, d_params, tpp, bs, s)
                  : launch_dpd_tpp<false, false>(*args, k, d_params, tpp, bs, s);
    }
