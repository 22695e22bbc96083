// aniso_forces.hip -- anisotropic pair force + torque for the two-patch Morse
// potential. Replaces HOOMD's
// gpu_compute_pair_aniso_forces<AnisoPairEvaluatorTwoPatchMorse>, requested by
// the reference at src/AnisoPotentialPairGPUKernel.cu.inc:21-25; per-pair
// arithmetic restated from src/AnisoPairEvaluatorTwoPatchMorse.h:127-216.
//
// Lane mapping as pair_kernel.hpp. Per neighbor: position (32 B) and
// quaternion (32 B) gathers; the patch director n = rotate(q, x^) of particle i
// is computed once per particle, that of j once per pair. Outputs: force
// (fx, fy, fz, e) and torque (tx, ty, tz, 0), both N x 4.
#include "pair_auto.hpp"
#include "xtiled.hpp"

namespace azp
{
struct TPMCoeff
    {
    double rcutsq, M_d, M_rinv, r_eq, omega, alpha, U_shift; // U_shift = U_Morse(r_cut) when mode == shift
    int repulsion, _pad;
    };

struct AnisoKArgs
    {
    PairKArgs p;
    const double* orientation;
    double* torque;
    };

__device__ __forceinline__ TPMCoeff tpm_prepare(const azp_tpm_params& p, double rcutsq, bool energy_shift)
    {
    TPMCoeff c;
    c.rcutsq = rcutsq;
    c.M_d = p.M_d;
    c.M_rinv = p.M_rinv;
    c.r_eq = p.r_eq;
    c.omega = p.omega;
    c.alpha = p.alpha;
    c.repulsion = p.repulsion ? 1 : 0;
    c._pad = 0;
    c.U_shift = 0.0;
    if (energy_shift)
        {
        const double rcut = sqrt(rcutsq);
        const double ex = exp(-(rcut - p.r_eq) * p.M_rinv);
        const double om = 1.0 - ex;
        c.U_shift = p.M_d * (om * om - 1.0);
        }
    return c;
    }

// rotate(q, (1,0,0)) with q = (s, u): (s^2 - |u|^2) x^ + 2 s (u x x^) + 2 u_x u
__device__ __forceinline__ double3 patch_director(const double4& q)
    {
    const double s = q.x, ux = q.y, uy = q.z, uz = q.w;
    const double c = s * s - (ux * ux + uy * uy + uz * uz);
    return make_double3(c + 2.0 * ux * ux, 2.0 * s * uz + 2.0 * ux * uy, -2.0 * s * uy + 2.0 * ux * uz);
    }
__device__ __forceinline__ double3 cross3(const double3& a, const double3& b)
    {
    return make_double3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
    }

// One TwoPatchMorse pair (src/AnisoPairEvaluatorTwoPatchMorse.h:127-215): force on i, the torque on i (before its
// accumulation), the pair energy. U = (U_Morse(r) - U_shift) Omega(gamma_i) Omega(gamma_j), gamma = u . n, u = dr / r,
// Omega(g) = 1 / (1 + exp(-omega (g^2 - alpha))). Written for the FP64 issue rate of gfx950: one reciprocal square
// root gives r and 1 / r; the reciprocals of the two sigmoid denominators are Newton-refined v_rcp_f64; and the
// perpendicular directors n_perp = -u x (u x n) = n - gamma u are never formed -- the force is assembled as
// F = a u + b_i n_i + b_j n_j with three scalar coefficients (9 FMAs instead of two double cross products).
__device__ __forceinline__ void tpm_pair(const TPMCoeff& c, const double3& n_i, const double3& n_j, double dx, double dy, double dz,
                                         double rsq, double (&F)[3], double (&T)[3], double& e)
    {
    const double rinv = fast_rsqrt(rsq);
    const double r = rsq * rinv;
    const double ux = dx * rinv, uy = dy * rinv, uz = dz * rinv;
    double UMorse = -c.M_d;
    double dUMorse_dr = 0.0;
    if (r > c.r_eq || c.repulsion)
        {
        const double Morse_exp = exp(-(r - c.r_eq) * c.M_rinv);
        const double one_minus_exp = 1.0 - Morse_exp;
        UMorse = c.M_d * __builtin_fma(one_minus_exp, one_minus_exp, -1.0);
        dUMorse_dr = 2.0 * c.M_d * c.M_rinv * Morse_exp * one_minus_exp;
        }
    const double gi = __builtin_fma(uz, n_i.z, __builtin_fma(uy, n_i.y, ux * n_i.x));
    const double gj = __builtin_fma(uz, n_j.z, __builtin_fma(uy, n_j.y, ux * n_j.x));
    const double ei = exp(-c.omega * __builtin_fma(gi, gi, -c.alpha));
    const double ej = exp(-c.omega * __builtin_fma(gj, gj, -c.alpha));
    const double Oi = fast_rcp(1.0 + ei), Oj = fast_rcp(1.0 + ej);
    const double OO = Oi * Oj;
    e = (UMorse - c.U_shift) * OO;
    const double w2 = 2.0 * c.omega * UMorse * OO;   // dU/dgamma_i = w2 gamma_i e_i Omega_i, likewise j
    const double dU_dgi = w2 * gi * ei * Oi;
    const double dU_dgj = w2 * gj * ej * Oj;
    const double bi = -rinv * dU_dgi, bj = -rinv * dU_dgj;
    const double a = -__builtin_fma(bi, gi, __builtin_fma(bj, gj, dUMorse_dr * OO));
    F[0] = __builtin_fma(a, ux, __builtin_fma(bi, n_i.x, bj * n_j.x));
    F[1] = __builtin_fma(a, uy, __builtin_fma(bi, n_i.y, bj * n_j.y));
    F[2] = __builtin_fma(a, uz, __builtin_fma(bi, n_i.z, bj * n_j.z));
    // torque on i: dU/dgamma_i (u x n_i)
    T[0] = dU_dgi * __builtin_fma(uy, n_i.z, -uz * n_i.y);
    T[1] = dU_dgi * __builtin_fma(uz, n_i.x, -ux * n_i.z);
    T[2] = dU_dgi * __builtin_fma(ux, n_i.y, -uy * n_i.x);
    }

template<int TPP, bool VIRIAL, bool SINGLE, bool WRAP>
__device__ __forceinline__ void aniso_loop(const AnisoKArgs& a, const TPMCoeff* __restrict__ s_coeff,
                                           const TPMCoeff& c0, uint32_t sub, uint32_t n, uint64_t head, double3 pi,
                                           double3 n_i, int typei, double (&f)[3], double (&t)[3], double& pe,
                                           double (&v)[6])
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
        TPMCoeff c;
        if (SINGLE)
            c = c0;
        else
            c = s_coeff[(uint32_t)typei * a.p.ntypes + (uint32_t)type_from_w(pj.w)];
        if (!(rsq > c.rcutsq)) // reference returns early on rsq > rcutsq (:135-136)
            {
            const double4 qj = load_scalar4(a.orientation, j);
            const double3 n_j = patch_director(qj);
            double F[3], T[3], e;
            tpm_pair(c, n_i, n_j, dx, dy, dz, rsq, F, T, e);
            const double Fx = F[0], Fy = F[1], Fz = F[2];
            f[0] += Fx; f[1] += Fy; f[2] += Fz;
            t[0] += T[0]; t[1] += T[1]; t[2] += T[2];
            pe += e;
            if (VIRIAL)
                {
                v[0] = __builtin_fma(dx, Fx, v[0]);
                v[1] = __builtin_fma(dy, Fx, v[1]);
                v[2] = __builtin_fma(dz, Fx, v[2]);
                v[3] = __builtin_fma(dy, Fy, v[3]);
                v[4] = __builtin_fma(dz, Fy, v[4]);
                v[5] = __builtin_fma(dz, Fz, v[5]);
                }
            }
        k = kn;
        j = jn;
        }
    }

template<int TPP, bool VIRIAL, bool SINGLE>
__global__ void __launch_bounds__(256) aniso_forces_kernel(const AnisoKArgs a, const azp_tpm_params* __restrict__ params)
    {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    TPMCoeff* s_coeff = reinterpret_cast<TPMCoeff*>(s_raw);
    const bool energy_shift = (a.p.shift_mode == AZP_SHIFT_SHIFT);
    TPMCoeff c0;
    if (SINGLE)
        c0 = tpm_prepare(params[0], a.p.rcutsq[0], energy_shift);
    else
        {
        const uint32_t ntp = a.p.ntypes * a.p.ntypes;
        for (uint32_t t = threadIdx.x; t < ntp; t += blockDim.x)
            s_coeff[t] = tpm_prepare(params[t], a.p.rcutsq[t], energy_shift);
        __syncthreads();
        }

    const uint32_t block = xcd_remap(blockIdx.x, a.p.nblocks_padded);
    const uint32_t idx = a.p.first + block * (blockDim.x / TPP) + threadIdx.x / TPP;
    const uint32_t sub = threadIdx.x % TPP;
    const bool active = idx < a.p.end;

    uint32_t n = 0;
    uint64_t head = 0;
    double3 pi = make_double3(0.0, 0.0, 0.0), n_i = make_double3(1.0, 0.0, 0.0);
    int typei = 0;
    if (active)
        {
        n = a.p.n_neigh[idx];
        head = a.p.head_list[idx];
        const double4 p = load_scalar4(a.p.pos, idx);
        pi = make_double3(p.x, p.y, p.z);
        typei = type_from_w(p.w);
        n_i = patch_director(load_scalar4(a.orientation, idx));
        }
    double f[3] = {0.0, 0.0, 0.0}, t[3] = {0.0, 0.0, 0.0}, pe = 0.0;
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    bool wrap = true;
    if (a.p.r_list_max > 0.0 && !a.p.box.triclinic)
        {
        const bool interior = !active || is_interior(a.p.box, pi.x, pi.y, pi.z, a.p.r_list_max);
        wrap = !__all(interior);
        }
    if (wrap)
        aniso_loop<TPP, VIRIAL, SINGLE, true>(a, s_coeff, c0, sub, n, head, pi, n_i, typei, f, t, pe, v);
    else
        aniso_loop<TPP, VIRIAL, SINGLE, false>(a, s_coeff, c0, sub, n, head, pi, n_i, typei, f, t, pe, v);

#pragma unroll
    for (int c = 0; c < 3; ++c)
        {
        f[c] = group_sum<TPP>(f[c]);
        t[c] = group_sum<TPP>(t[c]);
        }
    pe = group_sum<TPP>(pe);
    if (VIRIAL)
        {
#pragma unroll
        for (int c = 0; c < 6; ++c)
            v[c] = group_sum<TPP>(v[c]);
        }
    if (active && sub == 0)
        {
        store_scalar4(a.p.force, idx, f[0], f[1], f[2], 0.5 * pe);
        store_scalar4(a.torque, idx, t[0], t[1], t[2], 0.0);
        if (VIRIAL)
            {
#pragma unroll
            for (int c = 0; c < 6; ++c)
                a.p.virial[(uint64_t)c * a.p.virial_pitch + idx] = 0.5 * v[c];
            }
        }
    }

template<int TPP, bool VIRIAL, bool SINGLE>
static int launch_aniso_instance(const azp_aniso_args& args, AnisoKArgs k, const azp_tpm_params* d_params, uint32_t bs,
                                 hipStream_t stream)
    {
    const uint32_t groups = bs / TPP;
    uint32_t nblocks = (k.p.end - k.p.first + groups - 1) / groups;
    nblocks = (nblocks + 7u) & ~7u;
    k.p.nblocks_padded = nblocks;
    size_t lds = SINGLE ? 0 : sizeof(TPMCoeff) * (size_t)args.pair.ntypes * args.pair.ntypes;
    if (lds > 160 * 1024)
        return AZP_ERROR_TOO_MANY_TYPES;
    auto kern = aniso_forces_kernel<TPP, VIRIAL, SINGLE>;
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
static int launch_aniso_tpp(const azp_aniso_args& args, const AnisoKArgs& k, const azp_tpm_params* d_params,
                            uint32_t tpp, uint32_t bs, hipStream_t stream)
    {
    switch (tpp)
        {
    case 1: return launch_aniso_instance<1, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 2: return launch_aniso_instance<2, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 4: return launch_aniso_instance<4, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 8: return launch_aniso_instance<8, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 16: return launch_aniso_instance<16, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    case 32: return launch_aniso_instance<32, VIRIAL, SINGLE>(args, k, d_params, bs, stream);
    default: return AZP_ERROR_INVALID_ARGUMENT;
        }
    }

// ---- tile-staged form (xtiled.hpp): the patch director n_j = rotate(q_j, x^) of every
// staged particle is computed ONCE per tile and kept in LDS next to its position (the
// reference rotates per pair, src/AnisoPairEvaluatorTwoPatchMorse.h:145-146); same per-pair
// arithmetic as aniso_loop above ----
struct XTPM
    {
    typedef azp_tpm_params Params;
    typedef TPMCoeff Coeff;
    struct KExtra
        {
        const double* orientation;
        double* torque;
        };
    static constexpr int kExtra = 3; // n_j
    static constexpr bool kTag = false;
    static constexpr int kMinWaves = 3; // 48 B per slot: three workgroups per CU at 1,024 slots, <= 168 VGPRs
    struct Own
        {
        double3 n;
        };
    struct Acc
        {
        double f[3], t[3], pe;
        };
    static __device__ __forceinline__ Coeff prepare(const Params& p, double rcutsq, const KExtra&, uint32_t shift_mode)
        {
        return tpm_prepare(p, rcutsq, shift_mode == AZP_SHIFT_SHIFT);
        }
    static __device__ __forceinline__ void load_extra(const KExtra& x, uint32_t j, double (&e)[3], uint32_t&)
        {
        const double3 n = patch_director(load_scalar4(x.orientation, j));
        e[0] = n.x; e[1] = n.y; e[2] = n.z;
        }
    static __device__ __forceinline__ void load_own(const KExtra& x, uint32_t idx, Own& o)
        {
        o.n = patch_director(load_scalar4(x.orientation, idx));
        }
    static __device__ __forceinline__ void zero(Acc& a)
        {
        a.f[0] = a.f[1] = a.f[2] = a.t[0] = a.t[1] = a.t[2] = a.pe = 0.0;
        }
    static __device__ __forceinline__ bool in_range(const Coeff& c, double rsq) { return !(rsq > c.rcutsq); } // (:135-136)
    template<bool VIRIAL>
    static __device__ __forceinline__ void pair(const Coeff& c, const KExtra&, const Own& o, double dx, double dy, double dz, double rsq,
                                                const double (&nj)[3], uint32_t, Acc& a, double (&v)[6])
        {
        const double3 n_i = o.n;
        const double3 n_j = make_double3(nj[0], nj[1], nj[2]);
        double F[3], T[3], e;
        tpm_pair(c, n_i, n_j, dx, dy, dz, rsq, F, T, e);
        const double Fx = F[0], Fy = F[1], Fz = F[2];
        a.f[0] += Fx; a.f[1] += Fy; a.f[2] += Fz;
        a.t[0] += T[0]; a.t[1] += T[1]; a.t[2] += T[2];
        a.pe += e;
        if (VIRIAL)
            {
            v[0] = __builtin_fma(dx, Fx, v[0]);
            v[1] = __builtin_fma(dy, Fx, v[1]);
            v[2] = __builtin_fma(dz, Fx, v[2]);
            v[3] = __builtin_fma(dy, Fy, v[3]);
            v[4] = __builtin_fma(dz, Fy, v[4]);
            v[5] = __builtin_fma(dz, Fz, v[5]);
            }
        }
    static __device__ __forceinline__ void store(const Acc& a, const PairKArgs& p, const KExtra& x, uint32_t idx)
        {
        store_scalar4(p.force, idx, a.f[0], a.f[1], a.f[2], 0.5 * a.pe);
        store_scalar4(x.torque, idx, a.t[0], a.t[1], a.t[2], 0.0);
        }
    };
} // namespace azp

static int aniso_generic(const azp_aniso_args* args, const azp_tpm_params* d_params, void* stream);

extern "C" int azp_aniso_forces_planned_two_patch_morse(azp_pair_plan* plan_, const azp_aniso_args* args, const azp_tpm_params* d_params,
                                                        void* stream)
    {
    using namespace azp;
    if (!plan_ || !args)
        return AZP_ERROR_INVALID_ARGUMENT;
    const int bad = validate_pair_args(&args->pair, d_params);
    if (bad < 0) return bad;
    if (bad > 0) return AZP_SUCCESS;
    if (!args->d_orientation || !args->d_torque || args->pair.shift_mode == AZP_SHIFT_XPLOR)
        return AZP_ERROR_INVALID_ARGUMENT; // HOOMD aniso pairs accept "none" / "shift"
    const PairPlan& plan = *reinterpret_cast<const PairPlan*>(plan_);
    if (plan.builds == 0 || plan.N != args->pair.N || plan.nlist_ptr != args->pair.d_nlist || plan.head_ptr != args->pair.d_head_list)
        return AZP_ERROR_INVALID_ARGUMENT; // a plan compiled from a different list is a caller bug
    if (!xtiled_usable(plan, args->pair))
        return plan.from_cells ? AZP_ERROR_INVALID_ARGUMENT : aniso_generic(args, d_params, stream);
    XTPM::KExtra x;
    x.orientation = args->d_orientation;
    x.torque = args->d_torque;
    return launch_xtiled<XTPM>(plan, args->pair, x, d_params, static_cast<hipStream_t>(stream));
    }

// what gpu_compute_pair_aniso_forces<E> forwards to (src/AnisoPotentialPairGPUKernel.cu.inc:21-25): the tile-staged
// kernel from libazp's own plan cache (pair_auto.hpp) unless the caller asks for the generic kernel
extern "C" int azp_aniso_forces_two_patch_morse(const azp_aniso_args* args, const azp_tpm_params* d_params,
                                                void* stream)
    {
    using namespace azp;
    if (!args)
        return AZP_ERROR_INVALID_ARGUMENT;
    const int bad = validate_pair_args(&args->pair, d_params);
    if (bad < 0) return bad;
    if (bad > 0) return AZP_SUCCESS;
    if (!args->d_orientation || !args->d_torque || args->pair.shift_mode == AZP_SHIFT_XPLOR)
        return AZP_ERROR_INVALID_ARGUMENT; // HOOMD aniso pairs accept "none" / "shift"
    if (!auto_plan_wanted(args->pair))
        return aniso_generic(args, d_params, stream);
    XTPM::KExtra x;
    x.orientation = args->d_orientation;
    x.torque = args->d_torque;
    return auto_plan_run(
        args->pair, true, static_cast<hipStream_t>(stream),
        [&](const AutoLaunch& l)
            {
            if (!xtiled_usable(*l.plan, *l.args))
                return aniso_generic(args, d_params, stream);
            return launch_xtiled<XTPM>(*l.plan, *l.args, x, d_params, static_cast<hipStream_t>(stream), l.dyn);
            },
        [&]() { return aniso_generic(args, d_params, stream); });
    }

// the generic kernel (arguments validated by the caller)
static int aniso_generic(const azp_aniso_args* args, const azp_tpm_params* d_params, void* stream)
    {
    using namespace azp;
    AnisoKArgs k;
    k.p = make_pair_kargs(args->pair);
    k.orientation = args->d_orientation;
    k.torque = args->d_torque;
    const uint32_t tpp = choose_tpp(args->pair);
    const uint32_t bs = args->pair.block_size ? args->pair.block_size : 256u;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool single = (args->pair.ntypes == 1);
    if (args->pair.compute_virial)
        return single ? launch_aniso_tpp<true, true>(*args, k, d_params, tpp, bs, s)
                      : launch_aniso_tpp<true, false>(*args, k, d_params, tpp, bs, s);
    return single ? launch_aniso_tpp<false, true>(*args, k, d_params, tpp, bs, s)
                  : launch_aniso_tpp<false, false>(*args, k, d_params, tpp, bs, s);
    }
