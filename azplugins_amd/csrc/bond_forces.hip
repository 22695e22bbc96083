// bond_forces.hip -- two-body bonded forces over HOOMD's per-particle GPU bond
// table. Replaces gpu_compute_bond_forces<E, 2>, requested by the reference at
// src/PotentialBondGPUKernel.cu.inc:25-29 for E = BondEvaluatorDoubleWell and
// BondEvaluatorQuartic.
//
// One lane per particle (<= a handful of bonds each, ~84 B/particle of traffic:
// a pure streaming kernel). Table columns are particle-major
// (entry b of particle i at b * pitch + i) so every table read is coalesced;
// the partner position is the only gather. Per-bond-type parameters are staged
// in LDS. An evaluator that returns false (invalid parameters) raises the
// device flag word, as HOOMD's kernel does.
#include "evaluators.hpp"
#include "pair_kernel_host.hpp"

namespace azp
{
struct BondKArgs
    {
    double* force;
    double* virial;
    uint64_t virial_pitch;
    const double* pos;
    const azp_bond_entry* bondlist;
    const uint32_t* bond_pos;
    const uint32_t* n_bonds;
    uint64_t pitch;
    BoxDev box;
    uint32_t N;
    uint32_t n_bond_types;
    uint32_t compute_virial;
    uint32_t _pad;
    };

template<class E>
__global__ void __launch_bounds__(256) bond_forces_kernel(const BondKArgs a, const typename E::Params* __restrict__ params,
                                                          unsigned int* __restrict__ d_flags)
    {
    typedef typename E::Params Params;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    Params* s_params = reinterpret_cast<Params*>(s_raw);
    for (uint32_t t = threadIdx.x; t < a.n_bond_types; t += blockDim.x)
        s_params[t] = params[t];
    __syncthreads();

    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N)
        return;
    const uint32_t nb = a.n_bonds[idx];
    const double4 p = load_scalar4(a.pos, idx);
    double fx = 0.0, fy = 0.0, fz = 0.0, pe = 0.0;
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    auto one_bond = [&](const azp_bond_entry& ent, uint32_t my_pos, const double3& q)
        {
        // dx = x_a - x_b with a the first member of the bond
        double dx, dy, dz;
        if (my_pos == 0) { dx = p.x - q.x; dy = p.y - q.y; dz = p.z - q.z; }
        else { dx = q.x - p.x; dy = q.y - p.y; dz = q.z - p.z; }
        min_image(a.box, dx, dy, dz);
        const double rsq = dx * dx + dy * dy + dz * dz;
        double force_divr, bond_eng;
        const bool evaluated = E::eval(s_params[ent.type], rsq, force_divr, bond_eng);
        if (evaluated)
            {
            const double sgn = (my_pos == 0) ? 1.0 : -1.0;
            fx += sgn * dx * force_divr;
            fy += sgn * dy * force_divr;
            fz += sgn * dz * force_divr;
            pe += 0.5 * bond_eng;
            if (a.compute_virial)
                {
                const double fd2 = 0.5 * force_divr;
                v[0] += fd2 * dx * dx; v[1] += fd2 * dx * dy; v[2] += fd2 * dx * dz;
                v[3] += fd2 * dy * dy; v[4] += fd2 * dy * dz; v[5] += fd2 * dz * dz;
                }
            }
        else
            *d_flags = 1u;
        };
    // The first BATCH table columns of every lane are loaded together, then the partner
    // positions together: two dependent round trips per particle instead of two per bond
    // (a linear chain has <= 2 bonds per bead). The table reads are coalesced (column-major);
    // the partner position is the only gather.
    constexpr uint32_t BATCH = 4;
    azp_bond_entry ent[BATCH];
    uint32_t my_pos[BATCH];
#pragma unroll
    for (uint32_t b = 0; b < BATCH; ++b)
        {
        ent[b].idx = idx; ent[b].type = 0; my_pos[b] = 0;
        if (b < nb)
            {
            ent[b] = a.bondlist[(uint64_t)b * a.pitch + idx];
            my_pos[b] = a.bond_pos[(uint64_t)b * a.pitch + idx];
            }
        }
    double3 q[BATCH];
#pragma unroll
    for (uint32_t b = 0; b < BATCH; ++b)
        q[b] = load_scalar3_of4(a.pos, ent[b].idx); // unused slots re-read the lane's own (cached) row
#pragma unroll
    for (uint32_t b = 0; b < BATCH; ++b)
        if (b < nb)
            one_bond(ent[b], my_pos[b], q[b]);
    for (uint32_t b = BATCH; b < nb; ++b)
        {
        const azp_bond_entry e = a.bondlist[(uint64_t)b * a.pitch + idx];
        const uint32_t mp = a.bond_pos[(uint64_t)b * a.pitch + idx];
        one_bond(e, mp, load_scalar3_of4(a.pos, e.idx));
        }
    store_scalar4(a.force, idx, fx, fy, fz, pe);
    if (a.compute_virial)
        {
#pragma unroll
        for (int c = 0; c < 6; ++c)
            a.virial[(uint64_t)c * a.virial_pitch + idx] = v[c];
        }
    }

template<class E>
static int launch_bond(const azp_bond_args* args, const typename E::Params* d_params, unsigned int* d_flags,
                       void* stream)
    {
    if (!args || !d_params || !d_flags)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (args->N == 0)
        return AZP_SUCCESS;
    if (!args->d_force || !args->d_pos || !args->d_gpu_bondlist || !args->d_gpu_bond_pos || !args->d_gpu_n_bonds
        || args->pitch < args->N || args->n_bond_types == 0)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (args->compute_virial && (!args->d_virial || args->virial_pitch < args->N))
        return AZP_ERROR_INVALID_ARGUMENT;
    const size_t lds = sizeof(typename E::Params) * (size_t)args->n_bond_types;
    if (lds > 64 * 1024)
        return AZP_ERROR_TOO_MANY_TYPES;
    BondKArgs k;
    k.force = args->d_force;
    k.virial = args->d_virial;
    k.virial_pitch = args->virial_pitch;
    k.pos = args->d_pos;
    k.bondlist = args->d_gpu_bondlist;
    k.bond_pos = args->d_gpu_bond_pos;
    k.n_bonds = args->d_gpu_n_bonds;
    k.pitch = args->pitch;
    k.box = make_box_dev(args->box);
    k.N = args->N;
    k.n_bond_types = args->n_bond_types;
    k.compute_virial = args->compute_virial;
    k._pad = 0;
    const uint32_t bs = args->block_size ? args->block_size : 256u;
    if (bs % 64 || bs > 256)
        return AZP_ERROR_INVALID_ARGUMENT;
    const uint32_t grid = (args->N + bs - 1) / bs;
    LaunchInfo& li = last_launch();
    li.block_size = bs; li.tpp = 1; li.grid = grid; li.lds_bytes = (uint32_t)lds;
    hipLaunchKernelGGL(bond_forces_kernel<E>, dim3(grid), dim3(bs), lds, static_cast<hipStream_t>(stream), k, d_params,
                       d_flags);
    return (int)hipGetLastError();
    }
} // namespace azp

extern "C" int azp_bond_forces_double_well(const azp_bond_args* args, const azp_dw_params* d_params,
                                           unsigned int* d_flags, void* stream)
    {
    return azp::launch_bond<azp::EvalDoubleWell>(args, d_params, d_flags, stream);
    }

extern "C" int azp_bond_forces_quartic(const azp_bond_args* args, const azp_quartic_params* d_params,
                                       unsigned int* d_flags, void* stream)
    {
    return azp::launch_bond<azp::EvalQuartic>(args, d_params, d_flags, stream);
    }
