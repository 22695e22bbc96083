// azp_hoomd_kernel_drivers.cc -- HOOMD-blue v5 kernel drivers of the azplugins
// evaluators, implemented by libazp (include/azp.h). Replaces the explicit
// instantiations stamped from src/PotentialPairGPUKernel.cu.inc:25-28,
// src/PotentialPairDPDThermoGPUKernel.cu.inc:21-24,
// src/AnisoPotentialPairGPUKernel.cu.inc:21-25, src/PotentialBondGPUKernel.cu.inc:25-29.
//
// NOT COMPILED IN THIS REPOSITORY (HOOMD headers absent); see adapter/README.md.
#include "hoomd/md/AnisoPotentialPairGPU.cuh"
#include "hoomd/md/PotentialBondGPU.cuh"
#include "hoomd/md/PotentialPairDPDThermoGPU.cuh"
#include "hoomd/md/PotentialPairGPU.cuh"

#include "AnisoPairEvaluatorTwoPatchMorse.h"
#include "BondEvaluatorDoubleWell.h"
#include "BondEvaluatorQuartic.h"
#include "DPDPairEvaluatorGeneralWeight.h"
#include "PairEvaluatorColloid.h"
#include "PairEvaluatorExpandedYukawa.h"
#include "PairEvaluatorHertz.h"
#include "PairEvaluatorPerturbedLennardJones.h"

#include <azp.h>

static_assert(sizeof(hoomd::Scalar) == 8, "libazp is built for HOOMD_LONGREAL_SIZE == 64");
static_assert(sizeof(hoomd::azplugins::detail::PairEvaluatorPerturbedLennardJones::param_type) == sizeof(azp_plj_params), "param layout");
static_assert(sizeof(hoomd::azplugins::detail::PairEvaluatorHertz::param_type) == sizeof(azp_hertz_params), "param layout");
static_assert(sizeof(hoomd::azplugins::detail::PairEvaluatorExpandedYukawa::param_type) == sizeof(azp_yukawa_params), "param layout");
static_assert(sizeof(hoomd::azplugins::detail::PairEvaluatorColloid::param_type) == sizeof(azp_colloid_params), "param layout");
static_assert(sizeof(hoomd::azplugins::detail::DPDPairEvaluatorGeneralWeight::param_type) == sizeof(azp_dpd_params), "param layout");

namespace hoomd
    {
namespace md
    {
namespace kernel
    {
namespace
    {
azp_box to_azp_box(const BoxDim& box)
    {
    azp_box b;
    const Scalar3 L = box.getL();
    const uchar3 p = box.getPeriodic();
    b.L[0] = L.x; b.L[1] = L.y; b.L[2] = L.z;
    b.tilt[0] = box.getTiltFactorXY(); b.tilt[1] = box.getTiltFactorXZ(); b.tilt[2] = box.getTiltFactorYZ();
    b.periodic[0] = p.x; b.periodic[1] = p.y; b.periodic[2] = p.z; b._pad = 0;
    return b;
    }

hipError_t to_hip(int rc)
    {
    return rc > 0 ? static_cast<hipError_t>(rc) : (rc == 0 ? hipSuccess : hipErrorInvalidValue);
    }

// fields shared by pair_args_t, dpd_pair_args_t and a_pair_args_t
template<class Args> azp_pair_args common_pair_args(const Args& args)
    {
    azp_pair_args a = {};
    a.d_force = reinterpret_cast<double*>(args.d_force);
    a.d_virial = args.d_virial;
    a.virial_pitch = args.virial_pitch;
    a.N = args.N;
    a.n_max = args.n_max;
    a.d_pos = reinterpret_cast<const double*>(args.d_pos);
    a.box = to_azp_box(args.box);
    a.d_n_neigh = args.d_n_neigh;
    a.d_nlist = args.d_nlist;
    a.d_head_list = reinterpret_cast<const uint64_t*>(args.d_head_list);
    a.d_rcutsq = args.d_rcutsq;
    a.ntypes = args.ntypes;
    a.shift_mode = args.shift_mode; // 0 none, 1 shift, 2 xplor: same encoding
    a.compute_virial = args.compute_virial;
    a.block_size = 0;               // library default
    a.threads_per_particle = 0;     // 0: libazp's own cached tile plan; HOOMD's autotuner
                                    // value (args.threads_per_particle) would select the generic kernel
    a.flags = 0;
    return a;
    }
    } // namespace

#define AZP_PAIR_DRIVER(EVALUATOR, ENTRY, PARAMS)                                                            \
    template<> __attribute__((visibility("default"))) hipError_t                                             \
    gpu_compute_pair_forces<azplugins::detail::EVALUATOR>(const pair_args_t& args,                           \
                                                          const azplugins::detail::EVALUATOR::param_type* d_params) \
        {                                                                                                    \
        azp_pair_args a = common_pair_args(args);                                                            \
        a.d_ronsq = args.d_ronsq;                                                                            \
        a.size_nlist = args.size_neigh_list;                                                                 \
        return to_hip(ENTRY(&a, reinterpret_cast<const PARAMS*>(d_params), nullptr));                        \
        }

AZP_PAIR_DRIVER(PairEvaluatorColloid, azp_pair_forces_colloid, azp_colloid_params)
AZP_PAIR_DRIVER(PairEvaluatorExpandedYukawa, azp_pair_forces_expanded_yukawa, azp_yukawa_params)
AZP_PAIR_DRIVER(PairEvaluatorHertz, azp_pair_forces_hertz, azp_hertz_params)
AZP_PAIR_DRIVER(PairEvaluatorPerturbedLennardJones, azp_pair_forces_perturbed_lennard_jones, azp_plj_params)
#undef AZP_PAIR_DRIVER

// PotentialPairConservativeGeneralWeight is HOOMD's PotentialPair<DPDPairEvaluatorGeneralWeight>
// (src/export_PotentialPairDPDThermo.cc.inc:33-35) and has no GPU class in the reference.
template<> __attribute__((visibility("default"))) hipError_t
gpu_compute_dpd_forces<azplugins::detail::DPDPairEvaluatorGeneralWeight>(
    const dpd_pair_args_t& args,
    const azplugins::detail::DPDPairEvaluatorGeneralWeight::param_type* d_params)
    {
    azp_dpd_args d = {};
    d.pair = common_pair_args(args);
    d.pair.size_nlist = args.size_nlist;
    d.d_vel = reinterpret_cast<const double*>(args.d_vel);
    d.d_tag = args.d_tag;
    d.timestep = args.timestep;
    d.deltaT = args.deltaT;
    d.T = args.T;
    d.seed = args.seed;
    return to_hip(azp_dpd_forces_general_weight(&d, reinterpret_cast<const azp_dpd_params*>(d_params), nullptr));
    }

template<> __attribute__((visibility("default"))) hipError_t
gpu_compute_pair_aniso_forces<azplugins::detail::AnisoPairEvaluatorTwoPatchMorse>(
    const a_pair_args_t& args,
    const azplugins::detail::AnisoPairEvaluatorTwoPatchMorse::param_type* d_params,
    const azplugins::detail::AnisoPairEvaluatorTwoPatchMorse::shape_type* /* empty: src/AnisoPairEvaluator.h:65-85 */)
    {
    azp_aniso_args a = {};
    a.pair = common_pair_args(args);
    a.d_orientation = reinterpret_cast<const double*>(args.d_orientation);
    a.d_torque = reinterpret_cast<double*>(args.d_torque);
    return to_hip(azp_aniso_forces_two_patch_morse(&a, reinterpret_cast<const azp_tpm_params*>(d_params), nullptr));
    }

#define AZP_BOND_DRIVER(EVALUATOR, ENTRY, PARAMS)                                                            \
    template<> __attribute__((visibility("default"))) hipError_t                                             \
    gpu_compute_bond_forces<azplugins::detail::EVALUATOR, 2>(const bond_args_t<2>& args,                     \
                                                             const azplugins::detail::EVALUATOR::param_type* d_params, \
                                                             unsigned int* d_flags)                          \
        {                                                                                                    \
        azp_bond_args b = {};                                                                                \
        b.d_force = reinterpret_cast<double*>(args.d_force);                                                 \
        b.d_virial = args.d_virial;                                                                          \
        b.virial_pitch = args.virial_pitch;                                                                  \
        b.N = args.N;                                                                                        \
        b.n_max = args.n_max;                                                                                \
        b.d_pos = reinterpret_cast<const double*>(args.d_pos);                                               \
        b.box = to_azp_box(args.box);                                                                        \
        /* group_storage<2> = {idx[0] = partner, idx[1] = bond type} = azp_bond_entry */                     \
        b.d_gpu_bondlist = reinterpret_cast<const azp_bond_entry*>(args.d_gpu_bondlist);                     \
        b.d_gpu_bond_pos = args.d_gpu_bond_pos_list;                                                         \
        b.d_gpu_n_bonds = args.d_gpu_n_bonds;                                                                \
        b.pitch = args.gpu_table_indexer.getW();                                                             \
        b.n_bond_types = args.n_bond_types;                                                                  \
        b.compute_virial = 1; /* bond_args_t has no such flag: HOOMD's bond kernels always write the virial */ \
        return to_hip(ENTRY(&b, reinterpret_cast<const PARAMS*>(d_params), d_flags, nullptr));               \
        }

AZP_BOND_DRIVER(BondEvaluatorDoubleWell, azp_bond_forces_double_well, azp_dw_params)
AZP_BOND_DRIVER(BondEvaluatorQuartic, azp_bond_forces_quartic, azp_quartic_params)
#undef AZP_BOND_DRIVER

    } // namespace kernel
    } // namespace md
    } // namespace hoomd
