// pair_auto.hpp -- the tile plan behind the reference's own kernel-driver signature.
//
// azplugins binds   hipError_t gpu_compute_pair_forces<E>(const pair_args_t&, const param_type*)
// (src/PotentialPairGPUKernel.cu.inc:25-28): one call per step, borrowed device
// pointers, no notion of "the neighbor list was rebuilt". libazp's entry points
// azp_pair_forces_<evaluator> take exactly that information, so the LDS-staged tile
// kernel has to find out by itself whether the plan it compiled still describes the
// list it is handed. Per call:
//
//   1. one small kernel (pair_auto.hip: auto_check_kernel) computes
//        * a 64-bit fingerprint of the list: every n_neigh and head_list word, the
//          cutoff table, the box and N -- plus every list entry for lists of up to
//          2^22 entries, two sampled entries of every 8th row beyond that;
//        * the largest displacement of any particle (and any type change) since the
//          plan was compiled, against a copy of the positions taken at that time;
//   2. a readback of one partial triple per workgroup (<= 24 KiB; the one synchronisation of the call; HOOMD itself reads its
//      distance-check flag back every step);
//   3. fingerprint or types changed -> the plan is recompiled from the list (and the
//      positions are copied); else the displacement becomes the tile kernel's
//      displacement bound, so it stops every row before the Verlet-buffer shells that
//      cannot have come into range (exact, pair_plan.hpp: plan_shells_for);
//   4. the tile kernel (or the generic kernel when the list cannot be tiled).
//
// The plans live in a small process-wide cache keyed by the list's device pointers,
// N, the number of types and the cutoff table (one HOOMD neighbor list can serve
// several potentials). Callers that know when the list changes (the Python layer,
// HOOMD's NeighborList::getNumUpdates()) use the explicit azp_pair_plan_* API and pay
// neither the check nor the readback.
//
// Residual risk, stated: for lists of more than 2^22 entries a rebuild that leaves every
// row length, every row start and all sampled entries unchanged while changing some
// other entry would go unnoticed. For a million-particle system a rebuild changes on
// the order of a million row lengths; AZP_AUTO_PLAN=0 (environment) or
// AZP_PAIR_FLAG_NO_AUTO_PLAN (per call) selects the generic kernel instead.
#pragma once

#include "pair_tiled.hpp"

namespace azp
{
struct AutoPlanCheck
    {
    int status;              // AZP_SUCCESS or an error
    PairPlan* plan;          // null: use the generic kernel
    double displacement;     // largest displacement since the plan was compiled
    };

// pair_auto.hip: look up / (re)compile the cached plan for these arguments.
AutoPlanCheck auto_plan_prepare(const azp_pair_args& args, hipStream_t stream);
bool auto_plan_enabled();

// Entry point behind azp_pair_forces_<evaluator>.
template<class E> int launch_pair_entry(const azp_pair_args* args, const typename E::Params* d_params, void* stream)
    {
    const int bad = validate_pair_args(args, d_params);
    if (bad < 0) return bad;
    if (bad > 0) return AZP_SUCCESS;
    // an explicit launch shape (threads per particle) asks for the generic kernel
    if (args->threads_per_particle != 0 || (args->flags & AZP_PAIR_FLAG_NO_AUTO_PLAN) || !auto_plan_enabled())
        return launch_pair<E>(args, d_params, stream);
    const AutoPlanCheck chk = auto_plan_prepare(*args, static_cast<hipStream_t>(stream));
    if (chk.status != AZP_SUCCESS)
        return chk.status;
    if (!chk.plan || !chk.plan->valid)
        return launch_pair<E>(args, d_params, stream);
    azp_pair_args a = *args;
    a.has_displacement_bound = 1;
    a.displacement_bound = chk.displacement;
    // r_list_max is not part of pair_args_t: when the caller leaves it 0 the kernel decides per
    // tile from the staged positions whether the staged images are minimum images
    return launch_pair_planned<E>(reinterpret_cast<azp_pair_plan*>(chk.plan), &a, d_params, stream);
    }
} // namespace azp
