// pair_auto.hpp -- the tile plan behind the reference's own kernel-driver signatures.
//
// azplugins binds   hipError_t gpu_compute_pair_forces<E>(const pair_args_t&, const param_type*)
// (src/PotentialPairGPUKernel.cu.inc:25-28) and its siblings gpu_compute_dpd_forces<E>
// (src/PotentialPairDPDThermoGPUKernel.cu.inc:21-24) and gpu_compute_pair_aniso_forces<E>
// (src/AnisoPotentialPairGPUKernel.cu.inc:21-25): one call per step, borrowed device
// pointers, no notion of "the neighbor list was rebuilt". libazp's entry points
// azp_pair_forces_<evaluator>, azp_dpd_forces_general_weight and
// azp_aniso_forces_two_patch_morse take exactly that information, so the LDS-staged tile
// kernels have to find out by themselves whether the plan they compiled still describes the
// list they are handed. Per call (round 3: nothing on the caller's stream waits for the host):
//
//   1. one small kernel on the caller's stream (pair_auto.hip: auto_check_kernel; its last
//      workgroup folds the partial results) leaves in device memory
//        * a 64-bit fingerprint of the list -- every n_neigh and head_list word, the cutoff
//          table, the box and N, plus every list entry for lists of up to 2^22 entries and,
//          beyond that, two entries of every 8th row, the rows rotating from call to call so
//          that eight calls cover every row -- compared there with the value learned when the
//          plan was compiled: the "stale" word;
//        * the largest displacement of any particle (and any type change) since the plan was
//          compiled, against a copy of the positions taken at that time, and the number of
//          Verlet-buffer shells that displacement makes the tile kernel walk;
//   2. the tile kernel is launched right behind it, SPECULATIVELY: it reads the shell count
//      from device memory and its workgroups leave at once when the stale word is set;
//   3. the 48-byte result travels to the host on a side stream (event-ordered after the check
//      kernel only), and the host waits for THAT -- about the duration of the check kernel,
//      while the tile kernel is already running;
//   4. stale (or no plan yet): the plan is recompiled from the list, the positions are copied,
//      the fingerprints are learned, and the tile kernel is launched again -- stream order
//      makes its results the ones the caller sees.
//
// The plans live in a small process-wide cache keyed by the list's device pointers, N, the
// number of types, the cutoff table and the kind of kernel (one lane per particle for the DPD /
// TwoPatchMorse kernels). The cache lock is held from the lookup to the last launch of a call.
// Callers that know when the list changes (the Python layer, HOOMD's
// NeighborList::getNumUpdates()) either use the explicit azp_pair_plan_* API or pass
// azp_pair_args.list_generation (non-zero; compared instead of the fingerprint).
//
// Restrictions, stated: the call blocks the host for the duration of the check kernel (it
// cannot be captured into a HIP graph); all calls that share a list must use one stream. Residual
// risk of the sampled fingerprint (lists above 2^22 entries without list_generation): a rebuild
// that leaves every row length and every row start unchanged is noticed only when the rotating
// sample reaches a changed row (within eight calls). A million-particle rebuild changes on the
// order of a million row lengths. AZP_AUTO_PLAN=0 (environment) or AZP_PAIR_FLAG_NO_AUTO_PLAN
// (per call) selects the generic kernel instead.
#pragma once

#include <functional>

#include "pair_tiled.hpp"

namespace azp
{
// What the kernel launcher gets from the cache: the plan, the arguments to launch with, and -- for the
// speculative launch -- the device words the kernel reads (null for the launch after a recompile).
struct AutoLaunch
    {
    const PairPlan* plan;
    const azp_pair_args* args;  // displacement fields filled in
    const TileDyn* dyn;         // speculative launch: shell count / stale word in device memory
    };
typedef std::function<int(const AutoLaunch&)> AutoLauncher;

// pair_auto.hip. lanes_one: the kernel needs a plan with one lane per particle (xtiled kernels).
// launch_tiled is called once (plan current) or twice (speculative launch found stale); launch_generic
// when the list cannot be tiled. Returns an azp status.
int auto_plan_run(const azp_pair_args& args, bool lanes_one, hipStream_t stream, const AutoLauncher& launch_tiled,
                  const std::function<int()>& launch_generic);
bool auto_plan_enabled();

inline bool auto_plan_wanted(const azp_pair_args& a)
    {
    // an explicit launch shape (threads per particle) asks for the generic kernel
    return a.threads_per_particle == 0 && !(a.flags & AZP_PAIR_FLAG_NO_AUTO_PLAN) && auto_plan_enabled();
    }

// Entry point behind azp_pair_forces_<evaluator>.
template<class E> int launch_pair_entry(const azp_pair_args* args, const typename E::Params* d_params, void* stream)
    {
    const int bad = validate_pair_args(args, d_params);
    if (bad < 0) return bad;
    if (bad > 0) return AZP_SUCCESS;
    if (!auto_plan_wanted(*args))
        return launch_pair<E>(args, d_params, stream);
    // r_list_max is not part of pair_args_t: when the caller leaves it 0 the kernel decides per
    // tile from the staged positions whether the staged images are minimum images
    return auto_plan_run(
        *args, false, static_cast<hipStream_t>(stream),
        [&](const AutoLaunch& l)
            { return launch_pair_planned<E>(reinterpret_cast<azp_pair_plan*>(const_cast<PairPlan*>(l.plan)), l.args, d_params, stream, l.dyn); },
        [&]() { return launch_pair<E>(args, d_params, stream); });
    }
} // namespace azp
