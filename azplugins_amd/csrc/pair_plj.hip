// pair_plj.hip -- C-ABI entry points azp_pair_forces_perturbed_lennard_jones and
// azp_pair_forces_planned_perturbed_lennard_jones (see include/azp.h; kernels in
// pair_kernel.hpp / pair_tiled.hpp, arithmetic in evaluators.hpp).
#include "pair_auto.hpp"

extern "C" int azp_pair_forces_perturbed_lennard_jones(const azp_pair_args* args, const azp_plj_params* d_params, void* stream)
    {
    return azp::launch_pair_entry<azp::EvalPLJ>(args, d_params, stream);
    }

extern "C" int azp_pair_forces_planned_perturbed_lennard_jones(azp_pair_plan* plan, const azp_pair_args* args,
                                                const azp_plj_params* d_params, void* stream)
    {
    return azp::launch_pair_planned<azp::EvalPLJ>(plan, args, d_params, stream);
    }

#ifdef AZP_TIMELINE
namespace azp { __device__ unsigned long long g_timeline[8 * 4 * 16384]; }
extern "C" int azp_debug_timeline(unsigned long long* out, size_t n_words)
    {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(azp::g_timeline), n_words * sizeof(unsigned long long));
    }
#endif
