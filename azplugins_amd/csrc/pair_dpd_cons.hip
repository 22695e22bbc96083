// pair_dpd_cons.hip -- C-ABI entry points azp_pair_forces_dpd_conservative and
// azp_pair_forces_planned_dpd_conservative (see include/azp.h; kernels in
// pair_kernel.hpp / pair_tiled.hpp, arithmetic in evaluators.hpp).
#include "pair_auto.hpp"

extern "C" int azp_pair_forces_dpd_conservative(const azp_pair_args* args, const azp_dpd_params* d_params, void* stream)
    {
    return azp::launch_pair_entry<azp::EvalDPDConservative>(args, d_params, stream);
    }

extern "C" int azp_pair_forces_planned_dpd_conservative(azp_pair_plan* plan, const azp_pair_args* args,
                                                const azp_dpd_params* d_params, void* stream)
    {
    return azp::launch_pair_planned<azp::EvalDPDConservative>(plan, args, d_params, stream);
    }
