// pair_yukawa.hip -- C-ABI entry points azp_pair_forces_expanded_yukawa and
// azp_pair_forces_planned_expanded_yukawa (see include/azp.h; kernels in
// pair_kernel.hpp / pair_tiled.hpp, arithmetic in evaluators.hpp).
#include "pair_auto.hpp"

extern "C" int azp_pair_forces_expanded_yukawa(const azp_pair_args* args, const azp_yukawa_params* d_params, void* stream)
    {
    return azp::launch_pair_entry<azp::EvalYukawa>(args, d_params, stream);
    }

extern "C" int azp_pair_forces_planned_expanded_yukawa(azp_pair_plan* plan, const azp_pair_args* args,
                                                const azp_yukawa_params* d_params, void* stream)
    {
    return azp::launch_pair_planned<azp::EvalYukawa>(plan, args, d_params, stream);
    }
