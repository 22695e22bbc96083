// pair_colloid.hip -- C-ABI entry points azp_pair_forces_colloid and
// azp_pair_forces_planned_colloid (see include/azp.h; kernels in
// pair_kernel.hpp / pair_tiled.hpp, arithmetic in evaluators.hpp).
#include "pair_auto.hpp"

extern "C" int azp_pair_forces_colloid(const azp_pair_args* args, const azp_colloid_params* d_params, void* stream)
    {
    return azp::launch_pair_entry<azp::EvalColloid>(args, d_params, stream);
    }

extern "C" int azp_pair_forces_planned_colloid(azp_pair_plan* plan, const azp_pair_args* args,
                                                const azp_colloid_params* d_params, void* stream)
    {
    return azp::launch_pair_planned<azp::EvalColloid>(plan, args, d_params, stream);
    }
