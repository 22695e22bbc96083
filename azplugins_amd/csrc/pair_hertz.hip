// pair_hertz.hip -- C-ABI entry points azp_pair_forces_hertz and
// azp_pair_forces_planned_hertz (see include/azp.h; kernels in
// pair_kernel.hpp / pair_tiled.hpp, arithmetic in evaluators.hpp).
#include "pair_auto.hpp"

extern "C" int azp_pair_forces_hertz(const azp_pair_args* args, const azp_hertz_params* d_params, void* stream)
    {
    return azp::launch_pair_entry<azp::EvalHertz>(args, d_params, stream);
    }

extern "C" int azp_pair_forces_planned_hertz(azp_pair_plan* plan, const azp_pair_args* args,
                                                const azp_hertz_params* d_params, void* stream)
    {
    return azp::launch_pair_planned<azp::EvalHertz>(plan, args, d_params, stream);
    }
