// pair_colloid.hip -- C-ABI entry point azp_pair_forces_colloid
// (see include/azp.h; kernel in pair_kernel.hpp, arithmetic in evaluators.hpp).
#include "pair_kernel.hpp"

extern "C" int azp_pair_forces_colloid(const azp_pair_args* args, const azp_colloid_params* d_params, void* stream)
    {
    return azp::launch_pair<azp::EvalColloid>(args, d_params, stream);
    }
