// pair_hertz.hip -- C-ABI entry point azp_pair_forces_hertz
// (see include/azp.h; kernel in pair_kernel.hpp, arithmetic in evaluators.hpp).
#include "pair_kernel.hpp"

extern "C" int azp_pair_forces_hertz(const azp_pair_args* args, const azp_hertz_params* d_params, void* stream)
    {
    return azp::launch_pair<azp::EvalHertz>(args, d_params, stream);
    }
