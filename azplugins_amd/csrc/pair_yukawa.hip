// pair_yukawa.hip -- C-ABI entry point azp_pair_forces_expanded_yukawa
// (see include/azp.h; kernel in pair_kernel.hpp, arithmetic in evaluators.hpp).
#include "pair_kernel.hpp"

extern "C" int azp_pair_forces_expanded_yukawa(const azp_pair_args* args, const azp_yukawa_params* d_params, void* stream)
    {
    return azp::launch_pair<azp::EvalYukawa>(args, d_params, stream);
    }
