// pair_plj.hip -- C-ABI entry point azp_pair_forces_perturbed_lennard_jones
// (see include/azp.h; kernel in pair_kernel.hpp, arithmetic in evaluators.hpp).
#include "pair_kernel.hpp"

extern "C" int azp_pair_forces_perturbed_lennard_jones(const azp_pair_args* args, const azp_plj_params* d_params, void* stream)
    {
    return azp::launch_pair<azp::EvalPLJ>(args, d_params, stream);
    }
