// pair_dpd_cons.hip -- C-ABI entry point azp_pair_forces_dpd_conservative
// (see include/azp.h; kernel in pair_kernel.hpp, arithmetic in evaluators.hpp).
#include "pair_kernel.hpp"

extern "C" int azp_pair_forces_dpd_conservative(const azp_pair_args* args, const azp_dpd_params* d_params, void* stream)
    {
    return azp::launch_pair<azp::EvalDPDConservative>(args, d_params, stream);
    }
