// azp_host.cpp -- host-only parts of the C ABI: parameter-struct construction
// and inspection (what the reference's pybind11::dict constructors and
// asDict()/toPython() do), status strings, launch bookkeeping.
#include <cstdlib>
#include <cmath>
#include <cstring>

#include "../../include/azp.h"
#include "pair_kernel_host.hpp"

extern "C" {

// src/PairEvaluatorPerturbedLennardJones.h:33-45
void azp_plj_params_make(double epsilon, double sigma, double attraction_scale_factor, azp_plj_params* out)
    {
    const double sigma_2 = sigma * sigma;
    const double sigma_4 = sigma_2 * sigma_2;
    out->sigma_6 = sigma_2 * sigma_4;
    out->epsilon_x_4 = 4.0 * epsilon;
    out->attraction_scale_factor = attraction_scale_factor;
    out->rwcasq = std::pow(2.0, 1. / 3.) * sigma_2;
    }
// src/PairEvaluatorPerturbedLennardJones.h:47-54
void azp_plj_params_unpack(const azp_plj_params* p, double* epsilon, double* sigma, double* attraction_scale_factor)
    {
    *sigma = std::pow(p->sigma_6, 1. / 6.);
    *epsilon = p->epsilon_x_4 / 4.0;
    *attraction_scale_factor = p->attraction_scale_factor;
    }
// src/PairEvaluatorColloid.h:28-46
void azp_colloid_params_make(double A, double a_1, double a_2, double sigma, azp_colloid_params* out)
    {
    out->A = A;
    out->a_1 = a_1;
    out->a_2 = a_2;
    out->sigma_3 = sigma * sigma * sigma;
    }
void azp_colloid_params_unpack(const azp_colloid_params* p, double* A, double* a_1, double* a_2, double* sigma)
    {
    *A = p->A;
    *a_1 = p->a_1;
    *a_2 = p->a_2;
    *sigma = std::cbrt(p->sigma_3);
    }
// src/AnisoPairEvaluatorTwoPatchMorse.h:40-60
void azp_tpm_params_make(double M_d, double M_r, double r_eq, double omega, double alpha, int repulsion,
                         azp_tpm_params* out)
    {
    std::memset(out, 0, sizeof(*out));
    out->M_d = M_d;
    out->M_rinv = 1.0 / M_r;
    out->r_eq = r_eq;
    out->omega = omega;
    out->alpha = alpha;
    out->repulsion = repulsion ? 1 : 0;
    }
void azp_tpm_params_unpack(const azp_tpm_params* p, double* M_d, double* M_r, double* r_eq, double* omega,
                           double* alpha, int* repulsion)
    {
    *M_d = p->M_d;
    *M_r = 1.0 / p->M_rinv;
    *r_eq = p->r_eq;
    *omega = p->omega;
    *alpha = p->alpha;
    *repulsion = p->repulsion ? 1 : 0;
    }
// src/BondEvaluatorDoubleWell.h:33-49
void azp_dw_params_make(double r_0, double r_1, double U_1, double U_tilt, azp_dw_params* out)
    {
    out->r_1 = r_1;
    out->r_diff = r_1 - r_0;
    out->U_1 = U_1;
    out->U_tilt = U_tilt;
    }
void azp_dw_params_unpack(const azp_dw_params* p, double* r_0, double* r_1, double* U_1, double* U_tilt)
    {
    *r_0 = p->r_1 - p->r_diff;
    *r_1 = p->r_1;
    *U_1 = p->U_1;
    *U_tilt = p->U_tilt;
    }
// src/BondEvaluatorQuartic.h:36-66
void azp_quartic_params_make(double k, double r_0, double b_1, double b_2, double U_0, double sigma, double epsilon,
                             double delta, azp_quartic_params* out)
    {
    out->k = k;
    out->r_0 = r_0;
    out->b_1 = b_1;
    out->b_2 = b_2;
    out->U_0 = U_0;
    out->delta = delta;
    const double sigma_2 = sigma * sigma;
    const double sigma_4 = sigma_2 * sigma_2;
    out->sigma_6 = sigma_2 * sigma_4;
    out->epsilon_x_4 = 4.0 * epsilon;
    }
void azp_quartic_params_unpack(const azp_quartic_params* p, double* k, double* r_0, double* b_1, double* b_2,
                               double* U_0, double* sigma, double* epsilon, double* delta)
    {
    *k = p->k;
    *r_0 = p->r_0;
    *b_1 = p->b_1;
    *b_2 = p->b_2;
    *U_0 = p->U_0;
    *sigma = std::pow(p->sigma_6, 1. / 6.);
    *epsilon = p->epsilon_x_4 / 4.0;
    *delta = p->delta;
    }

int azp_version(void)
    {
    return AZP_VERSION_MAJOR * 1000 + AZP_VERSION_MINOR;
    }

const char* azp_status_string(int status)
    {
    switch (status)
        {
    case AZP_SUCCESS: return "success";
    case AZP_ERROR_INVALID_ARGUMENT: return "invalid argument";
    case AZP_ERROR_TOO_MANY_TYPES: return "per-type-pair coefficient table exceeds 160 KiB of LDS";
    case AZP_ERROR_NO_DEVICE: return "no HIP device";
    default: return status > 0 ? "HIP runtime error (value is hipError_t)" : "unknown status";
        }
    }

void azp_last_launch(uint32_t* block_size, uint32_t* threads_per_particle, uint32_t* grid, uint32_t* lds_bytes)
    {
    const azp::LaunchInfo& li = azp::last_launch();
    if (block_size) *block_size = li.block_size;
    if (threads_per_particle) *threads_per_particle = li.tpp;
    if (grid) *grid = li.grid;
    if (lds_bytes) *lds_bytes = li.lds_bytes;
    }

int azp_tuning_set(int key, int value)
    {
    int* slot = key == AZP_TUNE_ROW_PHASES ? &azp::tuning().row_phases
                : (key == AZP_TUNE_LOCAL_BOUND ? &azp::tuning().local_bound : (key == AZP_TUNE_SPLIT_TILES ? &azp::tuning().split_tiles : nullptr));
    if (!slot)
        return -1;
    const int old = *slot;
    *slot = value ? 1 : 0;
    return old;
    }

} // extern "C"

namespace azp
{
Tuning& tuning()
    {
    static Tuning t = []()
        {
        Tuning v;
        // off by default: measured on the north star (tools/ab_cycle.py) the phases issue 1 % fewer VALU
        // instructions and run 1.5 % SLOWER -- the kernel is held by its LDS gathers, not by VALU issue
        const char* e = std::getenv("AZP_ROW_PHASES");
        v.row_phases = (e && e[0] == '1');
        e = std::getenv("AZP_LOCAL_BOUND");
        v.local_bound = !(e && e[0] == '0');
        // (off by default: two launches pay the ramp-down of a launch twice -- the liquid's force kernel measured
        // 0.142-0.146 ms split against 0.135 ms in one launch of the 2,048-slot variant)
        e = std::getenv("AZP_SPLIT_TILES");
        v.split_tiles = (e && e[0] == '1');
        return v;
        }();
    return t;
    }

LaunchInfo& last_launch()
    {
    static thread_local LaunchInfo li = {0, 0, 0, 0};
    return li;
    }
} // namespace azp
