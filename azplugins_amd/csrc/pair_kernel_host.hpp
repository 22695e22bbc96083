// pair_kernel_host.hpp -- launch bookkeeping shared between host-only and
// device translation units.
#pragma once
#include <stdint.h>
namespace azp
{
struct LaunchInfo
    {
    uint32_t block_size, tpp, grid, lds_bytes;
    };
LaunchInfo& last_launch();
struct Tuning { int row_phases, local_bound, split_tiles; };
Tuning& tuning(); // azp_host.cpp (azp_tuning_set)
} // namespace azp
