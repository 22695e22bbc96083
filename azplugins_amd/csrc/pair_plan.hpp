// pair_plan.hpp -- "tile plan": a compiled form of a HOOMD neighbor list for the
// tile-staged pair kernel (pair_tiled.hpp).
//
// Why: the generic kernel gathers each neighbor position from L1/L2 (two tag
// lookups per neighbor per lane) and is bound by the L1 lookup rate, not by HBM
// or FP64 issue (profiles/r01_plj_generic_tpp8_summary.csv). The plan moves the
// gather into LDS:
//
//   tile   = TB consecutive (spatially sorted) particles = one workgroup
//   stage  = sorted unique list of every particle any tile member lists as a
//            neighbor; the force kernel loads those positions ONCE per tile with
//            coalesced-ish loads into LDS (SoA x|y|z), already shifted to the
//            periodic image nearest the tile, so the inner loop needs no
//            minimum-image arithmetic
//   cnl    = compiled neighbor list: per wave ("slice") a sliced-ELL array of
//            16-byte chunks, chunk (k, lane) at slice_head + k*64 + lane, each
//            holding 8 x u16 byte offsets (slot*8) into the LDS arrays; rows are
//            padded with offset 0 = a dummy slot parked far away. Index traffic:
//            2 B per neighbor, every load a full 1 KiB wave transaction.
//
// The plan is rebuilt whenever the neighbor list is rebuilt (every ~10-20 MD
// steps) and reused by every force call in between.
#pragma once
#include <cmath>
#include <vector>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azp.h"

namespace azp
{
constexpr uint32_t PLAN_HASH_CAP = 8192;     // largest LDS hash-set capacity per tile (build only)
constexpr uint32_t PLAN_MAX_STAGE = 2559;    // + dummy slot 0 => <= 2560 LDS slots (fill pass: 76 KB + 28 B/slot of LDS)
constexpr uint32_t PLAN_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t PLAN_BITMAP_WORDS = 80;    // 2560 slots / 32
constexpr uint32_t PLAN_ROWBUF = 512;         // compiled entries per row the builder can hold (rows longer than
                                              // this invalidate the plan)
constexpr double PLAN_FAR = 1.0e30;          // coordinate of the dummy slot
constexpr uint32_t PLAN_SHELLS = 8;          // the Verlet-buffer entries of every row are ordered into this many shells
                                              // of equal width by their separation when the plan is built
constexpr uint32_t PLAN_CLASSES = PLAN_SHELLS + 3; // row order: core | sure | near | shell 0 | ... | shell PLAN_SHELLS - 1
constexpr uint32_t PLAN_CLS_CORE = 0;  // closer than the evaluator's inner radius hint (azp_pair_args.d_rinnersq) at build time
constexpr uint32_t PLAN_CLS_SURE = 1;  // certainly closer than r_cut - r_buff at build time (one particle type only): such a pair
                                       // stays inside the cutoff while no particle has moved farther than r_buff / 2
constexpr uint32_t PLAN_CLS_NEAR = 2;  // inside the cutoff at build time, or too close to call
constexpr uint32_t PLAN_CLS_SHELL0 = 3; // + s: certainly >= r_cut + s w away at build time

struct PairPlan
    {
    // configuration chosen at build time
    uint32_t tpp = 0;          // lanes per particle (1, 2, 4)
    uint32_t tile = 0;         // particles per tile = 4 waves * 64 / tpp
    uint32_t cap = 0;          // LDS slots the force kernel must provide (1024 / 1536 / 1664 / 2048 / 2560)
    // what it was built for
    uint32_t N = 0, n_max = 0;
    const uint32_t* nlist_ptr = nullptr;
    const uint64_t* head_ptr = nullptr;
    uint64_t size_nlist = 0;
    bool valid = false;        // false => callers fall back to the generic kernel
    int invalid_reason = 0;    // 2: a tile's neighbor set exceeds PLAN_MAX_STAGE (e.g. unsorted particles)
    // sizes
    uint32_t n_tiles = 0, n_slices = 0;
    uint32_t max_stage = 0;
    uint64_t total_stage = 0, total_chunks = 0; // chunks in units of (64 lanes x 16 B)
    // device buffers (owned)
    uint32_t* d_tile_nstage = nullptr;     // n_tiles
    uint64_t* d_tile_head = nullptr;       // n_tiles
    uint32_t* d_stage_idx = nullptr;       // total_stage
    uint32_t* d_slice_K = nullptr;         // n_slices
    uint32_t* d_slice_Kend = nullptr;      // (PLAN_SHELLS + 1) x n_slices: chunks up to the end of the in-range entries [0] /
                                           // of buffer shell s [1 + s]
    uint32_t* d_slice_Kphase = nullptr;    // [2][n_slices]: chunks covering every entry of class core [0][slice]; chunks up to which
                                           // every row of the slice holds only entries of the classes core and sure [1][slice]
    size_t cap_kphase = 0;
    float core_r = 0.f;                    // entries outside class core were at least this far apart at build time (the inner
                                           // radius hint minus a margin for the single-precision test); 0: no core class
    float sure_r = 0.f;                    // entries of class sure were at most this far apart (margin included); 0: no such class
    double shell_width = 0.0;              // w: shell s holds entries with r_build >= r_cut + s w (certainly); 0: no shells
                                           // (no r_list_max hint at build time: all buffer entries sit in shell 0)
    float max_listed_r = 0.f;              // largest separation of a listed pair when the plan was built (single precision)
    double shell_hint_r_list = 0.0;        // > 0 and no azp_pair_args.r_list_max: an estimate of r_cut_max + r_buff that sizes
                                           // the shells (any value is exact; it only sets how finely the buffer is cut)
    uint64_t* d_slice_head = nullptr;      // n_slices (chunk units)
    uint4* d_cnl = nullptr;                // total_chunks * 64
    uint32_t* d_flags = nullptr;           // [1] stage overflow, [2] max staged set, [3] shell width, [4] max listed r^2 (float bits)
    size_t cap_tiles = 0, cap_slices = 0, cap_stage = 0, cap_cnl = 0, cap_kend = 0;
    uint64_t builds = 0;
    // plans compiled straight from the cell list (pair_plan_cells.hip)
    bool from_cells = false;
    uint32_t row_cap = 0;           // entries a compiled row can hold (fixed chunk capacity per slice)
    uint32_t max_row = 0;           // longest row seen (> row_cap: the caller retries with longer rows)
    uint16_t* d_raw = nullptr;      // raw rows (candidate number | class), scratch of the build
    uint32_t max_member_cells = 0;  // build from cells: most distinct cells under the members of one tile (limit 128)
    size_t cap_raw = 0;
    // build option of plans from cells (azp_pair_plan_set_balance): the rows of a tile are handed to its
    // lanes in the order of their in-range lengths, so that each wave gets rows of similar length
    // (the tile kernels run max-over-lanes heavy blocks per wave); d_perm[tile * 256 + lane] = member
    bool balance = false;
    bool balanced = false;          // this build is
    uint8_t* d_perm = nullptr;
    size_t cap_perm = 0;
    bool bank_order = true;         // build option (azp_pair_plan_set_bank_order)
    uint32_t stage_stride_hint = 0; // stage_idx entries to reserve per tile next time (last max + 25 %)
    // host copy of d_tile_nstage: a launch over a sub-range of tiles (domain-decomposed
    // runs: interior | boundary) picks the LDS variant from the tiles it covers
    std::vector<uint32_t> h_tile_nstage;
    // Two launches by staged-set size (pair_tiled.hpp: launch_tiled_cap). In a liquid the staged sets spread (1,400 ..
    // 1,850 at the north star's density): the largest one used to pick the LDS variant of the WHOLE launch (2,048 slots,
    // three workgroups per CU) although four tiles in five fit the 1,664-slot variant (four per CU). Tile numbers of the
    // two groups, small first; made on the first launch after a build from h_tile_nstage.
    mutable std::vector<uint32_t> h_tile_ids;
    mutable uint32_t* d_tile_ids = nullptr;
    mutable size_t cap_tile_ids = 0;
    mutable uint64_t tile_ids_build = 0;
    mutable uint32_t n_small_tiles = 0;
    };

// Buffer shells a launch has to walk, from the caller's displacement bound: an entry of
// shell s was at least r_cut + s w away when the plan was built, so it cannot be in range
// while 2 x bound <= s w. Exact, not a heuristic.
inline uint32_t plan_shells_for(const PairPlan& plan, const azp_pair_args& args)
    {
    if (!args.has_displacement_bound || !(args.displacement_bound >= 0.0))
        return PLAN_SHELLS; // unknown: whole rows
    if (args.displacement_bound == 0.0)
        return 0;
    if (!(plan.shell_width > 0.0))
        return PLAN_SHELLS;
    const double n = std::ceil(2.0 * args.displacement_bound * (1.0 + 1e-12) / plan.shell_width);
    return n >= (double)PLAN_SHELLS ? PLAN_SHELLS : (uint32_t)n;
    }

int plan_build(PairPlan& p, const azp_pair_args& args, hipStream_t s); // pair_plan.hip
int plan_build_from_cells(PairPlan& p, const azp_nlist_args& cells, const azp_pair_args& pair, hipStream_t s); // pair_plan_cells.hip
void plan_free(PairPlan& p);

inline uint32_t plan_cap_for(uint32_t max_stage)
    {
    // 1664 slots x 24 B is the most that still lets four workgroups share a CU's 160 KiB of LDS
    const uint32_t need = max_stage + 1;
    return need <= 1024 ? 1024 : (need <= 1536 ? 1536 : (need <= 1664 ? 1664 : (need <= 2048 ? 2048 : 2560)));
    }

} // namespace azp
