/*
 * azp.h -- C ABI of libazp: MI355X (gfx950) force-compute kernels for the
 * azplugins pair / bond potentials.
 *
 * This is the drop-in boundary. Every entry point replaces one kernel-driver
 * template instantiation that the reference (stattlab/azplugins v1.1.0)
 * requests from HOOMD-blue v5:
 *
 *   azp_pair_forces_*            <- hoomd::md::kernel::gpu_compute_pair_forces<E>
 *                                   (src/PotentialPairGPUKernel.cu.inc:25-28)
 *   azp_dpd_forces_general_weight<- gpu_compute_dpd_forces<DPDPairEvaluatorGeneralWeight>
 *                                   (src/PotentialPairDPDThermoGPUKernel.cu.inc:21-24)
 *   azp_aniso_forces_two_patch_morse
 *                                <- gpu_compute_pair_aniso_forces<AnisoPairEvaluatorTwoPatchMorse>
 *                                   (src/AnisoPotentialPairGPUKernel.cu.inc:21-25)
 *   azp_bond_forces_*            <- gpu_compute_bond_forces<E, 2>
 *                                   (src/PotentialBondGPUKernel.cu.inc:25-29)
 *
 * Conventions (all restated from HOOMD-blue's ForceCompute data model):
 *   - Scalar = double. Scalar4 arrays are 4 consecutive doubles.
 *   - d_pos[i]   = (x, y, z, type) with the integer type index stored in the
 *                  low 32 bits of w (HOOMD __scalar_as_int).
 *   - d_force[i] = (fx, fy, fz, energy); energy is this particle's half share.
 *   - d_virial   = 6 rows (xx, xy, xz, yy, yz, zz) of length virial_pitch.
 *   - d_orientation[i] = quaternion, scalar part first.
 *   - neighbor list: full storage; neighbors of i are
 *     d_nlist[d_head_list[i] + k], k < d_n_neigh[i]; indices may point at
 *     ghost particles (>= N, < n_max).
 *   - per-type-pair tables (rcutsq, ronsq, params) are indexed
 *     type_i * ntypes + type_j and must be symmetric.
 *   - outputs are OVERWRITTEN for all N local particles.
 *   - all pointers prefixed d_ are device pointers borrowed for the call. The
 *     bond, DPD, aniso, barrier, NVE, neighbor-list and generic pair kernels
 *     allocate nothing and never synchronise the stream; azp_pair_plan_build,
 *     azp_pair_plan_build_from_cells and the plan cache behind azp_pair_forces_*
 *     (see there) own device workspace and synchronise.
 *   - every function returns 0 on success, a positive hipError_t value if the
 *     launch failed, or a negative azp_status for invalid arguments; nothing
 *     throws across this boundary.
 */
#ifndef AZP_H_
#define AZP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZP_VERSION_MAJOR 0
#define AZP_VERSION_MINOR 1

typedef enum azp_status
    {
    AZP_SUCCESS = 0,
    AZP_ERROR_INVALID_ARGUMENT = -1,
    AZP_ERROR_TOO_MANY_TYPES = -2, /* per-type-pair table does not fit in LDS */
    AZP_ERROR_NO_DEVICE = -3
    } azp_status;

typedef enum azp_shift_mode
    {
    AZP_SHIFT_NONE = 0,
    AZP_SHIFT_SHIFT = 1,
    AZP_SHIFT_XPLOR = 2
    } azp_shift_mode;

/* HOOMD BoxDim fields the kernels need: box centred on the origin. */
typedef struct azp_box
    {
    double L[3];
    double tilt[3]; /* xy, xz, yz */
    int32_t periodic[3];
    int32_t _pad;
    } azp_box;

/* ---- parameter structs: byte-compatible with the reference's param_type ---- */

/* src/PairEvaluatorPerturbedLennardJones.h:57-66 */
typedef struct azp_plj_params { double sigma_6, epsilon_x_4, attraction_scale_factor, rwcasq; } azp_plj_params;
/* src/PairEvaluatorHertz.h:41-47 */
typedef struct azp_hertz_params { double epsilon; } azp_hertz_params;
/* src/PairEvaluatorExpandedYukawa.h:44-53 (aligned(32)) */
typedef struct azp_yukawa_params { double epsilon, kappa, delta, _pad; } azp_yukawa_params;
/* src/PairEvaluatorColloid.h:48-57 */
typedef struct azp_colloid_params { double A, a_1, a_2, sigma_3; } azp_colloid_params;
/* src/DPDPairEvaluatorGeneralWeight.h:53-62 (aligned(32)) */
typedef struct azp_dpd_params { double A, gamma, s, _pad; } azp_dpd_params;
/* src/AnisoPairEvaluatorTwoPatchMorse.h:63-69 */
typedef struct azp_tpm_params { double M_d, M_rinv, r_eq, omega, alpha; uint8_t repulsion; uint8_t _pad[7]; } azp_tpm_params;
/* src/BondEvaluatorDoubleWell.h:52-61 */
typedef struct azp_dw_params { double r_1, r_diff, U_1, U_tilt; } azp_dw_params;
/* src/BondEvaluatorQuartic.h:68-82 */
typedef struct azp_quartic_params { double k, r_0, b_1, b_2, U_0, sigma_6, epsilon_x_4, delta; } azp_quartic_params;

/* Host-side construction / inspection of the parameter structs: what the
 * reference does in each struct's pybind11::dict constructor and asDict() /
 * toPython() (file:line next to each struct above). Pure host functions. */
void azp_plj_params_make(double epsilon, double sigma, double attraction_scale_factor, azp_plj_params* out);
void azp_plj_params_unpack(const azp_plj_params* p, double* epsilon, double* sigma, double* attraction_scale_factor);
void azp_colloid_params_make(double A, double a_1, double a_2, double sigma, azp_colloid_params* out);
void azp_colloid_params_unpack(const azp_colloid_params* p, double* A, double* a_1, double* a_2, double* sigma);
void azp_tpm_params_make(double M_d, double M_r, double r_eq, double omega, double alpha, int repulsion,
                         azp_tpm_params* out);
void azp_tpm_params_unpack(const azp_tpm_params* p, double* M_d, double* M_r, double* r_eq, double* omega,
                           double* alpha, int* repulsion);
void azp_dw_params_make(double r_0, double r_1, double U_1, double U_tilt, azp_dw_params* out);
void azp_dw_params_unpack(const azp_dw_params* p, double* r_0, double* r_1, double* U_1, double* U_tilt);
void azp_quartic_params_make(double k, double r_0, double b_1, double b_2, double U_0, double sigma, double epsilon,
                             double delta, azp_quartic_params* out);
void azp_quartic_params_unpack(const azp_quartic_params* p, double* k, double* r_0, double* b_1, double* b_2,
                               double* U_0, double* sigma, double* epsilon, double* delta);

/* ---- pair forces ---- */

#define AZP_PAIR_FLAG_NO_AUTO_PLAN 1u /* azp_pair_forces_*: always the generic kernel (no cached tile plan) */

/* Mirrors hoomd::md::kernel::pair_args_t (minus charge and devprop). */
typedef struct azp_pair_args
    {
    double* d_force;             /* N x 4, overwritten                              */
    double* d_virial;            /* 6 x virial_pitch, overwritten if compute_virial */
    uint64_t virial_pitch;
    uint32_t N;                  /* local particles                                 */
    uint32_t n_max;              /* local + ghost particles addressable in d_pos    */
    const double* d_pos;         /* n_max x 4                                       */
    azp_box box;
    const uint32_t* d_n_neigh;   /* N                                               */
    const uint32_t* d_nlist;
    const uint64_t* d_head_list; /* N                                               */
    const double* d_rcutsq;      /* ntypes^2                                        */
    const double* d_ronsq;       /* ntypes^2 (xplor only; may be NULL otherwise)    */
    uint64_t size_nlist;         /* total entries in d_nlist (0 = unknown; tuning hint) */
    uint32_t ntypes;
    uint32_t shift_mode;         /* azp_shift_mode                                  */
    uint32_t compute_virial;
    uint32_t block_size;         /* 0 = library default                             */
    uint32_t threads_per_particle; /* 0 = library heuristic; else 1,2,4,8,16,32     */
    uint32_t flags;              /* AZP_PAIR_FLAG_*                                 */
    uint32_t range_first;        /* compute only particles [range_first, range_first + */
    uint32_t range_count;        /* range_count); range_count = 0 means all N. Lets a   */
                                 /* caller overlap the ghost exchange with the interior */
                                 /* particles (planned kernels round outwards to tiles) */
    double r_list_max;           /* optional: upper bound on the separation of any
                                    listed pair (r_cut_max + 2 r_buff); lets interior
                                    particles skip the minimum-image step. 0 = unknown. */
    const double* d_rinnersq;    /* optional, read by azp_pair_plan_build only: ntypes^2
                                    inner radii^2. Listed pairs closer than this when the
                                    plan is built are placed first in their rows, so that
                                    an evaluator's short-range branch (PerturbedLJ: the WCA
                                    core, r < 2^(1/6) sigma) is confined to the first chunks
                                    of a row. Ordering hint only; NULL = none.           */
    uint32_t has_displacement_bound; /* 0 (default): unknown, whole rows are processed        */
    uint32_t _pad3;
    double displacement_bound;   /* when has_displacement_bound != 0, read by the *_planned
                                    entry points: an upper bound on the distance ANY particle
                                    (local or ghost) has moved since azp_pair_plan_build. Rows
                                    end with the Verlet-buffer entries, split by their
                                    separation r_b at build time; an entry with
                                    r_b >= r_cut + m cannot be in range while
                                    2 * displacement_bound <= m, so the kernel stops before
                                    them (exact, not a heuristic):
                                      bound == 0 (positions unchanged): every buffer entry
                                            is skipped;
                                      2 * bound <= r_buff / 2: the outer half of the buffer
                                            shell is skipped (needs r_list_max at plan build);
                                      larger: whole rows.
                                    HOOMD: max |x - x_at_last_nlist_update|, the quantity
                                    NeighborList::distanceCheck compares with r_buff / 2.    */
    const float* d_displacement; /* optional (with has_displacement_bound != 0), read by the *_planned entry
                                    points: n_max per-particle upper bounds on the distance each particle has moved
                                    since the positions the plan's row classes refer to (azp_nlist_displacements
                                    writes them next to the global maximum). A tile of the kernel then stops its rows
                                    at the shells ITS OWN members and listed neighbors can have crossed (twice the
                                    largest of their displacements) instead of what the fastest particle of the whole
                                    system dictates. Exact; NULL = the global bound alone.  */
    uint64_t list_generation;    /* optional, read by the entry points that keep their own plan cache (azp_pair_forces_*,
                                    azp_dpd_forces_general_weight, azp_aniso_forces_two_patch_morse): a number the caller
                                    changes whenever it rewrites the neighbor list (HOOMD: NeighborList::getNumUpdates()
                                    + 1). Non-zero: the cached plan is recompiled exactly when the number changes and the
                                    list is not fingerprinted; 0 = unknown, the list is fingerprinted at every call. */
    const uint32_t* d_stale_flag; /* optional, read by the *_planned entry points together with the next field: the two
                                    words azp_nlist_distance_check / azp_nlist_displacements leave in device memory
                                    (d_flag and d_max_dist_sq_bits). The kernel then takes its displacement bound from
                                    there AT RUN TIME (sqrt of the double whose bits are in *d_displacement_sq_bits, plus
                                    displacement_bound_extra), and its workgroups leave at once when *d_stale_flag != 0
                                    (some particle moved farther than the check's limit: the list has to be rebuilt and
                                    the call repeated). A caller can so queue the force kernel right behind the check,
                                    before the host knows its result. has_displacement_bound / displacement_bound are
                                    ignored when these are set. */
    const unsigned long long* d_displacement_sq_bits;
    double displacement_bound_extra; /* added to every d_displacement entry: how far any particle had moved from those
                                    reference positions when the plan was built (0 in the usual flow: plan and list are
                                    built from the same positions) */
    } azp_pair_args;

/* The five entry points below take what gpu_compute_pair_forces<E> takes and nothing
 * else. With threads_per_particle == 0 (library's choice) they run the LDS-staged tile
 * kernel from a plan that libazp compiles and caches by itself (csrc/pair_auto.hpp): each
 * call fingerprints the list on the device (row lengths, row starts, cutoffs, box; the
 * entries themselves for lists of up to 2^22 entries, sampled beyond), measures the
 * largest displacement since the plan was compiled, reads the result (<= 24 KiB) back (ONE stream
 * synchronisation per call) and recompiles the plan when the list changed. A caller that
 * passes an explicit threads_per_particle (HOOMD's autotuner value), or sets
 * AZP_PAIR_FLAG_NO_AUTO_PLAN, or runs with AZP_AUTO_PLAN=0 in the environment, gets the
 * generic kernel, which never synchronises. Callers that know when the list changes use
 * the azp_pair_plan_* API below instead and pay neither check nor readback. */
int azp_pair_forces_perturbed_lennard_jones(const azp_pair_args* args, const azp_plj_params* d_params, void* stream);
int azp_pair_forces_hertz(const azp_pair_args* args, const azp_hertz_params* d_params, void* stream);
int azp_pair_forces_expanded_yukawa(const azp_pair_args* args, const azp_yukawa_params* d_params, void* stream);
int azp_pair_forces_colloid(const azp_pair_args* args, const azp_colloid_params* d_params, void* stream);
/* PotentialPairConservativeGeneralWeight (src/export_PotentialPairDPDThermo.cc.inc:33-35) */
int azp_pair_forces_dpd_conservative(const azp_pair_args* args, const azp_dpd_params* d_params, void* stream);

/* ---- tile plan: compiled neighbor list for the LDS-staged pair kernel ----
 * The generic azp_pair_forces_* kernels gather every neighbor position through
 * L1/L2. When the particle order is spatially sorted (HOOMD's SFC sorter), a
 * plan lets the kernels stage each tile's neighbor positions once in LDS and
 * address them with 16-bit tile-local indices (see csrc/pair_plan.hpp).
 * Build the plan whenever the neighbor list was rebuilt (HOOMD:
 * NeighborList::getNumUpdates() changed) -- it synchronises the stream and may
 * (re)allocate device workspace owned by the plan -- then call the *_planned
 * entry points every step. If the list cannot be tiled (a tile's neighbor set
 * exceeds 2559 particles or a row exceeds ~1000 entries, e.g. unsorted particle order) the plan is marked
 * invalid and the planned entry points run the generic kernel instead. Tiles
 * that are wide compared with the box, triclinic boxes, or calls without the
 * r_list_max hint re-apply the minimum image per pair (slower, still exact).
 * Results agree with the generic kernel to rounding. */
typedef struct azp_pair_plan azp_pair_plan; /* opaque */

typedef struct azp_pair_plan_info
    {
    int32_t valid;
    int32_t invalid_reason;      /* 0 none, 2: a tile lists more than 2559 distinct neighbors (or a row is too long);
                                    3, 4, 5: see azp_pair_plan_build_from_cells */
    uint32_t threads_per_particle;
    uint32_t tile_size;          /* particles per tile (workgroup) */
    uint32_t lds_slots;          /* staged-position capacity the kernel is instantiated for */
    uint32_t n_tiles;
    uint32_t max_stage;          /* largest staged set over all tiles */
    uint32_t max_row;            /* azp_pair_plan_build_from_cells: longest row found */
    uint64_t total_stage;        /* sum of staged-set sizes */
    uint64_t compiled_bytes;     /* size of the compiled 16-bit list */
    uint64_t builds;
    int32_t from_cells;          /* 1: compiled by azp_pair_plan_build_from_cells */
    uint32_t row_capacity;       /* ... with rows of this many entries */
    uint64_t list_id, head_id;   /* ... pass these as azp_pair_args.d_nlist / d_head_list to the *_planned
                                    entry points (the plan is its own list; there is no u32 list) */
    int32_t balanced;            /* 1: rows handed to the lanes in the order of their in-range lengths
                                    (azp_pair_plan_set_balance) */
    float core_radius;           /* row phases (one particle type): entries outside row class "core" were at least this
                                    far apart when the plan was built; 0: no such class */
    float sure_radius;           /* ... entries of row class "sure" at most this far apart; 0: no such class */
    uint32_t max_member_cells;   /* azp_pair_plan_build_from_cells, cells of the full list radius: the largest number of
                                    distinct cells the members of one tile sit in (beyond 128 the plan is refused with
                                    reason 4). It grows as the particles diffuse away from their sorted order: a caller
                                    that re-sorts them when it approaches the limit never sees the refusal */
    } azp_pair_plan_info;

int azp_pair_plan_create(azp_pair_plan** out);
void azp_pair_plan_destroy(azp_pair_plan* plan);
int azp_pair_plan_build(azp_pair_plan* plan, const azp_pair_args* args, void* stream);
/* Build option of azp_pair_plan_build_from_cells (default off): within every tile of 256 particles the
 * rows are handed to the force kernel's lanes longest in-range row first, so that each of its four waves
 * gets rows of similar in-range length. The tile kernels run as many heavy per-pair blocks per wave as
 * the LONGEST in-range row of the wave has entries; for potentials whose pair block is expensive and
 * whose rows are short and ragged (DPD thermostat: Philox per pair, 12.6 +- 3 in range) that is the
 * difference between max-over-256 and the quartile maxima. Results are those of the unbalanced plan. */
int azp_pair_plan_set_balance(azp_pair_plan* plan, int enabled);
/* Diagnostics: the staged-set size of the first n tiles of the last build (host copy, no device access);
 * returns the number of entries written. */
int azp_pair_plan_tile_stage(const azp_pair_plan* plan, uint32_t* out, uint32_t n);
/* (azp_pair_plan_build_from_cells, the plan compiled straight from the cell list, is declared with the
 * neighbor-list entry points below.) */
/* Build option: order every row bank-aware (conflict-poor LDS gathers; default on).
 * It adds ~10 % to the build and buys ~2-3 % per force call, so it pays for lists that
 * live for more than ~50 force calls; callers that rebuild more often turn it off. */
int azp_pair_plan_set_bank_order(azp_pair_plan* plan, int enabled);
int azp_pair_plan_query(const azp_pair_plan* plan, azp_pair_plan_info* info);
/* Diagnostics (copies from the device, synchronises): the row phases of the last build as means over the slices --
 * out[0] chunks that cover the core entries, out[1] chunks up to the end of the all-sure part of the rows, out[2]
 * chunks of the whole rows. */
int azp_pair_plan_phase_chunks(const azp_pair_plan* plan, float out[3]);

/* The plan cache behind azp_pair_forces_* (diagnostics and tests). */
typedef struct azp_auto_plan_stats
    {
    uint64_t calls;             /* calls that went through the cache                  */
    uint64_t compiles;          /* plan compilations (list changed / first use)      */
    uint64_t reuses;            /* calls served by an unchanged plan                 */
    uint64_t generic_fallbacks; /* calls whose list could not be tiled               */
    } azp_auto_plan_stats;
void azp_pair_auto_plan_get_stats(azp_auto_plan_stats* out);
void azp_pair_auto_plan_clear(void); /* frees every cached plan (synchronises the device) */

int azp_pair_forces_planned_perturbed_lennard_jones(azp_pair_plan* plan, const azp_pair_args* args,
                                                    const azp_plj_params* d_params, void* stream);
int azp_pair_forces_planned_hertz(azp_pair_plan* plan, const azp_pair_args* args, const azp_hertz_params* d_params,
                                  void* stream);
int azp_pair_forces_planned_expanded_yukawa(azp_pair_plan* plan, const azp_pair_args* args,
                                            const azp_yukawa_params* d_params, void* stream);
int azp_pair_forces_planned_colloid(azp_pair_plan* plan, const azp_pair_args* args, const azp_colloid_params* d_params,
                                    void* stream);
int azp_pair_forces_planned_dpd_conservative(azp_pair_plan* plan, const azp_pair_args* args,
                                             const azp_dpd_params* d_params, void* stream);

/* Mirrors hoomd::md::kernel::dpd_pair_args_t. */
typedef struct azp_dpd_args
    {
    azp_pair_args pair;
    const double* d_vel;   /* n_max x 4 (vx, vy, vz, mass) */
    const uint32_t* d_tag; /* n_max                        */
    uint64_t timestep;
    double deltaT;
    double T;              /* kT(timestep)                 */
    uint16_t seed;
    uint16_t _pad[3];
    } azp_dpd_args;

int azp_dpd_forces_general_weight(const azp_dpd_args* args, const azp_dpd_params* d_params, void* stream);
/* Tile-staged form: positions, velocities and tags of a tile's neighbors staged once in LDS
 * (csrc/xtiled.hpp); `plan` is compiled from args->pair with azp_pair_plan_build. Falls back to
 * the kernel above when the list cannot be tiled. */
int azp_dpd_forces_planned_general_weight(azp_pair_plan* plan, const azp_dpd_args* args, const azp_dpd_params* d_params,
                                          void* stream);

/* Mirrors hoomd::md::kernel::a_pair_args_t. */
typedef struct azp_aniso_args
    {
    azp_pair_args pair;
    const double* d_orientation; /* n_max x 4 */
    double* d_torque;            /* N x 4, overwritten */
    } azp_aniso_args;

int azp_aniso_forces_two_patch_morse(const azp_aniso_args* args, const azp_tpm_params* d_params, void* stream);
/* Tile-staged form: the patch director of every staged neighbor is computed once per tile
 * and kept in LDS with its position (csrc/xtiled.hpp). */
int azp_aniso_forces_planned_two_patch_morse(azp_pair_plan* plan, const azp_aniso_args* args, const azp_tpm_params* d_params,
                                             void* stream);

/* ---- bond forces ---- */

/* HOOMD group_storage<2>: one entry of the per-particle GPU bond table. */
typedef struct azp_bond_entry
    {
    uint32_t idx;  /* index of the other member of the bond */
    uint32_t type; /* bond type                             */
    } azp_bond_entry;

/* Mirrors hoomd::md::kernel::bond_args_t<2>. Table entry b of particle i is
 * d_gpu_bondlist[b * pitch + i], b < d_gpu_n_bonds[i]; d_gpu_bond_pos[b * pitch + i]
 * is i's position (0 or 1) inside that bond. */
typedef struct azp_bond_args
    {
    double* d_force;
    double* d_virial;
    uint64_t virial_pitch;
    uint32_t N;
    uint32_t n_max;
    const double* d_pos;
    azp_box box;
    const azp_bond_entry* d_gpu_bondlist;
    const uint32_t* d_gpu_bond_pos;
    const uint32_t* d_gpu_n_bonds;
    uint64_t pitch;
    uint32_t n_bond_types;
    uint32_t compute_virial;
    uint32_t block_size;
    uint32_t _pad;
    } azp_bond_args;

/* d_flags: one device word, set to 1 when an evaluator returned false
 * (invalid parameters) -- HOOMD turns that into "bond out of bounds". */
int azp_bond_forces_double_well(const azp_bond_args* args, const azp_dw_params* d_params, unsigned int* d_flags,
                                void* stream);
int azp_bond_forces_quartic(const azp_bond_args* args, const azp_quartic_params* d_params, unsigned int* d_flags,
                            void* stream);

/* ---- neighbor-list build (the step before the path) ----
 * Cell list -> full Verlet list in the layout the force kernels consume.
 * The reference consumes hoomd.md.nlist.Cell(buffer=...) (src/pytest/test_pair.py:337);
 * these kernels produce the same data: d_n_neigh, d_head_list (exclusive scan
 * of d_n_neigh, done by the caller) and d_nlist, rows compact and ordered by
 * cell, which is also the gather-friendly order for the force kernels.
 * Sequence: cell_assign -> (caller: stable sort of d_cell_of -> d_order,
 * d_cell_sorted) -> cell_bounds -> count -> (caller: exclusive scan) -> fill. */
typedef struct azp_cell_grid
    {
    double lo[3];        /* lower corner of the grid                     */
    double width[3];     /* cell width, >= largest r_list (>= half of it with azp_nlist_args.cell_subdivision = 2) */
    uint32_t dim[3];
    int32_t periodic[3]; /* 1: cell index wraps; 0: clamped (ghost slab) */
    } azp_cell_grid;

typedef struct azp_nlist_args
    {
    uint32_t N;                 /* rows are built for particles [0, N)        */
    uint32_t n_total;           /* particles binned (local + ghosts)          */
    const double* d_pos;        /* n_total x 4                                */
    azp_box box;
    azp_cell_grid grid;
    uint32_t ntypes;
    /* 0 / 1: cells at least as wide as the largest r_list (every entry point). 2: cells at least HALF as wide
     * (azp_nlist_cell_assign / cell_bounds / bin and azp_pair_plan_build_from_cells only: the plan compiler then
     * searches the 5 x 5 x 5 cells around a particle's own, cut down per particle to the cells its list sphere
     * reaches -- 0.39 x the candidate tests; azp_nlist_count / fill refuse such cells) */
    uint32_t cell_subdivision;
    const double* d_rlistsq;    /* ntypes^2, (r_cut + r_buff)^2; <= 0 disables the pair */
    uint32_t* d_cell_of;        /* n_total, written by cell_assign            */
    const uint32_t* d_cell_sorted; /* n_total, d_cell_of in ascending order   */
    const uint32_t* d_order;    /* n_total, particle indices in that order    */
    uint32_t* d_cell_start;     /* ncell + 1, written by cell_bounds          */
    const uint32_t* d_n_excl;   /* optional: N exclusion counts (NULL = none) */
    const uint32_t* d_excl;     /* entry e of particle i at e * excl_pitch + i */
    uint64_t excl_pitch;
    uint32_t* d_n_neigh;        /* N, written by count                        */
    const uint64_t* d_head_list; /* N, read by fill                           */
    uint32_t* d_nlist;          /* written by fill                            */
    /* single-pass mode of azp_nlist_fill (HOOMD's own protocol: rows of fixed
     * capacity, rebuilt with larger rows on overflow): when row_capacity > 0 the
     * fill also writes d_n_neigh[i] (the full count), stores at most row_capacity
     * entries per row and raises *d_max_neigh to the largest count seen
     * (atomic max; the caller zeroes it before the launch and must rebuild with
     * larger rows if it exceeds row_capacity). row_capacity = 0: rows are exact
     * (head_list from a scan of the count pass). */
    uint32_t row_capacity;
    uint32_t _pad2;
    uint32_t* d_max_neigh;
    } azp_nlist_args;

int azp_nlist_cell_assign(const azp_nlist_args* args, void* stream);
int azp_nlist_cell_bounds(const azp_nlist_args* args, void* stream);
/* The binning in one call: d_cell_of, d_order (the particles cell by cell, ascending index inside a cell: what a
 * stable sort by cell gives) and d_cell_start are written (d_cell_sorted is not used). A counting sort: histogram,
 * scan, scatter, and a per-cell sort that makes the result independent of the order of the atomics. Scratch:
 * d_cursor (ncell words), d_order_tmp (n_total words; also holds the per-workgroup totals of the scan when the
 * grid has more than 32,768 cells). Grids of many small cells (cell_subdivision = 2) get a multi-workgroup scan
 * and a thread-per-cell sort. */
int azp_nlist_bin(const azp_nlist_args* args, uint32_t* d_cursor, uint32_t* d_order_tmp, void* stream);
int azp_nlist_count(const azp_nlist_args* args, void* stream);
int azp_nlist_fill(const azp_nlist_args* args, void* stream);

/* The plan compiled straight from the cell list, in the pass that finds the neighbors
 * (csrc/pair_plan_cells.hip): no HOOMD-format list is produced or read. `cells`: the binned
 * particles as for azp_nlist_fill (d_pos, box, grid, d_rlistsq, d_cell_of, d_order,
 * d_cell_start, exclusions; d_n_neigh receives the row lengths; row_capacity = entries a row may
 * hold, 0 = 160; d_head_list / d_nlist / d_cell_sorted unused). `pair`: d_rcutsq, d_rinnersq,
 * r_list_max (required: sizes the buffer shells and decides where pairs must be re-imaged),
 * ntypes. Invalid plans (azp_pair_plan_query): invalid_reason 3 = a row exceeded row_capacity
 * (retry with max_row; hard limit 504), 2 = a tile stages more than 2559 particles, 4 / 5 =
 * particles not spatially sorted (the members of a tile of 256 sit in more than 128 cells, or
 * the cells around them number more than 512 / hold more than 8192 particles), 6 = tilted box
 * or more than 255 types -- build the u32 list and azp_pair_plan_build instead. The rows hold
 * the exact list plus, rarely, pairs a few 1e-6 r_list beyond it (single-precision acceptance
 * test with a margin that covers its own rounding); the force kernels' FP64 cutoff test ignores them. Synchronises the
 * stream once; owns device workspace (2 x row_capacity x 2 B per particle + the stage lists).
 * The *_planned entry points take list_id / head_id of azp_pair_plan_query as d_nlist /
 * d_head_list. */
int azp_pair_plan_build_from_cells(azp_pair_plan* plan, const azp_nlist_args* cells, const azp_pair_args* pair, void* stream);

/* Rebuild criterion (HOOMD NeighborList::distanceCheck restated): sets *d_flag to 1
 * when any of the n particles moved farther than sqrt(max_dist_sq) from its position
 * at the last build (minimum image in `box`); the caller zeroes *d_flag beforehand.
 * d_max_dist_sq_bits (optional, may be NULL; zeroed by the caller): receives the bit
 * pattern of the largest squared displacement as a double (atomic max on the bits,
 * which order like the values for non-negative doubles) -- the displacement bound the
 * planned force kernels accept (azp_pair_args.displacement_bound). */
int azp_nlist_distance_check(uint32_t n, const double* d_pos, const double* d_pos_at_build, const azp_box* box,
                             double max_dist_sq, uint32_t* d_flag, unsigned long long* d_max_dist_sq_bits, void* stream);
/* The same check that also writes every particle's own displacement (single precision, rounded up) to
 * d_displacement[0 .. n): the per-particle bounds azp_pair_args.d_displacement takes. */
int azp_nlist_displacements(uint32_t n, const double* d_pos, const double* d_pos_at_build, const azp_box* box,
                            double max_dist_sq, uint32_t* d_flag, unsigned long long* d_max_dist_sq_bits,
                            float* d_displacement, void* stream);

/* Sort keys of the particle sorter (the role of HOOMD's SFC sorter, which the tile plan relies
 * on): key of particle i = index of its cell along a blocked curve over a dims[0] x dims[1] x
 * dims[2] grid (block^3 cells per block, blocks and the cells inside them in row-major order), or,
 * with block = 0, the index along the Hilbert curve of a 2^b x 2^b x 2^b grid stretched over the box
 * (2^b >= the largest of dims, <= 1024); positions wrapped into the orthorhombic frame of `box`. */
int azp_sorter_keys(uint32_t n, const double* d_pos, const azp_box* box, const uint32_t* dims, uint32_t block, int32_t* d_keys,
                    void* stream);

/* ---- halo pack: dst[k, :] = src[idx[k], :] for k < n, rows of
 * row_doubles doubles (4 for positions / velocities / orientations). The send buffer of
 * the per-step ghost exchange; replaces the pack half of HOOMD's CommunicatorGPU. */
int azp_halo_pack(uint32_t n, const double* d_src, const int64_t* d_idx, uint32_t row_doubles, double* d_dst, void* stream);

/* Several per-particle arrays in ONE send buffer (DPD: positions + velocities + tags; aniso:
 * positions + orientations), so that a step needs a single collective. Packed row k holds, field
 * after field and each starting on an 8-byte boundary, row d_idx[k] of every field's array;
 * packed_row_bytes = sum of the fields' row sizes rounded up to 8. unpack writes packed row k to
 * row k of each field's destination (the caller passes the address of its first ghost row). */
#define AZP_HALO_MAX_FIELDS 4
typedef struct azp_halo_field
    {
    void* d_data;       /* pack: the source array; unpack: the first destination row */
    uint32_t row_bytes; /* bytes per particle, multiple of 4                         */
    uint32_t _pad;
    } azp_halo_field;
int azp_halo_pack_fields(uint32_t n, uint32_t n_fields, const azp_halo_field* fields, const int64_t* d_idx, void* d_packed,
                         uint32_t packed_row_bytes, void* stream);
int azp_halo_unpack_fields(uint32_t n, uint32_t n_fields, const azp_halo_field* fields, const void* d_packed,
                           uint32_t packed_row_bytes, void* stream);

/* ---- one-body harmonic barriers ----
 * Replaces the reference's own kernel driver
 *   azplugins::gpu::compute_harmonic_barrier<Evaluator>(...)   (src/HarmonicBarrierGPU.cuh:49-140,
 *   instantiated at src/HarmonicBarrierGPUKernel.cu.inc) for PlanarBarrierEvaluator
 *   (src/PlanarBarrierEvaluator.h:36-48) and SphericalBarrierEvaluator
 *   (src/SphericalBarrierEvaluator.h:36-51).
 * d_params: one (k, offset) pair per particle type (HOOMD Scalar2). Positions are
 * wrapped into the box before evaluation (src/HarmonicBarrier.h:167-169). The
 * virial is not computed by the reference (set to zero there); d_virial may be NULL. */
typedef struct azp_barrier_args
    {
    double* d_force;      /* N x 4, overwritten */
    double* d_virial;     /* 6 x virial_pitch, zeroed if not NULL */
    uint64_t virial_pitch;
    uint32_t N;
    uint32_t ntypes;
    const double* d_pos;  /* N x 4 */
    azp_box box;
    const double* d_params; /* ntypes x 2: k, offset */
    double location;      /* H (planar: y position) or R (spherical: radius) at this timestep */
    uint32_t block_size;
    uint32_t _pad;
    } azp_barrier_args;

int azp_external_planar_harmonic_barrier(const azp_barrier_args* args, void* stream);
int azp_external_spherical_harmonic_barrier(const azp_barrier_args* args, void* stream);
/* host-side validity checks of the evaluators (PlanarBarrierEvaluator::valid,
 * SphericalBarrierEvaluator::valid): 1 valid, 0 invalid */
int azp_planar_barrier_valid(double H, const azp_box* box);
int azp_spherical_barrier_valid(double R, const azp_box* box);

/* ---- velocity-Verlet NVE step ----
 * HOOMD's hoomd.md.methods.ConstantVolume without thermostat (the dummy
 * integrator of every reference test, src/pytest/test_pair.py:325-327), restated:
 * step one: v += a dt/2, x += v dt, wrap into the box (image counters updated);
 * step two: v += a dt/2 with the new net force. a = F / m, m = vel.w. */
typedef struct azp_nve_args
    {
    double* d_pos;            /* N x 4 (type in w is preserved) */
    double* d_vel;            /* N x 4 (vx, vy, vz, mass) */
    const double* d_net_force; /* N x 4 */
    int32_t* d_image;         /* N x 3 periodic image counters, may be NULL */
    azp_box box;
    double dt;
    uint32_t N;
    uint32_t block_size;
    } azp_nve_args;

int azp_integrate_nve_step_one(const azp_nve_args* args, void* stream);
int azp_integrate_nve_step_two(const azp_nve_args* args, void* stream);
/* Step two of one time step and step one of the next in one kernel: for a loop in which nothing reads the
 * velocities between the two (HOOMD calls integrateStepTwo and the next integrateStepOne back to back unless an
 * updater or analyzer is due). Same arithmetic in the same order as the two calls, one pass over the arrays. */
int azp_integrate_nve_step_two_one(const azp_nve_args* args, void* stream);
/* Net force of up to 8 force arrays (N x 4 each; HOOMD: Integrator::computeNetForce) in one pass:
 * d_out[i] = d_arrays[0][i] + d_arrays[1][i] + ... (d_arrays: HOST array of n_arrays device pointers). */
int azp_sum_forces(uint32_t n_rows, uint32_t n_arrays, const double* const* d_arrays, double* d_out, void* stream);

/* Rotational degrees of freedom of the same step ("+ rotational for
 * aniso"; the reference's aniso test gives its particles a moment of inertia,
 * src/pytest/test_pair_aniso.py:113-140): HOOMD's integrate_rotational_dof path restated -- the
 * symplectic NO_SQUISH quaternion scheme. q = d_orientation (scalar first), p = d_angmom (angular
 * momentum quaternion; body angular momentum = 1/2 conj(q) p), per-particle principal moments
 * d_inertia (N x 3; an axis with zero moment is not integrated), d_net_torque (N x 4, space
 * frame). step one: p += dt q t_body, free rotations about axes 3, 2, 1, 2, 3, q renormalised;
 * step two: p += dt q t_body. HOOMD-blue's source is absent here: PARITY UNPINNED (pinned by
 * conservation properties, tests/test_gpu_external_nve.py). */
typedef struct azp_nve_rot_args
    {
    double* d_orientation;      /* N x 4 */
    double* d_angmom;           /* N x 4 */
    const double* d_inertia;    /* N x 3 */
    const double* d_net_torque; /* N x 4 */
    double dt;
    uint32_t N;
    uint32_t block_size;
    } azp_nve_rot_args;

int azp_integrate_nve_rot_step_one(const azp_nve_rot_args* args, void* stream);
int azp_integrate_nve_rot_step_two(const azp_nve_rot_args* args, void* stream);

/* ---- misc ---- */
int azp_version(void);                    /* major * 1000 + minor       */
const char* azp_status_string(int status);
/* Resolved launch configuration of the last pair-force call on this thread
 * (for benchmarks / profiling reports). */
void azp_last_launch(uint32_t* block_size, uint32_t* threads_per_particle, uint32_t* grid, uint32_t* lds_bytes);
/* Process-wide switches of the tile kernels, for A/B measurements (results are identical either way):
 * AZP_TUNE_ROW_PHASES (default 0: it measures 1.5 % slower on the north star although it issues fewer
 * instructions): the test-free / core-test-free parts of a row (csrc/pair_tiled.hpp);
 * AZP_TUNE_LOCAL_BOUND (default 1): azp_pair_args.d_displacement is used when given;
 * AZP_TUNE_SPLIT_TILES (default 0: measured 5 % slower; AZP_SPLIT_TILES=1): plans whose largest staged set needs more than
 * 1,664 LDS slots are launched in two parts, the tiles that fit the 1,664-slot variant (four workgroups per CU) and the rest.
 * Returns the previous value, or -1 for an unknown key. The environment variables AZP_ROW_PHASES=1 /
 * AZP_LOCAL_BOUND=0 set the initial values. */
enum { AZP_TUNE_ROW_PHASES = 1, AZP_TUNE_LOCAL_BOUND = 2, AZP_TUNE_SPLIT_TILES = 3 };
int azp_tuning_set(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* AZP_H_ */
