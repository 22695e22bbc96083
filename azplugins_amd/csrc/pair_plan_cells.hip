// pair_plan_cells.hip -- the tile plan compiled STRAIGHT FROM THE CELL LIST, in the pass
// that finds the neighbors (SURVEY 8f row N1 + the plan of pair_plan.hpp in one kernel).
//
// pair_plan.hip compiles a plan from a finished HOOMD-format neighbor list: it has to
// rediscover, per tile, which particles the rows mention (LDS hash set + sort) and walks
// the u32 rows twice. An MD run that owns its neighbor search does not need the u32 list at
// all. One workgroup of 256 threads per tile of 256 consecutive particles, THREAD = MEMBER
// from the first phase to the last:
//
//   0. every member puts the 27 cells around its own into an LDS set (first member of a cell
//      only); the set, sorted by cell number, is the tile's candidate space: cells that are
//      consecutive along x are consecutive candidate ranges;
//   1. per distinct member cell, the <= 18 runs of candidates (3 x 3 rows of cells along x,
//      a row that wraps through the periodic boundary is two runs);
//   2. the candidates are staged in LDS in batches of 1,024 (single-precision positions
//      relative to the tile's reference particle); every thread walks the runs of its member
//      through the batch, four candidates per trip (packed FP32 math). An accepted candidate is appended to the
//      member's raw row (candidate number | class, 2 B, global scratch laid out [entry][member]
//      so that the 64 lanes of a wave write and later read one cache line); a second walk over
//      the finished row counts its classes and marks its candidates in a bitmap;
//   3. the bitmap, compacted (prefix of popcounts), gives every used candidate its LDS slot
//      of the force kernel; the stage list is written;
//   4. every thread turns its raw row into the force kernel's compiled row: entries ordered
//      class by class (a running cursor per class, seeded with the class totals of phase 2),
//      slot byte offsets, 16-byte chunks in the kernel's lane order, chunk counts per class
//      boundary for the displacement bound.
//
// No hash set of particles, no sort, no u32 rows, no host scan (slices have a fixed chunk
// capacity), no cross-lane compaction anywhere (each lane owns its row), one synchronisation
// (flags). The accepted set is a SUPERSET of the exact list by a hair (single-precision test
// with a 1e-5 margin on r_list^2 plus a bound on the rounding of the staged coordinates, see rl_extra):
// extra entries are buffer entries the force kernel's exact
// FP64 cutoff test ignores. Classes are conservative exactly as in pair_plan.hip.
//
// A second form of phases 0 - 2 works on cells of HALF the list radius (template parameter HALF, azp_nlist_args.
// cell_subdivision = 2; see PC_BATCH_H below): exact, 0.39 x the candidate tests, not faster -- off by default.
//
// Limits (the plan is marked invalid and the caller falls back to the list-based path): the
// members of a tile sit in more than 128 cells, or the cells around them number more than 512
// or hold more than 8,192 particles (particles not spatially sorted); more than 2,559 staged
// particles per tile; rows longer than the row capacity (reported back so that the caller
// retries with longer rows, HOOMD's own protocol for its list); tilted boxes.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "azp_device.hpp"
#include "pair_plan.hpp"

namespace azp
{
// Profiling variant (make variant SRC=pair_plan_cells NAME=pcprof DEFS=-DAZP_PLAN_CELLS_PROFILE, loaded through
// AZP_LIB_PATH): AZP_PLAN_CELLS_STOP=p in the environment leaves the kernel after phase p (1, 2, 4; + 256: no raw-row
// stores, + 512: no row walk) and the build reports the plan invalid (reason 7). Compiled out of libazp.so.
#ifdef AZP_PLAN_CELLS_PROFILE
#define PC_PROFILE_AND(x) && (x)
#else
#define PC_PROFILE_AND(x)
#endif

constexpr uint32_t PC_THREADS = 256;
constexpr uint32_t PC_BATCH = 1024;     // candidates staged at a time
constexpr uint32_t PC_MAXCAND = 8192;   // candidates of a tile's cells (13 bits of a raw entry; 12 when they suffice)
constexpr uint32_t PC_MAXCELLS = 512;   // distinct cells next to a tile's members
constexpr uint32_t PC_MAXMC = 128;      // distinct cells of the members themselves (more LDS here costs a workgroup per CU: 1.40 -> 1.66 ms)
constexpr uint32_t PC_RUNS = 18;        // 3 x 3 rows of cells, each at most two runs
constexpr uint32_t PC_SETA = 512, PC_SETB = 1024; // hash sets: member cells, their neighbor cells
constexpr uint32_t PC_ROWMAX = 512;     // largest row capacity + 8 (PLAN_ROWBUF)
constexpr uint32_t PC_EMPTY = 0xffffffffu;
constexpr uint32_t PC_CTAB = 1024;      // bins of the r^2 -> class table (one particle type)
constexpr uint32_t PC_WALK = 8;         // raw-row entries a thread has in flight in the row walks (latency-bound otherwise; 16 buys nothing more)
// Half-width cells (cells of width >= r_list / 2, azp_nlist_args.cell_subdivision = 2): the 5 x 5 x 5 cells around a member's
// own hold 15.6 r_list^3 instead of 27, and with cells that small it pays to cut the search per MEMBER: a row of cells
// (fixed y, z) farther from the member than r_list is skipped, the others are clipped along x to the cells the sphere
// reaches -- 84 of the 125 cells on average, 0.39 x the candidate tests of the full-width form. The tile's cells are a
// dense local grid (the members' cells + 2 on every side, at most PC_MAXGRID of them): no hash sets, no sort.
constexpr uint32_t PC_BATCH_H = 768;    // candidates staged at a time
constexpr uint32_t PC_MAXGRID = 2048;   // cells of a tile's local grid
constexpr uint32_t PC_RUNS_H = 25;      // 5 x 5 rows of cells, one run of candidates each

struct PlanCellsKArgs
    {
    const double* pos;
    const double* rlistsq;
    const double* rcutsq;
    const double* rinnersq;
    const uint32_t* cell_of;
    const uint32_t* order;
    const uint32_t* cell_start;
    const uint32_t* n_excl;
    const uint32_t* excl;
    uint64_t excl_pitch;
    uint32_t* n_neigh;
    uint16_t* raw;          // n_tiles x row_cap x 256: entry k of member h of a tile at [k][h]
    uint32_t* tile_nstage;
    uint64_t* tile_head;
    uint32_t* stage_idx;
    uint32_t* slice_K;
    uint32_t* slice_Kend;
    uint32_t* slice_Kphase; // [2][n_slices]: chunks covering class core; chunks holding only core / sure entries in every row
    uint64_t* slice_head;
    uint4* cnl;
    uint32_t* flags;        // [0] sure radius, [1] invalid, [2] max staged set, [3] shell width, [4] most member cells of a tile, [5] longest row, [6] reason, [7] core radius
    uint8_t* perm;          // balanced plans: lane -> member of the tile (n_tiles x 256); NULL: lane = member
    double r_list_max;
    BoxDev box;
    int dim[3], periodic[3];
    uint32_t N, n_total, ntypes, n_tiles;
    uint32_t row_cap;       // multiple of 8
    uint32_t stage_stride;
    uint32_t stop_after;    // AZP_PLAN_CELLS_PROFILE builds only (tools/plan_cells_probe.py): leave after this phase
    double glo[3], gwinv[3]; // half-width cells: the grid's lower corner, 1 / cell width
    float gw[3];             // cell widths
    };

// Distinct coordinates of the cells next to cell c along one axis, ascending: c - 1, c, c + 1
// wrapped (periodic) or clipped (not periodic); an axis of up to 3 cells lists every cell once.
__device__ __forceinline__ int axis_neighbors(int c, int dim, int periodic, int (&out)[3])
    {
    int n = 0;
    if (!periodic)
        {
        for (int o = -1; o <= 1; ++o)
            if (c + o >= 0 && c + o < dim)
                out[n++] = c + o;
        return n;
        }
    if (dim <= 3)
        {
        for (int q = 0; q < dim; ++q)
            out[n++] = q;
        return n;
        }
    int v0 = (c - 1 + dim) % dim, v1 = c, v2 = (c + 1) % dim;
    // ascending order (at most one of the three wrapped)
    if (v0 > v1) { const int t = v0; v0 = v1; v1 = v2; v2 = t; }       // c = 0: {dim - 1, 0, 1} -> {0, 1, dim - 1}
    else if (v2 < v1) { const int t = v2; v2 = v1; v1 = v0; v0 = t; }  // c = dim - 1: {dim - 2, dim - 1, 0} -> {0, dim - 2, dim - 1}
    out[0] = v0; out[1] = v1; out[2] = v2;
    return 3;
    }

// open-addressing set in LDS; returns 1 when this call put the key in, 0 when it was there, 2 when the set is full
__device__ __forceinline__ uint32_t set_insert(uint32_t* set, uint32_t size, uint32_t key)
    {
    uint32_t h = (key * 2654435761u) >> 12;
    for (uint32_t probe = 0; probe < size; ++probe)
        {
        h &= size - 1u;
        const uint32_t prev = atomicCAS(&set[h], PC_EMPTY, key);
        if (prev == PC_EMPTY)
            return 1u;
        if (prev == key)
            return 0u;
        ++h;
        }
    return 2u;
    }

// class of a listed pair at separation^2 rsq (pair_plan.hpp): core, near (inside the cutoff or
// too close to call), PLAN_CLS_SHELL0 + s buffer shell s <=> certainly >= r_cut + s w; rcsq_m = r_cut^2 * 1.0001,
// rcw = r_cut / w, rscale = 0.99995 / w. (Class sure is cut out of near by the caller: it needs an UPPER bound on r.)
__device__ __forceinline__ uint32_t pair_class(float rsq, float rcsq_m, float rin, float rcw, float rscale, float fmax_shell)
    {
    const float shf = floorf(__builtin_fmaf(__builtin_amdgcn_sqrtf(rsq), rscale, -rcw));
    const uint32_t shell = PLAN_CLS_SHELL0 + (uint32_t)fminf(fmaxf(shf, 0.f), fmax_shell); // NaN (w = 0) -> shell 0
    return !(rsq >= rcsq_m) ? ((rsq < rin) ? PLAN_CLS_CORE : PLAN_CLS_NEAR) : shell;
    }

// SINGLE: one particle type (cutoffs are constants, classes come from a table). HALF: cells of half the list radius
// (local grid, per-member runs); else cells of the full list radius (27 cells around each member cell).
template<bool SINGLE, bool HALF>
__global__ void __launch_bounds__(PC_THREADS) plan_cells_kernel(const PlanCellsKArgs a)
    {
    // One region, reused. Phase 0: hash sets + unsorted cells. Phase 2: the candidates of the
    // current batch (x, y, z, particle index; 16 B) + their types. Phases 3, 4: candidate -> slot.
    // Full-width cells: the per-thread class counters / cursors live behind it through phases 2 .. 4.
    // Half-width cells: the per-thread run tables and the prefix of the local grid's cell populations sit behind the
    // batch; counters, bitmap and slot table take the place of batch and run tables once the tests are done.
    constexpr uint32_t BATCH = HALF ? PC_BATCH_H : PC_BATCH;
    constexpr uint32_t CSTRIDE = BATCH + 4;                  // pad entries: candidates are read two / four at a time
    constexpr uint32_t CAND_BYTES = (CSTRIDE * 16 + BATCH + 32 + 15) / 16 * 16; // x | y | z | particle index | types
    constexpr uint32_t CUR_BYTES = PLAN_CLASSES * PC_THREADS * 2;
    constexpr uint32_t RUNG_OFF = CAND_BYTES;                                  // HALF: u16 [run][thread], first candidate
    constexpr uint32_t RUNL_OFF = RUNG_OFF + PC_RUNS_H * PC_THREADS * 2;       // HALF: u8 [run][thread], candidates in the run
    constexpr uint32_t PRE_OFF = RUNL_OFF + PC_RUNS_H * PC_THREADS;            // HALF: u16 [PC_MAXGRID + 1] candidates before each cell
    constexpr uint32_t CUR_OFF = HALF ? 2 * PC_MAXCAND : CAND_BYTES;
    constexpr uint32_t USED_OFF = CUR_OFF + CUR_BYTES;                         // HALF: bitmap, word bases, sort keys of balanced plans
    constexpr uint32_t WBASE_OFF = USED_OFF + PC_MAXCAND / 8;
    constexpr uint32_t KEY_OFF = WBASE_OFF + (PC_MAXCAND / 32 + 4) * 4;
    constexpr uint32_t REGION = HALF ? PRE_OFF + (PC_MAXGRID + 8) * 2 : CUR_OFF + CUR_BYTES; // 22,592 (full) / 36,448 (half)
    static_assert(HALF || CAND_BYTES >= 2 * PC_MAXCAND, "slot table does not fit");
    static_assert(HALF || CAND_BYTES >= (PC_SETA + PC_SETB + PC_MAXCELLS) * 4, "hash sets do not fit");
    static_assert(!HALF || KEY_OFF + PC_THREADS * 4 <= PRE_OFF, "half-width cells: the late tables overlap the grid prefix");
    static_assert(!HALF || PC_MAXGRID * 4 + 512 <= CAND_BYTES, "half-width cells: scan scratch does not fit");
    __shared__ __attribute__((aligned(16))) unsigned char s_region[REGION];
    float* cx = reinterpret_cast<float*>(s_region);
    float* cy = cx + CSTRIDE;
    float* cz = cy + CSTRIDE;
    uint32_t* cj = reinterpret_cast<uint32_t*>(cz + CSTRIDE);
    unsigned char* ctype = s_region + CSTRIDE * 16;
    uint32_t* setA = reinterpret_cast<uint32_t*>(s_region);
    uint32_t* setB = setA + PC_SETA;
    uint32_t* s_tmp = setB + PC_SETB;
    uint16_t* s_slot = reinterpret_cast<uint16_t*>(s_region);
    uint16_t* s_cur = reinterpret_cast<uint16_t*>(s_region + CUR_OFF); // [class][thread]
    __shared__ uint32_t s_used_f[HALF ? 1 : PC_MAXCAND / 32];
    __shared__ uint32_t s_wordbase_f[HALF ? 1 : PC_MAXCAND / 32 + 1];
    uint32_t* s_used = HALF ? reinterpret_cast<uint32_t*>(s_region + USED_OFF) : s_used_f;
    uint32_t* s_wordbase = HALF ? reinterpret_cast<uint32_t*>(s_region + WBASE_OFF) : s_wordbase_f;
    uint32_t* s_cells = reinterpret_cast<uint32_t*>(s_region + 8192); // phase 1 only: the cells in ascending order
    static_assert(HALF || 8192 + PC_MAXCELLS * 4 <= CAND_BYTES, "sorted cells do not fit");
    __shared__ uint32_t s_cfirst[HALF ? 1 : PC_MAXCELLS], s_coff[HALF ? 1 : PC_MAXCELLS + 8];
    __shared__ uint32_t s_mcl[HALF ? 1 : PC_MAXMC];                // the distinct cells of the members
    __shared__ uint32_t s_runs[HALF ? 1 : PC_MAXMC][PC_RUNS];      // per member cell: candidate ranges, g0 | g1 << 16
    uint16_t* s_rung = reinterpret_cast<uint16_t*>(s_region + RUNG_OFF);
    unsigned char* s_runl = s_region + RUNL_OFF;
    uint16_t* s_pre = reinterpret_cast<uint16_t*>(s_region + PRE_OFF);
    uint32_t* s_need = reinterpret_cast<uint32_t*>(s_region);                       // HALF, phases 0-1: cells some member reaches
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_region + 512);                  // HALF, phase 1: population of the needed cells
    __shared__ int s_glo[3], s_ghi[3];
    __shared__ uint32_t s_wtot[4];
    __shared__ unsigned char s_ctab[SINGLE ? PC_CTAB : 4];
    __shared__ uint32_t s_kend[4][PLAN_SHELLS + 1], s_smax[4], s_kcore[4], s_ksure[4];
    __shared__ float s_rcutsq[64], s_rinnersq[64], s_rcw[64], s_rlistsq[64];
    __shared__ uint32_t s_wide, s_bad, s_ncells, s_nmc, s_cmax;

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // (grid padded to a multiple of 8: each XCD compiles one contiguous eighth of the tiles, whose
    // candidate cells overlap, out of its own L2)
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= a.n_tiles)
        return;
    const uint32_t first = tile * 256u;
    const uint32_t count = min(256u, a.N - first);
    const bool member = tid < count;
    const uint32_t i = first + (member ? tid : 0u);
    const bool rc_cached = a.ntypes <= 8;
    const uint32_t ntp = a.ntypes * a.ntypes;
    const int dimx = a.dim[0], dimy = a.dim[1], dimz = a.dim[2];

    // shell width (as pair_plan.hip): r_buff / PLAN_SHELLS with r_buff = (r_list_max - r_cut_max) / 2
    float shell_w = 0.f;
        {
        double rc_max_sq = 0.0;
        for (uint32_t t = 0; t < ntp; ++t)
            rc_max_sq = fmax(rc_max_sq, a.rcutsq[t]);
        const double w = (a.r_list_max > 0.0) ? 0.5 * (a.r_list_max - sqrt(rc_max_sq)) / PLAN_SHELLS : 0.0;
        shell_w = (w > 0.0) ? (float)w : 0.f;
        if (blockIdx.x == 0 && tid == 0)
            a.flags[3] = (uint32_t)__float_as_int(shell_w);
        }
    const float shell_winv = (shell_w > 0.f) ? 1.0f / shell_w : 0.f;
    const float rscale = 0.99995f * shell_winv;

    // ---- phase 0: members, their cells, the set of cells next to them ----
    if (tid < 4)
        {
        s_smax[tid] = 0;
        s_kcore[tid] = 0;
        s_ksure[tid] = 0xffffffffu;
        for (uint32_t sh = 0; sh <= PLAN_SHELLS; ++sh)
            s_kend[tid][sh] = 0;
        }
    if (tid == 0)
        {
        s_wide = 0;
        s_bad = 0;
        s_ncells = 0;
        s_nmc = 0;
        s_cmax = 0;
        }
    if (!HALF)
        {
        for (uint32_t t = tid; t < PC_MAXCAND / 32; t += PC_THREADS)
            s_used[t] = 0;
        for (uint32_t t = tid; t < PC_SETA + PC_SETB; t += PC_THREADS)
            setA[t] = PC_EMPTY;
        }
    else
        {
        if (tid < 3)
            {
            s_glo[tid] = 0x7fffffff;
            s_ghi[tid] = -0x7fffffff;
            }
        for (uint32_t t = tid; t < PC_MAXGRID / 32; t += PC_THREADS)
            s_need[t] = 0;
        }
    if (!SINGLE && rc_cached && tid < ntp)
        {
        s_rcutsq[tid] = (float)a.rcutsq[tid] * 1.0001f;
        s_rinnersq[tid] = a.rinnersq ? (float)a.rinnersq[tid] : 0.f;
        s_rcw[tid] = sqrtf(fmaxf((float)a.rcutsq[tid], 0.f)) * shell_winv;
        // single-precision test with a margin: a superset of the exact list (r^2 <= r_list^2)
        s_rlistsq[tid] = a.rlistsq[tid] > 0.0 ? (float)a.rlistsq[tid] * 1.00001f : -1.f;
        }
    const double3 cref = load_scalar3_of4(a.pos, first);
    __syncthreads();
    float xi = 0.f, yi = 0.f, zi = 0.f;
    uint32_t mytype = 0, mycell = 0;
    int hd[3] = {0, 0, 0};          // half-width cells: my cell relative to the first member's
    float rel[3] = {0.f, 0.f, 0.f}; // and my position inside it, in cell widths
    int rcell[3] = {0, 0, 0};
    if (HALF)
        {
        const uint32_t c0 = a.cell_of[first];
        rcell[0] = (int)(c0 % (uint32_t)dimx);
        rcell[1] = (int)((c0 / (uint32_t)dimx) % (uint32_t)dimy);
        rcell[2] = (int)(c0 / (uint32_t)(dimx * dimy));
        }
    if (member)
        {
        const double4 p = load_scalar4(a.pos, i);
        double x = p.x - cref.x, y = p.y - cref.y, z = p.z - cref.z;
        // orthorhombic boxes only (the host refuses tilted ones)
        if (a.box.px) x = __builtin_fma(-a.box.Lx, rint(x * a.box.Lxinv), x);
        if (a.box.py) y = __builtin_fma(-a.box.Ly, rint(y * a.box.Lyinv), y);
        if (a.box.pz) z = __builtin_fma(-a.box.Lz, rint(z * a.box.Lzinv), z);
        // the staged images are minimum images for every member only if the tile and its
        // list radius fit into half the box (as in the force kernel); else re-image pairs
        if (!(a.r_list_max > 0.0) || (a.box.px && fabs(x) + a.r_list_max >= 0.5 * a.box.Lx)
            || (a.box.py && fabs(y) + a.r_list_max >= 0.5 * a.box.Ly) || (a.box.pz && fabs(z) + a.r_list_max >= 0.5 * a.box.Lz))
            s_wide = 1;
        xi = (float)x; yi = (float)y; zi = (float)z;
        atomicMax(&s_cmax, (uint32_t)__float_as_int(fmaxf(fabsf(xi), fmaxf(fabsf(yi), fabsf(zi))))); // (positive floats order as integers)
        mytype = (uint32_t)type_from_w(p.w);
        mycell = a.cell_of[i];
        if constexpr (HALF)
            {
            // my cell relative to the cell of the tile's first member (nearest image), my place inside it
            const int cc[3] = {(int)(mycell % (uint32_t)dimx), (int)((mycell / (uint32_t)dimx) % (uint32_t)dimy), (int)(mycell / (uint32_t)(dimx * dimy))};
            const double pk[3] = {p.x, p.y, p.z};
#pragma unroll
            for (int k = 0; k < 3; ++k)
                {
                int d = cc[k] - rcell[k];
                if (a.periodic[k])
                    {
                    const int half = a.dim[k] >> 1;
                    d = (d > half) ? d - a.dim[k] : ((d < -half) ? d + a.dim[k] : d);
                    }
                hd[k] = d;
                atomicMin(&s_glo[k], d);
                atomicMax(&s_ghi[k], d);
                const double u = (pk[k] - a.glo[k]) * a.gwinv[k];
                const double r = a.periodic[k] ? u - floor(u) : u - (double)cc[k]; // (a clamped particle lies beyond its cell: taking the nearest point of the cell is the conservative side)
                rel[k] = fminf(fmaxf((float)r, 0.f), 1.f);
                }
            }
        else
            {
            if (set_insert(setA, PC_SETA, mycell) == 1u)
                {
                // first member seen in this cell: its neighbor cells join the tile's cell set
                const uint32_t imc = atomicAdd(&s_nmc, 1u);
                if (imc < PC_MAXMC)
                    s_mcl[imc] = mycell;
                const int ccx = (int)(mycell % (uint32_t)dimx), ccy = (int)((mycell / (uint32_t)dimx) % (uint32_t)dimy), ccz = (int)(mycell / (uint32_t)(dimx * dimy));
                int nzs[3], nys[3], nxs[3];
                const int cz_n = axis_neighbors(ccz, dimz, a.periodic[2], nzs), cy_n = axis_neighbors(ccy, dimy, a.periodic[1], nys);
                const int cx_n = axis_neighbors(ccx, dimx, a.periodic[0], nxs);
                for (int qz = 0; qz < 3; ++qz)
                    for (int qy = 0; qy < 3; ++qy)
                        for (int qx = 0; qx < 3; ++qx)
                            {
                            if (qz >= cz_n || qy >= cy_n || qx >= cx_n || imc >= PC_MAXMC || *(volatile uint32_t*)&s_ncells > PC_MAXCELLS)
                                continue; // (beyond a limit: particles not sorted, stop filling the set)
                            const uint32_t nc = (uint32_t)((nzs[qz] * dimy + nys[qy]) * dimx + nxs[qx]);
                            const uint32_t r = set_insert(setB, PC_SETB, nc);
                            if (r == 1u)
                                {
                                const uint32_t idx = atomicAdd(&s_ncells, 1u);
                                if (idx < PC_MAXCELLS)
                                    s_tmp[idx] = nc;
                                }
                            else if (r == 2u)
                                atomicAdd(&s_ncells, PC_MAXCELLS + 1u);
                            }
                }
        
            }
        }
    __syncthreads();
    uint32_t ncell_blk = 0, n_mc = 0, NC = 0;
    // half-width cells: the tile's local grid (extent E, lower corner at cell gorg in unwrapped cell coordinates), my cell in it
    int E[3] = {1, 1, 1}, gorg[3] = {0, 0, 0}, lc3[3] = {0, 0, 0};
    uint32_t G = 0, nr = 0;
    // local cell -> cell of the grid
    auto cell_of_local = [&](uint32_t lc) -> uint32_t
        {
        const uint32_t t = lc / (uint32_t)E[0];
        int u[3] = {gorg[0] + (int)(lc - t * (uint32_t)E[0]), gorg[1] + (int)(t % (uint32_t)E[1]), gorg[2] + (int)(t / (uint32_t)E[1])};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            u[k] = (u[k] < 0) ? u[k] + a.dim[k] : ((u[k] >= a.dim[k]) ? u[k] - a.dim[k] : u[k]); // (periodic axes only leave the range, by dim / 2 + 2 <= dim at most)
        return (uint32_t)((u[2] * dimy + u[1]) * dimx + u[0]);
        };
    // candidate number -> particle
    auto cand_particle = [&](uint32_t g) -> uint32_t
        {
        if constexpr (HALF)
            {
            uint32_t sl = 0, sh = G; // last cell with pre[cell] <= g (empty and unused cells before it share its value)
            while (sh - sl > 1)
                {
                const uint32_t mid = (sl + sh) >> 1;
                if (s_pre[mid] <= g) sl = mid; else sh = mid;
                }
            return a.order[a.cell_start[cell_of_local(sl)] + (g - s_pre[sl])];
            }
        else
            {
            uint32_t sl = 0, sh = ncell_blk; // cell s with coff[s] <= g < coff[s + 1]
            while (sh - sl > 1)
                {
                const uint32_t mid = (sl + sh) >> 1;
                if (s_coff[mid] <= g) sl = mid; else sh = mid;
                }
            return a.order[s_cfirst[sl] + (g - s_coff[sl])];
            }
        };
    if constexpr (HALF)
        {
        // ---- phase 1 (half-width cells): local grid, the cells each member reaches, candidate numbers, per-member runs ----
        bool spread = false;
        uint64_t cells = 1;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            {
            int lo = s_glo[k] - 2, hi = s_ghi[k] + 2;
            if (!a.periodic[k])
                {
                lo = max(lo, -rcell[k]);
                hi = min(hi, a.dim[k] - 1 - rcell[k]);
                }
            E[k] = hi - lo + 1;
            gorg[k] = rcell[k] + lo;
            lc3[k] = hd[k] - lo;
            // (a tile that spans a whole periodic axis gets the cells at its ends twice, once on either side: two
            // candidates for one particle, never both in the 5 cells one member looks at as long as the axis has 5)
            spread = spread || (a.periodic[k] && a.dim[k] < 5) || E[k] > (int)PC_MAXGRID || E[k] < 1;
            cells *= (uint64_t)max(E[k], 1);
            }
        if (spread || cells > PC_MAXGRID)
            {
            if (tid == 0)
                {
                atomicOr(&a.flags[1], 1u);
                atomicMax(&a.flags[6], 4u); // members spread over too many cells: particles not spatially sorted (or a box of fewer than 5 cells)
                a.tile_nstage[tile] = 0;
                a.tile_head[tile] = (uint64_t)tile * a.stage_stride;
                }
            return;
            }
        G = (uint32_t)cells;
        // the largest list radius (the cells are at least half as wide), with a margin far above the rounding of rel[]
        float rlm = 0.f;
        for (uint32_t t = 0; t < ntp; ++t)
            rlm = fmaxf(rlm, (float)a.rlistsq[t]);
        const float Rm2 = rlm * 1.0002f;
        if (member)
            {
            // a row of cells (dy, dz) is searched if it comes within the list radius of ME, along x as far as the sphere reaches
            for (uint32_t r = 0; r < PC_RUNS_H; ++r)
                {
                const int dz = (int)(r / 5u) - 2, dy = (int)(r % 5u) - 2;
                const int yl = lc3[1] + dy, zl = lc3[2] + dz;
                uint32_t byte = 0xffu;
                if (yl >= 0 && yl < E[1] && zl >= 0 && zl < E[2])
                    {
                    const float fy = fmaxf(0.f, fmaxf((float)dy - rel[1], rel[1] - (float)(dy + 1))) * a.gw[1];
                    const float fz = fmaxf(0.f, fmaxf((float)dz - rel[2], rel[2] - (float)(dz + 1))) * a.gw[2];
                    const float d2 = fy * fy + fz * fz;
                    if (d2 <= Rm2)
                        {
                        const float h = sqrtf(Rm2 - d2) * (float)a.gwinv[0];
                        int xlo = max(-2, (int)floorf(rel[0] - h)), xhi = min(2, (int)floorf(rel[0] + h));
                        xlo = max(xlo, -lc3[0]);
                        xhi = min(xhi, E[0] - 1 - lc3[0]);
                        if (xlo <= xhi)
                            {
                            byte = (uint32_t)(xlo + 2) | ((uint32_t)(xhi + 2) << 3);
                            const uint32_t b0 = (uint32_t)((zl * E[1] + yl) * E[0] + lc3[0] + xlo);
                            const uint64_t m = ((1ull << (uint32_t)(xhi - xlo + 1)) - 1ull) << (b0 & 31u);
                            atomicOr(&s_need[b0 >> 5], (uint32_t)m);
                            if (m >> 32)
                                atomicOr(&s_need[(b0 >> 5) + 1u], (uint32_t)(m >> 32));
                            }
                        }
                    }
                s_runl[r * PC_THREADS + tid] = (unsigned char)byte;
                }
            }
        __syncthreads();
        for (uint32_t lc = tid; lc < G; lc += PC_THREADS)
            {
            uint32_t n = 0;
            if ((s_need[lc >> 5] >> (lc & 31u)) & 1u)
                {
                const uint32_t c = cell_of_local(lc);
                n = min(a.cell_start[c + 1] - a.cell_start[c], PC_MAXCAND + 1u);
                }
            s_cnt[lc] = n;
            }
        __syncthreads();
            {
            // exclusive scan of the populations, eight cells per thread
            constexpr uint32_t PER = PC_MAXGRID / PC_THREADS;
            uint32_t v[PER], sum = 0;
#pragma unroll
            for (uint32_t q = 0; q < PER; ++q)
                {
                const uint32_t t = tid * PER + q;
                v[q] = (t < G) ? s_cnt[t] : 0u;
                sum += v[q];
                }
            uint32_t incl = sum;
            for (int off = 1; off < 64; off <<= 1)
                {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
                if ((int)lane >= off)
                    incl += up;
                }
            if (lane == 63)
                s_wtot[wave] = incl;
            __syncthreads();
            uint32_t acc = incl - sum;
            for (uint32_t w = 0; w < wave; ++w)
                acc += s_wtot[w];
#pragma unroll
            for (uint32_t q = 0; q < PER; ++q)
                {
                const uint32_t t = tid * PER + q;
                if (t <= G)
                    s_pre[t] = (uint16_t)min(acc, 0xffffu); // (t == G: the total; more than PC_MAXCAND is refused below)
                acc += v[q];
                }
            if (tid == PC_THREADS - 1u)
                {
                s_ncells = acc;
                if (G == PC_MAXGRID)
                    s_pre[G] = (uint16_t)min(acc, 0xffffu);
                }
            }
        __syncthreads();
        NC = s_ncells;
        }
    else
        {
        ncell_blk = s_ncells;
        n_mc = s_nmc;
        if (tid == 0)
            atomicMax(&a.flags[4], n_mc); // (reported: a caller that watches it re-sorts its particles before the limit is hit)
        if (ncell_blk > PC_MAXCELLS || n_mc > PC_MAXMC)
            {
            if (tid == 0)
                {
                atomicOr(&a.flags[1], 1u);
                atomicMax(&a.flags[6], 4u); // members spread over too many cells: particles not spatially sorted
                a.tile_nstage[tile] = 0;
                a.tile_head[tile] = (uint64_t)tile * a.stage_stride;
                }
            return;
            }
        // ---- phase 1: the cells in ascending order, candidate ranges, the runs of every member cell ----
        for (uint32_t t = tid; t < ncell_blk; t += PC_THREADS)
            {
            const uint32_t c = s_tmp[t];
            uint32_t rank = 0;
            for (uint32_t u = 0; u < ncell_blk; ++u)
                rank += (s_tmp[u] < c) ? 1u : 0u;
            const uint32_t f = a.cell_start[c];
            s_cells[rank] = c;
            s_cfirst[rank] = f;
            s_coff[rank + 1] = a.cell_start[c + 1] - f;
            }
        __syncthreads();
        if (tid < 64)
            {
            // inclusive scan of up to 512 counts, 8 per lane
            constexpr uint32_t PER = PC_MAXCELLS / 64;
            uint32_t v[PER], sum = 0;
    #pragma unroll
            for (uint32_t q = 0; q < PER; ++q)
                {
                const uint32_t t = tid * PER + q;
                v[q] = (t < ncell_blk) ? s_coff[t + 1] : 0u;
                sum += v[q];
                }
            uint32_t incl = sum;
            for (int off = 1; off < 64; off <<= 1)
                {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
                if ((int)lane >= off)
                    incl += up;
                }
            uint32_t acc = incl - sum;
    #pragma unroll
            for (uint32_t q = 0; q < PER; ++q)
                {
                const uint32_t t = tid * PER + q;
                acc += v[q];
                if (t < ncell_blk)
                    s_coff[t + 1] = acc;
                }
            if (tid == 0)
                s_coff[0] = 0;
            }
        __syncthreads();
        NC = s_coff[ncell_blk];
        }
    if (NC > PC_MAXCAND)
        {
        if (tid == 0)
            {
            atomicOr(&a.flags[1], 1u);
            atomicMax(&a.flags[6], 5u); // too many particles in the cells around the tile
            a.tile_nstage[tile] = 0;
            a.tile_head[tile] = (uint64_t)tile * a.stage_stride;
            }
        return;
        }
    // a raw entry: candidate number | class. 4 bits of class (core, sure, near, 8 shells) when 12 bits hold the
    // candidate number; else 3 (shells 4 .. 7 are filed under 4 -- conservative, a shell is a lower bound)
    const uint32_t cbits = (NC <= 4096u) ? 4u : 3u;
    const uint32_t cmask = (1u << cbits) - 1u;
    const float fmax_shell = (cbits == 4u) ? (float)(PLAN_SHELLS - 1u) : 4.f;
    // Acceptance test in single precision: r^2 <= r_list^2 (1 + 1e-5) + e, where e bounds what rounding the
    // staged coordinates (|coordinate| <= cmax: the members' extent + two cells) to FP32 can do to r^2:
    // 2 sqrt(3) r cmax 2^-23 = 4.2e-7 r cmax; taken as 1e-6 r_list cmax. A superset of the exact list, never less.
    const float cmax = __int_as_float((int)s_cmax) + 2.f * (float)a.r_list_max;
    const float rl_extra = 1e-6f * (float)a.r_list_max * cmax;
    const float rl1 = SINGLE ? (a.rlistsq[0] > 0.0 ? (float)a.rlistsq[0] * 1.00001f + rl_extra : -1.f) : 0.f;
    if constexpr (HALF)
        {
        // my runs: the rows kept above, as candidate ranges (first candidate, length), empty ones dropped
        if (member)
            {
            const int rowbase = (lc3[2] * E[1] + lc3[1]) * E[0] + lc3[0];
            for (uint32_t r = 0; r < PC_RUNS_H; ++r)
                {
                const uint32_t byte = s_runl[r * PC_THREADS + tid];
                if (byte == 0xffu)
                    continue;
                const int dz = (int)(r / 5u) - 2, dy = (int)(r % 5u) - 2;
                const int b = rowbase + (dz * E[1] + dy) * E[0];
                const uint32_t g0 = s_pre[b + (int)(byte & 7u) - 2], g1 = s_pre[b + (int)(byte >> 3) - 1];
                uint32_t len = g1 - g0;
                if (len > 255u)
                    {
                    s_bad = 2; // a run of five cells holds more candidates than a byte counts: cells far denser than a liquid's
                    len = 255u;
                    }
                if (len)
                    {
                    s_rung[nr * PC_THREADS + tid] = (uint16_t)g0;
                    s_runl[nr * PC_THREADS + tid] = (unsigned char)len; // (nr <= r: behind the bytes still to be read)
                    ++nr;
                    }
                }
            }
        __syncthreads();
        if (s_bad)
            {
            if (tid == 0)
                {
                atomicOr(&a.flags[1], 1u);
                atomicMax(&a.flags[6], 5u);
                a.tile_nstage[tile] = 0;
                a.tile_head[tile] = (uint64_t)tile * a.stage_stride;
                }
            return;
            }
        }
    else
        {
        // the runs of a member cell: its 3 x 3 rows of cells along x; a row is one or two runs of consecutive
        // cells, consecutive in the sorted cell array too (every one of them is in the set): one contiguous
        // candidate range per run
        for (uint32_t t = tid; t < n_mc * PC_RUNS; t += PC_THREADS)
            {
            const uint32_t imc = t / PC_RUNS, q = t % PC_RUNS;
            const uint32_t mc = s_mcl[imc];
            const int mcx = (int)(mc % (uint32_t)dimx), mcy = (int)((mc / (uint32_t)dimx) % (uint32_t)dimy), mcz = (int)(mc / (uint32_t)(dimx * dimy));
            int nzs[3], nys[3], nxs[3];
            const int cz_n = axis_neighbors(mcz, dimz, a.periodic[2], nzs), cy_n = axis_neighbors(mcy, dimy, a.periodic[1], nys);
            const int cx_n = axis_neighbors(mcx, dimx, a.periodic[0], nxs);
            int run0[2] = {0, 0}, runlen[2] = {0, 0};
            int nruns = 0;
            for (int k = 0; k < 3; ++k)
                {
                if (k >= cx_n)
                    continue;
                if (nruns && nxs[k] == run0[nruns - 1] + runlen[nruns - 1])
                    ++runlen[nruns - 1];
                else
                    {
                    run0[nruns] = nxs[k];
                    runlen[nruns] = 1;
                    ++nruns;
                    }
                }
            const int row = (int)(q >> 1), r = (int)(q & 1u);
            const int qz = row / 3, qy = row % 3;
            uint32_t packed = 0; // empty run
            if (qz < cz_n && qy < cy_n && r < nruns)
                {
                const int zz = qz == 0 ? nzs[0] : (qz == 1 ? nzs[1] : nzs[2]);
                const int yy = qy == 0 ? nys[0] : (qy == 1 ? nys[1] : nys[2]);
                const uint32_t c = (uint32_t)((zz * dimy + yy) * dimx + (r ? run0[1] : run0[0]));
                uint32_t sl = 0, sh = ncell_blk; // s_cells[sl] == c (present by construction)
                while (sh - sl > 1)
                    {
                    const uint32_t mid = (sl + sh) >> 1;
                    if (s_cells[mid] <= c) sl = mid; else sh = mid;
                    }
                packed = s_coff[sl] | (s_coff[sl + (uint32_t)(r ? runlen[1] : runlen[0])] << 16); // NC <= 8192: 16 bits each
                }
            s_runs[imc][q] = packed;
            }
        }
    if (SINGLE)
        {
        // r^2 -> class, at the lower edge of each bin (classes grow with r: never too high a class)
        const float rcsq_m = (float)a.rcutsq[0] * 1.0001f, rin = a.rinnersq ? (float)a.rinnersq[0] : 0.f;
        const float rcw = sqrtf(fmaxf((float)a.rcutsq[0], 0.f)) * shell_winv;
        // class sure: certainly closer than r_cut - r_buff (r_buff = PLAN_SHELLS shell widths): the UPPER edge of the bin has
        // to clear the radius, which itself is taken 2e-4 short (the single-precision separation is good to ~2e-6)
        const float r_sure = (shell_w > 0.f) ? sqrtf(fmaxf((float)a.rcutsq[0], 0.f)) - (float)PLAN_SHELLS * shell_w - 2e-4f : 0.f;
        const float sure_sq = r_sure > 0.f ? r_sure * r_sure : 0.f;
        for (uint32_t t = tid; t < PC_CTAB; t += PC_THREADS)
            {
            const float lo = (float)t * (rl1 * (1.0f / PC_CTAB)) * 0.999999f;
            const float hi = (float)(t + 1u) * (rl1 * (1.0f / PC_CTAB)) * 1.000001f;
            uint32_t cls = pair_class(lo, rcsq_m, rin, rcw, rscale, fmax_shell);
            if (cls == PLAN_CLS_NEAR && hi < sure_sq)
                cls = PLAN_CLS_SURE;
            s_ctab[t] = (unsigned char)cls;
            }
        if (blockIdx.x == 0 && tid == 0)
            {
            // what the force kernel may rely on (margins for the single-precision test included)
            a.flags[0] = (uint32_t)__float_as_int(r_sure > 0.f ? r_sure + 1e-4f : 0.f);
            a.flags[7] = (uint32_t)__float_as_int(rin > 0.f ? sqrtf(rin) - 1e-4f : 0.f);
            }
        }
    // which member cell is mine
    uint32_t imc_mine = 0;
    if (!HALF)
        for (uint32_t u = 0; u < n_mc; ++u)
            imc_mine = (s_mcl[u] == mycell) ? u : imc_mine;
    const bool wide = s_wide != 0;
    const float bLx = (float)a.box.Lx, bLy = (float)a.box.Ly, bLz = (float)a.box.Lz;
    // (1 / L = 0 along a non-periodic axis: rint(0) = 0, nothing is subtracted)
    const float bLxi = a.box.px ? (float)a.box.Lxinv : 0.f, bLyi = a.box.py ? (float)a.box.Lyinv : 0.f, bLzi = a.box.pz ? (float)a.box.Lzinv : 0.f;
    const float tscale = SINGLE && rl1 > 0.f ? (float)PC_CTAB / rl1 : 0.f;
    uint16_t* raw_tile = a.raw + (uint64_t)tile * 256u * a.row_cap;
    const uint32_t trow = mytype * a.ntypes;
    const uint32_t nex = (a.n_excl && member) ? a.n_excl[i] : 0u;
    // the member's first four exclusions (bonded partners: two for a bead inside a chain) in registers: the accept path
    // compared every accepted candidate with the global table entry by entry (C3: 2.4 ms per build against 1.7 without bonds)
    uint32_t ex4[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
    for (uint32_t e = 0; e < 4u; ++e)
        if (e < nex)
            ex4[e] = a.excl[(uint64_t)e * a.excl_pitch + i];
#ifdef AZP_PLAN_CELLS_PROFILE
    if ((a.stop_after & 255u) == 1u)
        return;
#endif

    // ---- phase 2: stage a batch of candidates; every thread walks its member's runs through it ----
    uint32_t cnt = 0;
    if (!HALF) // (half-width cells: the counters take the place of the run tables once the tests are done)
        for (uint32_t t = 0; t < PLAN_CLASSES; ++t)
            s_cur[t * PC_THREADS + tid] = 0;
    // run q of my member: candidates [first, end)
    auto fetch_run = [&](uint32_t q, uint32_t& first_c, uint32_t& end_c)
        {
        if constexpr (HALF)
            {
            first_c = s_rung[q * PC_THREADS + tid];
            end_c = first_c + s_runl[q * PC_THREADS + tid];
            }
        else
            {
            const uint32_t packed = s_runs[imc_mine][q];
            first_c = packed & 0xffffu;
            end_c = packed >> 16;
            }
        };
    const uint32_t n_runs = HALF ? nr : PC_RUNS;
    uint32_t qres = 0; // my first run that reaches beyond the batches done so far
#ifdef AZP_PLAN_CELLS_PROFILE
    uint32_t prof_trips = 0, prof_tests = 0;
#endif
    for (uint32_t b0 = 0; b0 < NC; b0 += BATCH)
        {
        const uint32_t nbat = min(BATCH, NC - b0);
        __syncthreads(); // previous batch (first pass: the hash sets, the run table) settled
        for (uint32_t t = tid; t < nbat; t += PC_THREADS)
            {
            const uint32_t g = b0 + t;
            const uint32_t j = cand_particle(g);
            const double4 pj = load_scalar4(a.pos, j);
            double x = pj.x - cref.x, y = pj.y - cref.y, z = pj.z - cref.z;
            if (a.box.px) x = __builtin_fma(-a.box.Lx, rint(x * a.box.Lxinv), x);
            if (a.box.py) y = __builtin_fma(-a.box.Ly, rint(y * a.box.Lyinv), y);
            if (a.box.pz) z = __builtin_fma(-a.box.Lz, rint(z * a.box.Lzinv), z);
            cx[t] = (float)x; cy[t] = (float)y; cz[t] = (float)z;
            cj[t] = j;
            if (!SINGLE)
                ctype[t] = (unsigned char)type_from_w(pj.w);
            }
        __syncthreads();
        if (member)
            {
            // cursor: run q of my cell clipped to the batch is [g, l1). Two candidates per step (packed
            // single-precision math: one instruction per pair of differences / products)
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
            // (half-width cells -- the runs ascend: one that starts behind the batch ends the walk, and the next batch
            // resumes at the first run this one did not use up; without that every lane trudges through its remaining
            // runs at the end of every batch, each on its own trip of the wave. Full-width cells: the members of a cell
            // move in lock-step and the plain walk is cheaper)
            uint32_t q = qres, g = 0, l1 = 0;
            const uint32_t bend = b0 + nbat;
            while (g >= l1 && q < n_runs)
                {
                uint32_t rf, re;
                fetch_run(q, rf, re);
                if (HALF && rf >= bend)
                    {
                    q = n_runs;
                    break;
                    }
                qres = (HALF && re <= bend && qres == q) ? q + 1u : qres; // (an empty run behind one that reaches into the next batch must not step over it)
                ++q;
                g = max(rf, b0);
                l1 = min(re, bend);
                }
            // two candidates (c0, c0 + 1; numbers g0, g0 + 1; the first valid if va, the second if vb) against my member
            auto judge2 = [&](uint32_t g0, uint32_t c0, bool va, bool vb, f2 X, f2 Y, f2 Z)
                {
                uint32_t tpa = 0, tpb = 0;
                float rla = rl1, rlb = rl1;
                if (!SINGLE)
                    {
                    // (a pad entry holds no type: its byte must not index a table)
                    tpa = trow + (va ? (uint32_t)ctype[c0] : 0u);
                    tpb = trow + (vb ? (uint32_t)ctype[c0 + 1u] : 0u);
                    rla = rc_cached ? s_rlistsq[tpa] : (a.rlistsq[tpa] > 0.0 ? (float)a.rlistsq[tpa] * 1.00001f : -1.f);
                    rlb = rc_cached ? s_rlistsq[tpb] : (a.rlistsq[tpb] > 0.0 ? (float)a.rlistsq[tpb] * 1.00001f : -1.f);
                    rla = rla > 0.f ? rla + rl_extra : rla;
                    rlb = rlb > 0.f ? rlb + rl_extra : rlb;
                    }
                f2 dx = xi2 - X, dy = yi2 - Y, dz = zi2 - Z;
                if (wide)
                    {
                    const f2 kz = {rintf(dz.x * bLzi), rintf(dz.y * bLzi)}, ky = {rintf(dy.x * bLyi), rintf(dy.y * bLyi)};
                    const f2 kx = {rintf(dx.x * bLxi), rintf(dx.y * bLxi)};
                    dz = dz - kz * bLz;
                    dy = dy - ky * bLy;
                    dx = dx - kx * bLx;
                    }
                const f2 rsq = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                bool acca = va && rsq.x <= rla, accb = vb && rsq.y <= rlb; // rl < 0: the type pair is not listed
                if (acca || accb)
                    {
                    const uint32_t ja = cj[c0], jb = cj[c0 + 1u];
                    acca = acca && ja != i;
                    accb = accb && jb != i;
                    if (nex)
                        {
                        acca = acca && ja != ex4[0] && ja != ex4[1] && ja != ex4[2] && ja != ex4[3];
                        accb = accb && jb != ex4[0] && jb != ex4[1] && jb != ex4[2] && jb != ex4[3];
                        }
                    for (uint32_t e = 4u; e < nex; ++e)
                        {
                        const uint32_t x = a.excl[(uint64_t)e * a.excl_pitch + i];
                        acca = acca && x != ja;
                        accb = accb && x != jb;
                        }
                    uint32_t clsa, clsb;
                    if (SINGLE)
                        {
                        const f2 bin = rsq * tscale;
                        clsa = s_ctab[min((uint32_t)bin.x, PC_CTAB - 1u)];
                        clsb = s_ctab[min((uint32_t)bin.y, PC_CTAB - 1u)];
                        }
                    else
                        {
                        clsa = pair_class(rsq.x, rc_cached ? s_rcutsq[tpa] : (float)a.rcutsq[tpa] * 1.0001f,
                                          rc_cached ? s_rinnersq[tpa] : (a.rinnersq ? (float)a.rinnersq[tpa] : 0.f),
                                          rc_cached ? s_rcw[tpa] : sqrtf(fmaxf((float)a.rcutsq[tpa], 0.f)) * shell_winv, rscale, fmax_shell);
                        clsb = pair_class(rsq.y, rc_cached ? s_rcutsq[tpb] : (float)a.rcutsq[tpb] * 1.0001f,
                                          rc_cached ? s_rinnersq[tpb] : (a.rinnersq ? (float)a.rinnersq[tpb] : 0.f),
                                          rc_cached ? s_rcw[tpb] : sqrtf(fmaxf((float)a.rcutsq[tpb], 0.f)) * shell_winv, rscale, fmax_shell);
                        }
                    if (acca)
                        {
                        if (cnt < a.row_cap PC_PROFILE_AND(!(a.stop_after & 0x100u)))
                            raw_tile[cnt * 256u + tid] = (uint16_t)((g0 << cbits) | clsa);
                        ++cnt;
                        }
                    if (accb)
                        {
                        if (cnt < a.row_cap PC_PROFILE_AND(!(a.stop_after & 0x100u)))
                            raw_tile[cnt * 256u + tid] = (uint16_t)(((g0 + 1u) << cbits) | clsb);
                        ++cnt;
                        }
                    }
                };
            // full-width cells: four candidates per trip (the members of a cell walk long runs in lock-step: the loop
            // control, the run switch and the address arithmetic are paid once per four); half-width cells: two (runs
            // of a dozen candidates, every lane on its own)
            constexpr uint32_t PER_TRIP = HALF ? 2u : 4u;
            while (g < l1)
                {
                const uint32_t gc = g, c = g - b0;
                // (candidates beyond the run's end -- the pad entries / the next cell's first candidates -- are read and ignored)
                const bool v1 = gc + 1u < l1, v2 = PER_TRIP == 4u && gc + 2u < l1, v3 = PER_TRIP == 4u && gc + 3u < l1;
#ifdef AZP_PLAN_CELLS_PROFILE
                ++prof_trips;
                prof_tests += 1u + (v1 ? 1u : 0u) + (v2 ? 1u : 0u) + (v3 ? 1u : 0u);
#endif
                const f2 Xa = {cx[c], cx[c + 1u]}, Ya = {cy[c], cy[c + 1u]}, Za = {cz[c], cz[c + 1u]};
                f2 Xb = Xa, Yb = Ya, Zb = Za;
                if (PER_TRIP == 4u)
                    {
                    Xb = f2 {cx[c + 2u], cx[c + 3u]};
                    Yb = f2 {cy[c + 2u], cy[c + 3u]};
                    Zb = f2 {cz[c + 2u], cz[c + 3u]};
                    }
                g += PER_TRIP;
                while (g >= l1 && q < n_runs)
                    {
                    uint32_t rf, re;
                    fetch_run(q, rf, re);
                    if (HALF && rf >= bend)
                        {
                        q = n_runs;
                        break;
                        }
                    qres = (HALF && re <= bend && qres == q) ? q + 1u : qres; // (an empty run behind one that reaches into the next batch must not step over it)
                    ++q;
                    g = max(rf, b0);
                    l1 = min(re, bend);
                    }
                judge2(gc, c, true, v1, Xa, Ya, Za);
                if (PER_TRIP == 4u)
                    judge2(gc + 2u, c + 2u, v2, v3, Xb, Yb, Zb);
                }
            }
        }
#ifdef AZP_PLAN_CELLS_PROFILE
        {
        // (tools/plan_cells_probe.py: candidate tests, trips of the slowest lane of every wave, candidates and grid cells per tile)
        uint32_t tr = prof_trips, ts = prof_tests;
        for (int off = 32; off > 0; off >>= 1)
            {
            tr = max(tr, (uint32_t)__shfl_xor((int)tr, off, 64));
            ts += (uint32_t)__shfl_xor((int)ts, off, 64);
            }
        if (lane == 0)
            {
            atomicAdd(&a.flags[8], ts >> 4);
            atomicAdd(&a.flags[9], tr);
            }
        if (tid == 0)
            {
            atomicAdd(&a.flags[10], NC);
            atomicAdd(&a.flags[11], HALF ? G : ncell_blk);
            atomicAdd(&a.flags[12], (NC + BATCH - 1u) / BATCH);
            }
        }
#endif
    // every thread walks its raw row once: entries per class (the cursors of phase 4 start from these
    // totals) and the bitmap of the candidates somebody listed. Here and not in the loop above: all
    // lanes are busy, there only the accepting sixth was
    __syncthreads();
    if (HALF)
        {
        // (batch and run tables are done with: counters and bitmap move in)
        for (uint32_t t = 0; t < PLAN_CLASSES; ++t)
            s_cur[t * PC_THREADS + tid] = 0;
        for (uint32_t t = tid; t < PC_MAXCAND / 32; t += PC_THREADS)
            s_used[t] = 0;
        __syncthreads();
        }
    if (member PC_PROFILE_AND(!(a.stop_after & 0x200u)))
        {
        const uint32_t nk = min(cnt, a.row_cap);
        for (uint32_t k0 = 0; k0 < nk; k0 += PC_WALK)
            {
            uint32_t e[PC_WALK];
#pragma unroll
            for (uint32_t u = 0; u < PC_WALK; ++u)
                e[u] = (k0 + u < nk) ? (uint32_t)raw_tile[(k0 + u) * 256u + tid] : 0xffffffffu;
#pragma unroll
            for (uint32_t u = 0; u < PC_WALK; ++u)
                if (e[u] != 0xffffffffu)
                    {
                    ++s_cur[(e[u] & cmask) * PC_THREADS + tid];
                    const uint32_t g = e[u] >> cbits;
                    atomicOr(&s_used[g >> 5], 1u << (g & 31u));
                    }
            }
        }
    __syncthreads();
#ifdef AZP_PLAN_CELLS_PROFILE
    if ((a.stop_after & 255u) == 2u)
        return;
#endif
    // ---- phase 3: row lengths; slot numbers = rank among the marked candidates; stage list ----
    if (member)
        {
        a.n_neigh[i] = cnt;
        if (cnt > a.row_cap)
            {
            atomicOr(&a.flags[1], 1u);
            atomicMax(&a.flags[6], 3u); // row longer than the capacity: the caller retries with longer rows
            s_bad = 1;
            }
        }
    // the lane of the force kernel that gets my row: my own, or -- balanced plans -- my rank when the
    // members of the tile are ordered by the number of their in-range entries (longest first), so that
    // every wave of the force kernel gets rows of similar in-range length
    uint32_t pos_in_tile = tid;
    if (a.perm)
        {
        uint32_t* s_key = HALF ? reinterpret_cast<uint32_t*>(s_region + KEY_OFF) : &s_runs[0][0]; // (the run table is not needed any more)
        const uint32_t nin = member ? min((uint32_t)s_cur[tid] + (uint32_t)s_cur[PC_THREADS + tid] + (uint32_t)s_cur[2 * PC_THREADS + tid], 1023u) : 0u;
        const uint32_t key = ((1023u - nin) << 8) | tid;
        s_key[tid] = key;
        __syncthreads();
        uint32_t rank = 0;
        for (uint32_t u = 0; u < PC_THREADS; ++u)
            rank += (s_key[u] < key) ? 1u : 0u;
        pos_in_tile = rank;
        a.perm[(uint64_t)tile * 256u + rank] = (uint8_t)tid;
        }
    const uint32_t pw = pos_in_tile >> 6, pl = pos_in_tile & 63u; // the force kernel's wave (slice) and lane
    atomicMax(&s_smax[pw], member ? cnt : 0u);
    uint32_t longest = member ? cnt : 0u;
    for (int off = 32; off > 0; off >>= 1)
        longest = max(longest, (uint32_t)__shfl_xor((int)longest, off, 64));
    if (lane == 0)
        atomicMax(&a.flags[5], longest);
    if (tid < 64)
        {
        // exclusive scan of the popcounts of the bitmap words, four per lane
        constexpr uint32_t PER = PC_MAXCAND / 32 / 64;
        uint32_t w[PER], sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q)
            {
            w[q] = (uint32_t)__popc(s_used[PER * tid + q]);
            sum += w[q];
            }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1)
            {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
            if ((int)lane >= off)
                incl += v;
            }
        uint32_t acc = incl - sum;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q)
            {
            s_wordbase[PER * tid + q] = acc;
            acc += w[q];
            }
        if (tid == 63)
            s_wordbase[PC_MAXCAND / 32] = incl;
        }
    __syncthreads();
    const uint32_t n_stage = s_wordbase[PC_MAXCAND / 32];
    const bool bad = s_bad || n_stage > PLAN_MAX_STAGE || n_stage > a.stage_stride;
    if (tid == 0)
        {
        a.tile_nstage[tile] = bad ? 0u : n_stage;
        a.tile_head[tile] = (uint64_t)tile * a.stage_stride;
        atomicMax(&a.flags[2], n_stage);
        if (n_stage > PLAN_MAX_STAGE || n_stage > a.stage_stride)
            {
            atomicOr(&a.flags[1], 1u);
            atomicMax(&a.flags[6], 2u);
            }
        }
    if (bad)
        return;
    uint32_t* stage = a.stage_idx + (uint64_t)tile * a.stage_stride;
    for (uint32_t g = tid; g < NC; g += PC_THREADS)
        {
        const uint32_t w = s_used[g >> 5], bit = 1u << (g & 31u);
        if (w & bit)
            {
            const uint32_t slot = s_wordbase[g >> 5] + (uint32_t)__popc(w & (bit - 1u));
            s_slot[g] = (uint16_t)slot;
            stage[slot] = cand_particle(g);
            }
        }
    // class totals -> first row position of each class (the cursors of phase 4); chunk counts per class boundary
    const uint32_t n = member ? cnt : 0u;
    uint32_t before = 0;
    for (uint32_t c = 0; c < PLAN_CLASSES; ++c)
        {
        const uint32_t v = s_cur[c * PC_THREADS + tid];
        s_cur[c * PC_THREADS + tid] = (uint16_t)before;
        before += v;
        if (c >= PLAN_CLS_NEAR && member)
            atomicMax(&s_kend[pw][c - PLAN_CLS_NEAR], (before + 7u) / 8u); // [0]: through "near", [1 + s]: through shell s
        if (c == PLAN_CLS_CORE && member)
            atomicMax(&s_kcore[pw], (before + 7u) / 8u);
        if (c == PLAN_CLS_SURE && member)
            atomicMin(&s_ksure[pw], before / 8u);
        }
    __syncthreads();
#ifdef AZP_PLAN_CELLS_PROFILE
    if ((a.stop_after & 255u) == 4u)
        return;
#endif
    // ---- phase 4: raw rows -> compiled rows (class by class, 16-byte chunks in the force kernel's lane order) ----
    const uint32_t Kcap = a.row_cap / 8u;
    const uint32_t K = (s_smax[pw] + 7u) / 8u;
    unsigned char* out = reinterpret_cast<unsigned char*>(a.cnl + (uint64_t)(tile * 4u + pw) * Kcap * 64ull) + pl * 16u;
    for (uint32_t c = n >> 3; c < K; ++c) // the tail of the row up to the slice's rectangle: dummy slots
        *reinterpret_cast<uint4*>(out + c * 1024u) = make_uint4(0, 0, 0, 0);
    for (uint32_t k0 = 0; k0 < n; k0 += PC_WALK)
        {
        uint32_t e[PC_WALK], off[PC_WALK];
#pragma unroll
        for (uint32_t u = 0; u < PC_WALK; ++u)
            e[u] = (k0 + u < n) ? (uint32_t)raw_tile[(k0 + u) * 256u + tid] : 0xffffffffu;
#pragma unroll
        for (uint32_t u = 0; u < PC_WALK; ++u)
            off[u] = (e[u] != 0xffffffffu) ? ((uint32_t)s_slot[e[u] >> cbits] + 1u) * 8u : 0u;
#pragma unroll
        for (uint32_t u = 0; u < PC_WALK; ++u)
            if (e[u] != 0xffffffffu)
                {
                const uint32_t cls = e[u] & cmask;
                const uint32_t posn = s_cur[cls * PC_THREADS + tid];
                s_cur[cls * PC_THREADS + tid] = (uint16_t)(posn + 1u);
                *reinterpret_cast<uint16_t*>(out + (posn >> 3) * 1024u + (posn & 7u) * 2u) = (uint16_t)off[u];
                }
        }
    if (lane == 0)
        {
        const uint32_t slice = tile * 4u + wave;
        a.slice_K[slice] = (s_smax[wave] + 7u) / 8u;
        a.slice_head[slice] = (uint64_t)slice * Kcap;
        for (uint32_t sh = 0; sh <= PLAN_SHELLS; ++sh)
            a.slice_Kend[(PLAN_SHELLS + 1) * slice + sh] = s_kend[wave][sh];
        a.slice_Kphase[slice] = s_kcore[wave];
        a.slice_Kphase[a.n_tiles * 4u + slice] = (s_ksure[wave] == 0xffffffffu) ? 0u : s_ksure[wave]; // (a slice without a member)
        }
    }

#define AZP_HIP_TRY(expr)                      \
    do                                         \
        {                                      \
        hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) return (int)e_;  \
        } while (0)

template<class T> static hipError_t ensure_buf(T*& ptr, size_t& cap, size_t need)
    {
    if (need <= cap && ptr)
        return hipSuccess;
    if (ptr)
        {
        hipError_t e = hipFree(ptr);
        if (e != hipSuccess) return e;
        ptr = nullptr;
        }
    const size_t newcap = need + need / 8 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), newcap * sizeof(T));
    cap = (e == hipSuccess) ? newcap : 0;
    return e;
    }

int plan_build_from_cells(PairPlan& p, const azp_nlist_args& c, const azp_pair_args& pa, hipStream_t s)
    {
    p.valid = false;
    p.invalid_reason = 0;
    p.from_cells = true;
    p.N = c.N;
    p.n_max = c.n_total;
    p.size_nlist = 0;
    ++p.builds;
    if (c.N == 0)
        return AZP_SUCCESS;
    if (c.box.tilt[0] != 0.0 || c.box.tilt[1] != 0.0 || c.box.tilt[2] != 0.0 || c.ntypes > 255)
        {
        p.invalid_reason = 6; // cells of a tilted box are not binned by azp_nlist_cell_assign either; types are staged as bytes
        return AZP_SUCCESS;
        }
    uint32_t row_cap = c.row_capacity ? c.row_capacity : 160u;
    row_cap = std::min<uint32_t>((row_cap + 7u) & ~7u, PC_ROWMAX - 8u);
    p.tpp = 1;
    p.tile = 256;
    p.n_tiles = (c.N + 255u) / 256u;
    p.n_slices = p.n_tiles * 4;
    p.row_cap = row_cap;
    size_t cap_heads_t = p.d_tile_head ? p.cap_tiles : 0, cap_heads_s = p.d_slice_head ? p.cap_slices : 0;
    AZP_HIP_TRY(ensure_buf(p.d_tile_nstage, p.cap_tiles, p.n_tiles));
    AZP_HIP_TRY(ensure_buf(p.d_tile_head, cap_heads_t, p.n_tiles));
    AZP_HIP_TRY(ensure_buf(p.d_slice_K, p.cap_slices, p.n_slices));
    AZP_HIP_TRY(ensure_buf(p.d_slice_Kend, p.cap_kend, (PLAN_SHELLS + 1) * (size_t)p.n_slices));
    AZP_HIP_TRY(ensure_buf(p.d_slice_head, cap_heads_s, p.n_slices));
    AZP_HIP_TRY(ensure_buf(p.d_slice_Kphase, p.cap_kphase, 2 * (size_t)p.n_slices));
    size_t cap_flags = p.d_flags ? 16 : 0;
    AZP_HIP_TRY(ensure_buf(p.d_flags, cap_flags, 16)); // (8 .. 15: counters of the profiling build)
    p.total_chunks = (uint64_t)p.n_slices * (row_cap / 8u);
    AZP_HIP_TRY(ensure_buf(p.d_cnl, p.cap_cnl, (size_t)p.total_chunks * 64));
    AZP_HIP_TRY(ensure_buf(p.d_raw, p.cap_raw, (size_t)p.n_tiles * 256u * row_cap));
    p.balanced = false;
    if (p.balance)
        AZP_HIP_TRY(ensure_buf(p.d_perm, p.cap_perm, (size_t)p.n_tiles * 256u));

    PlanCellsKArgs k;
    k.pos = c.d_pos;
    k.rlistsq = c.d_rlistsq;
    k.rcutsq = pa.d_rcutsq;
    k.rinnersq = pa.d_rinnersq;
    k.cell_of = c.d_cell_of;
    k.order = c.d_order;
    k.cell_start = c.d_cell_start;
    k.n_excl = c.d_n_excl;
    k.excl = c.d_excl;
    k.excl_pitch = c.excl_pitch;
    k.n_neigh = c.d_n_neigh;
    k.raw = p.d_raw;
    k.tile_nstage = p.d_tile_nstage;
    k.tile_head = p.d_tile_head;
    k.slice_K = p.d_slice_K;
    k.slice_Kend = p.d_slice_Kend;
    k.slice_Kphase = p.d_slice_Kphase;
    k.slice_head = p.d_slice_head;
    k.cnl = p.d_cnl;
    k.flags = p.d_flags;
    k.perm = p.balance ? p.d_perm : nullptr;
    k.r_list_max = pa.r_list_max;
    k.box = make_box_dev(c.box);
    for (int q = 0; q < 3; ++q)
        {
        k.dim[q] = (int)c.grid.dim[q];
        k.periodic[q] = c.grid.periodic[q];
        }
    k.N = c.N;
    k.n_total = c.n_total;
    k.ntypes = c.ntypes;
    k.n_tiles = p.n_tiles;
    k.row_cap = row_cap;
#ifdef AZP_PLAN_CELLS_PROFILE
    const char* stop_env = getenv("AZP_PLAN_CELLS_STOP"); // (read at every build: tools/plan_cells_probe.py sets it after its warm-up run)
    const uint32_t stop_after = stop_env ? (uint32_t)atoi(stop_env) : 0u;
#else
    const uint32_t stop_after = 0;
#endif
    k.stop_after = stop_after;
    const bool half = c.cell_subdivision == 2;
    for (int q = 0; q < 3; ++q)
        {
        k.glo[q] = c.grid.lo[q];
        k.gwinv[q] = 1.0 / c.grid.width[q];
        k.gw[q] = (float)c.grid.width[q];
        }

    uint32_t stride = p.stage_stride_hint;
    if (stride == 0)
        stride = ((uint64_t)p.n_tiles * (PLAN_MAX_STAGE + 1) * 4 <= (256ull << 20)) ? PLAN_MAX_STAGE + 1 : 1536;
    uint32_t h_flags[16];
    for (;;)
        {
        stride = std::min<uint32_t>((stride + 63u) & ~63u, PLAN_MAX_STAGE + 1);
        AZP_HIP_TRY(ensure_buf(p.d_stage_idx, p.cap_stage, (size_t)p.n_tiles * stride));
        AZP_HIP_TRY(hipMemsetAsync(p.d_flags, 0, 16 * sizeof(uint32_t), s));
        k.stage_idx = p.d_stage_idx;
        k.stage_stride = stride;
        const dim3 grid((p.n_tiles + 7u) & ~7u), block(PC_THREADS);
        if (c.ntypes == 1 && half)
            hipLaunchKernelGGL((plan_cells_kernel<true, true>), grid, block, 0, s, k);
        else if (c.ntypes == 1)
            hipLaunchKernelGGL((plan_cells_kernel<true, false>), grid, block, 0, s, k);
        else if (half)
            hipLaunchKernelGGL((plan_cells_kernel<false, true>), grid, block, 0, s, k);
        else
            hipLaunchKernelGGL((plan_cells_kernel<false, false>), grid, block, 0, s, k);
        AZP_HIP_TRY(hipGetLastError());
        AZP_HIP_TRY(hipMemcpyAsync(h_flags, p.d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, s));
        p.h_tile_nstage.resize(p.n_tiles);
        AZP_HIP_TRY(hipMemcpyAsync(p.h_tile_nstage.data(), p.d_tile_nstage, sizeof(uint32_t) * p.n_tiles, hipMemcpyDeviceToHost, s));
        AZP_HIP_TRY(hipStreamSynchronize(s));
        p.max_stage = h_flags[2];
        p.max_row = h_flags[5];
        p.max_member_cells = h_flags[4];
#ifdef AZP_PLAN_CELLS_PROFILE
        fprintf(stderr, "plan_cells: %.1f candidate tests per particle, %.1f trips of the slowest lane per wave, per tile %.0f candidates, %.0f cells, %.2f batches\n",
                16.0 * h_flags[8] / c.N, (double)h_flags[9] / (p.n_tiles * 4.0), (double)h_flags[10] / p.n_tiles, (double)h_flags[11] / p.n_tiles,
                (double)h_flags[12] / p.n_tiles);
#endif
        if (stop_after)
            {
            p.invalid_reason = 7; // profiling run: the kernel left early, nothing to use
            return AZP_SUCCESS;
            }
        if (!h_flags[1])
            break;
        if (h_flags[6] == 2 && p.max_stage <= PLAN_MAX_STAGE && stride < PLAN_MAX_STAGE + 1)
            {
            stride = p.max_stage + p.max_stage / 8 + 64; // the stride was the problem: grow and redo
            continue;
            }
        p.invalid_reason = (int)h_flags[6]; // 2 stage set, 3 row capacity (max_row says how long), 4 / 5 cell block
        return AZP_SUCCESS;
        }
    p.stage_stride_hint = p.max_stage + p.max_stage / 16 + 32;
    p.total_stage = (uint64_t)p.n_tiles * stride;
        {
        float fm;
        __builtin_memcpy(&fm, &h_flags[3], sizeof(fm));
        p.shell_width = fm;
        __builtin_memcpy(&p.sure_r, &h_flags[0], sizeof(float));
        __builtin_memcpy(&p.core_r, &h_flags[7], sizeof(float));
        if (c.ntypes != 1)
            p.sure_r = p.core_r = 0.f; // (the phases serve the one-type split form of an evaluator only)
        }
    p.cap = plan_cap_for(p.max_stage);
    // identity of "the list" for the planned entry points: the plan's own raw rows and slice heads
    p.nlist_ptr = reinterpret_cast<const uint32_t*>(p.d_raw);
    p.head_ptr = p.d_slice_head;
    p.balanced = p.balance;
    p.valid = true;
    return AZP_SUCCESS;
    }
} // namespace azp

extern "C" int azp_pair_plan_build_from_cells(azp_pair_plan* plan, const azp_nlist_args* cells, const azp_pair_args* pair, void* stream)
    {
    if (!plan || !cells || !pair || !cells->d_pos || !cells->d_rlistsq || !cells->d_cell_of || !cells->d_order || !cells->d_cell_start
        || !cells->d_n_neigh || !pair->d_rcutsq || cells->ntypes == 0 || cells->ntypes != pair->ntypes || cells->n_total < cells->N)
        return AZP_ERROR_INVALID_ARGUMENT;
    for (int k = 0; k < 3; ++k)
        if (cells->grid.dim[k] == 0 || !(cells->grid.width[k] > 0.0))
            return AZP_ERROR_INVALID_ARGUMENT;
    if (cells->cell_subdivision > 2)
        return AZP_ERROR_INVALID_ARGUMENT;
    return azp::plan_build_from_cells(*reinterpret_cast<azp::PairPlan*>(plan), *cells, *pair, static_cast<hipStream_t>(stream));
    }
