// nlist.hip -- cell-list neighbor-list build on the GPU.
// Produces HOOMD-format full neighbor lists (n_neigh / head_list / nlist) that
// the force kernels consume; see include/azp.h for the call sequence.
#include "azp_device.hpp"

namespace azp
{
struct GridDev
    {
    double lo[3], winv[3];
    int dim[3];
    int periodic[3];
    };

static GridDev make_grid_dev(const azp_cell_grid& g)
    {
    GridDev d;
    for (int k = 0; k < 3; ++k)
        {
        d.lo[k] = g.lo[k];
        d.winv[k] = 1.0 / g.width[k];
        d.dim[k] = (int)g.dim[k];
        d.periodic[k] = g.periodic[k];
        }
    return d;
    }

__device__ __forceinline__ int cell_coord(const GridDev& g, int k, double x)
    {
    int c = (int)floor((x - g.lo[k]) * g.winv[k]);
    if (g.periodic[k])
        {
        c %= g.dim[k];
        if (c < 0) c += g.dim[k];
        }
    else
        c = min(max(c, 0), g.dim[k] - 1);
    return c;
    }

__global__ void __launch_bounds__(256) cell_assign_kernel(uint32_t n_total, const double* __restrict__ pos, GridDev g,
                                                          uint32_t* __restrict__ cell_of)
    {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total)
        return;
    const double3 p = load_scalar3_of4(pos, i);
    const int cx = cell_coord(g, 0, p.x), cy = cell_coord(g, 1, p.y), cz = cell_coord(g, 2, p.z);
    cell_of[i] = (uint32_t)((cz * g.dim[1] + cy) * g.dim[0] + cx);
    }

// cell_start[c] = first position in the sorted order whose cell id is >= c
__global__ void __launch_bounds__(256) cell_bounds_kernel(uint32_t n_total, uint32_t ncell,
                                                          const uint32_t* __restrict__ cell_sorted,
                                                          uint32_t* __restrict__ cell_start)
    {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncell)
        return;
    uint32_t lo = 0, hi = n_total;
    while (lo < hi)
        {
        const uint32_t mid = (lo + hi) >> 1;
        if (cell_sorted[mid] < c) lo = mid + 1; else hi = mid;
        }
    cell_start[c] = lo;
    }

// ---- binning in one call (azp_nlist_bin): counting sort of the particles by cell, stable in the particle index ----
// Lanes of a wave that hold the same key add to its counter with ONE atomic (consecutive particles of a spatially sorted
// system sit in the same cell: a plain atomic per lane serialises 32-deep on one address). Returns this lane's slot
// base + rank among the wave's lanes with that key.
__device__ __forceinline__ uint32_t wave_aggregated_add(uint32_t* counters, uint32_t key, bool active)
    {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t result = 0;
    unsigned long long todo = __ballot(active);
    while (todo)
        {
        const int leader = __builtin_ctzll(todo);
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader, 64);
        const unsigned long long same = __ballot(active && key == k0) & todo;
        uint32_t base = 0;
        if ((int)lane == leader)
            base = atomicAdd(&counters[k0], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (active && key == k0)
            result = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
        }
    return result;
    }

__global__ void __launch_bounds__(256) bin_assign_count_kernel(uint32_t n_total, const double* __restrict__ pos, GridDev g,
                                                               uint32_t* __restrict__ cell_of, uint32_t* __restrict__ count)
    {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n_total;
    uint32_t c = 0;
    if (active)
        {
        const double3 p = load_scalar3_of4(pos, i);
        const int cx = cell_coord(g, 0, p.x), cy = cell_coord(g, 1, p.y), cz = cell_coord(g, 2, p.z);
        c = (uint32_t)((cz * g.dim[1] + cy) * g.dim[0] + cx);
        cell_of[i] = c;
        }
    (void)wave_aggregated_add(count, c, active);
    }

// exclusive scan of ncell counts into cell_start[0 .. ncell] by ONE workgroup of 1,024 threads, 4,096 counters per
// trip (four per thread, wave scans by shuffles, one LDS hop for the 16 wave totals; the carry in a register);
// count[] is left holding the cell starts too (the scatter's cursors)
__global__ void __launch_bounds__(1024) bin_scan_kernel(uint32_t ncell, uint32_t* __restrict__ count, uint32_t* __restrict__ cell_start)
    {
    __shared__ uint32_t s_wave[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < ncell; c0 += 4096u)
        {
        const uint32_t c = c0 + 4u * t;
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            v[k] = (c + k < ncell) ? count[c + k] : 0u;
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1)
            {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
            if ((int)lane >= off)
                incl += up;
            }
        if (lane == 63u)
            s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < 16u; ++w)
            {
            const uint32_t x = s_wave[w];
            before += (w < wave) ? x : 0u;
            total += x;
            }
        __syncthreads();
        uint32_t acc = carry + before + incl - mine;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            {
            if (c + k < ncell)
                {
                cell_start[c + k] = acc;
                count[c + k] = acc;
                }
            acc += v[k];
            }
        carry += total;
        }
    if (t == 0)
        cell_start[ncell] = carry;
    }

// The same scan for grids of many cells (half-width cells: 2^18 at N = 2^20), where one workgroup's 64 dependent trips
// would take longer than the rest of the binning: every workgroup scans 4,096 cells on its own and leaves its total,
// one workgroup scans the totals, a third pass adds them in.
__global__ void __launch_bounds__(1024) bin_scan_local_kernel(uint32_t ncell, const uint32_t* __restrict__ count, uint32_t* __restrict__ cell_start,
                                                              uint32_t* __restrict__ block_total)
    {
    __shared__ uint32_t s_wave[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t c = blockIdx.x * 4096u + 4u * t;
    uint32_t v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k)
        v[k] = (c + k < ncell) ? count[c + k] : 0u;
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1)
        {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
        if ((int)lane >= off)
            incl += up;
        }
    if (lane == 63u)
        s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < 16u; ++w)
        {
        const uint32_t x = s_wave[w];
        before += (w < wave) ? x : 0u;
        total += x;
        }
    uint32_t acc = before + incl - mine;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k)
        {
        if (c + k < ncell)
            cell_start[c + k] = acc;
        acc += v[k];
        }
    if (t == 0)
        block_total[blockIdx.x] = total;
    }

__global__ void __launch_bounds__(256) bin_scan_add_kernel(uint32_t ncell, const uint32_t* __restrict__ block_start, uint32_t* __restrict__ count,
                                                           uint32_t* __restrict__ cell_start)
    {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c < ncell)
        {
        const uint32_t v = cell_start[c] + block_start[c >> 12];
        cell_start[c] = v;
        count[c] = v;
        }
    else if (c == ncell)
        cell_start[ncell] = block_start[(ncell + 4095u) >> 12]; // (the scan of the totals leaves the grand total behind the last)
    }

__global__ void __launch_bounds__(256) bin_scatter_kernel(uint32_t n_total, const uint32_t* __restrict__ cell_of,
                                                          uint32_t* __restrict__ cursor, uint32_t* __restrict__ order_tmp)
    {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n_total;
    const uint32_t c = active ? cell_of[i] : 0u;
    const uint32_t slot = wave_aggregated_add(cursor, c, active); // (any order of the WAVES inside a cell: the next kernel sorts it)
    if (active)
        order_tmp[slot] = i;
    }

// One wave per cell: the cell's particle indices in ascending order (a rank sort: an index's position is the number
// of smaller ones). Makes the permutation independent of the order in which the atomics above landed: neighbor rows,
// and with them the summation order of the forces, are reproducible from run to run.
__global__ void __launch_bounds__(256) bin_sort_cells_kernel(uint32_t ncell, const uint32_t* __restrict__ cell_start,
                                                             const uint32_t* __restrict__ order_tmp, uint32_t* __restrict__ order)
    {
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= ncell)
        return;
    const uint32_t b = cell_start[c], n = cell_start[c + 1] - b;
    for (uint32_t k0 = 0; k0 < n; k0 += 64u)
        {
        const uint32_t k = k0 + lane;
        const uint32_t mine = (k < n) ? order_tmp[b + k] : 0xffffffffu;
        uint32_t rank = 0;
        for (uint32_t q0 = 0; q0 < n; q0 += 64u)
            {
            const uint32_t other = (q0 + lane < n) ? order_tmp[b + q0 + lane] : 0xffffffffu;
            const uint32_t m = min(64u, n - q0);
            for (uint32_t q = 0; q < m; ++q)
                rank += ((uint32_t)__shfl((int)other, (int)q, 64) < mine) ? 1u : 0u;
            }
        if (k < n)
            order[b + rank] = mine;
        }
    }

// The same for grids of small cells (a few particles each: half-width cells): one THREAD per cell, the cell's entries
// ranked in registers (up to eight; longer cells straight from memory).
__global__ void __launch_bounds__(256) bin_sort_small_cells_kernel(uint32_t ncell, const uint32_t* __restrict__ cell_start,
                                                                   const uint32_t* __restrict__ order_tmp, uint32_t* __restrict__ order)
    {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= ncell)
        return;
    const uint32_t b = cell_start[c], n = cell_start[c + 1] - b;
    if (n <= 8u)
        {
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k)
            v[k] = (k < n) ? order_tmp[b + k] : 0xffffffffu;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k)
            {
            uint32_t rank = 0;
#pragma unroll
            for (uint32_t q = 0; q < 8u; ++q)
                rank += (v[q] < v[k]) ? 1u : 0u;
            if (k < n)
                order[b + rank] = v[k];
            }
        return;
        }
    for (uint32_t k = 0; k < n; ++k)
        {
        const uint32_t mine = order_tmp[b + k];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < n; ++q)
            rank += (order_tmp[b + q] < mine) ? 1u : 0u;
        order[b + rank] = mine;
        }
    }

struct NlistKArgs
    {
    const double* pos;
    const double* rlistsq;
    const uint32_t* cell_of;
    const uint32_t* order;
    const uint32_t* cell_start;
    const uint32_t* n_excl;
    const uint32_t* excl;
    uint64_t excl_pitch;
    uint32_t* n_neigh;
    const uint64_t* head_list;
    uint32_t* nlist;
    uint32_t* max_neigh;
    uint32_t row_capacity;
    BoxDev box;
    GridDev grid;
    uint32_t N;
    uint32_t ntypes;
    };

// ---------------------------------------------------------------------------
// Count / fill, one workgroup per cell.
//
// All particles of a cell share the same <= 27 candidate cells, so the workgroup
// stages those candidates once in LDS (positions as the image nearest to the cell
// centre, original index, type) with contiguous reads of the cell-sorted order,
// and each wave then takes one home particle at a time: the 64 lanes test 64
// consecutive staged candidates (conflict-free LDS reads), a ballot compacts the
// accepted ones, and the row is written as contiguous runs -- the list is
// streamed out once instead of with scattered 4-byte stores.
// FILL = false: count; FILL = true: write rows at head_list[i].
// MI = true: per-pair minimum image (boxes with < 4 cells along a periodic axis,
// triclinic boxes); MI = false: images resolved once per staged candidate.
// Dense cells are handled in batches of NL_CAP candidates x NL_HOME home particles.
// ---------------------------------------------------------------------------
constexpr uint32_t NL_CAP = 1280;
constexpr uint32_t NL_HOME = 1024;
constexpr uint32_t NL_THREADS = 512;
constexpr uint32_t NL_WAVES = NL_THREADS / 64;
constexpr uint32_t NL_PAD_IDX = 0xffffffffu;

struct NlStencil
    {
    uint32_t first[27];
    uint32_t off[28];
    };

template<bool FILL, bool MI> __global__ void __launch_bounds__(NL_THREADS) nlist_cell_kernel(const NlistKArgs a, uint32_t ncell,
                                                                                      uint32_t nblocks_pad8)
    {
    __shared__ double sx[NL_CAP + 128], sy[NL_CAP + 128], sz[NL_CAP + 128];
    __shared__ uint32_t sidx[NL_CAP + 128], styp[NL_CAP + 128];
    __shared__ uint32_t run[NL_HOME];
    __shared__ NlStencil st;

    const uint32_t cell = xcd_remap(blockIdx.x, nblocks_pad8);
    if (cell >= ncell)
        return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int dimx = a.grid.dim[0], dimy = a.grid.dim[1];
    const int cx = cell % dimx, cy = (cell / dimx) % dimy, cz = cell / (dimx * dimy);
    const uint32_t hs = a.cell_start[cell], nhome = a.cell_start[cell + 1] - hs;
    if (nhome == 0)
        return;

    if (tid < 27)
        {
        const int o[3] = {(int)(tid % 3) - 1, (int)((tid / 3) % 3) - 1, (int)(tid / 9) - 1};
        const int c[3] = {cx, cy, cz};
        int n[3];
        bool valid = true;
        for (int k = 0; k < 3; ++k)
            {
            const int d = a.grid.dim[k];
            if (a.grid.periodic[k])
                {
                if (d < 3)
                    {
                    // fewer than 3 cells: visit each cell of the axis exactly once
                    n[k] = o[k] + 1;
                    valid &= n[k] < d;
                    }
                else
                    n[k] = (c[k] + o[k] + d) % d;
                }
            else
                {
                n[k] = c[k] + o[k];
                valid &= (n[k] >= 0 && n[k] < d);
                }
            }
        uint32_t first = 0, cnt = 0;
        if (valid)
            {
            const uint32_t nc = (uint32_t)((n[2] * dimy + n[1]) * dimx + n[0]);
            first = a.cell_start[nc];
            cnt = a.cell_start[nc + 1] - first;
            }
        st.first[tid] = first;
        st.off[tid + 1] = cnt;
        }
    __syncthreads();
    if (tid == 0)
        {
        uint32_t acc = 0;
        st.off[0] = 0;
        for (int s = 1; s <= 27; ++s)
            {
            acc += st.off[s];
            st.off[s] = acc;
            }
        }
    __syncthreads();
    const uint32_t total = st.off[27];

    // cell centre: reference point of the staged images
    const double ccx = a.grid.lo[0] + (cx + 0.5) / a.grid.winv[0];
    const double ccy = a.grid.lo[1] + (cy + 0.5) / a.grid.winv[1];
    const double ccz = a.grid.lo[2] + (cz + 0.5) / a.grid.winv[2];
    const double rl_single = a.rlistsq[0];
    const bool one_type = (a.ntypes == 1);

    for (uint32_t hc = 0; hc < nhome; hc += NL_HOME)
        {
        const uint32_t nh = min(NL_HOME, nhome - hc);
        for (uint32_t t = tid; t < nh; t += NL_THREADS)
            run[t] = 0;
        for (uint32_t b0 = 0; b0 < total; b0 += NL_CAP)
            {
            const uint32_t nb = min(NL_CAP, total - b0);
            __syncthreads(); // previous batch fully consumed (and run[] zeroed)
            // pad to a multiple of 128 so that the test loop needs no bounds checks
            const uint32_t nb_pad = (nb + 127u) & ~127u;
            for (uint32_t t = nb + tid; t < nb_pad; t += NL_THREADS)
                {
                sx[t] = 1e150; sy[t] = 1e150; sz[t] = 1e150;
                sidx[t] = NL_PAD_IDX;
                styp[t] = 0;
                }
            for (uint32_t t = tid; t < nb; t += NL_THREADS)
                {
                const uint32_t g = b0 + t;
                int s = 0;
                while (g >= st.off[s + 1])
                    ++s;
                const uint32_t j = a.order[st.first[s] + (g - st.off[s])];
                const double4 pj = load_scalar4(a.pos, j);
                double x = pj.x, y = pj.y, z = pj.z;
                if (!MI)
                    {
                    if (a.box.px) x = __builtin_fma(-a.box.Lx, rint((x - ccx) * a.box.Lxinv), x);
                    if (a.box.py) y = __builtin_fma(-a.box.Ly, rint((y - ccy) * a.box.Lyinv), y);
                    if (a.box.pz) z = __builtin_fma(-a.box.Lz, rint((z - ccz) * a.box.Lzinv), z);
                    }
                sx[t] = x; sy[t] = y; sz[t] = z;
                sidx[t] = j;
                styp[t] = (uint32_t)type_from_w(pj.w);
                }
            __syncthreads();

            for (uint32_t h = wave; h < nh; h += NL_WAVES)
                {
                const uint32_t i = __builtin_amdgcn_readfirstlane(a.order[hs + hc + h]);
                if (i >= a.N)
                    continue; // ghost: no row
                const double4 pi = load_scalar4(a.pos, i);
                double xi = pi.x, yi = pi.y, zi = pi.z;
                if (!MI)
                    {
                    if (a.box.px) xi = __builtin_fma(-a.box.Lx, rint((xi - ccx) * a.box.Lxinv), xi);
                    if (a.box.py) yi = __builtin_fma(-a.box.Ly, rint((yi - ccy) * a.box.Lyinv), yi);
                    if (a.box.pz) zi = __builtin_fma(-a.box.Lz, rint((zi - ccz) * a.box.Lzinv), zi);
                    }
                const uint32_t typei = (uint32_t)type_from_w(pi.w);
                const uint32_t nex = a.n_excl ? a.n_excl[i] : 0u;
                uint32_t count = run[h];
                uint32_t* out = FILL ? a.nlist + a.head_list[i] : nullptr;
                for (uint32_t k = 0; k < nb_pad; k += 128)
                    {
                    bool accept[2];
                    uint32_t jj[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        {
                        const uint32_t c = k + 64u * u + lane;
                        const uint32_t j = sidx[c];
                        double dx = xi - sx[c], dy = yi - sy[c], dz = zi - sz[c];
                        if (MI)
                            min_image(a.box, dx, dy, dz);
                        const double rsq = dx * dx + dy * dy + dz * dz;
                        const double rl = one_type ? rl_single : a.rlistsq[typei * a.ntypes + styp[c]];
                        bool acc = (j != i) && (j != NL_PAD_IDX) && (rl > 0.0) && (rsq <= rl);
                        for (uint32_t e = 0; e < nex; ++e)
                            acc &= (a.excl[(uint64_t)e * a.excl_pitch + i] != j);
                        accept[u] = acc;
                        jj[u] = j;
                        }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        {
                        const uint64_t mask = __ballot(accept[u]);
                        if (FILL && accept[u])
                            {
                            const uint32_t below = __builtin_amdgcn_mbcnt_hi(
                                (uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            if (!a.row_capacity || count + below < a.row_capacity)
                                out[count + below] = jj[u];
                            }
                        count += (uint32_t)__popcll(mask);
                        }
                    }
                if (lane == 0)
                    run[h] = count;
                }
            }
        if (!FILL || a.row_capacity)
            {
            __syncthreads();
            uint32_t most = 0;
            for (uint32_t t = tid; t < nh; t += NL_THREADS)
                {
                const uint32_t i = a.order[hs + hc + t];
                if (i < a.N)
                    {
                    a.n_neigh[i] = run[t];
                    most = max(most, run[t]);
                    }
                }
            if (FILL && most > a.row_capacity)
                atomicMax(a.max_neigh, most); // overflow is rare: no reduction needed
            }
        __syncthreads();
        }
    }

__global__ void __launch_bounds__(256) distance_check_kernel(uint32_t n, const double* __restrict__ pos,
                                                             const double* __restrict__ pos0, BoxDev box, double max_dist_sq,
                                                             uint32_t* __restrict__ flag, unsigned long long* __restrict__ max_bits,
                                                             float* __restrict__ disp)
    {
    // grid-stride over at most 512 workgroups, each ending in at most one atomic
    double dsq_max = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        {
        const double3 p = load_scalar3_of4(pos, i), q = load_scalar3_of4(pos0, i);
        double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
        min_image(box, dx, dy, dz);
        const double dsq = dx * dx + dy * dy + dz * dz;
        dsq_max = fmax(dsq_max, dsq);
        if (disp)
            {
            // an UPPER bound in single precision (the tile kernels take maxima of these); NaN -> +inf
            const double d = sqrt(dsq) * (1.0 + 1e-15);
            disp[i] = (d == d) ? __double2float_ru(d) : __int_as_float(0x7f800000);
            }
        }
    for (int off = 32; off > 0; off >>= 1)
        dsq_max = fmax(dsq_max, __shfl_xor(dsq_max, off, 64));
    // one atomic per workgroup: 16 k waves hammering one address cost more than the reads
    __shared__ double s_wave_max[4];
    if ((threadIdx.x & 63) == 0)
        s_wave_max[threadIdx.x >> 6] = dsq_max;
    __syncthreads();
    if (threadIdx.x == 0)
        {
        dsq_max = fmax(fmax(s_wave_max[0], s_wave_max[1]), fmax(s_wave_max[2], s_wave_max[3]));
        if (dsq_max > max_dist_sq)
            atomicOr(flag, 1u);
        if (max_bits && dsq_max > 0.0)
            {
            // the bits of non-negative doubles order like the values; a plain read first
            // keeps most workgroups off the atomic
            const unsigned long long bits = (unsigned long long)__double_as_longlong(dsq_max);
            if (bits > *reinterpret_cast<volatile unsigned long long*>(max_bits))
                atomicMax(max_bits, bits);
            }
        }
    }

static int check_nlist_args(const azp_nlist_args* a)
    {
    if (!a || !a->d_pos || a->n_total < a->N || a->ntypes == 0)
        return AZP_ERROR_INVALID_ARGUMENT;
    for (int k = 0; k < 3; ++k)
        if (a->grid.dim[k] == 0 || !(a->grid.width[k] > 0.0))
            return AZP_ERROR_INVALID_ARGUMENT;
    return 0;
    }

static NlistKArgs make_nlist_kargs(const azp_nlist_args& a)
    {
    NlistKArgs k;
    k.pos = a.d_pos;
    k.rlistsq = a.d_rlistsq;
    k.cell_of = a.d_cell_of;
    k.order = a.d_order;
    k.cell_start = a.d_cell_start;
    k.n_excl = a.d_n_excl;
    k.excl = a.d_excl;
    k.excl_pitch = a.excl_pitch;
    k.n_neigh = a.d_n_neigh;
    k.head_list = a.d_head_list;
    k.nlist = a.d_nlist;
    k.max_neigh = a.d_max_neigh;
    k.row_capacity = a.row_capacity;
    k.box = make_box_dev(a.box);
    k.grid = make_grid_dev(a.grid);
    k.N = a.N;
    k.ntypes = a.ntypes;
    return k;
    }
} // namespace azp

extern "C" int azp_nlist_cell_assign(const azp_nlist_args* args, void* stream)
    {
    using namespace azp;
    if (check_nlist_args(args) || !args->d_cell_of)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (args->n_total == 0)
        return AZP_SUCCESS;
    const uint32_t grid = (args->n_total + 255u) / 256u;
    hipLaunchKernelGGL(cell_assign_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), args->n_total,
                       args->d_pos, make_grid_dev(args->grid), args->d_cell_of);
    return (int)hipGetLastError();
    }

extern "C" int azp_nlist_bin(const azp_nlist_args* args, uint32_t* d_cursor, uint32_t* d_order_tmp, void* stream)
    {
    using namespace azp;
    if (check_nlist_args(args) || !args->d_cell_of || !args->d_order || !args->d_cell_start || !d_cursor || !d_order_tmp)
        return AZP_ERROR_INVALID_ARGUMENT;
    const uint32_t ncell = args->grid.dim[0] * args->grid.dim[1] * args->grid.dim[2];
    const hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(d_cursor, 0, sizeof(uint32_t) * (size_t)ncell, s);
    if (e != hipSuccess)
        return (int)e;
    const uint32_t n = args->n_total;
    if (n)
        hipLaunchKernelGGL(bin_assign_count_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, n, args->d_pos, make_grid_dev(args->grid),
                           args->d_cell_of, d_cursor);
    const uint32_t nblk = (ncell + 4095u) / 4096u;
    if (nblk > 8u && nblk + 1u <= n)
        {
        // (the totals live in d_order_tmp, which the scatter fills only afterwards; bin_scan_kernel turns n totals into
        // their n starts + the grand total at [n])
        hipLaunchKernelGGL(bin_scan_local_kernel, dim3(nblk), dim3(1024), 0, s, ncell, d_cursor, args->d_cell_start, d_order_tmp);
        hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, s, nblk, d_order_tmp, d_order_tmp);
        hipLaunchKernelGGL(bin_scan_add_kernel, dim3((ncell + 1u + 255u) / 256u), dim3(256), 0, s, ncell, d_order_tmp, d_cursor, args->d_cell_start);
        }
    else
        hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, s, ncell, d_cursor, args->d_cell_start);
    if (n)
        {
        hipLaunchKernelGGL(bin_scatter_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, n, args->d_cell_of, d_cursor, d_order_tmp);
        if ((uint64_t)n <= 6ull * ncell)
            hipLaunchKernelGGL(bin_sort_small_cells_kernel, dim3((ncell + 255u) / 256u), dim3(256), 0, s, ncell, args->d_cell_start, d_order_tmp,
                               const_cast<uint32_t*>(args->d_order));
        else
            hipLaunchKernelGGL(bin_sort_cells_kernel, dim3((ncell + 3u) / 4u), dim3(256), 0, s, ncell, args->d_cell_start, d_order_tmp,
                               const_cast<uint32_t*>(args->d_order));
        }
    return (int)hipGetLastError();
    }

extern "C" int azp_nlist_cell_bounds(const azp_nlist_args* args, void* stream)
    {
    using namespace azp;
    if (check_nlist_args(args) || !args->d_cell_sorted || !args->d_cell_start)
        return AZP_ERROR_INVALID_ARGUMENT;
    const uint32_t ncell = args->grid.dim[0] * args->grid.dim[1] * args->grid.dim[2];
    const uint32_t grid = (ncell + 1 + 255u) / 256u;
    hipLaunchKernelGGL(cell_bounds_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), args->n_total,
                       ncell, args->d_cell_sorted, args->d_cell_start);
    return (int)hipGetLastError();
    }

static int nlist_scan(const azp_nlist_args* args, void* stream, bool fill)
    {
    using namespace azp;
    if (check_nlist_args(args) || !args->d_cell_of || !args->d_order || !args->d_cell_start || !args->d_rlistsq)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (fill ? (!args->d_head_list || !args->d_nlist) : !args->d_n_neigh)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (args->cell_subdivision > 1)
        return AZP_ERROR_INVALID_ARGUMENT; // the 27-cell search needs cells as wide as the list radius
    if (fill && args->row_capacity && (!args->d_n_neigh || !args->d_max_neigh))
        return AZP_ERROR_INVALID_ARGUMENT;
    if (args->N == 0)
        return AZP_SUCCESS;
    const NlistKArgs k = make_nlist_kargs(*args);
    const uint32_t ncell = args->grid.dim[0] * args->grid.dim[1] * args->grid.dim[2];
    const uint32_t grid = (ncell + 7u) & ~7u;
    // images can be resolved once per staged candidate when every periodic axis
    // has >= 4 cells (cell width + r_list <= L/2 with margin) and the box is orthorhombic
    bool mi = k.box.triclinic;
    for (int d = 0; d < 3; ++d)
        if (args->grid.periodic[d] && args->grid.dim[d] < 4)
            mi = true;
    const hipStream_t s = static_cast<hipStream_t>(stream);
    if (fill)
        {
        if (mi) hipLaunchKernelGGL((nlist_cell_kernel<true, true>), dim3(grid), dim3(NL_THREADS), 0, s, k, ncell, grid);
        else hipLaunchKernelGGL((nlist_cell_kernel<true, false>), dim3(grid), dim3(NL_THREADS), 0, s, k, ncell, grid);
        }
    else
        {
        if (mi) hipLaunchKernelGGL((nlist_cell_kernel<false, true>), dim3(grid), dim3(NL_THREADS), 0, s, k, ncell, grid);
        else hipLaunchKernelGGL((nlist_cell_kernel<false, false>), dim3(grid), dim3(NL_THREADS), 0, s, k, ncell, grid);
        }
    return (int)hipGetLastError();
    }

extern "C" int azp_nlist_count(const azp_nlist_args* args, void* stream) { return nlist_scan(args, stream, false); }
extern "C" int azp_nlist_fill(const azp_nlist_args* args, void* stream) { return nlist_scan(args, stream, true); }

extern "C" int azp_nlist_distance_check(uint32_t n, const double* d_pos, const double* d_pos_at_build, const azp_box* box,
                                        double max_dist_sq, uint32_t* d_flag, unsigned long long* d_max_dist_sq_bits,
                                        void* stream)
    {
    using namespace azp;
    if (!d_pos || !d_pos_at_build || !box || !d_flag || !(max_dist_sq >= 0.0))
        return AZP_ERROR_INVALID_ARGUMENT;
    if (n == 0)
        return AZP_SUCCESS;
    const uint32_t blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(distance_check_kernel, dim3(blocks < 512u ? blocks : 512u), dim3(256), 0, static_cast<hipStream_t>(stream), n, d_pos,
                       d_pos_at_build, make_box_dev(*box), max_dist_sq, d_flag, d_max_dist_sq_bits, (float*)nullptr);
    return (int)hipGetLastError();
    }

extern "C" int azp_nlist_displacements(uint32_t n, const double* d_pos, const double* d_pos_at_build, const azp_box* box,
                                       double max_dist_sq, uint32_t* d_flag, unsigned long long* d_max_dist_sq_bits,
                                       float* d_displacement, void* stream)
    {
    using namespace azp;
    if (!d_pos || !d_pos_at_build || !box || !d_flag || !d_displacement || !(max_dist_sq >= 0.0))
        return AZP_ERROR_INVALID_ARGUMENT;
    if (n == 0)
        return AZP_SUCCESS;
    const uint32_t blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(distance_check_kernel, dim3(blocks < 512u ? blocks : 512u), dim3(256), 0, static_cast<hipStream_t>(stream), n, d_pos,
                       d_pos_at_build, make_box_dev(*box), max_dist_sq, d_flag, d_max_dist_sq_bits, d_displacement);
    return (int)hipGetLastError();
    }
