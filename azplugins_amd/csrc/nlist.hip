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
    BoxDev box;
    GridDev grid;
    uint32_t N;
    uint32_t ntypes;
    };

// FILL = false: count neighbors; FILL = true: write them at head_list[i].
template<bool FILL> __global__ void __launch_bounds__(256) nlist_scan_kernel(const NlistKArgs a)
    {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N)
        return;
    const double4 pi = load_scalar4(a.pos, i);
    const int typei = type_from_w(pi.w);
    const uint32_t ci = a.cell_of[i];
    const int cx = ci % a.grid.dim[0], cy = (ci / a.grid.dim[0]) % a.grid.dim[1], cz = ci / (a.grid.dim[0] * a.grid.dim[1]);
    const uint32_t nex = a.n_excl ? a.n_excl[i] : 0u;
    uint32_t count = 0;
    uint32_t* out = FILL ? a.nlist + a.head_list[i] : nullptr;

    // offset ranges per axis: a periodic axis with fewer than 3 cells must not
    // visit the same cell twice
    int lo[3], hi[3];
    for (int k = 0; k < 3; ++k)
        {
        const int d = a.grid.dim[k];
        if (a.grid.periodic[k] && d < 3) { lo[k] = 0; hi[k] = d - 1; }
        else { lo[k] = -1; hi[k] = 1; }
        }
    for (int oz = lo[2]; oz <= hi[2]; ++oz)
        {
        int nz = cz + oz;
        if (a.grid.periodic[2]) nz = (nz + a.grid.dim[2]) % a.grid.dim[2];
        else if (nz < 0 || nz >= a.grid.dim[2]) continue;
        for (int oy = lo[1]; oy <= hi[1]; ++oy)
            {
            int ny = cy + oy;
            if (a.grid.periodic[1]) ny = (ny + a.grid.dim[1]) % a.grid.dim[1];
            else if (ny < 0 || ny >= a.grid.dim[1]) continue;
            for (int ox = lo[0]; ox <= hi[0]; ++ox)
                {
                int nx = cx + ox;
                if (a.grid.periodic[0]) nx = (nx + a.grid.dim[0]) % a.grid.dim[0];
                else if (nx < 0 || nx >= a.grid.dim[0]) continue;
                const uint32_t nc = (uint32_t)((nz * a.grid.dim[1] + ny) * a.grid.dim[0] + nx);
                const uint32_t qb = a.cell_start[nc], qe = a.cell_start[nc + 1];
                for (uint32_t q = qb; q < qe; ++q)
                    {
                    const uint32_t j = a.order[q];
                    if (j == i)
                        continue;
                    const double4 pj = load_scalar4(a.pos, j);
                    double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                    min_image(a.box, dx, dy, dz);
                    const double rsq = dx * dx + dy * dy + dz * dz;
                    const double rl = a.rlistsq[(uint32_t)typei * a.ntypes + (uint32_t)type_from_w(pj.w)];
                    if (rl <= 0.0 || rsq > rl)
                        continue;
                    bool excluded = false;
                    for (uint32_t e = 0; e < nex; ++e)
                        excluded |= (a.excl[(uint64_t)e * a.excl_pitch + i] == j);
                    if (excluded)
                        continue;
                    if (FILL)
                        out[count] = j;
                    ++count;
                    }
                }
            }
        }
    if (!FILL)
        a.n_neigh[i] = count;
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
    if (args->N == 0)
        return AZP_SUCCESS;
    const NlistKArgs k = make_nlist_kargs(*args);
    const uint32_t grid = (args->N + 255u) / 256u;
    if (fill)
        hipLaunchKernelGGL(nlist_scan_kernel<true>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), k);
    else
        hipLaunchKernelGGL(nlist_scan_kernel<false>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), k);
    return (int)hipGetLastError();
    }

extern "C" int azp_nlist_count(const azp_nlist_args* args, void* stream) { return nlist_scan(args, stream, false); }
extern "C" int azp_nlist_fill(const azp_nlist_args* args, void* stream) { return nlist_scan(args, stream, true); }
