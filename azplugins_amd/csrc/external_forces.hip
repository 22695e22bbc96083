// external_forces.hip -- one-body harmonic barrier forces:
// planar (src/PlanarBarrierEvaluator.h:36-48) and spherical
// (src/SphericalBarrierEvaluator.h:36-51) evaluators inside the loop of
// src/HarmonicBarrier.h:150-177 / src/HarmonicBarrierGPU.cuh:49-84.
// Pure streaming: 32 B in, 32 B out per particle, one lane per particle,
// per-type (k, offset) pairs in LDS.
#include <algorithm>

#include "azp_device.hpp"
#include "pair_kernel_host.hpp"

namespace azp
{
struct BarrierKArgs
    {
    double* force;
    const double* pos;
    const double* params;
    BoxDev box;
    double location;
    uint32_t N;
    uint32_t ntypes;
    };

// HOOMD BoxDim::wrap restated for one shift per axis (particles drift by much
// less than a box length between wraps)
__device__ __forceinline__ void wrap_into_box(const BoxDev& b, double& x, double& y, double& z)
    {
    if (b.pz)
        {
        const double h = 0.5 * b.Lz;
        if (z >= h) { z -= b.Lz; y -= b.Lz * b.yz; x -= b.Lz * b.xz; }
        else if (z < -h) { z += b.Lz; y += b.Lz * b.yz; x += b.Lz * b.xz; }
        }
    if (b.py)
        {
        const double h = 0.5 * b.Ly, s = z * b.yz;
        if (y >= h + s) { y -= b.Ly; x -= b.Ly * b.xy; }
        else if (y < -h + s) { y += b.Ly; x += b.Ly * b.xy; }
        }
    if (b.px)
        {
        const double h = 0.5 * b.Lx, s = y * b.xy + z * (b.xz - b.xy * b.yz);
        if (x >= h + s) x -= b.Lx;
        else if (x < -h + s) x += b.Lx;
        }
    }

template<bool SPHERICAL> __global__ void __launch_bounds__(256) barrier_kernel(const BarrierKArgs a)
    {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    double2* s_params = reinterpret_cast<double2*>(s_raw);
    for (uint32_t t = threadIdx.x; t < a.ntypes; t += blockDim.x)
        s_params[t] = make_double2(a.params[2 * t], a.params[2 * t + 1]);
    __syncthreads();
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N)
        return;
    const double4 p = load_scalar4(a.pos, idx);
    double x = p.x, y = p.y, z = p.z;
    wrap_into_box(a.box, x, y, z);
    const double2 prm = s_params[type_from_w(p.w)];
    double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
    if (SPHERICAL)
        {
        const double r = sqrt(x * x + y * y + z * z);
        const double dr = r - (a.location + prm.y);
        if (dr > 0.0)
            {
            const double k_dr = prm.x * dr;
            const double s = -(k_dr / r);
            fx = s * x; fy = s * y; fz = s * z;
            e = 0.5 * k_dr * dr;
            }
        }
    else
        {
        const double dy = y - (a.location + prm.y);
        if (dy > 0.0)
            {
            fy = -prm.x * dy;
            e = -0.5 * fy * dy;
            }
        }
    store_scalar4(a.force, idx, fx, fy, fz, e);
    }

template<bool SPHERICAL> static int launch_barrier(const azp_barrier_args* args, void* stream)
    {
    if (!args)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (args->N == 0)
        return AZP_SUCCESS;
    if (!args->d_force || !args->d_pos || !args->d_params || args->ntypes == 0)
        return AZP_ERROR_INVALID_ARGUMENT;
    const uint32_t bs = args->block_size ? args->block_size : 256u;
    if (bs % 64 || bs > 256)
        return AZP_ERROR_INVALID_ARGUMENT;
    const size_t lds = sizeof(double2) * (size_t)args->ntypes;
    if (lds > 64 * 1024)
        return AZP_ERROR_TOO_MANY_TYPES;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (args->d_virial)
        {
        // the reference sets the virial to zero (src/HarmonicBarrier.h:179-180)
        hipError_t e = hipMemsetAsync(args->d_virial, 0, sizeof(double) * 6 * args->virial_pitch, s);
        if (e != hipSuccess)
            return (int)e;
        }
    BarrierKArgs k;
    k.force = args->d_force;
    k.pos = args->d_pos;
    k.params = args->d_params;
    k.box = make_box_dev(args->box);
    k.location = args->location;
    k.N = args->N;
    k.ntypes = args->ntypes;
    const uint32_t grid = (args->N + bs - 1) / bs;
    LaunchInfo& li = last_launch();
    li.block_size = bs; li.tpp = 1; li.grid = grid; li.lds_bytes = (uint32_t)lds;
    hipLaunchKernelGGL(barrier_kernel<SPHERICAL>, dim3(grid), dim3(bs), lds, s, k);
    return (int)hipGetLastError();
    }

// ---------------------------------------------------------------------------
// velocity-Verlet NVE
// ---------------------------------------------------------------------------
struct NVEKArgs
    {
    double* pos;
    double* vel;
    const double* net_force;
    int32_t* image;
    BoxDev box;
    double dt;
    uint32_t N;
    };

// MODE 0: step two (v += a dt/2). 1: step one (v += a dt/2, x += v dt, wrap). 2: step two of one step and step
// one of the next in one pass over the arrays -- the two half kicks use the same force (no force evaluation lies
// between them) and are applied one after the other exactly as the two kernels apply them: bit-identical, 184
// instead of 280 bytes per particle.
template<int MODE> __global__ void __launch_bounds__(256) nve_kernel(const NVEKArgs a)
    {
    constexpr bool STEP_ONE = MODE != 0;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N)
        return;
    double4 v = load_scalar4(a.vel, idx);
    const double4 f = load_scalar4(a.net_force, idx);
    const double minv = 1.0 / v.w;
    const double hdt = 0.5 * a.dt;
    if (MODE == 2)
        {
        v.x += hdt * f.x * minv; v.y += hdt * f.y * minv; v.z += hdt * f.z * minv;
        }
    v.x += hdt * f.x * minv; v.y += hdt * f.y * minv; v.z += hdt * f.z * minv;
    store_scalar4(a.vel, idx, v.x, v.y, v.z, v.w);
    if (STEP_ONE)
        {
        const double4 p = load_scalar4(a.pos, idx);
        double x = p.x + a.dt * v.x, y = p.y + a.dt * v.y, z = p.z + a.dt * v.z;
        const double x0 = x, y0 = y, z0 = z;
        wrap_into_box(a.box, x, y, z);
        store_scalar4(a.pos, idx, x, y, z, p.w);
        if (a.image)
            {
            // which way was it wrapped (orthorhombic shortcut is exact; for triclinic
            // boxes the z shift is read off z, the y shift off y after removing z's tilt)
            const int iz = (z < z0) - (z > z0);
            const double y1 = y0 - iz * a.box.Lz * a.box.yz;
            const int iy = (y < y1) - (y > y1);
            const double x1 = x0 - iz * a.box.Lz * a.box.xz - iy * a.box.Ly * a.box.xy;
            const int ix = (x < x1) - (x > x1);
            a.image[3 * idx + 0] += ix; a.image[3 * idx + 1] += iy; a.image[3 * idx + 2] += iz;
            }
        }
    }

template<int MODE> static int launch_nve(const azp_nve_args* args, void* stream)
    {
    if (!args)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (args->N == 0)
        return AZP_SUCCESS;
    if (!args->d_pos || !args->d_vel || !args->d_net_force)
        return AZP_ERROR_INVALID_ARGUMENT;
    const uint32_t bs = args->block_size ? args->block_size : 256u;
    if (bs % 64 || bs > 256)
        return AZP_ERROR_INVALID_ARGUMENT;
    NVEKArgs k;
    k.pos = args->d_pos;
    k.vel = args->d_vel;
    k.net_force = args->d_net_force;
    k.image = args->d_image;
    k.box = make_box_dev(args->box);
    k.dt = args->dt;
    k.N = args->N;
    const uint32_t grid = (args->N + bs - 1) / bs;
    hipLaunchKernelGGL(nve_kernel<MODE>, dim3(grid), dim3(bs), 0, static_cast<hipStream_t>(stream), k);
    return (int)hipGetLastError();
    }
} // namespace azp

extern "C" int azp_external_planar_harmonic_barrier(const azp_barrier_args* args, void* stream)
    {
    return azp::launch_barrier<false>(args, stream);
    }
extern "C" int azp_external_spherical_harmonic_barrier(const azp_barrier_args* args, void* stream)
    {
    return azp::launch_barrier<true>(args, stream);
    }
extern "C" int azp_integrate_nve_step_one(const azp_nve_args* args, void* stream)
    {
    return azp::launch_nve<1>(args, stream);
    }
extern "C" int azp_integrate_nve_step_two(const azp_nve_args* args, void* stream)
    {
    return azp::launch_nve<0>(args, stream);
    }
extern "C" int azp_integrate_nve_step_two_one(const azp_nve_args* args, void* stream)
    {
    return azp::launch_nve<2>(args, stream);
    }

// Net force of several ForceComputes in ONE pass (HOOMD: Integrator::computeNetForce sums the forces' arrays):
// out = f[0] + f[1] + ... row by row, every array read once, the sum written once (a chain of framework
// element-wise operations would write and re-read the 32 MB accumulator once per force).
namespace azp
{
struct SumForcesArgs
    {
    const double* f[8];
    double* out;
    uint32_t n_rows, k;
    };
__global__ void __launch_bounds__(256) sum_forces_kernel(const SumForcesArgs a)
    {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n_rows)
        return;
    double4 v = load_scalar4(a.f[0], i);
    for (uint32_t q = 1; q < a.k; ++q)
        {
        const double4 w = load_scalar4(a.f[q], i);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
    store_scalar4(a.out, i, v.x, v.y, v.z, v.w);
    }
} // namespace azp

extern "C" int azp_sum_forces(uint32_t n_rows, uint32_t n_arrays, const double* const* d_arrays, double* d_out, void* stream)
    {
    using namespace azp;
    if (!d_arrays || !d_out || n_arrays == 0 || n_arrays > 8)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (n_rows == 0)
        return AZP_SUCCESS;
    SumForcesArgs a;
    for (uint32_t q = 0; q < 8; ++q)
        a.f[q] = (q < n_arrays) ? d_arrays[q] : nullptr;
    for (uint32_t q = 0; q < n_arrays; ++q)
        if (!a.f[q])
            return AZP_ERROR_INVALID_ARGUMENT;
    a.out = d_out;
    a.n_rows = n_rows;
    a.k = n_arrays;
    hipLaunchKernelGGL(sum_forces_kernel, dim3((n_rows + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
    }

// src/PlanarBarrierEvaluator.h:50-55: H inside [lo.y, hi.y), lo / hi = box.makeCoordinates((0,0,0)) /
// ((1,1,1)). HOOMD's makeCoordinates shears the fractional point, y += yz * z, so for a
// triclinic box the corners sit at y = -+(Ly / 2 + yz Lz / 2).
extern "C" int azp_planar_barrier_valid(double H, const azp_box* box)
    {
    if (!box)
        return 0;
    const double half = 0.5 * box->L[1] + box->tilt[2] * 0.5 * box->L[2];
    return (H >= -half && H < half) ? 1 : 0;
    }
// src/SphericalBarrierEvaluator.h:53-59: R >= 0 and 2 R <= nearest plane distance
extern "C" int azp_spherical_barrier_valid(double R, const azp_box* box)
    {
    if (!box)
        return 0;
    // nearest plane distances of a triclinic box (HOOMD BoxDim::getNearestPlaneDistance)
    const double xy = box->tilt[0], xz = box->tilt[1], yz = box->tilt[2];
    const double term = xy * yz - xz;
    const double dx = box->L[0] / sqrt(1.0 + xy * xy + term * term);
    const double dy = box->L[1] / sqrt(1.0 + yz * yz);
    const double dz = box->L[2];
    const double two_R = 2.0 * R;
    return (R >= 0.0 && dx >= two_R && dy >= two_R && dz >= two_R) ? 1 : 0;
    }

// ---- halo pack: gather rows into the send buffer ----
namespace azp
{
__global__ void __launch_bounds__(256) halo_pack_kernel(uint32_t n, const double* __restrict__ src, const int64_t* __restrict__ idx,
                                                        uint32_t row_doubles, double* __restrict__ dst)
    {
    // one lane per 16 bytes: a 4-double row is two lanes, consecutive lanes write consecutive addresses
    const uint32_t per_row = row_doubles / 2;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n * per_row)
        return;
    const uint32_t k = (uint32_t)(t / per_row), part = (uint32_t)(t % per_row);
    const double2 v = *reinterpret_cast<const double2*>(src + (uint64_t)idx[k] * row_doubles + 2 * part);
    *reinterpret_cast<double2*>(dst + (uint64_t)k * row_doubles + 2 * part) = v;
    }
} // namespace azp

extern "C" int azp_halo_pack(uint32_t n, const double* d_src, const int64_t* d_idx, uint32_t row_doubles, double* d_dst, void* stream)
    {
    if (n == 0)
        return AZP_SUCCESS; // a rank without peers: empty buffers, possibly null
    if (!d_src || !d_idx || !d_dst || row_doubles == 0 || (row_doubles & 1u))
        return AZP_ERROR_INVALID_ARGUMENT;
    const uint64_t lanes = (uint64_t)n * (row_doubles / 2);
    hipLaunchKernelGGL(azp::halo_pack_kernel, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                       d_src, d_idx, row_doubles, d_dst);
    return (int)hipGetLastError();
    }

// ---- packed halo rows: several per-particle arrays in ONE send buffer ----
namespace azp
{
struct HaloFieldsK
    {
    const unsigned char* src[AZP_HALO_MAX_FIELDS];
    unsigned char* dst[AZP_HALO_MAX_FIELDS];
    uint32_t row_bytes[AZP_HALO_MAX_FIELDS]; // multiple of 4
    uint32_t offset[AZP_HALO_MAX_FIELDS];    // byte offset of the field inside a packed row
    uint32_t n_fields;
    uint32_t packed_row_bytes;               // multiple of 8
    };

// one lane per 4-byte word of a packed row
template<bool PACK>
__global__ void __launch_bounds__(256) halo_fields_kernel(uint32_t n, const HaloFieldsK f, const int64_t* __restrict__ idx,
                                                          unsigned char* __restrict__ packed)
    {
    const uint32_t words = f.packed_row_bytes / 4;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n * words)
        return;
    const uint32_t k = (uint32_t)(t / words), byte = (uint32_t)(t % words) * 4u;
    uint32_t* slot = reinterpret_cast<uint32_t*>(packed + (uint64_t)k * f.packed_row_bytes + byte);
    for (uint32_t c = 0; c < f.n_fields; ++c)
        {
        if (byte >= f.offset[c] && byte < f.offset[c] + f.row_bytes[c])
            {
            if (PACK)
                *slot = *reinterpret_cast<const uint32_t*>(f.src[c] + (uint64_t)idx[k] * f.row_bytes[c] + (byte - f.offset[c]));
            else
                *reinterpret_cast<uint32_t*>(f.dst[c] + (uint64_t)k * f.row_bytes[c] + (byte - f.offset[c])) = *slot;
            return;
            }
        }
    if (PACK)
        *slot = 0u; // padding between fields
    }

template<bool PACK> int launch_halo_fields(uint32_t n, uint32_t n_fields, const azp_halo_field* fields, const int64_t* d_idx, void* d_packed,
                                           uint32_t packed_row_bytes, void* stream)
    {
    if (n == 0)
        return AZP_SUCCESS;
    if (!fields || !d_packed || n_fields == 0 || n_fields > AZP_HALO_MAX_FIELDS || (PACK && !d_idx) || (packed_row_bytes & 7u))
        return AZP_ERROR_INVALID_ARGUMENT;
    HaloFieldsK k;
    uint32_t off = 0;
    for (uint32_t c = 0; c < AZP_HALO_MAX_FIELDS; ++c)
        {
        k.src[c] = nullptr; k.dst[c] = nullptr; k.row_bytes[c] = 0; k.offset[c] = 0;
        }
    for (uint32_t c = 0; c < n_fields; ++c)
        {
        if (!fields[c].d_data || fields[c].row_bytes == 0 || (fields[c].row_bytes & 3u))
            return AZP_ERROR_INVALID_ARGUMENT;
        k.src[c] = static_cast<const unsigned char*>(fields[c].d_data);
        k.dst[c] = static_cast<unsigned char*>(fields[c].d_data);
        k.row_bytes[c] = fields[c].row_bytes;
        k.offset[c] = off;
        off += (fields[c].row_bytes + 7u) & ~7u; // every field starts on an 8-byte boundary
        }
    if (off != packed_row_bytes)
        return AZP_ERROR_INVALID_ARGUMENT;
    k.n_fields = n_fields;
    k.packed_row_bytes = packed_row_bytes;
    const uint64_t lanes = (uint64_t)n * (packed_row_bytes / 4);
    hipLaunchKernelGGL(halo_fields_kernel<PACK>, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, k,
                       d_idx, static_cast<unsigned char*>(d_packed));
    return (int)hipGetLastError();
    }
} // namespace azp

extern "C" int azp_halo_pack_fields(uint32_t n, uint32_t n_fields, const azp_halo_field* fields, const int64_t* d_idx, void* d_packed,
                                    uint32_t packed_row_bytes, void* stream)
    {
    return azp::launch_halo_fields<true>(n, n_fields, fields, d_idx, d_packed, packed_row_bytes, stream);
    }

extern "C" int azp_halo_unpack_fields(uint32_t n, uint32_t n_fields, const azp_halo_field* fields, const void* d_packed,
                                      uint32_t packed_row_bytes, void* stream)
    {
    return azp::launch_halo_fields<false>(n, n_fields, fields, nullptr, const_cast<void*>(d_packed), packed_row_bytes, stream);
    }

// ---- rotational half of the NVE step (azp_nve_rot_args, include/azp.h) ----
namespace azp
{
struct Quat
    {
    double s, x, y, z;
    };
__device__ __forceinline__ void free_rotation(int axis, Quat& p, Quat& q, double I, double dt)
    {
    Quat pk, qk;
    if (axis == 3)
        {
        pk = Quat {-p.z, p.y, -p.x, p.s};
        qk = Quat {-q.z, q.y, -q.x, q.s};
        }
    else if (axis == 2)
        {
        pk = Quat {-p.y, -p.z, p.s, p.x};
        qk = Quat {-q.y, -q.z, q.s, q.x};
        }
    else
        {
        pk = Quat {-p.x, p.s, p.z, -p.y};
        qk = Quat {-q.x, q.s, q.z, -q.y};
        }
    const double phi = 0.25 / I * (p.s * qk.s + p.x * qk.x + p.y * qk.y + p.z * qk.z);
    const double c = cos(dt * phi), sn = sin(dt * phi);
    p = Quat {c * p.s + sn * pk.s, c * p.x + sn * pk.x, c * p.y + sn * pk.y, c * p.z + sn * pk.z};
    q = Quat {c * q.s + sn * qk.s, c * q.x + sn * qk.x, c * q.y + sn * qk.y, c * q.z + sn * qk.z};
    }

template<bool STEP_ONE> __global__ void __launch_bounds__(256) nve_rot_kernel(const azp_nve_rot_args a)
    {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N)
        return;
    const double4 q4 = load_scalar4(a.d_orientation, i), p4 = load_scalar4(a.d_angmom, i), t4 = load_scalar4(a.d_net_torque, i);
    Quat q = {q4.x, q4.y, q4.z, q4.w}, p = {p4.x, p4.y, p4.z, p4.w};
    const double Ix = a.d_inertia[3ull * i], Iy = a.d_inertia[3ull * i + 1], Iz = a.d_inertia[3ull * i + 2];
    // torque into the body frame: rotate(conj(q), t)
    const double ux = -q.x, uy = -q.y, uz = -q.z;
    const double c = q.s * q.s - (ux * ux + uy * uy + uz * uz);
    const double d = 2.0 * (ux * t4.x + uy * t4.y + uz * t4.z);
    double tx = c * t4.x + 2.0 * q.s * (uy * t4.z - uz * t4.y) + d * ux;
    double ty = c * t4.y + 2.0 * q.s * (uz * t4.x - ux * t4.z) + d * uy;
    double tz = c * t4.z + 2.0 * q.s * (ux * t4.y - uy * t4.x) + d * uz;
    if (Ix == 0.0) tx = 0.0;
    if (Iy == 0.0) ty = 0.0;
    if (Iz == 0.0) tz = 0.0;
    // p += dt * q * (0, t)
    p.s += a.dt * -(q.x * tx + q.y * ty + q.z * tz);
    p.x += a.dt * (q.s * tx + (q.y * tz - q.z * ty));
    p.y += a.dt * (q.s * ty + (q.z * tx - q.x * tz));
    p.z += a.dt * (q.s * tz + (q.x * ty - q.y * tx));
    if (STEP_ONE)
        {
        if (Iz != 0.0) free_rotation(3, p, q, Iz, 0.5 * a.dt);
        if (Iy != 0.0) free_rotation(2, p, q, Iy, 0.5 * a.dt);
        if (Ix != 0.0) free_rotation(1, p, q, Ix, a.dt);
        if (Iy != 0.0) free_rotation(2, p, q, Iy, 0.5 * a.dt);
        if (Iz != 0.0) free_rotation(3, p, q, Iz, 0.5 * a.dt);
        const double n = 1.0 / sqrt(q.s * q.s + q.x * q.x + q.y * q.y + q.z * q.z);
        store_scalar4(a.d_orientation, i, q.s * n, q.x * n, q.y * n, q.z * n);
        }
    store_scalar4(a.d_angmom, i, p.s, p.x, p.y, p.z);
    }

template<bool STEP_ONE> static int launch_nve_rot(const azp_nve_rot_args* args, void* stream)
    {
    if (!args)
        return AZP_ERROR_INVALID_ARGUMENT;
    if (args->N == 0)
        return AZP_SUCCESS;
    if (!args->d_orientation || !args->d_angmom || !args->d_inertia || !args->d_net_torque)
        return AZP_ERROR_INVALID_ARGUMENT;
    const uint32_t bs = args->block_size ? args->block_size : 256u;
    if (bs % 64 || bs > 256)
        return AZP_ERROR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(nve_rot_kernel<STEP_ONE>, dim3((args->N + bs - 1) / bs), dim3(bs), 0, static_cast<hipStream_t>(stream), *args);
    return (int)hipGetLastError();
    }
} // namespace azp

extern "C" int azp_integrate_nve_rot_step_one(const azp_nve_rot_args* args, void* stream)
    {
    return azp::launch_nve_rot<true>(args, stream);
    }
extern "C" int azp_integrate_nve_rot_step_two(const azp_nve_rot_args* args, void* stream)
    {
    return azp::launch_nve_rot<false>(args, stream);
    }

// ---- particle sorter: blocked cell-curve keys in one launch ----
namespace azp
{
__global__ void __launch_bounds__(256) sorter_keys_kernel(uint32_t n, const double* __restrict__ pos, BoxDev box, double cell_width, uint32_t dimx,
                                                          uint32_t dimy, uint32_t dimz, uint32_t block, int32_t* __restrict__ keys)
    {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n)
        return;
    const double3 p = load_scalar3_of4(pos, i);
    // fractional coordinate in [0, 1): particles slightly outside the box wrap around
    double fx = p.x * box.Lxinv + 0.5, fy = p.y * box.Lyinv + 0.5, fz = p.z * box.Lzinv + 0.5;
    fx -= floor(fx); fy -= floor(fy); fz -= floor(fz);
    const uint32_t cx = min((uint32_t)(fx * dimx), dimx - 1), cy = min((uint32_t)(fy * dimy), dimy - 1), cz = min((uint32_t)(fz * dimz), dimz - 1);
    if (block == 0)
        {
        // Hilbert index (Skilling, "Programming the Hilbert curve", 2004: axes -> transpose, then the bits of the
        // three words interleaved, X[0] most significant) on a 2^b x 2^b x 2^b grid stretched over the whole box,
        // 2^b >= the largest of dims: the curve never leaves the box, so any run of consecutive cells is a compact
        // blob and 256 consecutive particles make a compact tile wherever the run starts. (On a dims grid that is
        // not a power of two the curve of the enclosing cube leaves and re-enters, and the tiles across such a gap
        // come in two distant pieces.)
        uint32_t b = 1;
        while ((1u << b) < max(dimx, max(dimy, dimz)))
            ++b;
        const uint32_t side = 1u << b;
        uint32_t X[3] = {min((uint32_t)(fx * side), side - 1), min((uint32_t)(fy * side), side - 1), min((uint32_t)(fz * side), side - 1)};
        const uint32_t M = 1u << (b - 1);
        for (uint32_t Q = M; Q > 1; Q >>= 1)
            {
            const uint32_t P = Q - 1;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                {
                if (X[k] & Q)
                    X[0] ^= P;
                else
                    {
                    const uint32_t t = (X[0] ^ X[k]) & P;
                    X[0] ^= t;
                    X[k] ^= t;
                    }
                }
            }
        X[1] ^= X[0];
        X[2] ^= X[1];
        uint32_t t = 0;
        for (uint32_t Q = M; Q > 1; Q >>= 1)
            if (X[2] & Q)
                t ^= Q - 1;
        X[0] ^= t; X[1] ^= t; X[2] ^= t;
        uint32_t h = 0;
        for (int bit = (int)b - 1; bit >= 0; --bit)
            h = (h << 3) | (((X[0] >> bit) & 1u) << 2) | (((X[1] >> bit) & 1u) << 1) | ((X[2] >> bit) & 1u);
        keys[i] = (int32_t)h;
        return;
        }
    const uint32_t nbx = (dimx + block - 1) / block, nby = (dimy + block - 1) / block;
    const uint32_t key = ((cz / block) * nby + (cy / block)) * nbx + (cx / block);
    const uint32_t inner = ((cz % block) * block + (cy % block)) * block + (cx % block);
    keys[i] = (int32_t)(key * (block * block * block) + inner);
    }
} // namespace azp

extern "C" int azp_sorter_keys(uint32_t n, const double* d_pos, const azp_box* box, const uint32_t* dims, uint32_t block, int32_t* d_keys,
                               void* stream)
    {
    if (n == 0)
        return AZP_SUCCESS;
    if (!d_pos || !box || !dims || !d_keys || dims[0] == 0 || dims[1] == 0 || dims[2] == 0)
        return AZP_ERROR_INVALID_ARGUMENT;
    uint64_t nkeys;
    if (block == 0)
        {
        uint32_t b = 1;
        while ((1u << b) < std::max(dims[0], std::max(dims[1], dims[2])))
            ++b;
        nkeys = 1ull << (3 * b); // Hilbert index in the enclosing 2^b cube
        }
    else
        nkeys = (uint64_t)((dims[0] + block - 1) / block) * ((dims[1] + block - 1) / block) * ((dims[2] + block - 1) / block) * block * block
                * block;
    if (nkeys > (1ull << 31))
        return AZP_ERROR_INVALID_ARGUMENT; // keys are int32 (torch's fast sort path)
    hipLaunchKernelGGL(azp::sorter_keys_kernel, dim3((n + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream), n, d_pos,
                       azp::make_box_dev(*box), 0.0, dims[0], dims[1], dims[2], block, d_keys);
    return (int)hipGetLastError();
    }
