// azp_device.hpp -- device-side building blocks shared by all force kernels
// (gfx950 / CDNA4, wave64). Box + minimum image, type extraction, DPP
// reductions inside a threads-per-particle group, fast FP64 reciprocal.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azp.h"

namespace azp
{
constexpr int WAVE = 64;

// ---------------------------------------------------------------------------
// Box (HOOMD BoxDim restated: centred on the origin, tilt factors xy, xz, yz)
// ---------------------------------------------------------------------------
struct BoxDev
    {
    double Lx, Ly, Lz;
    double Lxinv, Lyinv, Lzinv;
    double xy, xz, yz;
    int px, py, pz;
    int triclinic;
    };

inline BoxDev make_box_dev(const azp_box& b)
    {
    BoxDev d;
    d.Lx = b.L[0]; d.Ly = b.L[1]; d.Lz = b.L[2];
    d.Lxinv = 1.0 / b.L[0]; d.Lyinv = 1.0 / b.L[1]; d.Lzinv = 1.0 / b.L[2];
    d.xy = b.tilt[0]; d.xz = b.tilt[1]; d.yz = b.tilt[2];
    d.px = b.periodic[0]; d.py = b.periodic[1]; d.pz = b.periodic[2];
    d.triclinic = (b.tilt[0] != 0.0 || b.tilt[1] != 0.0 || b.tilt[2] != 0.0);
    return d;
    }

// Minimum image, rint form (mul + v_rndne_f64 + fma per axis: 3 FP64 ops, fewer
// than compare/select on 64-bit values). Agrees with the compare form of the
// CPU oracle away from exact ties at +-L/2.
__device__ __forceinline__ void min_image(const BoxDev& b, double& x, double& y, double& z)
    {
    if (!b.triclinic)
        {
        if (b.pz) z = __builtin_fma(-b.Lz, rint(z * b.Lzinv), z);
        if (b.py) y = __builtin_fma(-b.Ly, rint(y * b.Lyinv), y);
        if (b.px) x = __builtin_fma(-b.Lx, rint(x * b.Lxinv), x);
        }
    else
        {
        if (b.pz)
            {
            const double img = rint(z * b.Lzinv);
            z -= b.Lz * img; y -= b.Lz * b.yz * img; x -= b.Lz * b.xz * img;
            }
        if (b.py)
            {
            const double img = rint(y * b.Lyinv);
            y -= b.Ly * img; x -= b.Ly * b.xy * img;
            }
        if (b.px)
            x -= b.Lx * rint(x * b.Lxinv);
        }
    }

// true if a particle at (x,y,z) is farther than `margin` from every periodic
// face of an orthorhombic box, so that no listed neighbor can need wrapping.
__device__ __forceinline__ bool is_interior(const BoxDev& b, double x, double y, double z, double margin)
    {
    bool in = true;
    if (b.px) in = in && (fabs(x) < 0.5 * b.Lx - margin);
    if (b.py) in = in && (fabs(y) < 0.5 * b.Ly - margin);
    if (b.pz) in = in && (fabs(z) < 0.5 * b.Lz - margin);
    return in;
    }

// type index lives in the low 32 bits of pos.w
__device__ __forceinline__ int type_from_w(double w) { return __double2loint(w); }

// ---------------------------------------------------------------------------
// 16-byte loads of Scalar4 rows
// ---------------------------------------------------------------------------
__device__ __forceinline__ double4 load_scalar4(const double* base, uint32_t idx)
    {
    const double2* p = reinterpret_cast<const double2*>(base) + 2ull * idx;
    const double2 a = p[0];
    const double2 b = p[1];
    return make_double4(a.x, a.y, b.x, b.y);
    }
// x, y, z only (skips w when the caller does not need the type / mass)
__device__ __forceinline__ double3 load_scalar3_of4(const double* base, uint32_t idx)
    {
    const double2* p = reinterpret_cast<const double2*>(base) + 2ull * idx;
    const double2 a = p[0];
    const double z = base[4ull * idx + 2];
    return make_double3(a.x, a.y, z);
    }
__device__ __forceinline__ void store_scalar4(double* base, uint32_t idx, double x, double y, double z, double w)
    {
    double2* p = reinterpret_cast<double2*>(base) + 2ull * idx;
    p[0] = make_double2(x, y);
    p[1] = make_double2(z, w);
    }

// ---------------------------------------------------------------------------
// Butterfly reduction inside a group of TPP consecutive lanes (TPP <= 64,
// power of two). Steps 1/2/4/8 use DPP row operations (no LDS traffic);
// 16 and 32 cross rows and use the permute network.
// After the call every lane of the group holds the group sum.
// ---------------------------------------------------------------------------
template<int CTRL> __device__ __forceinline__ double dpp_move(double v)
    {
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
    }

template<int TPP> __device__ __forceinline__ double group_sum(double v)
    {
    if (TPP >= 2) v += dpp_move<0xB1>(v);  // quad_perm [1,0,3,2]
    if (TPP >= 4) v += dpp_move<0x4E>(v);  // quad_perm [2,3,0,1]
    if (TPP >= 8) v += dpp_move<0x141>(v); // row_half_mirror: quads 0<->1, 2<->3 (values uniform per quad)
    if (TPP >= 16) v += dpp_move<0x140>(v); // row_mirror: halves of the 16-lane row
    if (TPP >= 32) v += __shfl_xor(v, 16, WAVE);
    if (TPP >= 64) v += __shfl_xor(v, 32, WAVE);
    return v;
    }

// ---------------------------------------------------------------------------
// FP64 reciprocal: v_rcp_f64 seed (measured on gfx950: relative error 2^-24.4,
// ~14 issue cycles) + one third-order correction x (1 + e + e^2), e = 1 - a x:
// error e^3 ~ 2^-73, i.e. correct to rounding (<= 1 ulp), in 3 FMAs instead of
// the ~12-op IEEE division sequence or the 4 FMAs of two Newton steps.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double fast_rcp(double a)
    {
    const double x = __builtin_amdgcn_rcp(a);
    const double e = __builtin_fma(-a, x, 1.0);
    const double t = __builtin_fma(e, e, e);
    return __builtin_fma(x, t, x);
    }

// One Newton step only: relative error ~ (2^-24.4)^2 = 2e-15 (v_rcp_f64 is accurate
// to ~2^-24.4 on gfx950, tools/ubench.hip). One FMA cheaper than fast_rcp.
__device__ __forceinline__ double fast_rcp1(double a)
    {
    const double x = __builtin_amdgcn_rcp(a);
    const double e = __builtin_fma(-a, x, 1.0);
    return __builtin_fma(x, e, x);
    }

// 1 / sqrt(a), a > 0 and normal: v_rsq_f64 seed y = (1 + e) / sqrt(a) with |e| ~ 2^-23 and one third-order
// correction in h = 1 - a y^2: (1 - h)^(-1/2) = 1 + h / 2 + 3 h^2 / 8 + O(h^3), i.e. correct to rounding,
// in 5 FP64 operations + the quarter-rate seed -- against ~55 VALU instructions for 1.0 / sqrt(a) (the IEEE square
// root and the IEEE division are software sequences on gfx950).
__device__ __forceinline__ double fast_rsqrt(double a)
    {
    const double y = __builtin_amdgcn_rsq(a);
    const double h = __builtin_fma(-a * y, y, 1.0);
    return __builtin_fma(y * h, __builtin_fma(0.375, h, 0.5), y);
    }
// sqrt(a) for a >= 0 (a = 0 -> 0)
__device__ __forceinline__ double fast_sqrt(double a)
    {
    const double s = a * fast_rsqrt(a);
    return (a > 0.0) ? s : 0.0;
    }

// Move a wave-uniform value (computed with vector instructions, so living in
// VGPRs) into scalar registers.
template<class T> __device__ __forceinline__ T to_uniform(const T& v)
    {
    static_assert(sizeof(T) % 4 == 0, "to_uniform needs whole dwords");
    constexpr int W = sizeof(T) / 4;
    uint32_t w[W];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (int i = 0; i < W; ++i)
        w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[i]);
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
    }

// XCD-aware block remap: hardware deals blocks round-robin over the 8 XCDs, so
// blocks b and b+8 share an L2. Give each XCD one contiguous eighth of the
// (spatially sorted) particle range so neighbor gathers hit that XCD's L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblocks_padded8)
    {
    const uint32_t per_xcd = nblocks_padded8 >> 3;
    return (b & 7u) * per_xcd + (b >> 3);
    }

} // namespace azp
