// evaluators.hpp -- device restatement of the azplugins per-pair / per-bond
// arithmetic, written for the gfx950 kernels in this directory.
//
// Each isotropic pair evaluator E provides
//   E::Params                     raw per-type-pair struct (include/azp.h,
//                                 byte-compatible with the reference param_type)
//   E::Coeff                      everything that is constant per type pair,
//                                 derived ONCE per kernel (not per pair as in
//                                 the reference's evaluator constructors)
//   E::prepare(params, rcutsq, energy_shift) -> Coeff
//   E::eval(coeff, rsq, force_divr, pair_eng) -> bool evaluated
// Coeff tables live in LDS when ntypes > 1 and in registers when ntypes == 1.
#pragma once

#include "azp_device.hpp"

namespace azp
{
// ---------------------------------------------------------------------------
// PerturbedLennardJones -- src/PairEvaluatorPerturbedLennardJones.h:96-155
// Branch-free: the WCA / tail / cutoff decisions become selects so a wave
// never diverges inside the neighbor loop.
// ---------------------------------------------------------------------------
struct EvalPLJ
    {
    typedef azp_plj_params Params;
    struct Coeff
        {
        double rcutsq;    // effective cutoff^2; -1 when lj1 == 0 (":123 lj1 != 0" guard)
        double c12, c6;   // 12 lj1, 6 lj2
        double lj1, lj2;
        double lam;       // attraction_scale_factor
        double one_minus_lam;
        double wca_rsq;   // min(rwcasq, effective rcutsq): inside => WCA core AND inside the cutoff
        double wca_minus_tail; // (wca_shift - e_cut) - (-e_cut) = wca_shift
        double tail_add;  // -e_cut                 (energy offset in the scaled tail)
        double c12_lam, c6_lam; // lam c12, lam c6: the tail's force coefficients
        };
    static __device__ __forceinline__ Coeff prepare(const Params& p, double rcutsq, bool energy_shift)
        {
        Coeff c;
        const double lj1 = p.epsilon_x_4 * p.sigma_6 * p.sigma_6;
        const double lj2 = p.epsilon_x_4 * p.sigma_6;
        const double lam = p.attraction_scale_factor;
        const double wca_shift = p.epsilon_x_4 * (1.0 - lam) / 4.0;
        double e_cut = 0.0;
        if (energy_shift && rcutsq > 0.0) // r_cut = 0: the pair never interacts; keep the blended offsets finite
            {
            const double rcut2inv = 1.0 / rcutsq;
            const double rcut6inv = rcut2inv * rcut2inv * rcut2inv;
            e_cut = rcut6inv * (lj1 * rcut6inv - lj2);
            if (rcutsq < p.rwcasq)
                e_cut += wca_shift;
            else
                e_cut *= lam;
            }
        c.rcutsq = (lj1 != 0.0) ? rcutsq : -1.0;
        c.c12 = 12.0 * lj1;
        c.c6 = 6.0 * lj2;
        c.lj1 = lj1;
        c.lj2 = lj2;
        c.lam = lam;
        c.one_minus_lam = 1.0 - lam;
        c.wca_rsq = (p.rwcasq < c.rcutsq) ? p.rwcasq : c.rcutsq;
        c.tail_add = -e_cut;
        c.wca_minus_tail = (wca_shift - e_cut) - c.tail_add;
        c.c12_lam = lam * c.c12;
        c.c6_lam = lam * c.c6;
        return c;
        }
    static __device__ __forceinline__ bool eval(const Coeff& c, double rsq, double& force_divr, double& pair_eng)
        {
        // Selects are as expensive as FP64 ops on gfx950 (~4 issue cycles each, two
        // per 64-bit value), so the masks are folded into arithmetic:
        //   * outside the cutoff the HIGH word of the reciprocal is cleared (one 32-bit
        //     select): what is left is a denormal (< 1e-308) whose cube underflows to
        //     exactly 0, so r6inv, the force and the raw energy vanish exactly;
        //   * m = 1.0 inside the WCA core else 0.0 differs from 0.0 only in its high
        //     word (one 32-bit select); scale = lam + m (1 - lam), and the energy
        //     offsets are blended with m as well.
        const bool in = rsq < c.rcutsq;
        const bool wca = rsq < c.wca_rsq;
        const double x = fast_rcp(rsq);
        const double r2inv = __hiloint2double(in ? __double2hiint(x) : 0, __double2loint(x));
        const double m = __hiloint2double(wca ? 0x3ff00000 : 0, 0);
        const double r6inv = r2inv * r2inv * r2inv;
        const double f = r2inv * r6inv * __builtin_fma(c.c12, r6inv, -c.c6);
        const double e = r6inv * __builtin_fma(c.lj1, r6inv, -c.lj2);
        const double scale = __builtin_fma(m, c.one_minus_lam, c.lam);
        force_divr = f * scale;
        // energy offset: wca_add inside the core, tail_add in the tail, 0 outside
        const double tail = in ? c.tail_add : 0.0;
        pair_eng = __builtin_fma(e, scale, __builtin_fma(m, c.wca_minus_tail, tail));
        return in;
        }
    // Tile kernel, single type pair, no xplor: the two energy offsets (WCA shift in
    // the core, -e_cut inside the cutoff) are the same constants for every pair, so
    // the pairs are only COUNTED here (one add-with-carry each) and the offsets are
    // applied once per particle in finish_split -- 4 FP64-rate ops fewer per pair
    // than eval (no blended offset, no 64-bit select, one Newton step less), and the
    // reciprocals of a batch of four pairs share one v_rcp_f64 (rcp4).
    static constexpr bool kSplitEnergy = true;
    // waves per SIMD the tile kernel is compiled for (its register budget: 512 / kTileWaves VGPRs per lane)
#ifndef AZP_TILED_WAVES_PER_SIMD
    static constexpr int kTileWaves = 4;
#else
    static constexpr int kTileWaves = AZP_TILED_WAVES_PER_SIMD;
#endif
    // Reciprocals of four squared separations from ONE v_rcp_f64: R = 1 / (a b c d) (seed
    // + one Newton step, relative error ~2e-15 like fast_rcp1), then 1/(ab) = cd R,
    // 1/(cd) = ab R, 1/a = b / (ab), ...: 9 multiplies + 1 rcp + 2 fma instead of 4 rcp +
    // 8 fma. v_rcp_f64 issues at a quarter of the FP64 rate, so this is 60 instead of 96
    // issue cycles per four pairs (tools/ubench.hip). Range: the padding slot sits at 1e30
    // per axis (rsq = 3e60), so the product stays below 1e242; callers that re-image pairs
    // park the padding at rsq = 1e60. A zero separation poisons the four pairs of ITS lane
    // (the same particle's force, which is non-finite for r = 0 in any case).
    static __device__ __forceinline__ void rcp4(const double (&a)[4], double (&inv)[4])
        {
        const double ab = a[0] * a[1], cd = a[2] * a[3];
        const double R = fast_rcp1(ab * cd);
        const double iab = cd * R, icd = ab * R;
        inv[0] = a[1] * iab; inv[1] = a[0] * iab;
        inv[2] = a[3] * icd; inv[3] = a[2] * icd;
        }
    // x = 1 / rsq is supplied by the caller (rcp4 or fast_rcp1).
    static __device__ __forceinline__ void eval_split(const Coeff& c, double rsq, double x, double& force_divr, double& pe_raw,
                                                      uint32_t& n_wca, uint32_t& n_in)
        {
        const bool in = rsq < c.rcutsq;
        const bool wca = rsq < c.wca_rsq;
        const double r2inv = __hiloint2double(in ? __double2hiint(x) : 0, __double2loint(x));
        const double m = __hiloint2double(wca ? 0x3ff00000 : 0, 0);
        const double r6inv = r2inv * r2inv * r2inv;
        const double f = r2inv * r6inv * __builtin_fma(c.c12, r6inv, -c.c6);
        const double e = r6inv * __builtin_fma(c.lj1, r6inv, -c.lj2);
        const double scale = __builtin_fma(m, c.one_minus_lam, c.lam);
        force_divr = f * scale;
        pe_raw = __builtin_fma(e, scale, pe_raw);
        n_wca += wca ? 1u : 0u;
        n_in += in ? 1u : 0u;
        }
    // true when the pair needs eval_split's core/tail blend; eval_split_tail is valid
    // for every pair of a batch in which no lane saw a core pair (wave-uniform test)
    static __device__ __forceinline__ bool in_core(const Coeff& c, double rsq) { return rsq < c.wca_rsq; }
    // Tail-only form. The energy of a tail pair is lam (lj1 r6inv^2 - lj2 r6inv) with the
    // same constants for every pair, so only S2 = sum r6inv^2 and S1 = sum r6inv are
    // accumulated (2 ops instead of 3) and combined in finish_split; the pairs inside
    // the cutoff are counted only when the energy shift is non-zero (wave-uniform).
    static __device__ __forceinline__ void eval_split_tail(const Coeff& c, double rsq, double x, double& force_divr, double& s1,
                                                           double& s2, uint32_t& n_in, bool count_in)
        {
        const bool in = rsq < c.rcutsq;
        const double r2inv = __hiloint2double(in ? __double2hiint(x) : 0, __double2loint(x));
        const double r6inv = r2inv * r2inv * r2inv;
        force_divr = r2inv * r6inv * __builtin_fma(c.c12_lam, r6inv, -c.c6_lam);
        s2 = __builtin_fma(r6inv, r6inv, s2);
        s1 += r6inv;
        if (count_in)
            n_in += in ? 1u : 0u;
        }
    // A pair that is certainly inside the cutoff and outside the core (tile kernel, row phase "sure":
    // the plan's row classes and the caller's displacement bound say so): the tail form without its mask.
    static __device__ __forceinline__ void eval_split_sure(const Coeff& c, double x, double& force_divr, double& s1, double& s2)
        {
        const double r6inv = x * x * x;
        force_divr = x * r6inv * __builtin_fma(c.c12_lam, r6inv, -c.c6_lam);
        s2 = __builtin_fma(r6inv, r6inv, s2);
        s1 += r6inv;
        }
    // radius of the core (the separation below which in_core holds)
    static __device__ __forceinline__ double core_radius(const Coeff& c) { return sqrt(c.wca_rsq); }
    static __device__ __forceinline__ double finish_split(const Coeff& c, double pe_raw, double s1, double s2, uint32_t n_wca,
                                                          uint32_t n_in)
        {
        const double tail = c.lam * __builtin_fma(c.lj1, s2, -c.lj2 * s1);
        return __builtin_fma((double)n_wca, c.wca_minus_tail, __builtin_fma((double)n_in, c.tail_add, pe_raw + tail));
        }
    };

// ---------------------------------------------------------------------------
// Hertz -- src/PairEvaluatorHertz.h:93-110 (energy_shift has no effect)
// ---------------------------------------------------------------------------
struct EvalHertz
    {
    typedef azp_hertz_params Params;
    static constexpr bool kSplitEnergy = false;
#ifndef AZP_TW_HERTZ
#define AZP_TW_HERTZ 4
#endif
    static constexpr int kTileWaves = AZP_TW_HERTZ; // waves per SIMD of the tile kernel (register budget)
    struct Coeff
        {
        double rcutsq, epsilon, rcutinv, f_pref; // f_pref = 2.5 epsilon / r_cut
        };
    static __device__ __forceinline__ Coeff prepare(const Params& p, double rcutsq, bool)
        {
        Coeff c;
        c.rcutsq = (p.epsilon != 0.0) ? rcutsq : -1.0;
        c.epsilon = p.epsilon;
        c.rcutinv = 1.0 / sqrt(rcutsq);
        c.f_pref = 2.5 * p.epsilon * c.rcutinv;
        return c;
        }
    // U = eps x^(5/2), x = 1 - r / r_cut; F / r = (5 / 2) eps x^(3/2) / (r r_cut). One reciprocal square root
    // gives r and 1 / r, a second one x^(1/2): no division, no IEEE square root.
    static __device__ __forceinline__ bool eval(const Coeff& c, double rsq, double& force_divr, double& pair_eng)
        {
        force_divr = 0.0;
        pair_eng = 0.0;
        if (rsq < c.rcutsq)
            {
            const double rinv = fast_rsqrt(rsq);
            const double x = __builtin_fma(-rsq * rinv, c.rcutinv, 1.0);
            const double x3p2 = x * fast_sqrt(x);
            force_divr = c.f_pref * x3p2 * rinv;
            pair_eng = c.epsilon * x3p2 * x;
            return true;
            }
        return false;
        }
    };

// ---------------------------------------------------------------------------
// ExpandedYukawa -- src/PairEvaluatorExpandedYukawa.h:92-115
// ---------------------------------------------------------------------------
struct EvalYukawa
    {
    typedef azp_yukawa_params Params;
    static constexpr bool kSplitEnergy = false;
#ifndef AZP_TW_YUKAWA
#define AZP_TW_YUKAWA 3
#endif
    static constexpr int kTileWaves = AZP_TW_YUKAWA; // waves per SIMD of the tile kernel (register budget)
    struct Coeff
        {
        double rcutsq, epsilon, kappa, delta, e_cut;
        };
    static __device__ __forceinline__ Coeff prepare(const Params& p, double rcutsq, bool energy_shift)
        {
        Coeff c;
        c.rcutsq = (p.epsilon != 0.0) ? rcutsq : -1.0;
        c.epsilon = p.epsilon;
        c.kappa = p.kappa;
        c.delta = p.delta;
        c.e_cut = 0.0;
        if (energy_shift)
            {
            const double rcut_delta = sqrt(rcutsq) - p.delta;
            c.e_cut = p.epsilon * exp(-p.kappa * rcut_delta) / rcut_delta;
            }
        return c;
        }
    static __device__ __forceinline__ bool eval(const Coeff& c, double rsq, double& force_divr, double& pair_eng)
        {
        force_divr = 0.0;
        pair_eng = 0.0;
        if (rsq < c.rcutsq)
            {
            const double rinv = fast_rsqrt(rsq);
            const double r_delta = __builtin_fma(rsq, rinv, -c.delta);
            const double r_delta_inv = fast_rcp(r_delta);
            const double e = c.epsilon * exp(-c.kappa * r_delta) * r_delta_inv;
            force_divr = e * (c.kappa + r_delta_inv) * rinv;
            pair_eng = e - c.e_cut;
            return true;
            }
        return false;
        }
    };

// ---------------------------------------------------------------------------
// Colloid -- src/PairEvaluatorColloid.h:101-269 (the integrated Lennard-Jones potentials of Everaers and
// Ejtehadi). kind is fixed per type pair: 0 solvent-solvent, 1 colloid-solvent,
// 2 colloid-colloid (dispatch at :239-262).
// ---------------------------------------------------------------------------
struct EvalColloid
    {
    typedef azp_colloid_params Params;
    static constexpr bool kSplitEnergy = false;
#ifndef AZP_TW_COLLOID
#define AZP_TW_COLLOID 2
#endif
    static constexpr int kTileWaves = AZP_TW_COLLOID; // waves per SIMD of the tile kernel (register budget)
    // Everything that depends on the type pair alone is folded into the coefficients once per kernel; the pair
    // loop sees polynomials in Horner form and ONE reciprocal per branch (colloid-colloid: of the product of the
    // four surface-to-surface factors, shared four ways as EvalPLJ::rcp4 shares it among four pairs).
    struct Coeff
        {
        double rcutsq, e_cut;
        int kind, _pad;
        double A;
        // kind 0 (solvent-solvent): U = c1 r^-6 (s6 r^-6 - 1), F / r = r^-8 (f12 r^-6 - f6)
        double c1, s6, f12, f6;
        // kind 1 (colloid-solvent), a = the colloid's radius, m = a^2 - r^2 (< 0), t = r^2:
        //   U = pre / m^3 (2/9) (1 - s6 pe(t) / m^6),  F / r = (4/15) pre / m^4 (s6 pf(t) / m^6 - 5)
        double a2, pre, pe0, pe1, pe2, pf0, pf1, pf2;
        // kind 2 (colloid-colloid): S = a_1 + a_2, D = a_1 - a_2, k0 = a_1 a_2
        double S, D, k0, rep; // rep = A sigma^6 / 37800
        };

    static __device__ __forceinline__ double solvent_solvent(const Coeff& c, bool want_force, double& force_divr, double rsq)
        {
        const double r2inv = fast_rcp(rsq);
        const double r6inv = r2inv * r2inv * r2inv;
        if (want_force)
            force_divr = r2inv * r6inv * __builtin_fma(c.f12, r6inv, -c.f6);
        return c.c1 * r6inv * __builtin_fma(c.s6, r6inv, -1.0);
        }
    static __device__ __forceinline__ double colloid_solvent(const Coeff& c, bool want_force, double& force_divr, double rsq)
        {
        const double im = fast_rcp(c.a2 - rsq);
        const double im3 = im * im * im;
        const double w = c.s6 * im3 * im3; // sigma^6 / m^6
        const double u = c.pre * im3;      // sigma^3 A a^3 / m^3
        if (want_force)
            {
            const double pf = __builtin_fma(__builtin_fma(__builtin_fma(10.0, rsq, c.pf2), rsq, c.pf1), rsq, c.pf0);
            force_divr = (4.0 / 15.0) * u * im * __builtin_fma(pf, w, -5.0);
            }
        const double pe = __builtin_fma(__builtin_fma(__builtin_fma(1.0, rsq, c.pe2), rsq, c.pe1), rsq, c.pe0);
        return (2.0 / 9.0) * u * __builtin_fma(-pe, w, 1.0);
        }
    // One of the four repulsive terms: x = c +- r, xinv = 1 / x, s = +1 (c = S) or -1 (c = D).
    //   h = ((x + 5 c) x + 30 s k0) x^-7      (energy)
    //   g = (42 s k0 / x + 6 c + x) x^-7      (its r-derivative, up to the sign of dx / dr)
    static __device__ __forceinline__ void rep_term(double x, double xinv, double cc, double sk0, double& h, double& g)
        {
        const double x2 = xinv * xinv;
        const double x7 = x2 * x2 * x2 * xinv;
        h = __builtin_fma(x + 5.0 * cc, x, 30.0 * sk0) * x7;
        g = (__builtin_fma(42.0 * sk0, xinv, x) + 6.0 * cc) * x7;
        }
    static __device__ __forceinline__ double colloid_colloid(const Coeff& c, bool want_force, double& force_divr, double rsq)
        {
        const double rinv = fast_rsqrt(rsq);
        const double r = rsq * rinv;
        // the four factors of (S^2 - r^2) (D^2 - r^2) and their reciprocals from one v_rcp_f64
        const double p = c.S + r, q = c.S - r, u = c.D + r, w = c.D - r;
        const double pq = p * q, uw = u * w;
        const double R = fast_rcp(pq * uw);
        const double ipq = uw * R, iuw = pq * R; // 1 / (S^2 - r^2), 1 / (D^2 - r^2)
        const double ip = q * ipq, iq = p * ipq, iu = w * iuw, iw = u * iuw;
        double h0, h1, h2, h3, g0, g1, g2, g3;
        rep_term(p, ip, c.S, c.k0, h0, g0);
        rep_term(q, iq, c.S, c.k0, h1, g1);
        rep_term(u, iu, c.D, -c.k0, h2, g2);
        rep_term(w, iw, c.D, -c.k0, h3, g3);
        const double fR = c.rep * rinv;
        const double e_rep = fR * ((h0 - h1) - (h2 - h3));
        if (want_force)
            {
            const double dUR = __builtin_fma(e_rep, rinv, 5.0 * fR * ((g0 + g1) - (g2 + g3)));
            const double ta = __builtin_fma(2.0 * c.k0, ipq, 1.0) * ipq + __builtin_fma(2.0 * c.k0, iuw, -1.0) * iuw;
            // -dU/dr / r, attractive part: -(A / 3) r (...) / r
            force_divr = __builtin_fma(dUR, rinv, -(c.A / 3.0) * ta);
            }
        // ln((D^2 - r^2)^-1 / (S^2 - r^2)^-1) = ln((S^2 - r^2) / (D^2 - r^2)): the ratio is there already
        return e_rep + (c.A / 6.0) * (2.0 * c.k0 * (ipq + iuw) - log(pq * iuw));
        }
    static __device__ __forceinline__ double branch(const Coeff& c, bool want_force, double& force_divr, double rsq)
        {
        if (c.kind == 0)
            return solvent_solvent(c, want_force, force_divr, rsq);
        if (c.kind == 2)
            return colloid_colloid(c, want_force, force_divr, rsq);
        return colloid_solvent(c, want_force, force_divr, rsq);
        }

    static __device__ __forceinline__ Coeff prepare(const Params& p, double rcutsq, bool energy_shift)
        {
        Coeff c;
        const double s3 = p.sigma_3, s6 = s3 * s3;
        c.rcutsq = (p.A != 0.0) ? rcutsq : -1.0;
        c.A = p.A;
        c.kind = (p.a_1 == 0.0 && p.a_2 == 0.0) ? 0 : ((p.a_1 != 0.0 && p.a_2 != 0.0) ? 2 : 1);
        c._pad = 0;
        c.s6 = s6;
        c.c1 = p.A * s6 / 36.0;
        c.f12 = 12.0 * c.c1 * s6;
        c.f6 = 6.0 * c.c1;
        const double a = (p.a_1 > p.a_2) ? p.a_1 : p.a_2, a2 = a * a, a4 = a2 * a2;
        c.a2 = a2;
        c.pre = s3 * p.A * a * a2;
        // energy polynomial (t^3 + 4.2 a^2 t^2 + 3 a^4 t + a^6 / 3), force polynomial 2 (a^2 + t) (5 a^4 + 22 a^2 t + 5 t^2)
        c.pe2 = 4.2 * a2; c.pe1 = 3.0 * a4; c.pe0 = a4 * a2 / 3.0;
        c.pf2 = 54.0 * a2; c.pf1 = 54.0 * a4; c.pf0 = 10.0 * a4 * a2;
        c.S = p.a_1 + p.a_2;
        c.D = p.a_1 - p.a_2;
        c.k0 = p.a_1 * p.a_2;
        c.rep = p.A * s6 / 37800.0;
        c.e_cut = 0.0;
        if (energy_shift && p.A != 0.0)
            {
            double dummy;
            c.e_cut = branch(c, false, dummy, rcutsq);
            }
        return c;
        }
    static __device__ __forceinline__ bool eval(const Coeff& c, double rsq, double& force_divr, double& pair_eng)
        {
        force_divr = 0.0;
        pair_eng = 0.0;
        if (rsq < c.rcutsq)
            {
            pair_eng = branch(c, true, force_divr, rsq) - c.e_cut;
            return true;
            }
        return false;
        }
    };

// ---------------------------------------------------------------------------
// DPD conservative part -- src/DPDPairEvaluatorGeneralWeight.h:165-183
// (no A != 0 guard; energy_shift ignored). Also the base of the thermostat.
// ---------------------------------------------------------------------------
struct EvalDPDConservative
    {
    typedef azp_dpd_params Params;
    static constexpr bool kSplitEnergy = false;
#ifndef AZP_TW_DPDC
#define AZP_TW_DPDC 4
#endif
    static constexpr int kTileWaves = AZP_TW_DPDC; // waves per SIMD of the tile kernel (register budget)
    struct Coeff
        {
        double rcutsq, A, gamma, half_s, rcut, rcutinv;
        };
    static __device__ __forceinline__ Coeff prepare(const Params& p, double rcutsq, bool)
        {
        Coeff c;
        c.rcutsq = rcutsq;
        c.A = p.A;
        c.gamma = p.gamma;
        c.half_s = 0.5 * p.s;
        c.rcutinv = 1.0 / sqrt(rcutsq);
        c.rcut = 1.0 / c.rcutinv;
        return c;
        }
    static __device__ __forceinline__ bool eval(const Coeff& c, double rsq, double& force_divr, double& pair_eng)
        {
        force_divr = 0.0;
        pair_eng = 0.0;
        if (rsq < c.rcutsq)
            {
            const double rinv = fast_rsqrt(rsq);
            const double r = rsq * rinv;
            force_divr = c.A * (rinv - c.rcutinv);
            pair_eng = c.A * (c.rcut - r) - 0.5 * c.A * c.rcutinv * (c.rcutsq - rsq);
            return true;
            }
        return false;
        }
    };

// ---------------------------------------------------------------------------
// XPLOR smoothing applied around any isotropic evaluator (HOOMD PotentialPair
// restated; not pinned by a reference test).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void apply_xplor(double rsq, double ronsq, double rcutsq, double& force_divr,
                                            double& pair_eng)
    {
    if (rsq >= ronsq && rsq < rcutsq)
        {
        const double old_pair_eng = pair_eng;
        const double old_force_divr = force_divr;
        const double d = rcutsq - ronsq;
        const double xplor_denom_inv = 1.0 / (d * d * d);
        const double rsq_minus_r_cut_sq = rsq - rcutsq;
        const double s = rsq_minus_r_cut_sq * rsq_minus_r_cut_sq * (rcutsq + 2.0 * rsq - 3.0 * ronsq) * xplor_denom_inv;
        const double ds_dr_divr = 12.0 * (rsq - ronsq) * rsq_minus_r_cut_sq * xplor_denom_inv;
        pair_eng = old_pair_eng * s;
        force_divr = s * old_force_divr - ds_dr_divr * old_pair_eng;
        }
    }

// ---------------------------------------------------------------------------
// Philox4x32-10 (Random123 algorithm) and the HOOMD-style stream the DPD
// thermostat draws from: key = {id<<24 | ts[39:32]<<16 | seed, ts[31:0]},
// counter = {0, min tag, max tag, 0}; alpha = uniform(-1, 1].
// Call site restated: src/DPDPairEvaluatorGeneralWeight.h:213-233.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                              uint32_t k1)
    {
#pragma unroll
    for (int round = 0; round < 10; ++round)
        {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a lo and a hi multiply
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
        }
    }

__device__ __forceinline__ double dpd_alpha(uint16_t seed, uint32_t tag_i, uint32_t tag_j, uint64_t timestep)
    {
    const uint32_t oi = tag_i > tag_j ? tag_j : tag_i;
    const uint32_t oj = tag_i > tag_j ? tag_i : tag_j;
    const uint64_t ts = (uint64_t)(uint32_t)timestep; // reference passes unsigned int (:130-137)
    const uint32_t k0 = (200u << 24) | ((uint32_t)((ts >> 32) & 0xffu) << 16) | (uint32_t)seed;
    const uint32_t k1 = (uint32_t)(ts & 0xffffffffu);
    uint32_t c0 = 0, c1 = oi, c2 = oj, c3 = 0;
    philox4x32_10(c0, c1, c2, c3, k0, k1);
    const uint64_t u = ((uint64_t)c0 << 32) | (uint64_t)c1;
    const double u01 = (double)(u >> 11) * (1.0 / 9007199254740992.0) + (0.5 / 9007199254740992.0);
    return -1.0 + 2.0 * u01;
    }

// ---------------------------------------------------------------------------
// Bond evaluators
// ---------------------------------------------------------------------------
struct EvalDoubleWell // src/BondEvaluatorDoubleWell.h:96-113
    {
    typedef azp_dw_params Params;
    static constexpr bool kSplitEnergy = false;
    static __device__ __forceinline__ bool eval(const Params& p, double rsq, double& force_divr, double& bond_eng)
        {
        bond_eng = 0.0;
        force_divr = 0.0;
        if (p.r_diff == 0.0)
            return false;
        const double r = sqrt(rsq);
        const double x = (p.r_1 - r) / p.r_diff;
        const double x2 = x * x;
        const double y = 1.0 - x2;
        const double y2 = y * y;
        bond_eng = p.U_1 * y2 + p.U_tilt * (1.0 - x - y2);
        force_divr = (4.0 * x * y * (p.U_tilt - p.U_1) - p.U_tilt) / (p.r_diff * r);
        return true;
        }
    };

struct EvalQuartic // src/BondEvaluatorQuartic.h:113-200
    {
    typedef azp_quartic_params Params;
    static constexpr bool kSplitEnergy = false;
    static __device__ __forceinline__ bool eval(const Params& p, double rsq, double& force_divr, double& bond_eng)
        {
        const double lj1 = p.epsilon_x_4 * p.sigma_6 * p.sigma_6;
        const double lj2 = p.epsilon_x_4 * p.sigma_6;
        const double k = p.k, r_0 = p.r_0, b_1 = p.b_1, b_2 = p.b_2, U_0 = p.U_0, delta = p.delta;
        double f = 0.0, e = 0.0;
        bond_eng = 0.0;
        force_divr = 0.0;
        if (r_0 == 0.0)
            return false;
        double r_red = 1.0;
        if (delta == 0.0)
            {
            const double r2inv = 1.0 / rsq;
            const double r6inv = r2inv * r2inv * r2inv;
            const double sigma6inv = lj2 / lj1;
            if (lj1 != 0.0 && r6inv > sigma6inv / 2.0)
                {
                const double epsilon = lj2 * lj2 / 4.0 / lj1;
                f += r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
                e += r6inv * (lj1 * r6inv - lj2) + epsilon;
                }
            if (rsq < r_0 * r_0)
                r_red = sqrt(rsq) - r_0;
            }
        else
            {
            const double r = sqrt(rsq) - delta;
            const double r2inv = 1.0 / r / r;
            const double r6inv = r2inv * r2inv * r2inv;
            const double sigma6inv = lj2 / lj1;
            if (lj1 != 0.0 && r6inv > sigma6inv / 2.0)
                {
                f += r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2) / r / (r + delta);
                e += r6inv * (lj1 * r6inv - lj2) + p.epsilon_x_4 / 4.0;
                }
            if (r < r_0)
                r_red = r - r_0;
            }
        if (r_red < 0.0)
            {
            f += -1.0 * k * r_red * (4.0 * r_red * r_red - 3.0 * (b_1 + b_2) * r_red + 2.0 * b_1 * b_2)
                 / (r_red + r_0 + delta);
            e += k * (r_red - b_1) * (r_red - b_2) * r_red * r_red + U_0;
            }
        else
            e += U_0;
        force_divr = f;
        bond_eng = e;
        return true;
        }
    };

} // namespace azp
