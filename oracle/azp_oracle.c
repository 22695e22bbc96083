/*
 * azp_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the per-timestep force-compute hot path of
 * stattlab/azplugins v1.1.0 (HOOMD-blue 5.0.x plugin): the eight per-pair
 * "evaluator" functors the reference owns, plus the HOOMD-side outer loops
 * that call them (neighbor iteration, minimum image, shift/xplor modes,
 * half-energy split, virial, third-law scatter, bond-table walk, DPD setters,
 * aniso torque accumulation).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library. The product (azplugins_amd/, include/azp.h) never does.
 *
 * Pinning status
 * --------------
 *  - Evaluator arithmetic: PINNED against all 44 known-answer cases held by the
 *    reference's own tests (src/pytest/test_pair.py:22-306,
 *    test_pair_aniso.py:22-110, test_bond.py:20-193), transcribed as data in
 *    tests/golden/reference_cases.json and checked by tests/test_oracle_golden.py.
 *  - The reference's evaluator headers cannot be compiled here: they include
 *    hoomd/HOOMDMath.h, hoomd/VectorMath.h and hoomd/RandomNumbers.h from
 *    HOOMD-blue v5.0.1, which is absent from this image (no oracle/_ref).
 *  - PARITY UNPINNED (no reference test, no source in /root/reference): the
 *    HOOMD outer-loop conventions restated below beyond what the 2-particle
 *    tests show -- virial layout/values, xplor smoothing, type-pair indexing
 *    for T>1, periodic minimum image, and the DPD random stream (Seed/Counter
 *    packing and the uint->real mapping). For those this file is the
 *    definition; Philox4x32-10 itself is checked against Random123's
 *    published known-answer vectors.
 *
 * Third-party dependency restated: HOOMD-blue v5.0.1 (pinned at
 * /root/reference/.github/workflows/unit-test.yaml:11; >=5.0.0 at
 * /root/reference/CMakeLists.txt:8), classes PotentialPair, PotentialPairDPDThermo,
 * AnisoPotentialPair, PotentialBond, BoxDim, RandomGenerator (Random123 Philox4x32-10).
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; no -ffast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef double Scalar;

/* ------------------------------------------------------------------------- */
/* Parameter structs: field order follows the reference param structs.       */
/* ------------------------------------------------------------------------- */

/* src/PairEvaluatorPerturbedLennardJones.h:57-60 */
typedef struct { Scalar sigma_6, epsilon_x_4, attraction_scale_factor, rwcasq; } azo_plj_t;
/* src/PairEvaluatorHertz.h:41 */
typedef struct { Scalar epsilon; } azo_hertz_t;
/* src/PairEvaluatorExpandedYukawa.h:44-46 (aligned(32) => 32 bytes) */
typedef struct { Scalar epsilon, kappa, delta, _pad; } azo_yukawa_t;
/* src/PairEvaluatorColloid.h:48-51 */
typedef struct { Scalar A, a_1, a_2, sigma_3; } azo_colloid_t;
/* src/DPDPairEvaluatorGeneralWeight.h:53-55 (aligned(32) => 32 bytes) */
typedef struct { Scalar A, gamma, s, _pad; } azo_dpd_t;
/* src/AnisoPairEvaluatorTwoPatchMorse.h:63-68 (5 Scalars + bool, 48 bytes) */
typedef struct { Scalar M_d, M_rinv, r_eq, omega, alpha; uint8_t repulsion; uint8_t _pad[7]; } azo_tpm_t;
/* src/BondEvaluatorDoubleWell.h:52-55 */
typedef struct { Scalar r_1, r_diff, U_1, U_tilt; } azo_dw_t;
/* src/BondEvaluatorQuartic.h:68-75 */
typedef struct { Scalar k, r_0, b_1, b_2, U_0, sigma_6, epsilon_x_4, delta; } azo_quartic_t;

/* Host-side parameter construction (what the reference does in the
 * pybind11::dict constructors of each param struct). */

/* src/PairEvaluatorPerturbedLennardJones.h:33-45 */
void azo_make_plj(Scalar epsilon, Scalar sigma, Scalar lambda, azo_plj_t* p)
    {
    const Scalar sigma_2 = sigma * sigma;
    const Scalar sigma_4 = sigma_2 * sigma_2;
    p->sigma_6 = sigma_2 * sigma_4;
    p->epsilon_x_4 = 4.0 * epsilon;
    p->attraction_scale_factor = lambda;
    p->rwcasq = pow(2.0, 1. / 3.) * sigma_2;
    }
/* src/PairEvaluatorColloid.h:28-35 */
void azo_make_colloid(Scalar A, Scalar a_1, Scalar a_2, Scalar sigma, azo_colloid_t* p)
    {
    p->A = A; p->a_1 = a_1; p->a_2 = a_2; p->sigma_3 = sigma * sigma * sigma;
    }
/* src/AnisoPairEvaluatorTwoPatchMorse.h:40-48 */
void azo_make_tpm(Scalar M_d, Scalar M_r, Scalar r_eq, Scalar omega, Scalar alpha, int repulsion, azo_tpm_t* p)
    {
    memset(p, 0, sizeof(*p));
    p->M_d = M_d; p->M_rinv = 1.0 / M_r; p->r_eq = r_eq; p->omega = omega; p->alpha = alpha;
    p->repulsion = repulsion ? 1 : 0;
    }
/* src/BondEvaluatorDoubleWell.h:33-39 */
void azo_make_dw(Scalar r_0, Scalar r_1, Scalar U_1, Scalar U_tilt, azo_dw_t* p)
    {
    p->r_1 = r_1; p->r_diff = r_1 - r_0; p->U_1 = U_1; p->U_tilt = U_tilt;
    }
/* src/BondEvaluatorQuartic.h:36-52 */
void azo_make_quartic(Scalar k, Scalar r_0, Scalar b_1, Scalar b_2, Scalar U_0, Scalar sigma,
                      Scalar epsilon, Scalar delta, azo_quartic_t* p)
    {
    p->k = k; p->r_0 = r_0; p->b_1 = b_1; p->b_2 = b_2; p->U_0 = U_0; p->delta = delta;
    const Scalar sigma_2 = sigma * sigma;
    const Scalar sigma_4 = sigma_2 * sigma_2;
    p->sigma_6 = sigma_2 * sigma_4;
    p->epsilon_x_4 = 4.0 * epsilon;
    }

/* ------------------------------------------------------------------------- */
/* Isotropic pair evaluators. Signature mirrors                              */
/*   Evaluator(rsq, rcutsq, params).evalForceAndEnergy(force_divr, pair_eng, */
/*   energy_shift) -> bool           (src/PairEvaluator.h:72,98)             */
/* ------------------------------------------------------------------------- */
typedef int (*azo_pair_eval_fn)(const void* params, Scalar rsq, Scalar rcutsq, int energy_shift,
                                Scalar* force_divr, Scalar* pair_eng);

/* src/PairEvaluatorPerturbedLennardJones.h:96-104 (ctor), 117-155 */
int azo_eval_plj(const void* vp, Scalar rsq, Scalar rcutsq, int energy_shift, Scalar* force_divr, Scalar* pair_eng)
    {
    const azo_plj_t* p = (const azo_plj_t*)vp;
    const Scalar lj1 = p->epsilon_x_4 * p->sigma_6 * p->sigma_6;
    const Scalar lj2 = p->epsilon_x_4 * p->sigma_6;
    const Scalar lam = p->attraction_scale_factor;
    const Scalar rwcasq = p->rwcasq;
    const Scalar wca_shift = p->epsilon_x_4 * (1.0 - lam) / 4.0;
    if (rsq < rcutsq && lj1 != 0)
        {
        const Scalar r2inv = 1.0 / rsq;
        const Scalar r6inv = r2inv * r2inv * r2inv;
        Scalar f = r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
        Scalar e = r6inv * (lj1 * r6inv - lj2);
        if (rsq < rwcasq)
            e += wca_shift;
        else
            {
            f *= lam;
            e *= lam;
            }
        if (energy_shift)
            {
            const Scalar rcut2inv = 1.0 / rcutsq;
            const Scalar rcut6inv = rcut2inv * rcut2inv * rcut2inv;
            Scalar es = rcut6inv * (lj1 * rcut6inv - lj2);
            if (rcutsq < rwcasq)
                es += wca_shift;
            else
                es *= lam;
            e -= es;
            }
        *force_divr = f;
        *pair_eng = e;
        return 1;
        }
    return 0;
    }

/* src/PairEvaluatorHertz.h:93-110 (energy_shift ignored: U(rcut)=0) */
int azo_eval_hertz(const void* vp, Scalar rsq, Scalar rcutsq, int energy_shift, Scalar* force_divr, Scalar* pair_eng)
    {
    const azo_hertz_t* p = (const azo_hertz_t*)vp;
    (void)energy_shift;
    if (rsq < rcutsq && p->epsilon != 0.0)
        {
        const Scalar r = sqrt(rsq);
        const Scalar rcut = sqrt(rcutsq);
        const Scalar x = 1.0 - (r / rcut);
        const Scalar xsqrt = sqrt(x);
        const Scalar ex3p2 = p->epsilon * x * xsqrt;
        *force_divr = 2.5 * ex3p2 / (r * rcut);
        *pair_eng = ex3p2 * x;
        return 1;
        }
    return 0;
    }

/* src/PairEvaluatorExpandedYukawa.h:92-115 */
int azo_eval_yukawa(const void* vp, Scalar rsq, Scalar rcutsq, int energy_shift, Scalar* force_divr, Scalar* pair_eng)
    {
    const azo_yukawa_t* p = (const azo_yukawa_t*)vp;
    if (rsq < rcutsq && p->epsilon != 0.0)
        {
        const Scalar r = sqrt(rsq);
        const Scalar r_delta = r - p->delta;
        const Scalar r_delta_inv = 1.0 / r_delta;
        Scalar e = p->epsilon * exp(-p->kappa * r_delta) * r_delta_inv;
        *force_divr = e * (p->kappa + r_delta_inv) / r;
        if (energy_shift)
            {
            const Scalar rcut = sqrt(rcutsq);
            const Scalar rcut_delta = rcut - p->delta;
            e -= p->epsilon * exp(-p->kappa * rcut_delta) / rcut_delta;
            }
        *pair_eng = e;
        return 1;
        }
    return 0;
    }

/* src/PairEvaluatorColloid.h:101-113 */
static Scalar colloid_ss(const azo_colloid_t* p, int force, Scalar* force_divr, Scalar rsq)
    {
    const Scalar sigma_6 = p->sigma_3 * p->sigma_3;
    const Scalar r2inv = 1.0 / rsq;
    const Scalar r6inv = r2inv * r2inv * r2inv;
    const Scalar c1 = p->A * sigma_6 / 36.0;
    if (force)
        *force_divr = 6.0 * c1 * r2inv * r6inv * (2.0 * sigma_6 * r6inv - 1.0);
    return c1 * r6inv * (sigma_6 * r6inv - 1.0);
    }
/* src/PairEvaluatorColloid.h:125-152 */
static Scalar colloid_cs(const azo_colloid_t* p, int force, Scalar* force_divr, Scalar rsq)
    {
    const Scalar sigma_3 = p->sigma_3;
    const Scalar sigma_6 = sigma_3 * sigma_3;
    const Scalar a = (p->a_1 > p->a_2) ? p->a_1 : p->a_2;
    const Scalar asq = a * a;
    const Scalar asq_minus_rsq = asq - rsq;
    const Scalar rsqsq = rsq * rsq;
    const Scalar amr3 = asq_minus_rsq * asq_minus_rsq * asq_minus_rsq;
    const Scalar amr6 = amr3 * amr3;
    const Scalar fR = sigma_3 * p->A * a * asq / amr3;
    if (force)
        {
        *force_divr = (4.0 / 15.0) * fR
                      * (2.0 * (asq + rsq) * (asq * (5.0 * asq + 22.0 * rsq) + 5.0 * rsqsq) * sigma_6 / amr6 - 5.0)
                      / asq_minus_rsq;
        }
    return (2.0 / 9.0) * fR
           * (1.0 - (asq * (asq * (asq / 3.0 + 3.0 * rsq) + 4.2 * rsqsq) + rsq * rsqsq) * sigma_6 / amr6);
    }
/* src/PairEvaluatorColloid.h:164-220 */
static Scalar colloid_cc(const azo_colloid_t* p, int force, Scalar* force_divr, Scalar rsq)
    {
    const Scalar A = p->A, ai = p->a_1, aj = p->a_2;
    const Scalar sigma_6 = p->sigma_3 * p->sigma_3;
    const Scalar r = sqrt(rsq);
    const Scalar k0 = ai * aj;
    const Scalar k1 = ai + aj;
    const Scalar k2 = ai - aj;
    const Scalar k3 = k1 + r;
    const Scalar k4 = k1 - r;
    const Scalar k5 = k2 + r;
    const Scalar k6 = k2 - r;
    const Scalar k7 = 1.0 / (k3 * k4);
    const Scalar k8 = 1.0 / (k5 * k6);

    const Scalar k3inv = 1.0 / k3;
    Scalar g0 = k3inv * k3inv; g0 *= g0 * g0; g0 *= k3inv;
    const Scalar k4inv = 1.0 / k4;
    Scalar g1 = k4inv * k4inv; g1 *= g1 * g1; g1 *= k4inv;
    const Scalar k5inv = 1.0 / k5;
    Scalar g2 = k5inv * k5inv; g2 *= g2 * g2; g2 *= k5inv;
    const Scalar k6inv = 1.0 / k6;
    Scalar g3 = k6inv * k6inv; g3 *= g3 * g3; g3 *= k6inv;

    const Scalar h0 = ((k3 + 5.0 * k1) * k3 + 30.0 * k0) * g0;
    const Scalar h1 = ((k4 + 5.0 * k1) * k4 + 30.0 * k0) * g1;
    const Scalar h2 = ((k5 + 5.0 * k2) * k5 - 30.0 * k0) * g2;
    const Scalar h3 = ((k6 + 5.0 * k2) * k6 - 30.0 * k0) * g3;

    g0 *= 42.0 * k0 / k3 + 6.0 * k1 + k3;
    g1 *= 42.0 * k0 / k4 + 6.0 * k1 + k4;
    g2 *= -42.0 * k0 / k5 + 6.0 * k2 + k5;
    g3 *= -42.0 * k0 / k6 + 6.0 * k2 + k6;

    const Scalar fR = A * sigma_6 / r / 37800.0;
    Scalar pair_eng = fR * (h0 - h1 - h2 + h3);
    if (force)
        {
        const Scalar dUR = pair_eng / r + 5.0 * fR * (g0 + g1 - g2 - g3);
        const Scalar dUA = -A / 3.0 * r * ((2.0 * k0 * k7 + 1.0) * k7 + (2.0 * k0 * k8 - 1.0) * k8);
        *force_divr = (dUR + dUA) / r;
        }
    pair_eng += A / 6.0 * (2.0 * k0 * (k7 + k8) - log(k8 / k7));
    return pair_eng;
    }
/* src/PairEvaluatorColloid.h:233-269 */
int azo_eval_colloid(const void* vp, Scalar rsq, Scalar rcutsq, int energy_shift, Scalar* force_divr, Scalar* pair_eng)
    {
    const azo_colloid_t* p = (const azo_colloid_t*)vp;
    if (rsq < rcutsq && p->A != 0.0)
        {
        Scalar dummy = 0;
        if (p->a_1 == 0.0 && p->a_2 == 0.0)
            {
            *pair_eng = colloid_ss(p, 1, force_divr, rsq);
            if (energy_shift) *pair_eng -= colloid_ss(p, 0, &dummy, rcutsq);
            }
        else if (p->a_1 != 0.0 && p->a_2 != 0.0)
            {
            *pair_eng = colloid_cc(p, 1, force_divr, rsq);
            if (energy_shift) *pair_eng -= colloid_cc(p, 0, &dummy, rcutsq);
            }
        else
            {
            *pair_eng = colloid_cs(p, 1, force_divr, rsq);
            if (energy_shift) *pair_eng -= colloid_cs(p, 0, &dummy, rcutsq);
            }
        return 1;
        }
    return 0;
    }

/* src/DPDPairEvaluatorGeneralWeight.h:165-183 (conservative only; no A != 0 guard;
 * energy_shift ignored) */
int azo_eval_dpd_cons(const void* vp, Scalar rsq, Scalar rcutsq, int energy_shift, Scalar* force_divr, Scalar* pair_eng)
    {
    const azo_dpd_t* p = (const azo_dpd_t*)vp;
    (void)energy_shift;
    if (rsq < rcutsq)
        {
        const Scalar rinv = 1.0 / sqrt(rsq);
        const Scalar r = 1.0 / rinv;
        const Scalar rcutinv = 1.0 / sqrt(rcutsq);
        const Scalar rcut = 1.0 / rcutinv;
        *force_divr = p->A * (rinv - rcutinv);
        *pair_eng = p->A * (rcut - r) - 0.5 * p->A * rcutinv * (rcutsq - rsq);
        return 1;
        }
    return 0;
    }

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Random123; bundled inside HOOMD-blue as hoomd/extern).     */
/* Published algorithm: Salmon et al., SC'11. Constants from Random123       */
/* philox.h: M0=0xD2511F53, M1=0xCD9E8D57, W0=0x9E3779B9, W1=0xBB67AE85.     */
/* ------------------------------------------------------------------------- */
void azo_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4])
    {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int round = 0; round < 10; ++round)
        {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n1 = lo1;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        const uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
        }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }

/* HOOMD RandomGenerator(Seed(id, timestep, seed), Counter(a, b)) followed by
 * UniformDistribution<double>(-1, 1): restated from recollection of
 * hoomd/RandomNumbers.h (v5.0.x) -- PARITY UNPINNED (header absent).
 *   key  = { id<<24 | ((timestep>>32)&0xff)<<16 | seed ,  (uint32)timestep }
 *   ctr  = { d<<16 (d=0) , a , b , c (=0) }
 *   u64  = (uint64)out[0] << 32 | out[1]
 *   u01  = (u64 >> 11) * 2^-53 + 2^-54           (Random123 u01: (0,1])
 *   x    = a + (b-a) * u01
 * Call site: src/DPDPairEvaluatorGeneralWeight.h:213-233 (id = 200,
 * src/RNGIdentifiers.h:23; counter = (min tag, max tag)). */
Scalar azo_dpd_alpha(uint16_t seed, uint32_t tag_i, uint32_t tag_j, uint64_t timestep)
    {
    const uint32_t oi = tag_i > tag_j ? tag_j : tag_i;
    const uint32_t oj = tag_i > tag_j ? tag_i : tag_j;
    /* reference truncates the timestep to unsigned int in set_seed_ij_timestep
     * (src/DPDPairEvaluatorGeneralWeight.h:130-137) before Seed() widens it again */
    const uint64_t ts = (uint64_t)(uint32_t)timestep;
    uint32_t key[2], ctr[4], out[4];
    key[0] = ((uint32_t)200u << 24) | ((uint32_t)((ts >> 32) & 0xffu) << 16) | (uint32_t)seed;
    key[1] = (uint32_t)(ts & 0xffffffffu);
    ctr[0] = 0; ctr[1] = oi; ctr[2] = oj; ctr[3] = 0;
    azo_philox4x32_10(ctr, key, out);
    const uint64_t u = ((uint64_t)out[0] << 32) | (uint64_t)out[1];
    const Scalar u01 = (Scalar)(u >> 11) * (1.0 / 9007199254740992.0) + (0.5 / 9007199254740992.0);
    return -1.0 + 2.0 * u01;
    }

/* src/DPDPairEvaluatorGeneralWeight.h:198-255. alpha is passed in so the
 * deterministic terms can be checked in closed form. */
int azo_eval_dpd_thermo(const azo_dpd_t* p, Scalar rsq, Scalar rcutsq, Scalar rdotv, Scalar deltaT, Scalar T,
                        Scalar alpha, Scalar* force_divr, Scalar* force_divr_cons, Scalar* pair_eng)
    {
    if (rsq < rcutsq)
        {
        const Scalar rinv = 1.0 / sqrt(rsq);
        const Scalar r = 1.0 / rinv;
        const Scalar rcutinv = 1.0 / sqrt(rcutsq);
        const Scalar rcut = 1.0 / rcutinv;
        Scalar f = p->A * (rinv - rcutinv);
        *force_divr_cons = f;
        const Scalar wR = pow(1.0 - r * rcutinv, 0.5 * p->s) * rinv;
        f -= p->gamma * wR * wR * rdotv;
        /* fast::rsqrt(dt / (T*gamma*6)); T == 0 => rsqrt(inf) = 0 => no noise
         * (relied on by src/pytest/test_pair.py:331-332) */
        f += (1.0 / sqrt(deltaT / (T * p->gamma * 6.0))) * wR * alpha;
        *force_divr = f;
        *pair_eng = p->A * (rcut - r) - 0.5 * p->A * rcutinv * (rcutsq - rsq);
        return 1;
        }
    return 0;
    }

/* ------------------------------------------------------------------------- */
/* Anisotropic: TwoPatchMorse.  src/AnisoPairEvaluatorTwoPatchMorse.h:127-216 */
/* quaternion stored Scalar4 (x = scalar part, y,z,w = vector part).          */
/* ------------------------------------------------------------------------- */
static void quat_rotate_ex(const Scalar q[4], Scalar n[3])
    {
    /* rotate(q, v) with v = (1,0,0): (s^2 - |u|^2) v + 2 s (u x v) + 2 (u.v) u */
    const Scalar s = q[0], ux = q[1], uy = q[2], uz = q[3];
    const Scalar c = s * s - (ux * ux + uy * uy + uz * uz);
    /* u x (1,0,0) = (0, uz, -uy) */
    n[0] = c + 2.0 * ux * ux;
    n[1] = 2.0 * s * uz + 2.0 * ux * uy;
    n[2] = -2.0 * s * uy + 2.0 * ux * uz;
    }
static void cross3(const Scalar a[3], const Scalar b[3], Scalar c[3])
    {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
    }
int azo_eval_tpm(const azo_tpm_t* p, const Scalar dr[3], const Scalar qi[4], const Scalar qj[4], Scalar rcutsq,
                 int energy_shift, Scalar force[3], Scalar* pair_eng, Scalar torque_i[3], Scalar torque_j[3])
    {
    const Scalar rsq = dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2];
    if (rsq > rcutsq) /* strict '>' as in the reference (:135) */
        return 0;
    const Scalar rinv = 1.0 / sqrt(rsq);
    const Scalar r = 1.0 / rinv;
    const Scalar unitr[3] = {dr[0] * rinv, dr[1] * rinv, dr[2] * rinv};
    Scalar n_i[3], n_j[3];
    quat_rotate_ex(qi, n_i);
    quat_rotate_ex(qj, n_j);

    Scalar UMorse = -1.0 * p->M_d;
    Scalar dUMorse_dr = 0.0;
    if (r > p->r_eq || p->repulsion)
        {
        const Scalar Morse_exp = exp(-(r - p->r_eq) * p->M_rinv);
        const Scalar one_minus_exp = 1.0 - Morse_exp;
        UMorse = p->M_d * (one_minus_exp * one_minus_exp - 1.0);
        dUMorse_dr = 2.0 * p->M_d * p->M_rinv * Morse_exp * one_minus_exp;
        }
    const Scalar gamma_i = unitr[0] * n_i[0] + unitr[1] * n_i[1] + unitr[2] * n_i[2];
    const Scalar gamma_i_exp = exp(-p->omega * (gamma_i * gamma_i - p->alpha));
    const Scalar Omega_i = 1.0 / (1.0 + gamma_i_exp);
    const Scalar gamma_j = unitr[0] * n_j[0] + unitr[1] * n_j[1] + unitr[2] * n_j[2];
    const Scalar gamma_j_exp = exp(-p->omega * (gamma_j * gamma_j - p->alpha));
    const Scalar Omega_j = 1.0 / (1.0 + gamma_j_exp);

    Scalar e = UMorse * Omega_i * Omega_j;

    const Scalar dU_dr = dUMorse_dr * Omega_i * Omega_j;
    const Scalar dOmegai_dgi = 2.0 * p->omega * gamma_i * gamma_i_exp * Omega_i * Omega_i;
    const Scalar dOmegaj_dgj = 2.0 * p->omega * gamma_j * gamma_j_exp * Omega_j * Omega_j;
    const Scalar dU_dgi = dOmegai_dgi * UMorse * Omega_j;
    const Scalar dU_dgj = dOmegaj_dgj * UMorse * Omega_i;

    /* n_perp = cross(-unitr, cross(unitr, n)) */
    const Scalar munitr[3] = {-unitr[0], -unitr[1], -unitr[2]};
    Scalar rxni[3], rxnj[3], n_i_perp[3], n_j_perp[3];
    cross3(unitr, n_i, rxni);
    cross3(unitr, n_j, rxnj);
    cross3(munitr, rxni, n_i_perp);
    cross3(munitr, rxnj, n_j_perp);

    for (int k = 0; k < 3; ++k)
        {
        force[k] = -dU_dr * unitr[k] - rinv * (dU_dgi * n_i_perp[k] + dU_dgj * n_j_perp[k]);
        torque_i[k] = dU_dgi * rxni[k];
        torque_j[k] = dU_dgj * rxnj[k];
        }
    if (energy_shift)
        {
        const Scalar rcut = sqrt(rcutsq);
        const Scalar Morse_exp_shift = exp(-(rcut - p->r_eq) * p->M_rinv);
        const Scalar one_minus_exp_shift = 1.0 - Morse_exp_shift;
        const Scalar UMorse_shift = p->M_d * (one_minus_exp_shift * one_minus_exp_shift - 1.0);
        e -= UMorse_shift * Omega_i * Omega_j;
        }
    *pair_eng = e;
    return 1;
    }

/* ------------------------------------------------------------------------- */
/* Bond evaluators.  Evaluator(rsq, params).evalForceAndEnergy(force_divr,   */
/* bond_eng) -> bool   (src/BondEvaluator.h:84)                              */
/* ------------------------------------------------------------------------- */
typedef int (*azo_bond_eval_fn)(const void* params, Scalar rsq, Scalar* force_divr, Scalar* bond_eng);

/* src/BondEvaluatorDoubleWell.h:96-113 */
int azo_eval_double_well(const void* vp, Scalar rsq, Scalar* force_divr, Scalar* bond_eng)
    {
    const azo_dw_t* p = (const azo_dw_t*)vp;
    *bond_eng = 0;
    *force_divr = 0;
    if (p->r_diff == 0.0)
        return 0;
    const Scalar r = sqrt(rsq);
    const Scalar x = (p->r_1 - r) / p->r_diff;
    const Scalar x2 = x * x;
    const Scalar y = 1.0 - x2;
    const Scalar y2 = y * y;
    *bond_eng = p->U_1 * y2 + p->U_tilt * (1.0 - x - y2);
    *force_divr = (4.0 * x * y * (p->U_tilt - p->U_1) - p->U_tilt) / (p->r_diff * r);
    return 1;
    }

/* src/BondEvaluatorQuartic.h:113-124 (ctor), 129-200 */
int azo_eval_quartic(const void* vp, Scalar rsq, Scalar* force_divr, Scalar* bond_eng)
    {
    const azo_quartic_t* p = (const azo_quartic_t*)vp;
    const Scalar lj1 = p->epsilon_x_4 * p->sigma_6 * p->sigma_6;
    const Scalar lj2 = p->epsilon_x_4 * p->sigma_6;
    const Scalar epsilon_m = p->epsilon_x_4 / 4.0;
    const Scalar k = p->k, r_0 = p->r_0, b_1 = p->b_1, b_2 = p->b_2, U_0 = p->U_0, delta = p->delta;
    Scalar f = 0, e = 0;
    *bond_eng = 0;
    *force_divr = 0;
    if (r_0 == 0.0)
        return 0;
    Scalar r_red = 1.0;
    if (delta == 0.0)
        {
        const Scalar r2inv = 1.0 / rsq;
        const Scalar r6inv = r2inv * r2inv * r2inv;
        const Scalar sigma6inv = lj2 / lj1;
        if (lj1 != 0.0 && r6inv > sigma6inv / 2.0)
            {
            const Scalar epsilon = lj2 * lj2 / 4.0 / lj1; /* :150 recomputes epsilon */
            f += r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
            e += r6inv * (lj1 * r6inv - lj2) + epsilon;
            }
        if (rsq < r_0 * r_0)
            r_red = sqrt(rsq) - r_0;
        }
    else
        {
        const Scalar r = sqrt(rsq) - delta;
        const Scalar r2inv = 1.0 / r / r;
        const Scalar r6inv = r2inv * r2inv * r2inv;
        const Scalar sigma6inv = lj2 / lj1;
        if (lj1 != 0.0 && r6inv > sigma6inv / 2.0)
            {
            f += r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2) / r / (r + delta);
            e += r6inv * (lj1 * r6inv - lj2) + epsilon_m;
            }
        if (r < r_0)
            r_red = r - r_0;
        }
    if (r_red < 0.0)
        {
        f += -1.0 * k * r_red * (4 * r_red * r_red - 3 * (b_1 + b_2) * r_red + 2 * b_1 * b_2) / (r_red + r_0 + delta);
        e += k * (r_red - b_1) * (r_red - b_2) * r_red * r_red + U_0;
        }
    else
        e += U_0;
    *force_divr = f;
    *bond_eng = e;
    return 1;
    }

/* ------------------------------------------------------------------------- */
/* Box + minimum image (HOOMD BoxDim restated; box centred on the origin,    */
/* lo = -L/2, hi = +L/2; tilt factors xy, xz, yz). PARITY UNPINNED.          */
/* Orthorhombic boxes use HOOMD's CPU compare form; triclinic boxes use the  */
/* rint form. Both agree away from exact ties at +-L/2.                      */
/* ------------------------------------------------------------------------- */
typedef struct
    {
    Scalar L[3];
    Scalar tilt[3]; /* xy, xz, yz */
    int32_t periodic[3];
    int32_t _pad;
    } azo_box_t;

static inline void min_image(const azo_box_t* b, Scalar w[3])
    {
    const int tric = (b->tilt[0] != 0.0 || b->tilt[1] != 0.0 || b->tilt[2] != 0.0);
    if (!tric)
        {
        for (int k = 0; k < 3; ++k)
            if (b->periodic[k])
                {
                const Scalar hi = 0.5 * b->L[k];
                if (w[k] >= hi)
                    w[k] -= b->L[k];
                else if (w[k] < -hi)
                    w[k] += b->L[k];
                }
        }
    else
        {
        if (b->periodic[2])
            {
            const Scalar img = rint(w[2] / b->L[2]);
            w[2] -= b->L[2] * img;
            w[1] -= b->L[2] * b->tilt[2] * img;
            w[0] -= b->L[2] * b->tilt[1] * img;
            }
        if (b->periodic[1])
            {
            const Scalar img = rint(w[1] / b->L[1]);
            w[1] -= b->L[1] * img;
            w[0] -= b->L[1] * b->tilt[0] * img;
            }
        if (b->periodic[0])
            w[0] -= b->L[0] * rint(w[0] / b->L[0]);
        }
    }

void azo_min_image(const azo_box_t* b, Scalar w[3]) { min_image(b, w); }

/* particle type lives in the low 32 bits of pos.w (HOOMD __scalar_as_int on a
 * double Scalar: union of int and double; usage in-repo at src/HarmonicBarrier.h:163-165) */
static inline int32_t type_of(const Scalar* pos4)
    {
    int32_t t;
    memcpy(&t, &pos4[3], sizeof(int32_t));
    return t;
    }

/* ------------------------------------------------------------------------- */
/* Outer loops (HOOMD PotentialPair::computeForces restated).                */
/* shift_mode: 0 none, 1 shift, 2 xplor.                                     */
/* virial layout: 6 rows (xx, xy, xz, yy, yz, zz) x virial_pitch.            */
/* Type-pair table index: typ_i * ntypes + typ_j (tables are symmetric:      */
/* setParams(a,b) fills both (a,b) and (b,a)).                               */
/* ------------------------------------------------------------------------- */
typedef struct
    {
    int64_t N;          /* local particles (forces are written for these)    */
    int64_t n_ghost;    /* ghosts appended after N in pos (never receive force) */
    const Scalar* pos;  /* (N+n_ghost) x 4                                   */
    azo_box_t box;
    const uint32_t* n_neigh;   /* N                                          */
    const uint32_t* nlist;
    const uint64_t* head_list; /* N                                          */
    int32_t ntypes;
    int32_t shift_mode;
    const Scalar* rcutsq;      /* ntypes^2                                   */
    const Scalar* ronsq;       /* ntypes^2                                   */
    int32_t half_list;         /* 1: each pair stored once, third-law scatter */
    int32_t compute_virial;
    Scalar* force;             /* N x 4 (fx, fy, fz, energy), overwritten    */
    Scalar* virial;            /* 6 x virial_pitch, overwritten (may be NULL) */
    int64_t virial_pitch;
    } azo_pair_args_t;

static inline void apply_xplor(int evaluated, Scalar rsq, Scalar ronsq, Scalar rcutsq, Scalar* force_divr,
                               Scalar* pair_eng)
    {
    if (evaluated && rsq >= ronsq && rsq < rcutsq)
        {
        const Scalar old_pair_eng = *pair_eng;
        const Scalar old_force_divr = *force_divr;
        const Scalar d = rcutsq - ronsq;
        const Scalar xplor_denom_inv = 1.0 / (d * d * d);
        const Scalar rsq_minus_r_cut_sq = rsq - rcutsq;
        const Scalar s = rsq_minus_r_cut_sq * rsq_minus_r_cut_sq * (rcutsq + 2.0 * rsq - 3.0 * ronsq) * xplor_denom_inv;
        const Scalar ds_dr_divr = 12.0 * (rsq - ronsq) * rsq_minus_r_cut_sq * xplor_denom_inv;
        *pair_eng = old_pair_eng * s;
        *force_divr = s * old_force_divr - ds_dr_divr * old_pair_eng;
        }
    }

static void zero_outputs(int64_t N, Scalar* force, Scalar* virial, int64_t pitch, int cv)
    {
    memset(force, 0, sizeof(Scalar) * 4 * (size_t)N);
    if (cv && virial)
        memset(virial, 0, sizeof(Scalar) * 6 * (size_t)pitch);
    }

void azo_pair_forces(const azo_pair_args_t* a, azo_pair_eval_fn eval, const void* params, size_t param_stride)
    {
    const int64_t N = a->N;
    zero_outputs(N, a->force, a->virial, a->virial_pitch, a->compute_virial);
    for (int64_t i = 0; i < N; ++i)
        {
        const Scalar* pi = a->pos + 4 * i;
        const int32_t typei = type_of(pi);
        Scalar fi[3] = {0, 0, 0}, pei = 0, vi[6] = {0, 0, 0, 0, 0, 0};
        const uint64_t head = a->head_list[i];
        const uint32_t nn = a->n_neigh[i];
        for (uint32_t k = 0; k < nn; ++k)
            {
            const uint32_t j = a->nlist[head + k];
            const Scalar* pj = a->pos + 4 * (int64_t)j;
            Scalar dx[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
            const int32_t typej = type_of(pj);
            min_image(&a->box, dx);
            const Scalar rsq = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
            const int64_t tp = (int64_t)typei * a->ntypes + typej;
            const Scalar rcutsq = a->rcutsq[tp];
            const Scalar ronsq = a->ronsq[tp];
            int energy_shift = 0;
            if (a->shift_mode == 1)
                energy_shift = 1;
            else if (a->shift_mode == 2 && ronsq > rcutsq)
                energy_shift = 1;
            Scalar force_divr = 0, pair_eng = 0;
            const int evaluated = eval((const char*)params + param_stride * (size_t)tp, rsq, rcutsq, energy_shift,
                                       &force_divr, &pair_eng);
            if (evaluated)
                {
                if (a->shift_mode == 2)
                    apply_xplor(evaluated, rsq, ronsq, rcutsq, &force_divr, &pair_eng);
                Scalar pv[6] = {0, 0, 0, 0, 0, 0};
                if (a->compute_virial)
                    {
                    const Scalar fd2 = 0.5 * force_divr;
                    pv[0] = fd2 * dx[0] * dx[0]; pv[1] = fd2 * dx[0] * dx[1]; pv[2] = fd2 * dx[0] * dx[2];
                    pv[3] = fd2 * dx[1] * dx[1]; pv[4] = fd2 * dx[1] * dx[2]; pv[5] = fd2 * dx[2] * dx[2];
                    for (int c = 0; c < 6; ++c) vi[c] += pv[c];
                    }
                fi[0] += dx[0] * force_divr; fi[1] += dx[1] * force_divr; fi[2] += dx[2] * force_divr;
                pei += pair_eng * 0.5;
                if (a->half_list && (int64_t)j < N)
                    {
                    Scalar* fj = a->force + 4 * (int64_t)j;
                    fj[0] -= dx[0] * force_divr; fj[1] -= dx[1] * force_divr; fj[2] -= dx[2] * force_divr;
                    fj[3] += pair_eng * 0.5;
                    if (a->compute_virial)
                        for (int c = 0; c < 6; ++c) a->virial[c * a->virial_pitch + j] += pv[c];
                    }
                }
            }
        Scalar* fo = a->force + 4 * i;
        fo[0] += fi[0]; fo[1] += fi[1]; fo[2] += fi[2]; fo[3] += pei;
        if (a->compute_virial)
            for (int c = 0; c < 6; ++c) a->virial[c * a->virial_pitch + i] += vi[c];
        }
    }

/* Full-list loop parallelised over particles (cpu_baseline leg (ii): all cores).
 * Same arithmetic per (i, j) as azo_pair_forces with half_list = 0. */
void azo_pair_forces_omp(const azo_pair_args_t* a, azo_pair_eval_fn eval, const void* params, size_t param_stride,
                         int nthreads)
    {
    const int64_t N = a->N;
    (void)nthreads;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i)
        {
        const Scalar* pi = a->pos + 4 * i;
        const int32_t typei = type_of(pi);
        Scalar fi[3] = {0, 0, 0}, pei = 0, vi[6] = {0, 0, 0, 0, 0, 0};
        const uint64_t head = a->head_list[i];
        const uint32_t nn = a->n_neigh[i];
        for (uint32_t k = 0; k < nn; ++k)
            {
            const uint32_t j = a->nlist[head + k];
            const Scalar* pj = a->pos + 4 * (int64_t)j;
            Scalar dx[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
            const int32_t typej = type_of(pj);
            min_image(&a->box, dx);
            const Scalar rsq = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
            const int64_t tp = (int64_t)typei * a->ntypes + typej;
            const Scalar rcutsq = a->rcutsq[tp];
            const Scalar ronsq = a->ronsq[tp];
            int energy_shift = (a->shift_mode == 1) || (a->shift_mode == 2 && ronsq > rcutsq);
            Scalar force_divr = 0, pair_eng = 0;
            const int evaluated = eval((const char*)params + param_stride * (size_t)tp, rsq, rcutsq, energy_shift,
                                       &force_divr, &pair_eng);
            if (evaluated)
                {
                if (a->shift_mode == 2)
                    apply_xplor(evaluated, rsq, ronsq, rcutsq, &force_divr, &pair_eng);
                if (a->compute_virial)
                    {
                    const Scalar fd2 = 0.5 * force_divr;
                    vi[0] += fd2 * dx[0] * dx[0]; vi[1] += fd2 * dx[0] * dx[1]; vi[2] += fd2 * dx[0] * dx[2];
                    vi[3] += fd2 * dx[1] * dx[1]; vi[4] += fd2 * dx[1] * dx[2]; vi[5] += fd2 * dx[2] * dx[2];
                    }
                fi[0] += dx[0] * force_divr; fi[1] += dx[1] * force_divr; fi[2] += dx[2] * force_divr;
                pei += pair_eng * 0.5;
                }
            }
        Scalar* fo = a->force + 4 * i;
        fo[0] = fi[0]; fo[1] = fi[1]; fo[2] = fi[2]; fo[3] = pei;
        if (a->compute_virial)
            for (int c = 0; c < 6; ++c) a->virial[c * a->virial_pitch + i] = vi[c];
        }
    }

/* entry points by evaluator id (for ctypes) */
enum { AZO_PLJ = 0, AZO_HERTZ = 1, AZO_YUKAWA = 2, AZO_COLLOID = 3, AZO_DPD_CONS = 4 };

static azo_pair_eval_fn pair_fn(int id, size_t* stride)
    {
    switch (id)
        {
    case AZO_PLJ: *stride = sizeof(azo_plj_t); return azo_eval_plj;
    case AZO_HERTZ: *stride = sizeof(azo_hertz_t); return azo_eval_hertz;
    case AZO_YUKAWA: *stride = sizeof(azo_yukawa_t); return azo_eval_yukawa;
    case AZO_COLLOID: *stride = sizeof(azo_colloid_t); return azo_eval_colloid;
    case AZO_DPD_CONS: *stride = sizeof(azo_dpd_t); return azo_eval_dpd_cons;
    default: *stride = 0; return NULL;
        }
    }

int azo_pair_forces_by_id(int id, const azo_pair_args_t* a, const void* params, int nthreads)
    {
    size_t stride;
    azo_pair_eval_fn fn = pair_fn(id, &stride);
    if (!fn) return -1;
    if (nthreads == 0)
        azo_pair_forces(a, fn, params, stride);
    else
        {
        if (a->half_list) return -2;
        azo_pair_forces_omp(a, fn, params, stride, nthreads);
        }
    return 0;
    }

int azo_pair_eval_by_id(int id, const void* param, Scalar rsq, Scalar rcutsq, int energy_shift, Scalar* f, Scalar* e)
    {
    size_t stride;
    azo_pair_eval_fn fn = pair_fn(id, &stride);
    if (!fn) return -1;
    *f = 0; *e = 0;
    return fn(param, rsq, rcutsq, energy_shift, f, e);
    }

/* ------------------------------------------------------------------------- */
/* DPD thermostat loop (HOOMD PotentialPairDPDThermo::computeForces restated) */
/* ------------------------------------------------------------------------- */
typedef struct
    {
    azo_pair_args_t base;
    const Scalar* vel;    /* (N+n_ghost) x 4 (vx, vy, vz, mass)               */
    const uint32_t* tag;  /* (N+n_ghost)                                      */
    uint16_t seed;
    uint16_t _pad[3];
    uint64_t timestep;
    Scalar deltaT;
    Scalar T;
    } azo_dpd_args_t;

void azo_dpd_forces(const azo_dpd_args_t* d, const azo_dpd_t* params)
    {
    const azo_pair_args_t* a = &d->base;
    const int64_t N = a->N;
    zero_outputs(N, a->force, a->virial, a->virial_pitch, a->compute_virial);
    for (int64_t i = 0; i < N; ++i)
        {
        const Scalar* pi = a->pos + 4 * i;
        const Scalar* vi_ = d->vel + 4 * i;
        const int32_t typei = type_of(pi);
        Scalar fi[3] = {0, 0, 0}, pei = 0, vi[6] = {0, 0, 0, 0, 0, 0};
        const uint64_t head = a->head_list[i];
        const uint32_t nn = a->n_neigh[i];
        for (uint32_t k = 0; k < nn; ++k)
            {
            const uint32_t j = a->nlist[head + k];
            const Scalar* pj = a->pos + 4 * (int64_t)j;
            const Scalar* vj = d->vel + 4 * (int64_t)j;
            Scalar dx[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
            const Scalar dv[3] = {vi_[0] - vj[0], vi_[1] - vj[1], vi_[2] - vj[2]};
            const int32_t typej = type_of(pj);
            min_image(&a->box, dx);
            const Scalar rsq = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
            const Scalar rdotv = dx[0] * dv[0] + dx[1] * dv[1] + dx[2] * dv[2];
            const int64_t tp = (int64_t)typei * a->ntypes + typej;
            const Scalar rcutsq = a->rcutsq[tp];
            Scalar force_divr = 0, force_divr_cons = 0, pair_eng = 0;
            int evaluated = 0;
            if (rsq < rcutsq)
                {
                const Scalar alpha = azo_dpd_alpha(d->seed, d->tag[i], d->tag[j], d->timestep);
                evaluated = azo_eval_dpd_thermo(params + tp, rsq, rcutsq, rdotv, d->deltaT, d->T, alpha, &force_divr,
                                                &force_divr_cons, &pair_eng);
                }
            if (evaluated)
                {
                Scalar pv[6] = {0, 0, 0, 0, 0, 0};
                if (a->compute_virial)
                    {
                    /* virial from the conservative part only (:193-194) */
                    const Scalar fd2 = 0.5 * force_divr_cons;
                    pv[0] = fd2 * dx[0] * dx[0]; pv[1] = fd2 * dx[0] * dx[1]; pv[2] = fd2 * dx[0] * dx[2];
                    pv[3] = fd2 * dx[1] * dx[1]; pv[4] = fd2 * dx[1] * dx[2]; pv[5] = fd2 * dx[2] * dx[2];
                    for (int c = 0; c < 6; ++c) vi[c] += pv[c];
                    }
                fi[0] += dx[0] * force_divr; fi[1] += dx[1] * force_divr; fi[2] += dx[2] * force_divr;
                pei += pair_eng * 0.5;
                if (a->half_list && (int64_t)j < N)
                    {
                    Scalar* fj = a->force + 4 * (int64_t)j;
                    fj[0] -= dx[0] * force_divr; fj[1] -= dx[1] * force_divr; fj[2] -= dx[2] * force_divr;
                    fj[3] += pair_eng * 0.5;
                    if (a->compute_virial)
                        for (int c = 0; c < 6; ++c) a->virial[c * a->virial_pitch + j] += pv[c];
                    }
                }
            }
        Scalar* fo = a->force + 4 * i;
        fo[0] += fi[0]; fo[1] += fi[1]; fo[2] += fi[2]; fo[3] += pei;
        if (a->compute_virial)
            for (int c = 0; c < 6; ++c) a->virial[c * a->virial_pitch + i] += vi[c];
        }
    }

/* ------------------------------------------------------------------------- */
/* Anisotropic loop (HOOMD AnisoPotentialPair::computeForces restated).      */
/* dr = r_i - r_j; force from evaluate() acts on i, -force on j; torque_i on */
/* i, torque_j on j (src/pytest/test_pair_aniso.py:113-168 pins signs).      */
/* ------------------------------------------------------------------------- */
typedef struct
    {
    azo_pair_args_t base;
    const Scalar* orientation; /* (N+n_ghost) x 4, scalar part first         */
    Scalar* torque;            /* N x 4 (tx, ty, tz, 0), overwritten         */
    } azo_aniso_args_t;

void azo_aniso_forces_tpm(const azo_aniso_args_t* g, const azo_tpm_t* params)
    {
    const azo_pair_args_t* a = &g->base;
    const int64_t N = a->N;
    zero_outputs(N, a->force, a->virial, a->virial_pitch, a->compute_virial);
    memset(g->torque, 0, sizeof(Scalar) * 4 * (size_t)N);
    for (int64_t i = 0; i < N; ++i)
        {
        const Scalar* pi = a->pos + 4 * i;
        const Scalar* qi = g->orientation + 4 * i;
        const int32_t typei = type_of(pi);
        Scalar fi[3] = {0, 0, 0}, ti[3] = {0, 0, 0}, pei = 0, vi[6] = {0, 0, 0, 0, 0, 0};
        const uint64_t head = a->head_list[i];
        const uint32_t nn = a->n_neigh[i];
        for (uint32_t k = 0; k < nn; ++k)
            {
            const uint32_t j = a->nlist[head + k];
            const Scalar* pj = a->pos + 4 * (int64_t)j;
            const Scalar* qj = g->orientation + 4 * (int64_t)j;
            Scalar dx[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
            const int32_t typej = type_of(pj);
            min_image(&a->box, dx);
            const int64_t tp = (int64_t)typei * a->ntypes + typej;
            const Scalar rcutsq = a->rcutsq[tp];
            Scalar force[3] = {0, 0, 0}, tqi[3] = {0, 0, 0}, tqj[3] = {0, 0, 0}, pair_eng = 0;
            const int evaluated
                = azo_eval_tpm(params + tp, dx, qi, qj, rcutsq, a->shift_mode == 1, force, &pair_eng, tqi, tqj);
            if (evaluated)
                {
                Scalar pv[6] = {0, 0, 0, 0, 0, 0};
                if (a->compute_virial)
                    {
                    pv[0] = 0.5 * dx[0] * force[0]; pv[1] = 0.5 * dx[1] * force[0]; pv[2] = 0.5 * dx[2] * force[0];
                    pv[3] = 0.5 * dx[1] * force[1]; pv[4] = 0.5 * dx[2] * force[1]; pv[5] = 0.5 * dx[2] * force[2];
                    for (int c = 0; c < 6; ++c) vi[c] += pv[c];
                    }
                for (int c = 0; c < 3; ++c) { fi[c] += force[c]; ti[c] += tqi[c]; }
                pei += pair_eng * 0.5;
                if (a->half_list && (int64_t)j < N)
                    {
                    Scalar* fj = a->force + 4 * (int64_t)j;
                    Scalar* tj = g->torque + 4 * (int64_t)j;
                    for (int c = 0; c < 3; ++c) { fj[c] -= force[c]; tj[c] += tqj[c]; }
                    fj[3] += pair_eng * 0.5;
                    if (a->compute_virial)
                        for (int c = 0; c < 6; ++c) a->virial[c * a->virial_pitch + j] += pv[c];
                    }
                }
            }
        Scalar* fo = a->force + 4 * i;
        Scalar* to = g->torque + 4 * i;
        fo[0] += fi[0]; fo[1] += fi[1]; fo[2] += fi[2]; fo[3] += pei;
        to[0] += ti[0]; to[1] += ti[1]; to[2] += ti[2];
        if (a->compute_virial)
            for (int c = 0; c < 6; ++c) a->virial[c * a->virial_pitch + i] += vi[c];
        }
    }

/* ------------------------------------------------------------------------- */
/* Bond loop (HOOMD PotentialBond::computeForces restated): loop over bonds  */
/* (a, b, type); dx = x_a - x_b min-imaged; half the bond energy to each     */
/* member; returns the number of bonds whose evaluator returned false        */
/* (HOOMD raises "bond out of bounds" / sets d_flags).                       */
/* ------------------------------------------------------------------------- */
typedef struct
    {
    int64_t N;
    int64_t n_ghost;
    const Scalar* pos;
    azo_box_t box;
    int64_t n_bonds;
    const uint32_t* bonds;     /* n_bonds x 2 (particle indices a, b)        */
    const uint32_t* bond_type; /* n_bonds                                    */
    int32_t n_bond_types;
    int32_t compute_virial;
    Scalar* force;
    Scalar* virial;
    int64_t virial_pitch;
    } azo_bond_args_t;

int azo_bond_forces(const azo_bond_args_t* a, azo_bond_eval_fn eval, const void* params, size_t stride)
    {
    const int64_t N = a->N;
    int bad = 0;
    zero_outputs(N, a->force, a->virial, a->virial_pitch, a->compute_virial);
    for (int64_t b = 0; b < a->n_bonds; ++b)
        {
        const int64_t ia = a->bonds[2 * b], ib = a->bonds[2 * b + 1];
        const Scalar* pa = a->pos + 4 * ia;
        const Scalar* pb = a->pos + 4 * ib;
        Scalar dx[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
        min_image(&a->box, dx);
        const Scalar rsq = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
        Scalar force_divr = 0, bond_eng = 0;
        const int evaluated = eval((const char*)params + stride * a->bond_type[b], rsq, &force_divr, &bond_eng);
        if (!evaluated)
            {
            ++bad;
            continue;
            }
        bond_eng *= 0.5;
        Scalar pv[6] = {0, 0, 0, 0, 0, 0};
        if (a->compute_virial)
            {
            const Scalar fd2 = 0.5 * force_divr;
            pv[0] = fd2 * dx[0] * dx[0]; pv[1] = fd2 * dx[0] * dx[1]; pv[2] = fd2 * dx[0] * dx[2];
            pv[3] = fd2 * dx[1] * dx[1]; pv[4] = fd2 * dx[1] * dx[2]; pv[5] = fd2 * dx[2] * dx[2];
            }
        if (ia < N)
            {
            Scalar* f = a->force + 4 * ia;
            f[0] += dx[0] * force_divr; f[1] += dx[1] * force_divr; f[2] += dx[2] * force_divr; f[3] += bond_eng;
            if (a->compute_virial)
                for (int c = 0; c < 6; ++c) a->virial[c * a->virial_pitch + ia] += pv[c];
            }
        if (ib < N)
            {
            Scalar* f = a->force + 4 * ib;
            f[0] -= dx[0] * force_divr; f[1] -= dx[1] * force_divr; f[2] -= dx[2] * force_divr; f[3] += bond_eng;
            if (a->compute_virial)
                for (int c = 0; c < 6; ++c) a->virial[c * a->virial_pitch + ib] += pv[c];
            }
        }
    return bad;
    }

int azo_bond_forces_by_id(int id, const azo_bond_args_t* a, const void* params)
    {
    if (id == 0) return azo_bond_forces(a, azo_eval_double_well, params, sizeof(azo_dw_t));
    if (id == 1) return azo_bond_forces(a, azo_eval_quartic, params, sizeof(azo_quartic_t));
    return -1;
    }

int azo_bond_eval_by_id(int id, const void* param, Scalar rsq, Scalar* f, Scalar* e)
    {
    if (id == 0) return azo_eval_double_well(param, rsq, f, e);
    if (id == 1) return azo_eval_quartic(param, rsq, f, e);
    return -1;
    }

/* ------------------------------------------------------------------------- */
/* Cell-list neighbor list (HOOMD NeighborListBinned restated): for test     */
/* inputs only. Orthorhombic periodic boxes. r_list per type pair.           */
/* full (half_list = 0): every (i,j) and (j,i); half: stored once at the     */
/* lower index. Neighbors of i are emitted in ascending j.                   */
/* Two-pass API: call with nlist == NULL to get counts, then fill.           */
/* ------------------------------------------------------------------------- */
static int cmp_u32(const void* a, const void* b)
    {
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
    }

int64_t azo_build_nlist(int64_t N, int64_t n_total, const Scalar* pos, const azo_box_t* box, int32_t ntypes,
                        const Scalar* r_list /* ntypes^2 */, int32_t half_list, const uint32_t* excl_n,
                        const uint32_t* excl /* N x excl_stride */, int32_t excl_stride, uint32_t* n_neigh,
                        uint64_t* head_list, uint32_t* nlist /* may be NULL */)
    {
    Scalar rmax = 0;
    for (int t = 0; t < ntypes * ntypes; ++t)
        if (r_list[t] > rmax) rmax = r_list[t];
    int dim[3];
    Scalar w[3];
    for (int k = 0; k < 3; ++k)
        {
        dim[k] = (int)floor(box->L[k] / rmax);
        if (dim[k] < 1) dim[k] = 1;
        w[k] = box->L[k] / dim[k];
        }
    const int64_t ncell = (int64_t)dim[0] * dim[1] * dim[2];
    int64_t* cell_start = (int64_t*)calloc((size_t)ncell + 1, sizeof(int64_t));
    uint32_t* cell_of = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n_total);
    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n_total);
    for (int64_t i = 0; i < n_total; ++i)
        {
        int c[3];
        for (int k = 0; k < 3; ++k)
            {
            Scalar f = (pos[4 * i + k] + 0.5 * box->L[k]) / w[k];
            int ci = (int)floor(f);
            if (box->periodic[k])
                {
                ci %= dim[k];
                if (ci < 0) ci += dim[k];
                }
            else
                {
                if (ci < 0) ci = 0;
                if (ci >= dim[k]) ci = dim[k] - 1;
                }
            c[k] = ci;
            }
        cell_of[i] = (uint32_t)((c[2] * dim[1] + c[1]) * dim[0] + c[0]);
        cell_start[cell_of[i] + 1]++;
        }
    for (int64_t c = 0; c < ncell; ++c) cell_start[c + 1] += cell_start[c];
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)ncell);
    memcpy(fill, cell_start, sizeof(int64_t) * (size_t)ncell);
    for (int64_t i = 0; i < n_total; ++i) order[fill[cell_of[i]]++] = (uint32_t)i;
    free(fill);

    int64_t total = 0;
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * 4096);
    size_t tmp_cap = 4096;
    for (int64_t i = 0; i < N; ++i)
        {
        const Scalar* pi = pos + 4 * i;
        const int32_t ti = type_of(pi);
        const uint32_t ci = cell_of[i];
        const int cx = ci % dim[0], cy = (ci / dim[0]) % dim[1], cz = ci / (dim[0] * dim[1]);
        size_t cnt = 0;
        /* visit each distinct neighbor cell once (dims < 3 alias) */
        int64_t seen[27];
        int nseen = 0;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dxx = -1; dxx <= 1; ++dxx)
                    {
                    int nx = cx + dxx, ny = cy + dy, nz = cz + dz;
                    if (box->periodic[0]) nx = (nx + dim[0]) % dim[0]; else if (nx < 0 || nx >= dim[0]) continue;
                    if (box->periodic[1]) ny = (ny + dim[1]) % dim[1]; else if (ny < 0 || ny >= dim[1]) continue;
                    if (box->periodic[2]) nz = (nz + dim[2]) % dim[2]; else if (nz < 0 || nz >= dim[2]) continue;
                    const int64_t nc = ((int64_t)nz * dim[1] + ny) * dim[0] + nx;
                    int dup = 0;
                    for (int s = 0; s < nseen; ++s) dup |= (seen[s] == nc);
                    if (dup) continue;
                    seen[nseen++] = nc;
                    for (int64_t q = cell_start[nc]; q < cell_start[nc + 1]; ++q)
                        {
                        const uint32_t j = order[q];
                        if ((int64_t)j == i) continue;
                        if (half_list && (int64_t)j < i) continue;
                        const Scalar* pj = pos + 4 * (int64_t)j;
                        Scalar d[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
                        min_image(box, d);
                        const Scalar rsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                        const Scalar rl = r_list[(int64_t)ti * ntypes + type_of(pj)];
                        if (rl <= 0 || rsq > rl * rl) continue;
                        int ex = 0;
                        if (excl_n)
                            for (uint32_t e = 0; e < excl_n[i]; ++e) ex |= (excl[i * excl_stride + e] == j);
                        if (ex) continue;
                        if (cnt == tmp_cap)
                            {
                            tmp_cap *= 2;
                            tmp = (uint32_t*)realloc(tmp, sizeof(uint32_t) * tmp_cap);
                            }
                        tmp[cnt++] = j;
                        }
                    }
        n_neigh[i] = (uint32_t)cnt;
        head_list[i] = (uint64_t)total;
        if (nlist)
            {
            qsort(tmp, cnt, sizeof(uint32_t), cmp_u32);
            memcpy(nlist + total, tmp, sizeof(uint32_t) * cnt);
            }
        total += (int64_t)cnt;
        }
    free(tmp);
    free(cell_start);
    free(cell_of);
    free(order);
    return total;
    }

/* ------------------------------------------------------------------------- */
/* Counter-based synthetic-data RNG: SplitMix64 finaliser of (seed, tag,     */
/* component). Mirrored bit-for-bit by azplugins_amd/synthetic.py (numpy).   */
/* ------------------------------------------------------------------------- */
uint64_t azo_hash64(uint64_t seed, uint64_t tag, uint64_t comp)
    {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + tag * 0xBF58476D1CE4E5B9ull + comp * 0x94D049BB133111EBull
                 + 0x2545F4914F6CDD1Dull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    /* second round so that nearby (tag, comp) decorrelate fully */
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
    }
/* uniform in [0,1) with 53 bits */
Scalar azo_u01(uint64_t seed, uint64_t tag, uint64_t comp)
    {
    return (Scalar)(azo_hash64(seed, tag, comp) >> 11) * (1.0 / 9007199254740992.0);
    }
void azo_u01_array(uint64_t seed, uint64_t tag0, int64_t n, uint64_t comp, Scalar* out)
    {
    for (int64_t i = 0; i < n; ++i) out[i] = azo_u01(seed, tag0 + (uint64_t)i, comp);
    }

/* ------------------------------------------------------------------------- */
/* One-body harmonic barriers.                            */
/* Evaluators: src/PlanarBarrierEvaluator.h:36-48,                           */
/* src/SphericalBarrierEvaluator.h:36-51; loop: src/HarmonicBarrier.h:150-180 */
/* (positions wrapped into the box first; virial not computed).              */
/* Pinned by src/pytest/test_external.py:95-223 (tests/golden).              */
/* ------------------------------------------------------------------------- */
static void wrap_into_box(const azo_box_t* b, Scalar w[3])
    {
    /* HOOMD BoxDim::wrap restated, one shift per axis */
    if (b->periodic[2])
        {
        const Scalar h = 0.5 * b->L[2];
        if (w[2] >= h) { w[2] -= b->L[2]; w[1] -= b->L[2] * b->tilt[2]; w[0] -= b->L[2] * b->tilt[1]; }
        else if (w[2] < -h) { w[2] += b->L[2]; w[1] += b->L[2] * b->tilt[2]; w[0] += b->L[2] * b->tilt[1]; }
        }
    if (b->periodic[1])
        {
        const Scalar h = 0.5 * b->L[1], s = w[2] * b->tilt[2];
        if (w[1] >= h + s) { w[1] -= b->L[1]; w[0] -= b->L[1] * b->tilt[0]; }
        else if (w[1] < -h + s) { w[1] += b->L[1]; w[0] += b->L[1] * b->tilt[0]; }
        }
    if (b->periodic[0])
        {
        const Scalar h = 0.5 * b->L[0], s = w[1] * b->tilt[0] + w[2] * (b->tilt[1] - b->tilt[0] * b->tilt[2]);
        if (w[0] >= h + s) w[0] -= b->L[0];
        else if (w[0] < -h + s) w[0] += b->L[0];
        }
    }

void azo_barrier_forces(int spherical, int64_t N, const Scalar* pos, const azo_box_t* box, const Scalar* params /* ntypes x 2 */,
                        Scalar location, Scalar* force)
    {
    for (int64_t i = 0; i < N; ++i)
        {
        Scalar p[3] = {pos[4 * i], pos[4 * i + 1], pos[4 * i + 2]};
        const int32_t t = type_of(pos + 4 * i);
        const Scalar k = params[2 * t], offset = params[2 * t + 1];
        wrap_into_box(box, p);
        Scalar* f = force + 4 * i;
        f[0] = f[1] = f[2] = f[3] = 0;
        if (spherical)
            {
            const Scalar r = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
            const Scalar dr = r - (location + offset);
            if (dr > 0.0)
                {
                const Scalar k_dr = k * dr;
                f[0] = -(k_dr / r) * p[0]; f[1] = -(k_dr / r) * p[1]; f[2] = -(k_dr / r) * p[2];
                f[3] = 0.5 * k_dr * dr;
                }
            }
        else
            {
            const Scalar dy = p[1] - (location + offset);
            if (dy > 0.0)
                {
                f[1] = -k * dy;
                f[3] = -0.5 * f[1] * dy;
                }
            }
        }
    }

/* ------------------------------------------------------------------------- */
/* Velocity-Verlet NVE (HOOMD TwoStepConstantVolume without */
/* thermostat restated; PARITY UNPINNED beyond the DPD <kT> test).           */
/* ------------------------------------------------------------------------- */
void azo_nve_step(int step_one, int64_t N, Scalar* pos, Scalar* vel, const Scalar* net_force, const azo_box_t* box, Scalar dt)
    {
    for (int64_t i = 0; i < N; ++i)
        {
        const Scalar minv = 1.0 / vel[4 * i + 3];
        for (int k = 0; k < 3; ++k)
            vel[4 * i + k] += 0.5 * dt * net_force[4 * i + k] * minv;
        if (step_one)
            {
            Scalar p[3];
            for (int k = 0; k < 3; ++k)
                p[k] = pos[4 * i + k] + dt * vel[4 * i + k];
            wrap_into_box(box, p);
            for (int k = 0; k < 3; ++k)
                pos[4 * i + k] = p[k];
            }
        }
    }

/* Rotational half of the velocity-Verlet NVE step (HOOMD TwoStepConstantVolume with
 * integrate_rotational_dof, restated from recollection of hoomd/md/TwoStepConstantVolume.cc:
 * the symplectic NO_SQUISH scheme of Miller et al. / Kamberaj et al.). HOOMD-blue's source is
 * absent, so this restatement is PARITY UNPINNED; the reference only exercises it through
 * `moment_inertia` in src/pytest/test_pair_aniso.py:113-140. Pinned here by: free rotation
 * conserves |L| and the rotational kinetic energy, and total energy is conserved with the
 * TwoPatchMorse torques (tests/test_gpu_external_nve.py).
 *   q: orientation (scalar first), p: angular-momentum quaternion (body angular momentum
 *   s = 1/2 conj(q) p), I: principal moments, t: net torque (space frame).
 *   step one: p += dt q t_body; free rotations 3 (dt/2), 2 (dt/2), 1 (dt), 2 (dt/2), 3 (dt/2);
 *             q renormalised.   step two: p += dt q t_body.
 * An axis with zero moment of inertia is not integrated. */
static void q_mul_vec(const Scalar* q, const Scalar* t, Scalar* out) /* (s, v) * (0, t) */
    {
    out[0] = -(q[1] * t[0] + q[2] * t[1] + q[3] * t[2]);
    out[1] = q[0] * t[0] + (q[2] * t[2] - q[3] * t[1]);
    out[2] = q[0] * t[1] + (q[3] * t[0] - q[1] * t[2]);
    out[3] = q[0] * t[2] + (q[1] * t[1] - q[2] * t[0]);
    }
static void rotate_conj(const Scalar* q, const Scalar* v, Scalar* out) /* rotate(conj(q), v) */
    {
    const Scalar s = q[0], ux = -q[1], uy = -q[2], uz = -q[3];
    const Scalar c = s * s - (ux * ux + uy * uy + uz * uz);
    const Scalar d = 2.0 * (ux * v[0] + uy * v[1] + uz * v[2]);
    out[0] = c * v[0] + 2.0 * s * (uy * v[2] - uz * v[1]) + d * ux;
    out[1] = c * v[1] + 2.0 * s * (uz * v[0] - ux * v[2]) + d * uy;
    out[2] = c * v[2] + 2.0 * s * (ux * v[1] - uy * v[0]) + d * uz;
    }
static void free_rotation(int axis, Scalar* p, Scalar* q, Scalar I, Scalar dt)
    {
    Scalar pk[4], qk[4];
    if (axis == 3)
        {
        pk[0] = -p[3]; pk[1] = p[2]; pk[2] = -p[1]; pk[3] = p[0];
        qk[0] = -q[3]; qk[1] = q[2]; qk[2] = -q[1]; qk[3] = q[0];
        }
    else if (axis == 2)
        {
        pk[0] = -p[2]; pk[1] = -p[3]; pk[2] = p[0]; pk[3] = p[1];
        qk[0] = -q[2]; qk[1] = -q[3]; qk[2] = q[0]; qk[3] = q[1];
        }
    else
        {
        pk[0] = -p[1]; pk[1] = p[0]; pk[2] = p[3]; pk[3] = -p[2];
        qk[0] = -q[1]; qk[1] = q[0]; qk[2] = q[3]; qk[3] = -q[2];
        }
    const Scalar phi = 0.25 / I * (p[0] * qk[0] + p[1] * qk[1] + p[2] * qk[2] + p[3] * qk[3]);
    const Scalar c = cos(dt * phi), sn = sin(dt * phi);
    for (int k = 0; k < 4; ++k)
        {
        p[k] = c * p[k] + sn * pk[k];
        q[k] = c * q[k] + sn * qk[k];
        }
    }
void azo_nve_rot_step(int step_one, int64_t N, Scalar* orientation, Scalar* angmom, const Scalar* inertia, const Scalar* net_torque,
                      Scalar dt)
    {
    for (int64_t i = 0; i < N; ++i)
        {
        Scalar* q = orientation + 4 * i;
        Scalar* p = angmom + 4 * i;
        const Scalar* I = inertia + 3 * i;
        Scalar t[3], qt[4];
        rotate_conj(q, net_torque + 4 * i, t);
        for (int k = 0; k < 3; ++k)
            if (I[k] == 0.0)
                t[k] = 0.0;
        q_mul_vec(q, t, qt);
        for (int k = 0; k < 4; ++k)
            p[k] += dt * qt[k];
        if (step_one)
            {
            if (I[2] != 0.0) free_rotation(3, p, q, I[2], 0.5 * dt);
            if (I[1] != 0.0) free_rotation(2, p, q, I[1], 0.5 * dt);
            if (I[0] != 0.0) free_rotation(1, p, q, I[0], dt);
            if (I[1] != 0.0) free_rotation(2, p, q, I[1], 0.5 * dt);
            if (I[2] != 0.0) free_rotation(3, p, q, I[2], 0.5 * dt);
            const Scalar n = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            for (int k = 0; k < 4; ++k)
                q[k] *= n;
            }
        }
    }

size_t azo_sizeof(int what)
    {
    switch (what)
        {
    case 0: return sizeof(azo_plj_t);
    case 1: return sizeof(azo_hertz_t);
    case 2: return sizeof(azo_yukawa_t);
    case 3: return sizeof(azo_colloid_t);
    case 4: return sizeof(azo_dpd_t);
    case 5: return sizeof(azo_tpm_t);
    case 6: return sizeof(azo_dw_t);
    case 7: return sizeof(azo_quartic_t);
    case 10: return sizeof(azo_box_t);
    case 11: return sizeof(azo_pair_args_t);
    case 12: return sizeof(azo_dpd_args_t);
    case 13: return sizeof(azo_aniso_args_t);
    case 14: return sizeof(azo_bond_args_t);
    default: return 0;
        }
    }
