"""ctypes front-end of the CPU oracle (``oracle/azp_oracle.c``).

TEST INFRASTRUCTURE ONLY. May be imported by ``tests/``, by
``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg -- never
by anything under ``azplugins_amd/`` (the product).

All arrays are numpy, float64 / uint32 / uint64, C-contiguous.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    src = os.path.join(_HERE, "azp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


# --------------------------------------------------------------------------
# structs
# --------------------------------------------------------------------------
class Box(C.Structure):
    _fields_ = [("L", C.c_double * 3), ("tilt", C.c_double * 3), ("periodic", C.c_int32 * 3), ("_pad", C.c_int32)]


def make_box(L, tilt=(0.0, 0.0, 0.0), periodic=(1, 1, 1)):
    b = Box()
    if np.isscalar(L):
        L = (L, L, L)
    for k in range(3):
        b.L[k] = float(L[k])
        b.tilt[k] = float(tilt[k])
        b.periodic[k] = int(periodic[k])
    return b


class PairArgs(C.Structure):
    _fields_ = [
        ("N", C.c_int64),
        ("n_ghost", C.c_int64),
        ("pos", C.c_void_p),
        ("box", Box),
        ("n_neigh", C.c_void_p),
        ("nlist", C.c_void_p),
        ("head_list", C.c_void_p),
        ("ntypes", C.c_int32),
        ("shift_mode", C.c_int32),
        ("rcutsq", C.c_void_p),
        ("ronsq", C.c_void_p),
        ("half_list", C.c_int32),
        ("compute_virial", C.c_int32),
        ("force", C.c_void_p),
        ("virial", C.c_void_p),
        ("virial_pitch", C.c_int64),
    ]


class DPDArgs(C.Structure):
    _fields_ = [
        ("base", PairArgs),
        ("vel", C.c_void_p),
        ("tag", C.c_void_p),
        ("seed", C.c_uint16),
        ("_pad", C.c_uint16 * 3),
        ("timestep", C.c_uint64),
        ("deltaT", C.c_double),
        ("T", C.c_double),
    ]


class AnisoArgs(C.Structure):
    _fields_ = [("base", PairArgs), ("orientation", C.c_void_p), ("torque", C.c_void_p)]


class BondArgs(C.Structure):
    _fields_ = [
        ("N", C.c_int64),
        ("n_ghost", C.c_int64),
        ("pos", C.c_void_p),
        ("box", Box),
        ("n_bonds", C.c_int64),
        ("bonds", C.c_void_p),
        ("bond_type", C.c_void_p),
        ("n_bond_types", C.c_int32),
        ("compute_virial", C.c_int32),
        ("force", C.c_void_p),
        ("virial", C.c_void_p),
        ("virial_pitch", C.c_int64),
    ]


PAIR_IDS = {"PerturbedLennardJones": 0, "Hertz": 1, "ExpandedYukawa": 2, "Colloid": 3, "DPDConservative": 4}
PAIR_PARAM_DOUBLES = {0: 4, 1: 1, 2: 4, 3: 4, 4: 4}
BOND_IDS = {"DoubleWell": 0, "Quartic": 1}
SHIFT_MODES = {"none": 0, "shift": 1, "xplor": 2}


def _declare(l):
    d = C.c_double
    pd = C.POINTER(C.c_double)
    l.azo_pair_eval_by_id.argtypes = [C.c_int, C.c_void_p, d, d, C.c_int, pd, pd]
    l.azo_pair_eval_by_id.restype = C.c_int
    l.azo_bond_eval_by_id.argtypes = [C.c_int, C.c_void_p, d, pd, pd]
    l.azo_bond_eval_by_id.restype = C.c_int
    l.azo_pair_forces_by_id.argtypes = [C.c_int, C.POINTER(PairArgs), C.c_void_p, C.c_int]
    l.azo_pair_forces_by_id.restype = C.c_int
    l.azo_dpd_forces.argtypes = [C.POINTER(DPDArgs), C.c_void_p]
    l.azo_dpd_forces.restype = None
    l.azo_aniso_forces_tpm.argtypes = [C.POINTER(AnisoArgs), C.c_void_p]
    l.azo_aniso_forces_tpm.restype = None
    l.azo_bond_forces_by_id.argtypes = [C.c_int, C.POINTER(BondArgs), C.c_void_p]
    l.azo_bond_forces_by_id.restype = C.c_int
    l.azo_dpd_alpha.argtypes = [C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint64]
    l.azo_dpd_alpha.restype = d
    l.azo_eval_dpd_thermo.argtypes = [C.c_void_p, d, d, d, d, d, d, pd, pd, pd]
    l.azo_eval_dpd_thermo.restype = C.c_int
    l.azo_eval_tpm.argtypes = [C.c_void_p, pd, pd, pd, d, C.c_int, pd, pd, pd, pd]
    l.azo_eval_tpm.restype = C.c_int
    l.azo_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    l.azo_philox4x32_10.restype = None
    l.azo_build_nlist.argtypes = [
        C.c_int64, C.c_int64, C.c_void_p, C.POINTER(Box), C.c_int32, C.c_void_p, C.c_int32,
        C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
    ]
    l.azo_build_nlist.restype = C.c_int64
    l.azo_hash64.argtypes = [C.c_uint64] * 3
    l.azo_hash64.restype = C.c_uint64
    l.azo_u01_array.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_uint64, C.c_void_p]
    l.azo_u01_array.restype = None
    l.azo_sizeof.argtypes = [C.c_int]
    l.azo_sizeof.restype = C.c_size_t
    l.azo_min_image.argtypes = [C.POINTER(Box), pd]
    l.azo_min_image.restype = None
    for name, n in (("azo_make_plj", 3), ("azo_make_colloid", 4), ("azo_make_dw", 4), ("azo_make_quartic", 8)):
        getattr(l, name).argtypes = [d] * n + [C.c_void_p]
        getattr(l, name).restype = None
    l.azo_make_tpm.argtypes = [d] * 5 + [C.c_int, C.c_void_p]
    l.azo_make_tpm.restype = None
    l.azo_barrier_forces.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.POINTER(Box), C.c_void_p, d, C.c_void_p]
    l.azo_barrier_forces.restype = None
    l.azo_nve_step.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Box), d]
    l.azo_nve_step.restype = None


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# --------------------------------------------------------------------------
# parameter packing: user-level dict (reference Python API keys) -> raw struct
# as float64 array (one row per type pair / bond type)
# --------------------------------------------------------------------------
def pack_pair_params(name, d):
    """Return the raw parameter struct (np.float64 row) for one type pair."""
    l = lib()
    if name == "PerturbedLennardJones":
        out = np.zeros(4)
        l.azo_make_plj(d["epsilon"], d["sigma"], d["attraction_scale_factor"], _p(out))
    elif name == "Hertz":
        out = np.array([float(d["epsilon"])])
    elif name == "ExpandedYukawa":
        out = np.array([d["epsilon"], d["kappa"], d["delta"], 0.0], dtype=np.float64)
    elif name == "Colloid":
        out = np.zeros(4)
        l.azo_make_colloid(d["A"], d["a_1"], d["a_2"], d["sigma"], _p(out))
    elif name in ("DPDConservative", "DPDGeneralWeight"):
        out = np.array([d["A"], d["gamma"], d["s"], 0.0], dtype=np.float64)
    elif name == "TwoPatchMorse":
        out = np.zeros(6)
        l.azo_make_tpm(d["M_d"], d["M_r"], d["r_eq"], d["omega"], d["alpha"], int(bool(d["repulsion"])), _p(out))
    else:
        raise KeyError(name)
    return out


def pack_bond_params(name, d):
    l = lib()
    if name == "DoubleWell":
        out = np.zeros(4)
        l.azo_make_dw(d["r_0"], d["r_1"], d["U_1"], d["U_tilt"], _p(out))
    elif name == "Quartic":
        out = np.zeros(8)
        l.azo_make_quartic(d["k"], d["r_0"], d["b_1"], d["b_2"], d["U_0"], d["sigma"], d["epsilon"],
                           d.get("delta", 0.0), _p(out))
    else:
        raise KeyError(name)
    return out


# --------------------------------------------------------------------------
# scalar evaluators
# --------------------------------------------------------------------------
def eval_pair(name, params, r, r_cut, energy_shift=False):
    """(evaluated, force_divr, pair_eng) for one pair at distance r."""
    p = pack_pair_params(name, params) if isinstance(params, dict) else np.ascontiguousarray(params, dtype=np.float64)
    f = C.c_double(0)
    e = C.c_double(0)
    ok = lib().azo_pair_eval_by_id(PAIR_IDS[name], _p(p), r * r, r_cut * r_cut, int(energy_shift), C.byref(f),
                                   C.byref(e))
    return bool(ok), f.value, e.value


def eval_bond(name, params, r):
    p = pack_bond_params(name, params) if isinstance(params, dict) else np.ascontiguousarray(params, dtype=np.float64)
    f = C.c_double(0)
    e = C.c_double(0)
    ok = lib().azo_bond_eval_by_id(BOND_IDS[name], _p(p), r * r, C.byref(f), C.byref(e))
    return bool(ok), f.value, e.value


def eval_dpd_thermo(params, r, r_cut, rdotv, dt, kT, alpha):
    p = pack_pair_params("DPDGeneralWeight", params) if isinstance(params, dict) else params
    f, fc, e = C.c_double(0), C.c_double(0), C.c_double(0)
    ok = lib().azo_eval_dpd_thermo(_p(p), r * r, r_cut * r_cut, rdotv, dt, kT, alpha, C.byref(f), C.byref(fc),
                                   C.byref(e))
    return bool(ok), f.value, fc.value, e.value


def eval_tpm(params, dr, qi, qj, r_cut, energy_shift=False):
    p = pack_pair_params("TwoPatchMorse", params) if isinstance(params, dict) else params
    dr = np.ascontiguousarray(dr, dtype=np.float64)
    qi = np.ascontiguousarray(qi, dtype=np.float64)
    qj = np.ascontiguousarray(qj, dtype=np.float64)
    force = np.zeros(3)
    ti = np.zeros(3)
    tj = np.zeros(3)
    e = C.c_double(0)
    pd = C.POINTER(C.c_double)
    ok = lib().azo_eval_tpm(_p(p), dr.ctypes.data_as(pd), qi.ctypes.data_as(pd), qj.ctypes.data_as(pd),
                            r_cut * r_cut, int(energy_shift), force.ctypes.data_as(pd), C.byref(e),
                            ti.ctypes.data_as(pd), tj.ctypes.data_as(pd))
    return bool(ok), force, e.value, ti, tj


def philox4x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().azo_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def dpd_alpha(seed, tag_i, tag_j, timestep):
    return lib().azo_dpd_alpha(seed, tag_i, tag_j, timestep)


def hash64(seed, tag, comp):
    return lib().azo_hash64(seed, tag, comp)


def u01_array(seed, tag0, n, comp):
    out = np.empty(n, dtype=np.float64)
    lib().azo_u01_array(seed, tag0, n, comp, _p(out))
    return out


# --------------------------------------------------------------------------
# neighbor list
# --------------------------------------------------------------------------
def build_nlist(pos, box, r_list, N=None, ntypes=1, half=False, exclusions=None):
    """Cell-list neighbor list. ``pos`` is (n_total, 4) float64 with the type
    in the low 32 bits of w. Returns (n_neigh u32[N], head u64[N], nlist u32)."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n_total = pos.shape[0]
    if N is None:
        N = n_total
    rl = np.ascontiguousarray(np.broadcast_to(np.asarray(r_list, dtype=np.float64), (ntypes, ntypes)).reshape(-1))
    n_neigh = np.zeros(N, dtype=np.uint32)
    head = np.zeros(N, dtype=np.uint64)
    excl_n = excl = None
    stride = 0
    if exclusions is not None:
        excl_n, excl = exclusions
        excl_n = np.ascontiguousarray(excl_n, dtype=np.uint32)
        excl = np.ascontiguousarray(excl, dtype=np.uint32)
        stride = excl.shape[1]
    b = box if isinstance(box, Box) else make_box(*box)
    total = lib().azo_build_nlist(N, n_total, _p(pos), C.byref(b), ntypes, _p(rl), int(half), _p(excl_n), _p(excl),
                                  stride, _p(n_neigh), _p(head), None)
    nlist = np.zeros(max(total, 1), dtype=np.uint32)
    lib().azo_build_nlist(N, n_total, _p(pos), C.byref(b), ntypes, _p(rl), int(half), _p(excl_n), _p(excl), stride,
                          _p(n_neigh), _p(head), _p(nlist))
    return n_neigh, head, nlist[:total] if total else nlist[:0]


# --------------------------------------------------------------------------
# force loops
# --------------------------------------------------------------------------
def _pair_args(pos, box, nl, N, ntypes, r_cut, r_on, mode, half, virial):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n_total = pos.shape[0]
    if N is None:
        N = n_total
    n_neigh, head, nlist = nl
    rc = np.broadcast_to(np.asarray(r_cut, dtype=np.float64), (ntypes, ntypes))
    ro = np.broadcast_to(np.asarray(r_on, dtype=np.float64), (ntypes, ntypes))
    rcutsq = np.ascontiguousarray((rc * rc).reshape(-1))
    ronsq = np.ascontiguousarray((ro * ro).reshape(-1))
    force = np.zeros((N, 4))
    vir = np.zeros((6, N)) if virial else None
    a = PairArgs()
    a.N = N
    a.n_ghost = n_total - N
    a.pos = pos.ctypes.data
    a.box = box if isinstance(box, Box) else make_box(*box)
    a.n_neigh = n_neigh.ctypes.data
    a.nlist = nlist.ctypes.data if nlist.size else None
    a.head_list = head.ctypes.data
    a.ntypes = ntypes
    a.shift_mode = SHIFT_MODES[mode] if isinstance(mode, str) else int(mode)
    a.rcutsq = rcutsq.ctypes.data
    a.ronsq = ronsq.ctypes.data
    a.half_list = int(half)
    a.compute_virial = int(bool(virial))
    a.force = force.ctypes.data
    a.virial = vir.ctypes.data if virial else None
    a.virial_pitch = N
    keep = (pos, n_neigh, head, nlist, rcutsq, ronsq, force, vir)
    return a, force, vir, keep


def pair_forces(name, pos, box, nl, params, r_cut, r_on=0.0, mode="none", ntypes=1, N=None, half=False,
                virial=False, nthreads=0):
    """HOOMD-equivalent pair loop. ``params``: (ntypes*ntypes, k) float64 raw
    structs (see pack_pair_params). nthreads=0: serial loop (HOOMD's per-rank
    CPU execution model, supports half lists); >0: OpenMP over particles."""
    params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
    a, force, vir, keep = _pair_args(pos, box, nl, N, ntypes, r_cut, r_on, mode, half, virial)
    rc = lib().azo_pair_forces_by_id(PAIR_IDS[name], C.byref(a), _p(params), nthreads)
    if rc != 0:
        raise RuntimeError("oracle pair loop failed: %d" % rc)
    return (force, vir) if virial else force


def dpd_forces(pos, vel, tag, box, nl, params, r_cut, kT, dt, seed, timestep, ntypes=1, N=None, half=False,
               virial=False):
    params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
    vel = np.ascontiguousarray(vel, dtype=np.float64)
    tag = np.ascontiguousarray(tag, dtype=np.uint32)
    a, force, vir, keep = _pair_args(pos, box, nl, N, ntypes, r_cut, 0.0, "none", half, virial)
    d = DPDArgs()
    d.base = a
    d.vel = vel.ctypes.data
    d.tag = tag.ctypes.data
    d.seed = seed
    d.timestep = timestep
    d.deltaT = dt
    d.T = kT
    lib().azo_dpd_forces(C.byref(d), _p(params))
    return (force, vir) if virial else force


def aniso_forces_tpm(pos, orientation, box, nl, params, r_cut, mode="none", ntypes=1, N=None, half=False,
                     virial=False):
    params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
    orientation = np.ascontiguousarray(orientation, dtype=np.float64)
    a, force, vir, keep = _pair_args(pos, box, nl, N, ntypes, r_cut, 0.0, mode, half, virial)
    torque = np.zeros((a.N, 4))
    g = AnisoArgs()
    g.base = a
    g.orientation = orientation.ctypes.data
    g.torque = torque.ctypes.data
    lib().azo_aniso_forces_tpm(C.byref(g), _p(params))
    return (force, torque, vir) if virial else (force, torque)


def bond_forces(name, pos, box, bonds, bond_type, params, N=None, virial=False):
    """Returns (force, n_bad[, virial]). ``bonds`` (n_bonds, 2) particle indices."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n_total = pos.shape[0]
    if N is None:
        N = n_total
    bonds = np.ascontiguousarray(bonds, dtype=np.uint32).reshape(-1, 2)
    bond_type = np.ascontiguousarray(bond_type, dtype=np.uint32)
    params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
    force = np.zeros((N, 4))
    vir = np.zeros((6, N)) if virial else None
    a = BondArgs()
    a.N = N
    a.n_ghost = n_total - N
    a.pos = pos.ctypes.data
    a.box = box if isinstance(box, Box) else make_box(*box)
    a.n_bonds = bonds.shape[0]
    a.bonds = bonds.ctypes.data if bonds.size else None
    a.bond_type = bond_type.ctypes.data if bond_type.size else None
    a.n_bond_types = params.shape[0]
    a.compute_virial = int(bool(virial))
    a.force = force.ctypes.data
    a.virial = vir.ctypes.data if virial else None
    a.virial_pitch = N
    bad = lib().azo_bond_forces_by_id(BOND_IDS[name], C.byref(a), _p(params))
    return (force, bad, vir) if virial else (force, bad)


def pos4(xyz, types=None):
    """(n,3) coordinates + integer types -> HOOMD-style (n,4) Scalar4 array with
    the type index stored in the low 32 bits of w."""
    xyz = np.asarray(xyz, dtype=np.float64)
    out = np.zeros((xyz.shape[0], 4), dtype=np.float64)
    out[:, :3] = xyz
    if types is not None:
        w = np.zeros(xyz.shape[0], dtype=np.int64)
        w[:] = np.asarray(types, dtype=np.int64) & 0xFFFFFFFF
        out[:, 3] = w.view(np.float64)
    return out


def barrier_forces(kind, pos, box, params, location):
    """One-body harmonic barrier ("planar" or "spherical"); params (ntypes, 2) = k, offset."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
    force = np.zeros((pos.shape[0], 4))
    b = box if isinstance(box, Box) else make_box(*box)
    lib().azo_barrier_forces(int(kind == "spherical"), pos.shape[0], _p(pos), C.byref(b), _p(params), float(location), _p(force))
    return force


def nve_step(step_one, pos, vel, net_force, box, dt):
    """In-place velocity-Verlet half step on (n,4) pos / vel arrays."""
    b = box if isinstance(box, Box) else make_box(*box)
    lib().azo_nve_step(int(bool(step_one)), pos.shape[0], _p(pos), _p(vel), _p(np.ascontiguousarray(net_force)), C.byref(b), float(dt))


def nve_rot_step(step_one, orientation, angmom, inertia, net_torque, dt):
    """Rotational half of the NVE step, in place on copies; returns (orientation, angmom).
    PARITY UNPINNED against HOOMD (see azp_oracle.c: azo_nve_rot_step)."""
    q = np.ascontiguousarray(orientation, dtype=np.float64).copy()
    p = np.ascontiguousarray(angmom, dtype=np.float64).copy()
    I = np.ascontiguousarray(inertia, dtype=np.float64)
    t = np.ascontiguousarray(net_torque, dtype=np.float64)
    l = lib()
    l.azo_nve_rot_step.restype = None
    l.azo_nve_rot_step.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    l.azo_nve_rot_step(int(bool(step_one)), q.shape[0], q.ctypes.data, p.ctypes.data, I.ctypes.data, t.ctypes.data, float(dt))
    return q, p

