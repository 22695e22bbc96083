"""ctypes binding of ``libazp.so`` (the C ABI declared in ``include/azp.h``).

The product path has no CPU fallback: if the HIP library is missing or a call
fails, an exception is raised.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZP_LIB_PATH", os.path.join(_HERE, "libazp.so"))  # env override: kernel experiments only
CSRC = os.path.join(_HERE, "csrc")


class AzpError(RuntimeError):
    pass


# ---------------------------------------------------------------------------
# structs (field order = include/azp.h)
# ---------------------------------------------------------------------------
class Box(C.Structure):
    _fields_ = [("L", C.c_double * 3), ("tilt", C.c_double * 3), ("periodic", C.c_int32 * 3), ("_pad", C.c_int32)]


class PairArgs(C.Structure):
    _fields_ = [
        ("d_force", C.c_void_p),
        ("d_virial", C.c_void_p),
        ("virial_pitch", C.c_uint64),
        ("N", C.c_uint32),
        ("n_max", C.c_uint32),
        ("d_pos", C.c_void_p),
        ("box", Box),
        ("d_n_neigh", C.c_void_p),
        ("d_nlist", C.c_void_p),
        ("d_head_list", C.c_void_p),
        ("d_rcutsq", C.c_void_p),
        ("d_ronsq", C.c_void_p),
        ("size_nlist", C.c_uint64),
        ("ntypes", C.c_uint32),
        ("shift_mode", C.c_uint32),
        ("compute_virial", C.c_uint32),
        ("block_size", C.c_uint32),
        ("threads_per_particle", C.c_uint32),
        ("flags", C.c_uint32),
        ("range_first", C.c_uint32),
        ("range_count", C.c_uint32),
        ("r_list_max", C.c_double),
        ("d_rinnersq", C.c_void_p),
        ("has_displacement_bound", C.c_uint32),
        ("_pad3", C.c_uint32),
        ("displacement_bound", C.c_double),
        ("d_displacement", C.c_void_p),
        ("list_generation", C.c_uint64),
        ("d_stale_flag", C.c_void_p),
        ("d_displacement_sq_bits", C.c_void_p),
        ("displacement_bound_extra", C.c_double),
    ]


class DPDArgs(C.Structure):
    _fields_ = [
        ("pair", PairArgs),
        ("d_vel", C.c_void_p),
        ("d_tag", C.c_void_p),
        ("timestep", C.c_uint64),
        ("deltaT", C.c_double),
        ("T", C.c_double),
        ("seed", C.c_uint16),
        ("_pad", C.c_uint16 * 3),
    ]


class AnisoArgs(C.Structure):
    _fields_ = [("pair", PairArgs), ("d_orientation", C.c_void_p), ("d_torque", C.c_void_p)]


class BondArgs(C.Structure):
    _fields_ = [
        ("d_force", C.c_void_p),
        ("d_virial", C.c_void_p),
        ("virial_pitch", C.c_uint64),
        ("N", C.c_uint32),
        ("n_max", C.c_uint32),
        ("d_pos", C.c_void_p),
        ("box", Box),
        ("d_gpu_bondlist", C.c_void_p),
        ("d_gpu_bond_pos", C.c_void_p),
        ("d_gpu_n_bonds", C.c_void_p),
        ("pitch", C.c_uint64),
        ("n_bond_types", C.c_uint32),
        ("compute_virial", C.c_uint32),
        ("block_size", C.c_uint32),
        ("_pad", C.c_uint32),
    ]


class BarrierArgs(C.Structure):
    _fields_ = [
        ("d_force", C.c_void_p),
        ("d_virial", C.c_void_p),
        ("virial_pitch", C.c_uint64),
        ("N", C.c_uint32),
        ("ntypes", C.c_uint32),
        ("d_pos", C.c_void_p),
        ("box", Box),
        ("d_params", C.c_void_p),
        ("location", C.c_double),
        ("block_size", C.c_uint32),
        ("_pad", C.c_uint32),
    ]


class NVEArgs(C.Structure):
    _fields_ = [
        ("d_pos", C.c_void_p),
        ("d_vel", C.c_void_p),
        ("d_net_force", C.c_void_p),
        ("d_image", C.c_void_p),
        ("box", Box),
        ("dt", C.c_double),
        ("N", C.c_uint32),
        ("block_size", C.c_uint32),
    ]


class NVERotArgs(C.Structure):
    _fields_ = [
        ("d_orientation", C.c_void_p),
        ("d_angmom", C.c_void_p),
        ("d_inertia", C.c_void_p),
        ("d_net_torque", C.c_void_p),
        ("dt", C.c_double),
        ("N", C.c_uint32),
        ("block_size", C.c_uint32),
    ]


class PlanInfo(C.Structure):
    _fields_ = [
        ("valid", C.c_int32),
        ("invalid_reason", C.c_int32),
        ("threads_per_particle", C.c_uint32),
        ("tile_size", C.c_uint32),
        ("lds_slots", C.c_uint32),
        ("n_tiles", C.c_uint32),
        ("max_stage", C.c_uint32),
        ("max_row", C.c_uint32),
        ("total_stage", C.c_uint64),
        ("compiled_bytes", C.c_uint64),
        ("builds", C.c_uint64),
        ("from_cells", C.c_int32),
        ("row_capacity", C.c_uint32),
        ("list_id", C.c_uint64),
        ("head_id", C.c_uint64),
        ("balanced", C.c_int32),
        ("core_radius", C.c_float),
        ("sure_radius", C.c_float),
        ("max_member_cells", C.c_uint32),
    ]


class AutoPlanStats(C.Structure):
    _fields_ = [("calls", C.c_uint64), ("compiles", C.c_uint64), ("reuses", C.c_uint64), ("generic_fallbacks", C.c_uint64)]


PAIR_FLAG_NO_AUTO_PLAN = 1


class HaloField(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("row_bytes", C.c_uint32), ("_pad", C.c_uint32)]


class CellGrid(C.Structure):
    _fields_ = [("lo", C.c_double * 3), ("width", C.c_double * 3), ("dim", C.c_uint32 * 3), ("periodic", C.c_int32 * 3)]


class NlistArgs(C.Structure):
    _fields_ = [
        ("N", C.c_uint32),
        ("n_total", C.c_uint32),
        ("d_pos", C.c_void_p),
        ("box", Box),
        ("grid", CellGrid),
        ("ntypes", C.c_uint32),
        ("cell_subdivision", C.c_uint32),
        ("d_rlistsq", C.c_void_p),
        ("d_cell_of", C.c_void_p),
        ("d_cell_sorted", C.c_void_p),
        ("d_order", C.c_void_p),
        ("d_cell_start", C.c_void_p),
        ("d_n_excl", C.c_void_p),
        ("d_excl", C.c_void_p),
        ("excl_pitch", C.c_uint64),
        ("d_n_neigh", C.c_void_p),
        ("d_head_list", C.c_void_p),
        ("d_nlist", C.c_void_p),
        ("row_capacity", C.c_uint32),
        ("_pad2", C.c_uint32),
        ("d_max_neigh", C.c_void_p),
    ]


# every symbol include/azp.h declares: name -> (restype, argtypes)
_D = C.c_double
_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int)
_VP = C.c_void_p
SYMBOLS = {
    "azp_plj_params_make": (None, [_D] * 3 + [_VP]),
    "azp_plj_params_unpack": (None, [_VP] + [_PD] * 3),
    "azp_colloid_params_make": (None, [_D] * 4 + [_VP]),
    "azp_colloid_params_unpack": (None, [_VP] + [_PD] * 4),
    "azp_tpm_params_make": (None, [_D] * 5 + [C.c_int, _VP]),
    "azp_tpm_params_unpack": (None, [_VP] + [_PD] * 5 + [_PI]),
    "azp_dw_params_make": (None, [_D] * 4 + [_VP]),
    "azp_dw_params_unpack": (None, [_VP] + [_PD] * 4),
    "azp_quartic_params_make": (None, [_D] * 8 + [_VP]),
    "azp_quartic_params_unpack": (None, [_VP] + [_PD] * 8),
    "azp_pair_forces_perturbed_lennard_jones": (C.c_int, [C.POINTER(PairArgs), _VP, _VP]),
    "azp_pair_forces_hertz": (C.c_int, [C.POINTER(PairArgs), _VP, _VP]),
    "azp_pair_forces_expanded_yukawa": (C.c_int, [C.POINTER(PairArgs), _VP, _VP]),
    "azp_pair_forces_colloid": (C.c_int, [C.POINTER(PairArgs), _VP, _VP]),
    "azp_pair_forces_dpd_conservative": (C.c_int, [C.POINTER(PairArgs), _VP, _VP]),
    "azp_pair_plan_create": (C.c_int, [C.POINTER(_VP)]),
    "azp_pair_plan_destroy": (None, [_VP]),
    "azp_pair_plan_build": (C.c_int, [_VP, C.POINTER(PairArgs), _VP]),
    "azp_pair_plan_build_from_cells": (C.c_int, [_VP, C.POINTER(NlistArgs), C.POINTER(PairArgs), _VP]),
    "azp_pair_plan_set_bank_order": (C.c_int, [_VP, C.c_int]),
    "azp_pair_plan_set_balance": (C.c_int, [_VP, C.c_int]),
    "azp_pair_plan_tile_stage": (C.c_int, [_VP, C.POINTER(C.c_uint32), C.c_uint32]),
    "azp_pair_plan_query": (C.c_int, [_VP, C.POINTER(PlanInfo)]),
    "azp_pair_plan_phase_chunks": (C.c_int, [_VP, C.POINTER(C.c_float)]),
    "azp_tuning_set": (C.c_int, [C.c_int, C.c_int]),
    "azp_sum_forces": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), _VP, _VP]),
    "azp_pair_auto_plan_get_stats": (None, [C.POINTER(AutoPlanStats)]),
    "azp_pair_auto_plan_clear": (None, []),
    "azp_pair_forces_planned_perturbed_lennard_jones": (C.c_int, [_VP, C.POINTER(PairArgs), _VP, _VP]),
    "azp_pair_forces_planned_hertz": (C.c_int, [_VP, C.POINTER(PairArgs), _VP, _VP]),
    "azp_pair_forces_planned_expanded_yukawa": (C.c_int, [_VP, C.POINTER(PairArgs), _VP, _VP]),
    "azp_pair_forces_planned_colloid": (C.c_int, [_VP, C.POINTER(PairArgs), _VP, _VP]),
    "azp_pair_forces_planned_dpd_conservative": (C.c_int, [_VP, C.POINTER(PairArgs), _VP, _VP]),
    "azp_dpd_forces_general_weight": (C.c_int, [C.POINTER(DPDArgs), _VP, _VP]),
    "azp_aniso_forces_two_patch_morse": (C.c_int, [C.POINTER(AnisoArgs), _VP, _VP]),
    "azp_dpd_forces_planned_general_weight": (C.c_int, [_VP, C.POINTER(DPDArgs), _VP, _VP]),
    "azp_aniso_forces_planned_two_patch_morse": (C.c_int, [_VP, C.POINTER(AnisoArgs), _VP, _VP]),
    "azp_bond_forces_double_well": (C.c_int, [C.POINTER(BondArgs), _VP, _VP, _VP]),
    "azp_bond_forces_quartic": (C.c_int, [C.POINTER(BondArgs), _VP, _VP, _VP]),
    "azp_nlist_cell_assign": (C.c_int, [C.POINTER(NlistArgs), _VP]),
    "azp_nlist_cell_bounds": (C.c_int, [C.POINTER(NlistArgs), _VP]),
    "azp_nlist_bin": (C.c_int, [C.POINTER(NlistArgs), _VP, _VP, _VP]),
    "azp_nlist_count": (C.c_int, [C.POINTER(NlistArgs), _VP]),
    "azp_nlist_fill": (C.c_int, [C.POINTER(NlistArgs), _VP]),
    "azp_sorter_keys": (C.c_int, [C.c_uint32, _VP, C.POINTER(Box), C.POINTER(C.c_uint32), C.c_uint32, _VP, _VP]),
    "azp_halo_pack": (C.c_int, [C.c_uint32, _VP, _VP, C.c_uint32, _VP, _VP]),
    "azp_halo_pack_fields": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(HaloField), _VP, _VP, C.c_uint32, _VP]),
    "azp_halo_unpack_fields": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(HaloField), _VP, C.c_uint32, _VP]),
    "azp_nlist_distance_check": (C.c_int, [C.c_uint32, _VP, _VP, C.POINTER(Box), _D, _VP, _VP, _VP]),
    "azp_nlist_displacements": (C.c_int, [C.c_uint32, _VP, _VP, C.POINTER(Box), _D, _VP, _VP, _VP, _VP]),
    "azp_external_planar_harmonic_barrier": (C.c_int, [C.POINTER(BarrierArgs), _VP]),
    "azp_external_spherical_harmonic_barrier": (C.c_int, [C.POINTER(BarrierArgs), _VP]),
    "azp_planar_barrier_valid": (C.c_int, [_D, C.POINTER(Box)]),
    "azp_spherical_barrier_valid": (C.c_int, [_D, C.POINTER(Box)]),
    "azp_integrate_nve_step_one": (C.c_int, [C.POINTER(NVEArgs), _VP]),
    "azp_integrate_nve_step_two": (C.c_int, [C.POINTER(NVEArgs), _VP]),
    "azp_integrate_nve_step_two_one": (C.c_int, [C.POINTER(NVEArgs), _VP]),
    "azp_integrate_nve_rot_step_one": (C.c_int, [C.POINTER(NVERotArgs), _VP]),
    "azp_integrate_nve_rot_step_two": (C.c_int, [C.POINTER(NVERotArgs), _VP]),
    "azp_version": (C.c_int, []),
    "azp_status_string": (C.c_char_p, [C.c_int]),
    "azp_last_launch": (None, [C.POINTER(C.c_uint32)] * 4),
}

_lib = None


def lib():
    """Load libazp.so; raise (never fall back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AzpError(
                "libazp.so not found at %s: build it with `make -C %s` (or __graft_entry__.build()); "
                "there is no CPU fallback" % (LIB_PATH, CSRC)
            )
        # One HIP runtime per process: torch ships its own libamdhip64.so.7 and
        # must be loaded first so that libazp binds to the same runtime (the
        # device pointers and streams handed across the C ABI are torch's).
        import torch  # noqa: F401

        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


_ext = None


def ext_module():
    """The compiled ``_azplugins`` module (pybind11, built by ``make -C csrc``): the class
    names the reference registers (src/module.cc:110-166), holding the host-side parameter
    tables in C++. libazp (and with it torch's HIP runtime) is loaded first."""
    global _ext
    if _ext is None:
        lib()
        try:
            from . import _azplugins as m
        except ImportError as e:
            raise AzpError("the _azplugins extension module is missing (%s): build it with `make -C %s`" % (e, CSRC))
        _ext = m
    return _ext


def check(rc, what="libazp call"):
    if rc != 0:
        msg = lib().azp_status_string(rc).decode()
        raise AzpError("%s failed: status %d (%s)" % (what, rc, msg))


def raw_stream(device):
    """The current HIP stream of ``device`` as an integer handle (what the C ABI takes);
    torch.cuda.current_stream(device).cuda_stream without the Stream object (4 us per launch)."""
    import torch

    try:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)
    except AttributeError:  # private hook missing in this torch build
        return torch.cuda.current_stream(device).cuda_stream


def make_box(L, tilt=(0.0, 0.0, 0.0), periodic=(1, 1, 1)):
    b = Box()
    try:
        L = (float(L),) * 3
    except TypeError:
        pass
    for k in range(3):
        b.L[k] = float(L[k])
        b.tilt[k] = float(tilt[k])
        b.periodic[k] = int(periodic[k])
    return b


class PairPlan:
    """Owner of one azp_pair_plan handle."""

    def __init__(self):
        self._h = _VP()
        check(lib().azp_pair_plan_create(C.byref(self._h)), "azp_pair_plan_create")

    def build(self, args, stream):
        check(lib().azp_pair_plan_build(self._h, C.byref(args), stream), "azp_pair_plan_build")

    def build_from_cells(self, cells, pair, stream):
        check(lib().azp_pair_plan_build_from_cells(self._h, C.byref(cells), C.byref(pair), stream), "azp_pair_plan_build_from_cells")

    def tile_stage(self):
        """Staged-set size of every tile of the last build (numpy uint32)."""
        import numpy as np

        n = self.info()["n_tiles"]
        out = np.zeros(max(n, 1), dtype=np.uint32)
        m = lib().azp_pair_plan_tile_stage(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out[:max(m, 0)]

    def set_balance(self, enabled):
        check(lib().azp_pair_plan_set_balance(self._h, int(bool(enabled))), "azp_pair_plan_set_balance")

    def set_bank_order(self, enabled):
        check(lib().azp_pair_plan_set_bank_order(self._h, int(bool(enabled))), "azp_pair_plan_set_bank_order")

    def info(self):
        i = PlanInfo()
        check(lib().azp_pair_plan_query(self._h, C.byref(i)), "azp_pair_plan_query")
        return {f[0]: getattr(i, f[0]) for f in PlanInfo._fields_ if f[0] != "_pad"}

    def phase_chunks(self):
        """Diagnostics: mean chunks per slice covering the core entries / up to the end of the all-sure part / whole rows."""
        out = (C.c_float * 3)()
        check(lib().azp_pair_plan_phase_chunks(self._h, out), "azp_pair_plan_phase_chunks")
        return [float(x) for x in out]

    @property
    def handle(self):
        return self._h

    def __del__(self):
        try:
            if self._h:
                lib().azp_pair_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def auto_plan_stats():
    """Counters of the plan cache behind the HOOMD-signature entry points."""
    st = AutoPlanStats()
    lib().azp_pair_auto_plan_get_stats(C.byref(st))
    return {f[0]: int(getattr(st, f[0])) for f in AutoPlanStats._fields_}


def last_launch():
    vals = [C.c_uint32(0) for _ in range(4)]
    lib().azp_last_launch(*[C.byref(v) for v in vals])
    return dict(block_size=vals[0].value, threads_per_particle=vals[1].value, grid=vals[2].value,
                lds_bytes=vals[3].value)
