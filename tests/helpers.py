"""Test helpers: call libazp's C ABI directly on numpy inputs (uploaded with
torch) and build random multi-type configurations."""

import ctypes as C

import numpy as np

from azplugins_amd import _lib
from azplugins_amd import synthetic as syn

ENTRY = {
    "PerturbedLennardJones": "azp_pair_forces_perturbed_lennard_jones",
    "Hertz": "azp_pair_forces_hertz",
    "ExpandedYukawa": "azp_pair_forces_expanded_yukawa",
    "Colloid": "azp_pair_forces_colloid",
    "DPDConservative": "azp_pair_forces_dpd_conservative",
}
SHIFT = {"none": 0, "shift": 1, "xplor": 2}


def _dev(a, dtype=None):
    import torch

    a = np.ascontiguousarray(a if dtype is None else np.asarray(a, dtype=dtype))
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    if a.size == 0:
        return torch.zeros(1, dtype=torch.from_numpy(np.zeros(1, a.dtype)).dtype, device="cuda:0")
    return torch.from_numpy(a).to("cuda:0")


def gpu_pair_args(pos, box, nl, ntypes, r_cut, r_on, mode, virial, N=None, tpp=0, block_size=0, r_list_max=0.0, auto_plan=False):
    """auto_plan=False: azp_pair_forces_* run the generic kernel (AZP_PAIR_FLAG_NO_AUTO_PLAN);
    True: the HOOMD-signature call as the adapter makes it (plan cache inside libazp)."""
    import torch

    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n_total = pos.shape[0]
    N = n_total if N is None else N
    n_neigh, head, nlist = nl
    rc = np.broadcast_to(np.asarray(r_cut, dtype=np.float64), (ntypes, ntypes))
    ro = np.broadcast_to(np.asarray(r_on, dtype=np.float64), (ntypes, ntypes))
    t = dict(
        pos=_dev(pos), n_neigh=_dev(n_neigh, np.uint32), head=_dev(head, np.uint64), nlist=_dev(nlist, np.uint32),
        rcutsq=_dev((rc * rc).reshape(-1)), ronsq=_dev((ro * ro).reshape(-1)),
        force=torch.full((N, 4), float("nan"), dtype=torch.float64, device="cuda:0"),
        virial=torch.full((6, N), float("nan"), dtype=torch.float64, device="cuda:0"),
    )
    a = _lib.PairArgs()
    a.d_force = t["force"].data_ptr()
    a.d_virial = t["virial"].data_ptr()
    a.virial_pitch = N
    a.N = N
    a.n_max = n_total
    a.d_pos = t["pos"].data_ptr()
    a.box = box if isinstance(box, _lib.Box) else _lib.make_box(*box)
    a.d_n_neigh = t["n_neigh"].data_ptr()
    a.d_nlist = t["nlist"].data_ptr()
    a.d_head_list = t["head"].data_ptr()
    a.d_rcutsq = t["rcutsq"].data_ptr()
    a.d_ronsq = t["ronsq"].data_ptr()
    a.size_nlist = int(np.asarray(nlist).size)
    a.ntypes = ntypes
    a.shift_mode = SHIFT[mode]
    a.compute_virial = int(bool(virial))
    a.block_size = block_size
    a.threads_per_particle = tpp
    a.flags = 0 if auto_plan else _lib.PAIR_FLAG_NO_AUTO_PLAN
    a.r_list_max = r_list_max
    return a, t


def _stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


def _finish(t, virial):
    import torch

    torch.cuda.synchronize()
    f = t["force"].cpu().numpy()
    return (f, t["virial"].cpu().numpy()) if virial else f


def gpu_pair_forces(name, pos, box, nl, params, r_cut, r_on=0.0, mode="none", ntypes=1, N=None, virial=False, tpp=0,
                    block_size=0, r_list_max=0.0, planned=False, plan_info=None, r_inner=None, auto_plan=False):
    """planned=True: build a tile plan from the list and use the *_planned entry
    point; plan_info (a dict) receives azp_pair_plan_query's answer. auto_plan=True: the
    plain entry point with libazp's own plan cache (what the HOOMD adapter calls)."""
    a, t = gpu_pair_args(pos, box, nl, ntypes, r_cut, r_on, mode, virial, N, tpp, block_size, r_list_max, auto_plan)
    p = _dev(np.atleast_2d(params).astype(np.float64))
    if r_inner is not None:  # plan row-ordering hint (azp_pair_args.d_rinnersq)
        ri = _dev(np.broadcast_to(np.asarray(r_inner, dtype=np.float64) ** 2, (ntypes, ntypes)).reshape(-1).copy())
        a.d_rinnersq = ri.data_ptr()
    if planned:
        plan = _lib.PairPlan()
        plan.build(a, _stream())
        if plan_info is not None:
            plan_info.update(plan.info())
        entry = ENTRY[name].replace("azp_pair_forces_", "azp_pair_forces_planned_")
        _lib.check(getattr(_lib.lib(), entry)(plan.handle, C.byref(a), p.data_ptr(), _stream()), entry)
        out = _finish(t, virial)
        del plan
        return out
    fn = getattr(_lib.lib(), ENTRY[name])
    _lib.check(fn(C.byref(a), p.data_ptr(), _stream()), ENTRY[name])
    return _finish(t, virial)


def _planned_call(entry, a, args_struct, p, plan_info):
    """Build a tile plan from the list in ``a`` and call a *_planned entry point."""
    plan = _lib.PairPlan()
    plan.build(a, _stream())
    if plan_info is not None:
        plan_info.update(plan.info())
    _lib.check(getattr(_lib.lib(), entry)(plan.handle, C.byref(args_struct), p.data_ptr(), _stream()), entry)
    import torch

    torch.cuda.synchronize()
    del plan


def gpu_dpd_forces(pos, vel, tag, box, nl, params, r_cut, kT, dt, seed, timestep, ntypes=1, N=None, virial=False,
                   tpp=0, r_list_max=0.0, planned=False, plan_info=None, displacement_bound=None):
    a, t = gpu_pair_args(pos, box, nl, ntypes, r_cut, 0.0, "none", virial, N, tpp, 0, r_list_max)
    if displacement_bound is not None:
        a.has_displacement_bound, a.displacement_bound = 1, displacement_bound
    p = _dev(np.atleast_2d(params).astype(np.float64))
    v = _dev(vel, np.float64)
    tg = _dev(tag, np.uint32)
    d = _lib.DPDArgs()
    d.pair = a
    d.d_vel = v.data_ptr()
    d.d_tag = tg.data_ptr()
    d.timestep = timestep
    d.deltaT = dt
    d.T = kT
    d.seed = seed
    if planned:
        _planned_call("azp_dpd_forces_planned_general_weight", d.pair, d, p, plan_info)
    else:
        _lib.check(_lib.lib().azp_dpd_forces_general_weight(C.byref(d), p.data_ptr(), _stream()), "dpd")
    return _finish(t, virial)


def gpu_aniso_forces(pos, orientation, box, nl, params, r_cut, mode="none", ntypes=1, N=None, virial=False, tpp=0,
                     r_list_max=0.0, planned=False, plan_info=None, displacement_bound=None):
    import torch

    a, t = gpu_pair_args(pos, box, nl, ntypes, r_cut, 0.0, mode, virial, N, tpp, 0, r_list_max)
    if displacement_bound is not None:
        a.has_displacement_bound, a.displacement_bound = 1, displacement_bound
    p = _dev(np.atleast_2d(params).astype(np.float64))
    q = _dev(orientation, np.float64)
    tq = torch.full((a.N, 4), float("nan"), dtype=torch.float64, device="cuda:0")
    g = _lib.AnisoArgs()
    g.pair = a
    g.d_orientation = q.data_ptr()
    g.d_torque = tq.data_ptr()
    if planned:
        _planned_call("azp_aniso_forces_planned_two_patch_morse", g.pair, g, p, plan_info)
    else:
        _lib.check(_lib.lib().azp_aniso_forces_two_patch_morse(C.byref(g), p.data_ptr(), _stream()), "aniso")
    out = _finish(t, virial)
    torque = tq.cpu().numpy()
    return (out[0], torque, out[1]) if virial else (out, torque)


def bond_table(N, bonds, bond_type):
    """Per-particle GPU bond table (column-major) from a flat bond list."""
    bonds = np.asarray(bonds, dtype=np.int64).reshape(-1, 2)
    nb = np.zeros(N, dtype=np.uint32)
    for a_, b_ in bonds:
        if a_ < N:
            nb[a_] += 1
        if b_ < N:
            nb[b_] += 1
    width = max(int(nb.max()) if N else 0, 1)
    table = np.zeros((width, N, 2), dtype=np.uint32)
    bpos = np.zeros((width, N), dtype=np.uint32)
    fill = np.zeros(N, dtype=np.int64)
    for (a_, b_), t in zip(bonds, bond_type):
        for me, other, which in ((a_, b_, 0), (b_, a_, 1)):
            if me < N:
                table[fill[me], me] = (other, t)
                bpos[fill[me], me] = which
                fill[me] += 1
    return table, bpos, nb


def gpu_bond_forces(name, pos, box, bonds, bond_type, params, N=None, virial=False):
    import torch

    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n_total = pos.shape[0]
    N = n_total if N is None else N
    table, bpos, nb = bond_table(N, bonds, bond_type)
    t = dict(pos=_dev(pos), table=_dev(table), bpos=_dev(bpos), nb=_dev(nb),
             force=torch.full((N, 4), float("nan"), dtype=torch.float64, device="cuda:0"),
             virial=torch.full((6, N), float("nan"), dtype=torch.float64, device="cuda:0"),
             flags=torch.zeros(1, dtype=torch.int32, device="cuda:0"))
    p = _dev(np.atleast_2d(params).astype(np.float64))
    a = _lib.BondArgs()
    a.d_force = t["force"].data_ptr()
    a.d_virial = t["virial"].data_ptr()
    a.virial_pitch = N
    a.N = N
    a.n_max = n_total
    a.d_pos = t["pos"].data_ptr()
    a.box = box if isinstance(box, _lib.Box) else _lib.make_box(*box)
    a.d_gpu_bondlist = t["table"].data_ptr()
    a.d_gpu_bond_pos = t["bpos"].data_ptr()
    a.d_gpu_n_bonds = t["nb"].data_ptr()
    a.pitch = N
    a.n_bond_types = np.atleast_2d(params).shape[0]
    a.compute_virial = int(bool(virial))
    entry = {"DoubleWell": "azp_bond_forces_double_well", "Quartic": "azp_bond_forces_quartic"}[name]
    _lib.check(getattr(_lib.lib(), entry)(C.byref(a), p.data_ptr(), t["flags"].data_ptr(), _stream()), entry)
    torch.cuda.synchronize()
    f = t["force"].cpu().numpy()
    flag = int(t["flags"].item())
    return (f, flag, t["virial"].cpu().numpy()) if virial else (f, flag)


# ---------------------------------------------------------------------------
# configurations
# ---------------------------------------------------------------------------
def lattice_config(n_side, a, jitter, seed, ntypes=1, L_scale=None):
    """Jittered simple-cubic lattice with hashed type assignment."""
    xyz, L = syn.simple_cubic(n_side, n_side, n_side, a, jitter, seed)
    n = xyz.shape[0]
    typeid = (syn.hash64(seed + 1000, np.arange(n, dtype=np.uint64), 7) % np.uint64(ntypes)).astype(np.int64)
    return syn.pos4(xyz, typeid), L, typeid


def sym_table(ntypes, fn):
    """Symmetric per-type-pair table built from fn(i, j) for i <= j."""
    rows = [[None] * ntypes for _ in range(ntypes)]
    for i in range(ntypes):
        for j in range(i, ntypes):
            rows[i][j] = rows[j][i] = fn(i, j)
    return rows


def rel_err(a, b):
    """max |a - b| / max |b| (norm-wise) and the worst per-row relative error."""
    a = np.asarray(a)
    b = np.asarray(b)
    scale = np.abs(b).max()
    return np.abs(a - b).max() / (scale if scale > 0 else 1.0)


# ---------------------------------------------------------------------------
# binned particles (azp_nlist_args) for the plan-from-cells compiler
# ---------------------------------------------------------------------------
def gpu_cells(pos, box, r_list, ntypes=1, N=None, exclusions=None, row_capacity=0, sub=1):
    """Bin ``pos`` (n_total x 4, ghosts after the N locals) into cells of width >= max r_list / sub with
    libazp's own kernels (azp_nlist_cell_assign / _cell_bounds) and return (azp_nlist_args, keepalive).
    ``box``: (L, tilt, periodic) as for gpu_pair_args; ``exclusions``: (n_excl, excl[N, max]). ``sub`` = 2: cells of
    half the list radius (azp_nlist_args.cell_subdivision) where the plan compiler can take them (>= 5 cells along
    every periodic axis), else cells of the full radius as the product does."""
    import torch

    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n_total = pos.shape[0]
    N = n_total if N is None else N
    rl = np.broadcast_to(np.asarray(r_list, dtype=np.float64), (ntypes, ntypes))
    a = _lib.NlistArgs()
    a.N, a.n_total, a.ntypes = N, n_total, ntypes
    t = dict(pos=_dev(pos), rlistsq=_dev((rl * rl).reshape(-1)))
    a.d_pos = t["pos"].data_ptr()
    a.box = box if isinstance(box, _lib.Box) else _lib.make_box(*box)
    L = tuple(a.box.L)
    periodic = tuple(a.box.periodic)
    ncell = 1
    if sub == 2 and any(periodic[k] and int(np.floor(L[k] / (0.5 * rl.max()))) < 5 for k in range(3)):
        sub = 1
    a.cell_subdivision = sub
    for k in range(3):
        dim = max(int(np.floor(L[k] / (rl.max() / sub))), 1)
        a.grid.dim[k] = dim
        a.grid.width[k] = L[k] / dim
        a.grid.lo[k] = -0.5 * L[k]
        a.grid.periodic[k] = 1 if periodic[k] else 0
        ncell *= dim
    a.d_rlistsq = t["rlistsq"].data_ptr()
    t["cell_of"] = torch.empty(n_total, dtype=torch.int32, device="cuda:0")
    a.d_cell_of = t["cell_of"].data_ptr()
    l = _lib.lib()
    _lib.check(l.azp_nlist_cell_assign(C.byref(a), _stream()), "azp_nlist_cell_assign")
    t["cell_sorted"], order = torch.sort(t["cell_of"], stable=True)
    t["order"] = order.to(torch.int32)
    t["cell_start"] = torch.empty(ncell + 1, dtype=torch.int32, device="cuda:0")
    a.d_cell_sorted = t["cell_sorted"].data_ptr()
    a.d_order = t["order"].data_ptr()
    a.d_cell_start = t["cell_start"].data_ptr()
    _lib.check(l.azp_nlist_cell_bounds(C.byref(a), _stream()), "azp_nlist_cell_bounds")
    if exclusions is not None:
        n_excl, excl = exclusions
        t["n_excl"] = _dev(n_excl, np.uint32)
        t["excl"] = _dev(np.ascontiguousarray(np.asarray(excl, dtype=np.uint32).T))  # [max][N]
        a.d_n_excl = t["n_excl"].data_ptr()
        a.d_excl = t["excl"].data_ptr()
        a.excl_pitch = N
    t["n_neigh"] = torch.zeros(max(N, 1), dtype=torch.int32, device="cuda:0")
    a.d_n_neigh = t["n_neigh"].data_ptr()
    a.row_capacity = row_capacity
    return a, t
