"""Run-time domain decomposition on CPU (gloo; world_size 2, 4 and 8): particles move
step after step, cross sub-box faces and the periodic boundary, migrate and have their
ghosts re-selected at every "neighbor-list rebuild"; the packed per-step exchange keeps
the ghost rows current in between. With the oracle standing in for the GPU kernels, the
union of the per-rank forces must equal the single-domain forces at every step -- for
the PerturbedLJ pair force, for the DPD thermostat (positions + velocities + tags: both
owners of a cross-rank pair draw the same random number,
src/DPDPairEvaluatorGeneralWeight.h:213-231) and for TwoPatchMorse (positions +
orientations, forces and torques)."""

import os
import socket

import numpy as np
import pytest

from azplugins_amd import decomposition as dd
from azplugins_amd import synthetic as syn

R_BUFF = 0.4
STEPS = 7
REBUILD_EVERY = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _system(kind, world):
    """Global description every rank (and the checker) can regenerate."""
    if kind == "dpd":
        n = 6000 if world <= 4 else 12000
        cfg = syn.config_dpd(n)
        cfg["r_cut"], cfg["r_buff"] = 1.0, R_BUFF
    elif kind == "tpm":
        cfg = syn.config_tpm(8, 8, 16 if world <= 4 else 20)
        # lattice planes sit half a spacing from the slab faces: shift them next to the faces so that
        # the drift carries particles across
        cfg["xyz"] = syn.wrap(cfg["xyz"] + np.array([0.0, 0.0, 0.45]), cfg["L"])
    elif kind == "chains":
        # linear chains of 8 beads along x (C3's topology at reduced size): bonds cross the rank faces
        cfg = syn.config_chains(16, 12, 12, 8)
        cfg["r_cut"] = 2.5
    else:
        cfg = syn.config_plj_sc(12 if world <= 4 else 16)
        cfg["r_cut"] = 2.5
    n = cfg["xyz"].shape[0]
    tag = np.arange(n, dtype=np.uint64)
    # drift: every particle moves 0.09 per step in its own hashed direction (<= r_buff / 2 over two steps)
    v = np.stack([syn.normal(41, tag, c) for c in range(3)], axis=1)
    v *= (0.09 / np.linalg.norm(v, axis=1))[:, None]
    cfg["drift"] = v
    if "vel" not in cfg:
        cfg["vel"] = np.stack([syn.normal(42, tag, c) for c in range(3)], axis=1)
    if "orientation" not in cfg:
        cfg["orientation"] = syn.random_quaternions(n, 43)
    return cfg


def _positions(cfg, step):
    return syn.wrap(cfg["xyz"] + step * cfg["drift"], cfg["L"])


def _forces(oracle, kind, cfg, pos, vel, q, tag, box, nl, N, step):
    if kind == "chains":
        # DoubleWell bonds (row A8 / A9): the bonds this rank evaluates, by local index (azplugins_amd.state.localize_bonds),
        # forces of the local rows only -- a ghost member's share belongs to its owner's evaluation of the same bond
        from azplugins_amd.state import localize_bonds

        group, typeid = localize_bonds(tag.astype(np.int64), N, cfg["bonds"], np.zeros(len(cfg["bonds"]), dtype=np.uint32))
        p = oracle.pack_bond_params("DoubleWell", cfg["bond_params"])
        f, bad = oracle.bond_forces("DoubleWell", pos, box, group, typeid, p)
        assert bad == 0
        return f[:N]
    if kind == "dpd":
        p = oracle.pack_pair_params("DPDGeneralWeight", cfg["params"])
        return oracle.dpd_forces(pos, vel, tag, box, nl, p, cfg["r_cut"], 1.0, 0.01, 7, step, N=N)
    if kind == "tpm":
        p = oracle.pack_pair_params("TwoPatchMorse", cfg["params"])
        f, t = oracle.aniso_forces_tpm(pos, q, box, nl, p, cfg["r_cut"], mode="shift", N=N)
        return np.concatenate([f, t], axis=1)
    p = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    return oracle.pair_forces("PerturbedLennardJones", pos, box, nl, p, cfg["r_cut"], mode="shift", N=N)


def _worker(rank, world, port, out_dir, kind):
    import torch
    import torch.distributed as dist

    import oracle
    from azplugins_amd.domain import DeviceDomain

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _system(kind, world)
    n = cfg["xyz"].shape[0]
    r_list = cfg["r_cut"] + R_BUFF
    dec = dd.Decomposition(cfg["L"], world, r_list)
    mine = np.flatnonzero(dec.owner(cfg["xyz"]) == rank)
    vel4 = np.zeros((mine.size, 4))
    vel4[:, :3] = cfg["vel"][mine]
    vel4[:, 3] = 1.0
    arrays = dict(pos=torch.from_numpy(syn.pos4(cfg["xyz"][mine])), vel=torch.from_numpy(vel4),
                  orientation=torch.from_numpy(np.ascontiguousarray(cfg["orientation"][mine])),
                  tag=torch.from_numpy(mine.astype(np.int32)), image=torch.zeros((mine.size, 3), dtype=torch.int32))
    dom = DeviceDomain(dec, rank, arrays, density=n / float(np.prod(cfg["L"])))
    dom.rebuild()
    box = oracle.make_box(cfg["L"])
    halo = {"dpd": ["pos", "vel"], "tpm": ["pos", "orientation"]}.get(kind, ["pos"])
    drift = torch.from_numpy(cfg["drift"])
    L = torch.from_numpy(np.asarray(cfg["L"], dtype=np.float64))
    nl = None
    migrated = 0
    out = {}
    for step in range(STEPS):
        N = dom.N_local
        if step > 0:
            # owners move their particles (deterministic in the tag) and change their velocities;
            # ghost rows are stale until the exchange
            tag = dom.arrays["tag"][:N].to(torch.int64)
            x = dom.arrays["pos"][:N, :3] + drift[tag]
            dom.arrays["pos"][:N, :3] = x - L * torch.floor(x / L + 0.5)
            dom.arrays["vel"][:N, :3] = torch.from_numpy(cfg["vel"])[tag] * (1.0 + 0.1 * step)
            if step % REBUILD_EVERY == 0:
                dom.rebuild()  # migration + ghost re-selection (+ a full exchange)
                migrated += dom.num_migrated
                nl = None
            else:
                dom.exchange(halo)
        N = dom.N_local
        pos = dom.arrays["pos"].numpy()
        tag_all = dom.arrays["tag"].numpy().astype(np.int64)
        # every row, ghosts included, holds its particle's current global data
        assert np.allclose(pos[:, :3], _positions(cfg, step)[tag_all], atol=1e-12)
        if kind == "dpd":
            assert np.array_equal(dom.arrays["vel"].numpy()[:, :3], cfg["vel"][tag_all] * (1.0 + 0.1 * step if step else 1.0))
        if kind == "tpm":
            assert np.array_equal(dom.arrays["orientation"].numpy(), cfg["orientation"][tag_all])
        if nl is None:
            nl = oracle.build_nlist(pos, box, r_list, N=N)
            # interior particles (first n_interior rows) list no ghost
            n_neigh, head, lst = nl
            if dom.n_interior:
                last = int(head[dom.n_interior - 1]) + int(n_neigh[dom.n_interior - 1])
                assert (lst[:last] < N).all()
        f = _forces(oracle, kind, cfg, pos, dom.arrays["vel"].numpy(), dom.arrays["orientation"].numpy(),
                    dom.arrays["tag"].numpy().view(np.uint32), box, nl, N, step)
        out["tag%d" % step] = tag_all[:N]
        out["f%d" % step] = f
    out["migrated"] = np.array([migrated])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("plj", 2), ("plj", 4), ("plj", 8), ("dpd", 2), ("dpd", 8), ("tpm", 4), ("chains", 2), ("chains", 4)])
def test_migration_ghosts_and_packed_exchange_gloo(kind, world, tmp_path, oracle):
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), kind), nprocs=world, join=True)
    cfg = _system(kind, world)
    n = cfg["xyz"].shape[0]
    box = oracle.make_box(cfg["L"])
    data = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert sum(int(d["migrated"][0]) for d in data) > 0  # particles did change owner during the run
    vel4 = np.zeros((n, 4))
    vel4[:, 3] = 1.0
    tags = np.arange(n, dtype=np.uint32)
    for step in range(STEPS):
        pos = syn.pos4(_positions(cfg, step))
        vel4[:, :3] = cfg["vel"] * (1.0 + 0.1 * step if step else 1.0)
        nl = oracle.build_nlist(pos, box, cfg["r_cut"] + R_BUFF)
        f_ref = _forces(oracle, kind, cfg, pos, vel4, cfg["orientation"], tags, box, nl, n, step)
        f = np.full_like(f_ref, np.nan)
        seen = np.zeros(n, dtype=int)
        for d in data:
            f[d["tag%d" % step]] = d["f%d" % step]
            seen[d["tag%d" % step]] += 1
        assert np.all(seen == 1), "step %d: every particle has exactly one owner" % step
        assert np.abs(f - f_ref).max() <= 1e-11 * np.abs(f_ref).max(), "step %d" % step


def _global_snapshot():
    """A system with every per-particle field set to something of its own, tags that are not the indices, two types and bonds."""
    from azplugins_amd.state import Snapshot

    cfg = syn.config_chains(8, 8, 8, 8)
    n = cfg["xyz"].shape[0]
    idx = np.arange(n, dtype=np.uint64)
    tag = ((idx * 7919) % n).astype(np.uint32)   # a permutation: 7919 is prime and does not divide n
    assert np.unique(tag).size == n
    snap = Snapshot.from_arrays(cfg["xyz"], cfg["L"], typeid=(idx % 2).astype(np.uint32), types=("A", "B"), tag=tag,
                                velocity=np.stack([syn.normal(3, idx, c) for c in range(3)], axis=1),
                                orientation=syn.random_quaternions(n, 5),
                                bonds=cfg["bonds"], moment_inertia=np.stack([0.1 + syn.u01(4, idx, c) for c in range(3)], axis=1),
                                angmom=np.stack([syn.normal(6, idx, c) for c in range(4)], axis=1))
    snap.particles.mass[:] = 1.0 + syn.u01(8, idx, 0)
    return snap, cfg


def _distribute_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    snap, cfg = _global_snapshot() if rank == 0 else (None, syn.config_chains(8, 8, 8, 8))
    dec = dd.Decomposition(cfg["L"], world, 2.9)
    local, n_global, topo = dd.distribute_snapshot(snap, dec, root=0)
    p = local.particles
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), n_global=np.array([n_global]), position=p.position, typeid=p.typeid, orientation=p.orientation,
             velocity=p.velocity, mass=p.mass, moment_inertia=p.moment_inertia, angmom=p.angmom, tag=p.tag, types=np.array(p.types),
             L=local.configuration.box.L, bond_tags=topo["bond_tags"], bond_typeid=topo["bond_typeid"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_distribute_snapshot_gloo(world, tmp_path):
    """Only rank 0 holds the snapshot: every rank receives exactly the particles its sub-box owns, field by field, in the
    snapshot's order, and the bond topology by tag (HOOMD: create_state_from_snapshot under MPI)."""
    import torch.multiprocessing as mp

    mp.spawn(_distribute_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    snap, cfg = _global_snapshot()
    g = snap.particles
    dec = dd.Decomposition(cfg["L"], world, 2.9)
    owner = dec.owner(g.position)
    seen = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        mine = np.flatnonzero(owner == r)
        assert int(d["n_global"][0]) == g.N and list(d["types"]) == ["A", "B"] and np.array_equal(d["L"], np.asarray(cfg["L"], dtype=np.float64))
        for name in ("position", "typeid", "orientation", "velocity", "mass", "moment_inertia", "angmom", "tag"):
            assert np.array_equal(d[name], getattr(g, name)[mine]), (r, name)
        assert np.array_equal(d["bond_tags"], g.tag.astype(np.int64)[snap.bonds.group.astype(np.int64)])
        assert np.array_equal(d["bond_typeid"], snap.bonds.typeid)
        seen += mine.size
    assert seen == g.N
