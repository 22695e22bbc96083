"""Domain decomposition + halo exchange on CPU (gloo, world_size 2, 4 and 8) --
the N > 1 path of bench.py, with the oracle standing in for the GPU kernels:
the union of per-rank forces (own particles, own + ghost positions, full lists)
must equal the single-domain forces."""

import os
import socket

import numpy as np
import pytest

from azplugins_amd import decomposition as dd
from azplugins_amd import synthetic as syn


def test_choose_grid():
    L = np.array([10.0, 10.0, 10.0])
    assert dd.choose_grid(1, L) == (1, 1, 1)
    assert sorted(dd.choose_grid(2, L)) == [1, 1, 2]
    assert sorted(dd.choose_grid(4, L)) == [1, 2, 2]
    assert dd.choose_grid(8, L) == (2, 2, 2)
    # C5: 76.8 x 76.8 x 153.6 box over 4 GPUs -> slabs along the long axis
    assert dd.choose_grid(4, np.array([76.8, 76.8, 153.6])) == (1, 1, 4)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_partition_is_exact_and_lists_match(world):
    cfg = syn.config_plj_sc(16)
    r_ghost = 3.4
    dec = dd.Decomposition(cfg["L"], world, r_ghost)
    doms = [dd.RankDomain(dec, r, cfg["xyz"]) for r in range(world)]
    owned = np.concatenate([d.local_gid for d in doms])
    assert np.array_equal(np.sort(owned), np.arange(cfg["xyz"].shape[0]))
    for r, d in enumerate(doms):
        assert d.recv_counts[r] == 0 and d.send_counts[r] == 0
        off = 0
        for q in range(world):
            # what r receives from q is exactly what q sends to r, in the same order
            got = d.ghost_gid[off: off + d.recv_counts[q]]
            off += d.recv_counts[q]
            qd = doms[q]
            soff = int(qd.send_counts[:r].sum())
            sent = qd.local_gid[qd.send_idx[soff: soff + qd.send_counts[r]]]
            assert np.array_equal(got, sent)
        # ghost shell is complete: every particle within r_ghost of a local one is present
        have = np.zeros(cfg["xyz"].shape[0], dtype=bool)
        have[d.all_gid] = True
        sample = d.local_gid[:: max(1, d.N_local // 50)]
        for i in sample:
            dx = cfg["xyz"] - cfg["xyz"][i]
            dx -= cfg["L"] * np.round(dx / cfg["L"])
            near = np.flatnonzero((dx * dx).sum(axis=1) <= r_ghost**2)
            assert have[near].all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _n_side(world):
    return 12 if world <= 4 else 16  # 2x2x2: keep every sub-box wider than two ghost shells


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = syn.config_plj_sc(_n_side(world))
    r_cut, r_buff = 2.5, 0.3
    dec = dd.Decomposition(cfg["L"], world, r_cut + r_buff)
    dom = dd.RankDomain(dec, rank, cfg["xyz"])
    gid = dom.all_gid

    def moved(g):  # displacement a "time step" later: deterministic in the global id
        return cfg["xyz"][g] + 0.05 * np.stack([np.sin(g * 0.37), np.cos(g * 0.11), np.sin(g * 0.23 + 1.0)], axis=1)

    pos = torch.from_numpy(syn.pos4(cfg["xyz"][gid]))
    vel = torch.zeros((gid.size, 4), dtype=torch.float64)
    # owners move their particles; ghosts are stale until the exchange
    pos[: dom.N_local, :3] = torch.from_numpy(moved(dom.local_gid))
    vel[: dom.N_local, 0] = torch.from_numpy(dom.local_gid.astype(np.float64))
    halo = dd.HaloExchange(dom, "cpu")
    halo.exchange(pos, vel)
    assert np.array_equal(pos[dom.N_local:, :3].numpy(), moved(dom.ghost_gid))
    assert np.array_equal(vel[dom.N_local:, 0].numpy(), dom.ghost_gid.astype(np.float64))
    # the split pack / transfer form used to overlap the exchange with interior forces
    pos2 = pos.clone()
    pos2[dom.N_local:] = 0.0
    halo.transfer(halo.pack(pos2))
    assert torch.equal(pos2, pos)

    # per-rank force compute (oracle as the kernel stand-in), global box min image
    box = oracle.make_box(cfg["L"])
    p = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos.numpy(), box, r_cut + r_buff, N=dom.N_local)
    # interior particles come first and list no ghost: their forces need no halo.
    # (the domain was classified on the unmoved snapshot; particles moved by <= 0.087
    # since, so test the ones that are still deeper than the list radius)
    n_neigh, head, nlist = nl
    assert np.all(dec.depth(cfg["xyz"][dom.local_gid[: dom.n_interior]], rank) >= dec.r_ghost)
    # (whole tiles of 256 only: up to 255 interior particles are computed with the boundary)
    deep = dec.depth(cfg["xyz"][dom.local_gid[dom.n_interior:]], rank) >= dec.r_ghost
    assert dom.n_interior % 256 == 0 and deep.sum() < 256 and not deep[deep.sum():].any()
    deep = np.flatnonzero(dec.depth(pos[: dom.N_local, :3].numpy(), rank) >= dec.r_ghost)
    for i in deep:
        assert (nlist[int(head[i]): int(head[i]) + int(n_neigh[i])] < dom.N_local).all()
    assert (nlist >= dom.N_local).any()
    f = oracle.pair_forces("PerturbedLennardJones", pos.numpy(), box, nl, p, r_cut, mode="shift", N=dom.N_local)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gid=dom.local_gid, force=f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_halo_exchange_and_forces_gloo(world, tmp_path, oracle):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cfg = syn.config_plj_sc(_n_side(world))
    g = np.arange(cfg["xyz"].shape[0])
    xyz = cfg["xyz"] + 0.05 * np.stack([np.sin(g * 0.37), np.cos(g * 0.11), np.sin(g * 0.23 + 1.0)], axis=1)
    pos = syn.pos4(xyz)
    box = oracle.make_box(cfg["L"])
    p = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box, 2.8)
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, nl, p, 2.5, mode="shift")
    f = np.zeros_like(f_ref)
    seen = np.zeros(len(f_ref), dtype=int)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        f[d["gid"]] = d["force"]
        seen[d["gid"]] += 1
    assert np.all(seen == 1)
    assert np.abs(f - f_ref).max() <= 1e-12 * np.abs(f_ref).max()
