"""A domain-decomposed NVE run on the GPU path: two ranks (two processes sharing the one
GPU of the test box, gloo for the collectives; the product backend is RCCL) run
Simulation.run with a DeviceDomain attached -- per-step packed halo exchange, collective
rebuild decision, particle migration and ghost re-selection at every neighbor-list rebuild,
tile plan recompiled after each -- and must end where a single-domain run of the same
initial state ends (positions and velocities by tag). Covers the PerturbedLJ tile kernel
with its displacement bound on moving ghosts, and the DPD thermostat (both owners of a
cross-rank pair must draw the same random number)."""

import os
import socket

import numpy as np
import pytest

from azplugins_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _config(kind):
    if kind == "dpd":
        cfg = syn.config_dpd(8000)
        cfg["steps"] = 30
    else:
        cfg = syn.config_plj_sc(16)
        n = cfg["xyz"].shape[0]
        tag = np.arange(n, dtype=np.uint64)
        v = np.stack([syn.normal(51, tag, c) for c in range(3)], axis=1) * np.sqrt(1.5)
        cfg["vel"] = v - v.mean(axis=0)
        cfg["steps"] = 40
        cfg["dt"] = 0.005
    return cfg


def _potential(azp, kind, cfg, nl):
    if kind == "dpd":
        pot = azp.pair.DPDGeneralWeight(nlist=nl, kT=cfg["kT"], default_r_cut=cfg["r_cut"])
    else:
        pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
    pot.params[("A", "A")] = cfg["params"]
    return pot


def _worker(rank, world, port, out_dir, kind):
    import torch
    import torch.distributed as dist

    import azplugins_amd as azp
    from azplugins_amd import decomposition as dd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = _config(kind)
    dec = dd.Decomposition(cfg["L"], world, cfg["r_cut"] + cfg["r_buff"])
    sim, dom = dd.rank_simulation(cfg, dec, rank, "cuda:0", seed=cfg.get("seed", 1))
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    pot = _potential(azp, kind, cfg, nl)
    sim.operations.integrator = azp.Integrator(dt=cfg["dt"], forces=[pot], methods=[azp.ConstantVolume()])
    sim.run(cfg["steps"])
    torch.cuda.synchronize()
    st = sim.state
    N = st.N
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), tag=st.tag[:N].cpu().numpy().view(np.uint32), pos=st.pos[:N, :3].cpu().numpy(),
             vel=st.vel[:N, :3].cpu().numpy(), rebuilds=np.array([dom.num_rebuilds]), builds=np.array([nl.num_builds]),
             plan_valid=np.array([(pot.plan_info or {}).get("valid", -1)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["plj", "dpd"])
def test_decomposed_md_run_matches_single_domain(kind, tmp_path):
    import torch
    import torch.multiprocessing as mp

    import azplugins_amd as azp

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), kind), nprocs=world, join=True)
    cfg = _config(kind)
    n = cfg["xyz"].shape[0]
    sim = azp.Simulation(device="cuda:0", seed=cfg.get("seed", 1))
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], tag=cfg.get("tag"), velocity=cfg["vel"]))
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    pot = _potential(azp, kind, cfg, nl)
    sim.operations.integrator = azp.Integrator(dt=cfg["dt"], forces=[pot], methods=[azp.ConstantVolume()])
    sim.operations.tuners.clear()
    sim.run(cfg["steps"])
    torch.cuda.synchronize()
    tag = sim.state.tag.cpu().numpy().view(np.uint32).astype(np.int64)
    ref_pos = np.zeros((n, 3))
    ref_vel = np.zeros((n, 3))
    ref_pos[tag] = sim.state.pos[:, :3].cpu().numpy()
    ref_vel[tag] = sim.state.vel[:, :3].cpu().numpy()
    got_pos = np.full((n, 3), np.nan)
    got_vel = np.full((n, 3), np.nan)
    rebuilds = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        got_pos[d["tag"].astype(np.int64)] = d["pos"]
        got_vel[d["tag"].astype(np.int64)] = d["vel"]
        rebuilds.append(int(d["rebuilds"][0]))
        if kind == "plj":
            assert int(d["plan_valid"][0]) == 1  # the tile kernel ran on every rank
    assert min(rebuilds) >= 3, "the run must cross several neighbor-list rebuilds (with migration): %r" % rebuilds
    assert nl.num_builds >= 3
    L = np.asarray(cfg["L"])
    dx = got_pos - ref_pos
    dx -= L * np.round(dx / L)
    assert np.all(np.isfinite(got_pos)) and np.abs(dx).max() < 1e-9, np.abs(dx).max()
    assert np.abs(got_vel - ref_vel).max() < 1e-8 * max(1.0, np.abs(ref_vel).max())
