"""Per-workgroup timeline of the tiled PerturbedLJ kernel (debug build with
-DAZP_TIMELINE, see pair_tiled.hpp): where does the launch lose time -- staging,
imbalance between the waves of a tile, the tail of the launch?

    make -C azplugins_amd/csrc timeline
    AZP_LIB_PATH=tools/libazp_timeline.so python tools/timeline.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import azplugins_amd as azp
from azplugins_amd import _lib, synthetic as syn

cfg = syn.config_north_star(64)
sim = azp.Simulation(device="cuda:0", seed=1)
sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
nl = azp.nlist.Cell(buffer=0.4)
pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=3.0)
pot.params[("A", "A")] = cfg["params"]
sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
sim.run(0)
for _ in range(20):
    pot.compute(0)
torch.cuda.synchronize()
pot.compute(0)
torch.cuda.synchronize()
n_tiles = pot.plan_info["n_tiles"]
raw = np.zeros(n_tiles * 32, dtype=np.uint64)
lib = C.CDLL(os.environ["AZP_LIB_PATH"])
lib.azp_debug_timeline.argtypes = [C.c_void_p, C.c_size_t]
assert _lib.lib().azp_debug_timeline(raw.ctypes.data_as(C.c_void_p), raw.size) == 0
t = raw.reshape(n_tiles, 4, 8)
t0 = t[:, :, 0].astype(np.int64); t1 = t[:, :, 1].astype(np.int64); t2 = t[:, :, 2].astype(np.int64)
hw = t[:, 0, 3]
origin = t0.min()
us = 1e-2  # 100 MHz ticks -> microseconds
span = (t2.max() - origin) * us
print("tiles %d, launch span %.1f us" % (n_tiles, span))
ta, tb, tc, td = (t[:, :, k].astype(np.int64) for k in (4, 5, 6, 7))
for name, x in (("entry->coeffs", (ta - t0)[:, :3]), ("index loads", (tb - ta)[:, :3]), ("position loads", tc - tb), ("image+LDS+own", td - tc), ("barrier", t1 - td)):
    x = x.reshape(-1) * us
    print("  %-16s mean %.2f us  p10 %.2f  p90 %.2f" % (name, x.mean(), *np.percentile(x, [10, 90])))
clk = t[:, 3, 4].astype(np.int64) / ((t2 - t0)[:, 3] * us)  # shader ticks per us = MHz
print("  shader clock during the launch: mean %.0f MHz (p10 %.0f, p90 %.0f)" % (clk.mean(), *np.percentile(clk, [10, 90])))
first_round = (t0[:, 0] - origin) * us < 2.0
print("  stage phase of tiles started in the first 2 us: %.2f us (n=%d); later tiles: %.2f us" % (
    ((t1 - t0)[:, 0] * us)[first_round].mean(), first_round.sum(), ((t1 - t0)[:, 0] * us)[~first_round].mean()))
stage = (t1 - t0)[:, 0] * us
loop = (t2 - t1) * us
wg_end = t2.max(axis=1); wg_start = t0.min(axis=1)
dur = (wg_end - wg_start) * us
print("stage phase  : mean %.2f us  p10 %.2f  p90 %.2f  max %.2f" % (stage.mean(), *np.percentile(stage, [10, 90]), stage.max()))
print("wave loop    : mean %.2f us  p10 %.2f  p90 %.2f  max %.2f" % (loop.mean(), *np.percentile(loop, [10, 90]), loop.max()))
print("tile (WG)    : mean %.2f us  p10 %.2f  p90 %.2f  max %.2f" % (dur.mean(), *np.percentile(dur, [10, 90]), dur.max()))
spread = (t2.max(axis=1) - t2.min(axis=1)) * us
print("wave finish spread inside a tile: mean %.2f us, p90 %.2f" % (spread.mean(), np.percentile(spread, 90)))
# wave-slot occupancy over time: resident waves (entry -> loop done) / (1024 SIMD x 4)
edges = np.linspace(0, span, 41)
starts = (t0 - origin).reshape(-1) * us; ends = (t2 - origin).reshape(-1) * us
occ = []
for a, b in zip(edges[:-1], edges[1:]):
    overlap = np.clip(np.minimum(ends, b) - np.maximum(starts, a), 0, None).sum() / (b - a)
    occ.append(overlap / 4096.0)
print("resident waves / 4096 slots per %.1f us bin:" % (edges[1] - edges[0]))
print(" ".join("%.2f" % o for o in occ))
print("mean occupancy over the launch: %.3f" % (np.clip(ends - starts, 0, None).sum() / span / 4096.0))
# in-loop fraction (waves doing pair work)
inloop = (loop.reshape(-1)).sum() / span / 4096.0
print("mean fraction of wave slots inside the pair loop: %.3f" % inloop)
xcc = (t[:, 0, 3] >> np.uint64(32)).astype(np.int64) & 0xF
for x in range(8):
    m = xcc == x
    if m.any():
        print("xcc %d: %4d tiles, last finish %.1f us" % (x, m.sum(), (wg_end[m].max() - origin) * us))

# ---- would a longest-first tile order shorten the tail? ----
nn = nl.n_neigh.cpu().numpy().astype(np.int64)
K = np.ceil(nn.reshape(n_tiles, 4, 64).max(axis=2) / 8.0)  # chunks per slice
work = K.sum(axis=1)
print("chunks per tile: mean %.1f  min %d  max %d; corr(duration, chunks) = %.2f" % (
    work.mean(), work.min(), work.max(), np.corrcoef(dur, work)[0, 1]))
order_start = np.argsort(wg_start)
print("duration by launch order (quarters): " + " ".join("%.1f" % dur[order_start[q * 1024:(q + 1) * 1024]].mean() for q in range(4)))


def makespan(durs, slots=128):
    import heapq
    h = [0.0] * slots
    heapq.heapify(h)
    for d in durs:
        heapq.heappush(h, heapq.heappop(h) + d)
    return max(h)


for x in range(8):
    m = np.where(xcc == x)[0]
    if m.size == 0:
        continue
    d = dur[m]
    print("xcc %d: list-scheduling makespan in launch order %.1f us, longest-first %.1f us, ideal %.1f us" % (
        x, makespan(d[np.argsort(wg_start[m])]), makespan(np.sort(d)[::-1]), d.sum() / 128))
    if x >= 1:
        break
