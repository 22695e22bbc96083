import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
import helpers as H
from azplugins_amd import synthetic as syn

def run(a, rl, tpp, name="Hertz"):
    pos, L, typeid = H.lattice_config(20, a, 0.1 * a, seed=31, ntypes=1)
    box = oracle.make_box(L)
    nl = oracle.build_nlist(pos, box, rl)
    params = np.array([oracle.pack_pair_params(name, dict(epsilon=1.0))])
    ref = oracle.pair_forces(name, pos, box, nl, params, rl - 0.3, nthreads=8)
    info = {}
    f = H.gpu_pair_forces(name, pos, (L,), nl, params, rl - 0.3, planned=True, plan_info=info, tpp=tpp, r_list_max=rl)
    nan = np.isnan(f).any(axis=1)
    bad = ~nan & (np.abs(f - ref).max(axis=1) > 1e-9 * np.abs(ref).max())
    print("a", a, "rl", rl, "tpp", tpp, "mean n %.1f" % nl[0].mean(), "nan", nan.sum(), "bad", bad.sum(), "stage", info["max_stage"], "cap", info["lds_slots"], "tile", info["tile_size"])

run(1.6, 2.0, 1)   # tpp1, small stage
run(1.6, 3.0, 1)
run(1.6, 3.5, 1)
run(1.1, 3.5, 2)   # tpp2, big stage
run(1.1, 3.5, 4)
run(1.1, 4.4, 4)
run(1.1, 2.0, 1)
