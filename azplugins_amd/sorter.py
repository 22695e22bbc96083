"""Particle sorter ("+ particle SFC sort"): the role of
``hoomd.update.ParticleSorter``. Reorders the local particles along a Hilbert
curve through small cells so that any 256 consecutive particles form a compact tile -- the order the
tile plan (``pair_plan.hpp``) and the neighbor-list build rely on. Tags, images,
velocities, orientations travel with the particles; bonds are re-indexed.

Keys and the permutation are computed with torch ops on the device (this runs
every few hundred steps, not on the per-step path)."""

import numpy as np

from . import _lib


class ParticleSorter:
    def __init__(self, trigger_period=500, block=4, particles_per_block=256, curve="hilbert"):
        """``curve``: "hilbert" (cells of ~4 particles along a Hilbert curve: any 256 consecutive particles
        form a compact blob) or "blocks" (row-major blocks of ``block``^3 cells, row-major inside)."""
        if curve not in ("hilbert", "blocks"):
            raise ValueError("curve must be 'hilbert' or 'blocks'")
        self.curve = curve
        self.trigger_period = int(trigger_period)
        self.block = int(block)
        self.particles_per_block = int(particles_per_block)
        self.num_sorts = 0

    def keys(self, state):
        """int32 sort key per local particle: (block index, position inside the block), one
        libazp kernel (a chain of a dozen torch element-wise ops paid 170-550 ms of lazy kernel
        loading on the first sort of a run)."""
        import ctypes as C

        import torch

        N = state.N
        volume = float(np.prod(state.box.L))
        # cell width such that block^3 cells hold about particles_per_block particles
        w = (self.particles_per_block * volume / max(N, 1)) ** (1.0 / 3.0) / self.block
        dims = (C.c_uint32 * 3)(*[max(int(np.floor(L / w)), 1) for L in state.box.L])
        keys = torch.empty(N, dtype=torch.int32, device=state.device)
        box = state.box.to_c()
        stream = torch.cuda.current_stream(state.device).cuda_stream
        _lib.check(_lib.lib().azp_sorter_keys(N, state.pos.data_ptr(), C.byref(box), dims, 0 if self.curve == "hilbert" else self.block,
                                              keys.data_ptr(), stream), "azp_sorter_keys")
        return keys

    def sort(self, sim):
        """Reorder the state of ``sim`` in place; returns the permutation applied
        (new index -> old index) as a device tensor."""
        import torch

        import time

        t0 = time.perf_counter()
        st = sim.state
        if st.n_ghost:
            raise _lib.AzpError("ParticleSorter: decomposed states keep their interior | boundary | ghost order")
        N = st.N
        order = torch.sort(self.keys(st), stable=True).indices
        for name in ("pos", "vel", "orientation", "tag", "image", "angmom", "inertia"):
            a = getattr(st, name)
            a[:N] = a[:N].index_select(0, order)
        if st.n_bonds:
            inv = torch.empty(N, dtype=torch.int64, device=st.device)
            inv[order] = torch.arange(N, dtype=torch.int64, device=st.device)
            # (the 10^6-entry lookup on the device, and the result stays there: State.bond_group fetches it when asked)
            st.set_bond_group_device(inv[st.bond_group_device()])
            st._bond_table = None
        st.position_generation += 1
        st.order_generation = getattr(st, "order_generation", 0) + 1
        self.num_sorts += 1
        self.host_seconds = getattr(self, "host_seconds", 0.0) + time.perf_counter() - t0  # launch-side time only
        return order
