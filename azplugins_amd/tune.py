"""Tuners around the hot path (reduced ``hoomd.md.tune``).

``NeighborListBuffer`` -- HOOMD's ``hoomd.md.tune.NeighborListBuffer`` restated in reduced form: the neighbor list's
``buffer`` (r_buff) trades fewer, dearer list rebuilds against longer rows in every force call, and the best value
depends on the potential, the density, the temperature and the time step. The tuner runs short stretches of the
simulation at a grid of buffer values, measures the step rate of each, and leaves the best one set. With the tile
kernels' Verlet-buffer shells (rows stop at the shells a step's displacement can have crossed) the optimum sits at
wider buffers than with a plain list: north-star liquid 0.4 -> 0.7, 0.43 -> 0.40 ms per step.
"""

import time


class NeighborListBuffer:
    def __init__(self, nlist, candidates=(0.3, 0.4, 0.5, 0.6, 0.7, 0.8), steps=60):
        """``nlist``: the list to tune; ``candidates``: buffer values to try (each for ``steps`` time steps, after one
        untimed rebuild cycle)."""
        self.nlist = nlist
        self.candidates = tuple(float(c) for c in candidates)
        self.steps = int(steps)
        self.results = {}
        self.best = None

    def tune(self, sim):
        """Advance ``sim`` by len(candidates) * 2 * steps time steps in total; returns the chosen buffer."""
        import torch

        nl = self.nlist
        self.results = {}
        def set_buffer(b):
            nl.buffer = b
            nl._consumers_changed()     # the list on hand was built for another r_cut + buffer: rebuild
            for f in nl._consumers:     # (tables that depend on the buffer: the tile plan's inner-radius hint)
                f._mark_dirty()

        for b in self.candidates:
            set_buffer(b)
            sim.run(self.steps)         # untimed: first rebuilds at this width, plan capacities learned
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sim.run(self.steps)
            torch.cuda.synchronize()
            self.results[b] = self.steps / (time.perf_counter() - t0)
        self.best = max(self.results, key=self.results.get)
        set_buffer(self.best)
        return self.best


__all__ = ["NeighborListBuffer"]
