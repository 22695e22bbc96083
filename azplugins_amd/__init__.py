"""azplugins_amd -- MI355X-native force-compute hot path of azplugins.

``pair`` / ``bond`` mirror ``hoomd.azplugins.pair`` / ``hoomd.azplugins.bond``
(class names, parameters, modes). The compute path is libazp.so (hand-written
HIP for gfx950, C ABI in ``include/azp.h``); there is no CPU fallback.
"""

from . import _lib, bond, external, nlist, pair, sorter, synthetic, tune
from ._lib import AzpError
from .simulation import All, ConstantVolume, Integrator, Simulation
from .sorter import ParticleSorter
from .state import (Box, Snapshot, State, bonded_two_particle_snapshot, lattice_snapshot, two_particle_snapshot)

__version__ = "0.1.0"

__all__ = ["All", "AzpError", "Box", "ParticleSorter", "ConstantVolume", "Integrator", "Simulation", "Snapshot", "State", "bond", "external", "nlist",
           "pair", "synthetic", "tune", "two_particle_snapshot", "bonded_two_particle_snapshot", "lattice_snapshot"]
