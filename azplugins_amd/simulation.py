"""Minimal simulation driver: the slice of ``hoomd.Simulation`` /
``hoomd.md.Integrator`` the reference's tests exercise -- attach forces to a
state, ``run(0)`` to evaluate them, and a velocity-Verlet NVE step
(``hoomd.md.methods.ConstantVolume``) so thermostatting pair forces can be
validated as in src/pytest/test_pair_dpd.py.
"""

import warnings

import numpy as np

from . import _lib
from .state import State


class All:
    """``hoomd.filter.All``: every particle. The only filter the NVE kernels implement."""

    def __eq__(self, other):
        return isinstance(other, All)

    def __hash__(self):
        return hash("All")


class ConstantVolume:
    """NVE integration method (velocity Verlet) on all particles
    (``hoomd.md.methods.ConstantVolume(filter=hoomd.filter.All())`` without a
    thermostat, the dummy integrator of the reference's tests,
    src/pytest/test_pair.py:325-327). Any other filter is rejected: the kernels
    integrate all N particles."""

    def __init__(self, filter=None):
        if filter is not None and not isinstance(filter, All):
            raise _lib.AzpError("ConstantVolume: only filter=All() (or None) is supported, got %r" % (filter,))
        self.filter = All() if filter is None else filter


class Integrator:
    """``hoomd.md.Integrator`` reduced: ``dt``, ``forces``, ``methods``,
    ``integrate_rotational_dof`` (orientations and angular momenta of particles with a
    non-zero moment of inertia are integrated with the net torque, HOOMD's default False)."""

    def __init__(self, dt, forces=None, methods=None, integrate_rotational_dof=False):
        self.dt = float(dt)
        self.forces = list(forces) if forces is not None else []
        self.methods = list(methods) if methods is not None else []
        self.integrate_rotational_dof = bool(integrate_rotational_dof)


class _Operations:
    def __init__(self):
        self.integrator = None
        # HOOMD puts a ParticleSorter into sim.operations.tuners by default; so does this
        # (remove it from the list, or set trigger_period = 0, to keep the initial order)
        from .sorter import ParticleSorter

        self.tuners = [ParticleSorter(trigger_period=200)]


class Simulation:
    def __init__(self, device="cuda:0", seed=None):
        self.device = device
        self.seed = seed
        self.state = None
        self.timestep = 0
        self.operations = _Operations()
        self._attached = []
        self.domain = None  # azplugins_amd.domain.DeviceDomain of a decomposed run (attach_domain)

    def attach_domain(self, domain):
        """Domain-decomposed run: ``domain`` (a rebuilt ``DeviceDomain``) owns the particle
        arrays. Every step the ghost rows of the arrays the forces read are exchanged; when
        any rank's distance check asks for a neighbor-list rebuild, all ranks migrate their
        particles and re-select their ghosts first (HOOMD: Communicator::migrateParticles /
        exchangeGhosts ahead of NeighborList::compute)."""
        if self.state.n_bonds and self.state.bond_tags is None:
            raise _lib.AzpError("attach_domain: a bonded system needs its topology by tag (State.set_global_bonds) -- the "
                                "index-based bond table of a single-domain state does not survive a migration")
        # every per-particle array that the integrator or a force touches must migrate with the particles
        # (an array left behind keeps its old size and order while N changes under it)
        need = ["pos", "vel", "tag", "image"]
        integ = self.operations.integrator
        if integ is not None:
            for f in integ.forces:
                need += [n for n in getattr(f, "_halo_fields", ()) if n not in need]
            if getattr(integ, "integrate_rotational_dof", False):
                need += ["orientation", "angmom", "inertia"]
        missing = [n for n in need if n not in domain.names]
        if missing:
            raise _lib.AzpError("attach_domain: the domain does not carry %s (DeviceDomain(arrays=...) must hold every array "
                                "the integrator and the forces use)" % ", ".join(missing))
        self.domain = domain
        domain.attach_state(self.state)
        self.operations.tuners.clear()  # the domain keeps its own interior | boundary | ghost order

    def _halo_fields(self):
        names = ["pos"]
        for f in self.operations.integrator.forces:
            for n in getattr(f, "_halo_fields", ()):
                if n not in names and n in self.domain.names:
                    names.append(n)
        return names

    def _wire_domain(self):
        """Point the neighbor lists of the attached forces at the domain's collective hooks."""
        dom = self.domain

        def before_rebuild(state):
            dom.rebuild()
            dom.attach_state(state)

        for f in self.operations.integrator.forces:
            nl = getattr(f, "nlist", None)
            if nl is not None:
                nl.reduce_flag = dom.all_reduce_flag
                nl.before_rebuild = before_rebuild

    def _warn_if_seed_unset(self):
        if self.seed is None:
            warnings.warn("Simulation.seed is not set, using default seed=0", RuntimeWarning)
            self.seed = 0

    def create_state_from_snapshot(self, snapshot):
        self.state = State(snapshot, self.device)
        return self.state

    @property
    def dt(self):
        integ = self.operations.integrator
        return integ.dt if integ is not None else 0.0

    def _attach_all(self):
        integ = self.operations.integrator
        if integ is None:
            raise _lib.AzpError("Simulation.operations.integrator is not set")
        if self.state is None:
            raise _lib.AzpError("Simulation has no state; call create_state_from_snapshot first")
        for f in integ.forces:
            if f not in self._attached or f._state is not self.state:
                f._attach(self)
                if f not in self._attached:
                    self._attached.append(f)

    def _compute_forces(self):
        import torch

        st = self.state
        forces = self.operations.integrator.forces
        if self.domain is not None:
            # decomposed runs: a neighbor-list rebuild migrates particles (N and every index change). Bring every list
            # up to date BEFORE any force buffer is sized or summed, so that all forces of this step see one order
            seen = []
            for f in forces:
                nl = getattr(f, "nlist", None)
                if nl is not None and all(nl is not s for s in seen):
                    seen.append(nl)
                    nl.compute(st)
            if len(seen) > 1:
                raise _lib.AzpError("decomposed runs support one neighbor list (a second list's rebuild would migrate "
                                    "particles under the first)")
        if len(forces) == 1:
            # a single force: its own array is the net force (no 32 MB zero + add per step)
            forces[0].compute(self.timestep)
            st.net_force = forces[0].force_tensor
            return
        if st.net_force.shape[0] != st.N or any(st.net_force is f.force_tensor for f in forces):
            st.net_force = torch.zeros((st.N, 4), dtype=torch.float64, device=st.device)
        for f in forces:
            f.compute(self.timestep)
        # one pass over the forces' arrays (azp_sum_forces) instead of a zero + one read-modify-write per force
        import ctypes as C

        for k0 in range(0, len(forces), 7):
            grp = forces[k0:k0 + 7]
            ptrs = ([st.net_force.data_ptr()] if k0 else []) + [f.force_tensor.data_ptr() for f in grp]
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            _lib.check(_lib.lib().azp_sum_forces(st.N, len(ptrs), arr, st.net_force.data_ptr(), _lib.raw_stream(st.device)), "azp_sum_forces")

    def run(self, steps):
        import torch

        self._attach_all()
        integ = self.operations.integrator
        st = self.state
        if self.domain is not None:
            self._wire_domain()
        self._compute_forces()
        if steps == 0 or not integ.methods:
            return
        if len(integ.methods) != 1 or not isinstance(integ.methods[0], ConstantVolume):
            raise _lib.AzpError("Integrator.methods must hold exactly one ConstantVolume (all particles); got %r" % (integ.methods,))
        import ctypes as C

        a = _lib.NVEArgs()
        a.box = st.box.to_c()
        a.dt = integ.dt
        lib = _lib.lib()
        stream = _lib.raw_stream(st.device)

        rot = None
        if integ.integrate_rotational_dof:
            if self.domain is not None and not all(n in self.domain.names for n in ("orientation", "angmom", "inertia")):
                raise _lib.AzpError("integrate_rotational_dof in a decomposed run: the domain must carry orientation, angmom "
                                    "and inertia (they migrate with the particles)")
            rot = _lib.NVERotArgs()
            rot.dt = integ.dt

        def net_torque():
            forces = integ.forces
            if len(forces) == 1:
                return forces[0].torque_tensor
            t = forces[0].torque_tensor.clone()
            for f in forces[1:]:
                t += f.torque_tensor
            return t

        def point_at_state():
            # (the arrays are replaced when particles migrate between ranks or are re-sorted)
            a.d_pos = st.pos.data_ptr()
            a.d_vel = st.vel.data_ptr()
            a.d_net_force = st.net_force.data_ptr()
            a.d_image = st.image.data_ptr()
            a.N = st.N

        def rotational_step(one):
            torque = net_torque()  # kept alive until the kernel is queued
            rot.d_orientation = st.orientation.data_ptr()
            rot.d_angmom = st.angmom.data_ptr()
            rot.d_inertia = st.inertia.data_ptr()
            rot.d_net_torque = torque.data_ptr()
            rot.N = st.N
            fn = lib.azp_integrate_nve_rot_step_one if one else lib.azp_integrate_nve_rot_step_two
            _lib.check(fn(C.byref(rot), stream), "azp_integrate_nve_rot_step")
            return torque

        # (a bond force examines the "evaluator rejected its parameters" flag of step k when step k + 1 is queued, and
        # once more after the loop: no host round trip behind every launch)
        deferred = [f for f in integ.forces if hasattr(f, "defer_flag_check")]
        for f in deferred:
            f.defer_flag_check = True
        for k in range(steps):
            # velocity Verlet (libazp kernels): v += a dt/2, x += v dt, wrap | forces | v += a dt/2. Inside a run
            # nothing reads the velocities between step two of one step and step one of the next: they are one
            # kernel (same arithmetic, one pass over the arrays); the last step two comes after the loop
            point_at_state()
            if k == 0:
                _lib.check(lib.azp_integrate_nve_step_one(C.byref(a), stream), "azp_integrate_nve_step_one")
            else:
                _lib.check(lib.azp_integrate_nve_step_two_one(C.byref(a), stream), "azp_integrate_nve_step_two_one")
            if rot is not None:
                if k:
                    rotational_step(False)  # (step two of the previous step: the torques are still its own)
                rotational_step(True)
            if self.domain is not None:
                self.domain.exchange(self._halo_fields())  # ghost rows follow their owners' particles
            st.position_generation += 1
            self.timestep += 1
            # re-index the particles between the position update and the force
            # evaluation: every per-particle array that survives the step is permuted,
            # the forces are recomputed in the new order
            # (also when a tile plan could not be compiled from the cells because the members of some tile have drifted
            # apart -- a DPD fluid diffuses a cell width in ~200 steps: sorting now costs 0.5 ms, the list-based rebuilds
            # that would follow until the sorter's next period 2.5 ms each; at most once per 20 steps)
            lists = [f.nlist for f in integ.forces if getattr(f, "nlist", None) is not None]
            wanted = any(getattr(nl, "_sort_wanted", False) for nl in lists)
            for tuner in self.operations.tuners:
                due = tuner.trigger_period > 0 and self.timestep % tuner.trigger_period == 0
                on_demand = wanted and tuner.trigger_period > 0 and self.timestep - getattr(tuner, "_last_sort_step", -(10 ** 9)) >= 20
                if (due or on_demand) and st.n_ghost == 0:
                    tuner.sort(self)
                    tuner._last_sort_step = self.timestep
                    for nl in lists:
                        nl._sort_wanted = False
                        nl._fused_failures = 0
                        if getattr(nl, "_fused_auto_off", False):
                            nl.fused, nl._fused_auto_off = True, False
            self._compute_forces()
        point_at_state()
        _lib.check(lib.azp_integrate_nve_step_two(C.byref(a), stream), "azp_integrate_nve_step_two")
        if rot is not None:
            rotational_step(False)
        for f in deferred:
            f.defer_flag_check = False
            f.check_flags(wait=True)

    def kinetic_temperature(self):
        """Instantaneous kT = 2 KE / (3 N - 3) (HOOMD ThermodynamicQuantities)."""
        st = self.state
        v = st.vel[: st.N]
        ke = 0.5 * float((v[:, 3] * (v[:, :3] ** 2).sum(dim=1)).sum().item())
        return 2.0 * ke / (3 * st.N - 3)

    def rotational_kinetic_energy(self):
        """sum_k s_k^2 / (2 I_k) over the axes with I_k != 0, s = 1/2 conj(q) p the body-frame
        angular momentum (HOOMD ComputeThermo)."""
        import torch

        st = self.state
        q, p, I = st.orientation[: st.N], st.angmom[: st.N], st.inertia[: st.N]
        qs, qv = q[:, 0:1], q[:, 1:4]
        ps, pv = p[:, 0:1], p[:, 1:4]
        s_v = 0.5 * (qs * pv - ps * qv - torch.linalg.cross(qv, pv))  # vector part of conj(q) p / 2
        ke = torch.where(I != 0.0, s_v * s_v / torch.where(I != 0.0, I, torch.ones_like(I)), torch.zeros_like(I))
        return 0.5 * float(ke.sum().item())

    def thermalize_particle_momenta(self, kT, seed=12345):
        """Maxwell-Boltzmann velocities with zero total momentum."""
        import torch

        from .synthetic import normal

        st = self.state
        tag = np.arange(st.N, dtype=np.uint64)
        m = st.vel[: st.N, 3].cpu().numpy()
        v = np.stack([normal(seed, tag, c) for c in range(3)], axis=1) * np.sqrt(kT / m)[:, None]
        v -= (v * m[:, None]).sum(axis=0) / m.sum()
        st.vel[: st.N, :3] = torch.from_numpy(v).to(st.device)
