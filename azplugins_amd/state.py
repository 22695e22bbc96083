"""Host snapshot and device state: the minimal stand-in for the HOOMD-blue
objects the reference's force classes are attached to (``hoomd.Snapshot``,
``hoomd.State`` / ``ParticleData``). Device memory is held in torch tensors
(plumbing only); the layouts are HOOMD's:

* ``pos``   (n_max, 4) float64: x, y, z, type index in the low 32 bits of w
* ``vel``   (n_max, 4) float64: vx, vy, vz, mass
* ``orientation`` (n_max, 4) float64 quaternion, scalar part first
* ``tag``   (n_max,) uint32 (stored as int32 bit pattern)
* ``angmom`` (n_max, 4) float64 angular-momentum quaternion, ``inertia`` (n_max, 3) principal moments
"""

import numpy as np

from . import _lib
from .synthetic import pos4 as _pos4


class Box:
    """Triclinic periodic box centred on the origin (HOOMD ``BoxDim``)."""

    def __init__(self, Lx, Ly=None, Lz=None, xy=0.0, xz=0.0, yz=0.0, periodic=(True, True, True)):
        self.Lx = float(Lx)
        self.Ly = float(Lx if Ly is None else Ly)
        self.Lz = float(Lx if Lz is None else Lz)
        self.xy, self.xz, self.yz = float(xy), float(xz), float(yz)
        self.periodic = tuple(bool(p) for p in periodic)

    @classmethod
    def cube(cls, L):
        return cls(L, L, L)

    @classmethod
    def from_box(cls, box):
        if isinstance(box, Box):
            return box
        box = list(box)
        if len(box) == 3:
            return cls(*box)
        return cls(*box[:6])

    @property
    def L(self):
        return np.array([self.Lx, self.Ly, self.Lz])

    @property
    def is_triclinic(self):
        return self.xy != 0.0 or self.xz != 0.0 or self.yz != 0.0

    def to_c(self):
        key = (self.Lx, self.Ly, self.Lz, self.xy, self.xz, self.yz, self.periodic)
        if getattr(self, "_c_key", None) != key:  # (called for every launch: build the struct once per box)
            self._c = _lib.make_box(key[:3], key[3:6], [int(p) for p in self.periodic])
            self._c_key = key
        return self._c

    def __repr__(self):
        return "Box(Lx=%g, Ly=%g, Lz=%g, xy=%g, xz=%g, yz=%g)" % (self.Lx, self.Ly, self.Lz, self.xy, self.xz, self.yz)


class _Particles:
    def __init__(self):
        self._N = 0
        self.types = ["A"]
        self._alloc(0)

    def _alloc(self, n):
        self.position = np.zeros((n, 3))
        self.typeid = np.zeros(n, dtype=np.uint32)
        self.orientation = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (n, 1))
        self.velocity = np.zeros((n, 3))
        self.mass = np.ones(n)
        self.moment_inertia = np.zeros((n, 3))
        self.angmom = np.zeros((n, 4))
        self.tag = np.arange(n, dtype=np.uint32)

    @property
    def N(self):
        return self._N

    @N.setter
    def N(self, n):
        self._N = int(n)
        self._alloc(self._N)


class _Bonds:
    def __init__(self):
        self._N = 0
        self.types = []
        self.group = np.zeros((0, 2), dtype=np.uint32)
        self.typeid = np.zeros(0, dtype=np.uint32)

    @property
    def N(self):
        return self._N

    @N.setter
    def N(self, n):
        self._N = int(n)
        self.group = np.zeros((self._N, 2), dtype=np.uint32)
        self.typeid = np.zeros(self._N, dtype=np.uint32)


class _Configuration:
    def __init__(self):
        self.box = Box(1.0)
        self.dimensions = 3


class Snapshot:
    """Host-side system description with ``hoomd.Snapshot``'s attribute names
    (``particles.N/position/typeid/types/orientation/velocity/mass``,
    ``bonds.N/group/typeid/types``, ``configuration.box``)."""

    def __init__(self):
        self.particles = _Particles()
        self.bonds = _Bonds()
        self.configuration = _Configuration()

    @classmethod
    def from_arrays(cls, xyz, box, typeid=None, types=("A",), orientation=None, velocity=None, tag=None, bonds=None,
                    bond_typeid=None, bond_types=("A-A",), moment_inertia=None, angmom=None):
        s = cls()
        xyz = np.asarray(xyz, dtype=np.float64)
        s.particles.N = xyz.shape[0]
        s.particles.position[:] = xyz
        s.particles.types = list(types)
        if typeid is not None:
            s.particles.typeid[:] = typeid
        if orientation is not None:
            s.particles.orientation[:] = orientation
        if velocity is not None:
            s.particles.velocity[:] = velocity
        if tag is not None:
            s.particles.tag[:] = tag
        if moment_inertia is not None:
            s.particles.moment_inertia[:] = moment_inertia
        if angmom is not None:
            s.particles.angmom[:] = angmom
        s.configuration.box = Box.from_box(box)
        if bonds is not None:
            bonds = np.asarray(bonds, dtype=np.uint32).reshape(-1, 2)
            s.bonds.N = bonds.shape[0]
            s.bonds.group[:] = bonds
            s.bonds.types = list(bond_types)
            if bond_typeid is not None:
                s.bonds.typeid[:] = bond_typeid
        return s


def two_particle_snapshot(particle_types=("A",), d=1.0, L=20.0):
    """HOOMD conftest's ``two_particle_snapshot_factory`` restated: two particles
    at (-d/2, 0, 0) and (+d/2, 0, 0) in a cubic box (used by every 2-particle
    reference test, e.g. src/pytest/test_pair.py:319-321)."""
    s = Snapshot()
    s.particles.N = 2
    s.particles.types = list(particle_types)
    s.particles.position[:] = [[-d / 2.0, 0.0, 0.0], [d / 2.0, 0.0, 0.0]]
    s.configuration.box = Box.cube(L)
    return s


def bonded_two_particle_snapshot(bond_types=None, **kwargs):
    """src/conftest.py:10-24 restated: one bond [0, 1] of type "A-A"."""
    s = two_particle_snapshot(**kwargs)
    s.bonds.N = 1
    s.bonds.types = list(bond_types) if bond_types is not None else ["A-A"]
    s.bonds.group[0] = [0, 1]
    return s


def lattice_snapshot(particle_types=("A",), n=10, a=0.6):
    """HOOMD conftest's ``lattice_snapshot_factory`` restated: n^3 simple cubic,
    box n*a (src/pytest/test_pair_dpd.py:15)."""
    s = Snapshot()
    s.particles.N = n**3
    s.particles.types = list(particle_types)
    g = (np.arange(n) + 0.5) * a - 0.5 * n * a
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    s.particles.position[:] = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    s.configuration.box = Box.cube(n * a)
    return s


def localize_bonds(tag, n_local, bond_tags, bond_typeid):
    """The bonds of a global topology (pairs of particle tags) that one rank of a decomposed run evaluates, as index
    pairs over its rows (``tag``: the tags of its local rows [0, n_local) followed by its ghost rows): every bond
    with at least one LOCAL member. Returns (bond_group uint32 (n, 2), typeid)."""
    tag = np.asarray(tag, dtype=np.int64)
    bond_tags = np.asarray(bond_tags, dtype=np.int64).reshape(-1, 2)
    n_glob = int(max(bond_tags.max() + 1 if bond_tags.size else 0, tag.max() + 1 if tag.size else 0))
    rtag = np.full(n_glob + 1, -1, dtype=np.int64)
    # (a particle can sit on a rank more than once: as a local and as its own periodic ghost; the lowest row wins
    # -- local before ghost -- and bonds are evaluated with the minimum image)
    rtag[tag[::-1]] = np.arange(tag.size - 1, -1, -1)
    ia, ib = rtag[bond_tags[:, 0]], rtag[bond_tags[:, 1]]
    mine = ((ia >= 0) & (ia < n_local)) | ((ib >= 0) & (ib < n_local))
    if np.any(mine & ((ia < 0) | (ib < 0))):
        raise _lib.AzpError("a bonded partner of a local particle is neither local nor a ghost on this rank: the ghost "
                            "shell (r_cut + buffer) is narrower than a bond")
    group = np.stack([ia[mine], ib[mine]], axis=1).astype(np.uint32).reshape(-1, 2)
    return group, np.asarray(bond_typeid, dtype=np.uint32)[mine]


class State:
    """Device-resident particle data (HOOMD ``ParticleData`` + ``BondData``)."""

    def __init__(self, snapshot, device, n_local=None):
        """``n_local``: in a domain-decomposed run the snapshot lists this rank's
        local particles first and its ghosts after them; forces are computed
        for the first ``n_local`` only."""
        import torch

        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.AzpError("azplugins_amd states live on an MI355X (device 'cuda:N'); there is no CPU path")
        p = snapshot.particles
        self.N = p.N if n_local is None else int(n_local)
        self.n_ghost = p.N - self.N
        self.types = list(p.types)
        self.box = Box.from_box(snapshot.configuration.box)
        f64 = torch.float64
        self.pos = torch.from_numpy(_pos4(p.position, p.typeid)).to(self.device)
        vel = np.zeros((p.N, 4))
        vel[:, :3] = p.velocity
        vel[:, 3] = p.mass
        self.vel = torch.from_numpy(vel).to(self.device)
        self.orientation = torch.from_numpy(np.ascontiguousarray(p.orientation, dtype=np.float64)).to(self.device)
        self.tag = torch.from_numpy(np.ascontiguousarray(p.tag, dtype=np.uint32).view(np.int32)).to(self.device)
        # rotational degrees of freedom (HOOMD ParticleData: angmom Scalar4, moment_inertia Scalar3)
        self.angmom = torch.from_numpy(np.ascontiguousarray(p.angmom, dtype=np.float64)).to(self.device)
        self.inertia = torch.from_numpy(np.ascontiguousarray(p.moment_inertia, dtype=np.float64)).to(self.device)
        self.net_force = torch.zeros((self.N, 4), dtype=f64, device=self.device)
        self.image = torch.zeros((p.N, 3), dtype=torch.int32, device=self.device)
        b = snapshot.bonds
        self.bond_types = list(b.types)
        self.bond_group = np.ascontiguousarray(b.group, dtype=np.uint32).reshape(-1, 2)
        self.bond_typeid = np.ascontiguousarray(b.typeid, dtype=np.uint32)
        self._bond_table = None
        # decomposed runs (set_global_bonds): the whole topology by particle TAG, replicated on every rank; the
        # index-based table above is rebuilt from it whenever particles migrate (relocalize_bonds)
        self.bond_tags = None
        self.bond_tags_typeid = None
        self.position_generation = 0  # bumped whenever positions change

    @property
    def n_max(self):
        return self.N + self.n_ghost

    # The bond members live in two places: a host array (HOOMD's snapshot layout, uint32 (n, 2)) and a device tensor
    # (int64 (n, 2)). Whoever writes one invalidates the other; the copy across happens when the other one is asked for --
    # the particle sorter re-indexes a million bonds on the device and the bond table is built there, so inside a run
    # nothing travels (a sort used to cost two 8 MB transfers and two numpy passes, ~20 ms of idle GPU).
    @property
    def bond_group(self):
        if self._bond_group_host is None:
            import torch

            self._bond_group_host = self._bond_group_dev.to(torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 2)
        return self._bond_group_host

    @bond_group.setter
    def bond_group(self, group):
        self._bond_group_host = np.ascontiguousarray(group, dtype=np.uint32).reshape(-1, 2)
        self._bond_group_dev = None

    @property
    def n_bonds(self):
        g = self._bond_group_host if self._bond_group_host is not None else self._bond_group_dev
        return int(g.shape[0])

    def bond_group_device(self):
        """The bond members as an int64 (n, 2) tensor on the state's device."""
        if self._bond_group_dev is None:
            import torch

            self._bond_group_dev = torch.from_numpy(self._bond_group_host.astype(np.int64)).to(self.device).reshape(-1, 2)
        return self._bond_group_dev

    def set_bond_group_device(self, group):
        self._bond_group_dev = group.reshape(-1, 2)
        self._bond_group_host = None

    @property
    def typeid_host(self):
        return self.pos[: self.N, 3].cpu().numpy().view(np.int64).astype(np.int64) & 0xFFFFFFFF

    def set_global_bonds(self, bond_tags, bond_typeid, bond_types):
        """Domain-decomposed runs: the bonds of the WHOLE system as pairs of particle tags (replicated on every
        rank; HOOMD's BondData migrates its groups with their members, a static topology of 12 B per bond can simply
        be everywhere). ``relocalize_bonds`` turns it into this rank's index-based table."""
        self.bond_tags = np.ascontiguousarray(bond_tags, dtype=np.int64).reshape(-1, 2)
        self.bond_tags_typeid = np.ascontiguousarray(bond_typeid, dtype=np.uint32)
        self.bond_types = list(bond_types)
        self.relocalize_bonds()

    def relocalize_bonds(self):
        """(Re)build ``bond_group`` -- index pairs over local + ghost rows -- from the tags now on this rank: every
        bond with at least one LOCAL member (a bond is evaluated by the rank(s) owning a member); its
        partner must be on the rank, as a local or a ghost (the ghost shell is at least one bond length wide)."""
        tag = self.tag[: self.n_max].cpu().numpy().view(np.uint32).astype(np.int64)
        self.bond_group, self.bond_typeid = localize_bonds(tag, self.N, self.bond_tags, self.bond_tags_typeid)
        self._bond_table = None

    def bond_table(self):
        """HOOMD's per-particle GPU bond table (``BondData::getGPUTable``):
        column-major entries (partner index, bond type), the particle's position
        in the bond, and the per-particle bond count."""
        import torch

        if self._bond_table is None:
            # built on the device (one stable sort of the 2 x n_bonds member entries): the table is rebuilt whenever
            # the particle sorter re-indexes the particles or a decomposed run migrates them, and numpy's
            # scatter-add took 0.14 s for the 10^6 bonds of C3
            N = self.N
            dev = self.device
            g = self.bond_group_device()
            bt = torch.from_numpy(self.bond_typeid.astype(np.int64)).to(dev)
            nbnd = g.shape[0]
            # one entry per (bond, member); members that are ghosts here get their rows on their owner's rank
            member = torch.cat([g[:, 0], g[:, 1]]) if nbnd else torch.zeros(0, dtype=torch.int64, device=dev)
            partner = torch.cat([g[:, 1], g[:, 0]]) if nbnd else member
            which = torch.cat([torch.zeros(nbnd, dtype=torch.int64, device=dev), torch.ones(nbnd, dtype=torch.int64, device=dev)])
            btype = torch.cat([bt, bt]) if nbnd else member
            keep = member < N
            member, partner, which, btype = member[keep], partner[keep], which[keep], btype[keep]
            nb = torch.bincount(member, minlength=N)[:N] if member.numel() else torch.zeros(N, dtype=torch.int64, device=dev)
            width = max(int(nb.max().item()) if (N and member.numel()) else 0, 1)
            table = torch.zeros((width, N, 2), dtype=torch.int32, device=dev)
            bpos = torch.zeros((width, N), dtype=torch.int32, device=dev)
            if member.numel():
                order = torch.sort(member, stable=True).indices  # bond order inside a particle: slot 0 entries first, as HOOMD
                m = member[order]
                start = torch.cumsum(nb, 0) - nb
                slot = torch.arange(m.numel(), device=dev) - start[m]
                table[slot, m, 0] = partner[order].to(torch.int32)
                table[slot, m, 1] = btype[order].to(torch.int32)
                bpos[slot, m] = which[order].to(torch.int32)
            self._bond_table = dict(table=table, bond_pos=bpos, n_bonds=nb.to(torch.int32), pitch=N, width=width)
        return self._bond_table

    def exclusion_table(self):
        """Bonded partners as neighbor-list exclusions (HOOMD's default
        ``exclusions=('bond',)``): (n_excl int32[N], excl int32[width, N])."""
        t = self.bond_table()
        return t["n_bonds"], t["table"][:, :, 0].contiguous(), t["pitch"]
