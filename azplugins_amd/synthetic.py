"""Synthetic particle configurations for the parity tests and benchmarks.

Everything random comes from a counter-based hash (SplitMix64 finaliser of
``(seed, tag, component)``), so any rank can generate exactly its shard, and
the build container and the GPU box agree bit-for-bit. ``numpy.random`` is not
used. The same hash exists in C in the test oracle; ``tests/`` checks that the
two agree.

Configurations follow SURVEY.md section 8(d) / BASELINE.json ``configs``:

======  =====================================================================
C1      Hertz, N=4,096 uniform random, cubic L=16, r_cut=1.0
C2      PerturbedLJ, N=262,144 = 64^3 simple cubic (rho*=0.8) + jitter
NS      PerturbedLJ north star, N=1,048,576 = 64^3 FCC cells x 4 + jitter
C3      32,768 chains x 32 beads (DoubleWell bonds + PerturbedLJ)
C4      DPD, N=2,097,152 uniform random, rho=3.0, velocities ~ N(0, kT)
C5      TwoPatchMorse, N=524,288 = 64 x 64 x 128 simple cubic a=1.2, random quaternions
======  =====================================================================

Particles are emitted in a spatially sorted order (blocks of 4x4x4 lattice
sites / cells, then x-fastest inside a block), which is what HOOMD's SFC
particle sorter would give the force kernels.
"""

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)
_K = np.uint64(0x2545F4914F6CDD1D)


def hash64(seed, tag, comp):
    """SplitMix64-style hash of (seed, tag, comp); vectorised over ``tag``."""
    with np.errstate(over="ignore"):
        tag = np.asarray(tag, dtype=np.uint64)
        z = np.uint64(seed) * _G + tag * _M1 + np.uint64(comp) * _M2 + _K
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
        z = z + _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def u01(seed, tag, comp):
    """Uniform [0, 1) doubles with 53 random bits."""
    return (hash64(seed, tag, comp) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed, tag, comp):
    """Standard normal via Box-Muller on two hash streams (comp, comp + 64)."""
    u1 = 1.0 - u01(seed, tag, comp)  # (0, 1]
    u2 = u01(seed, tag, comp + 64)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def pos4(xyz, typeid=None):
    """(n,3) coordinates (+ integer type ids) -> HOOMD Scalar4 layout, the type
    index in the low 32 bits of w."""
    xyz = np.asarray(xyz, dtype=np.float64)
    out = np.zeros((xyz.shape[0], 4), dtype=np.float64)
    out[:, :3] = xyz
    if typeid is not None:
        w = np.zeros(xyz.shape[0], dtype=np.int64)
        w[:] = np.asarray(typeid, dtype=np.int64) & 0xFFFFFFFF
        out[:, 3] = w.view(np.float64)
    return out


def wrap(xyz, L):
    """Wrap coordinates into the centred box [-L/2, L/2)."""
    L = np.asarray(L, dtype=np.float64)
    out = xyz - L * np.floor(xyz / L + 0.5)
    # guard the upper edge against rounding
    out = np.where(out >= 0.5 * L, out - L, out)
    return out


def blocked_order(nx, ny, nz, b=4):
    """Permutation of lattice sites (x fastest) that visits them in b x b x b
    blocks: a cheap space-filling order."""
    ix, iy, iz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    ix, iy, iz = ix.ravel(), iy.ravel(), iz.ravel()
    key = (((iz // b) * ((ny + b - 1) // b) + (iy // b)) * ((nx + b - 1) // b) + (ix // b)).astype(np.int64)
    inner = ((iz % b) * b + (iy % b)) * b + (ix % b)
    order = np.lexsort((inner, key))
    return ix[order], iy[order], iz[order]


def _jitter(seed, n, amp):
    tag = np.arange(n, dtype=np.uint64)
    return np.stack([(2.0 * u01(seed, tag, c) - 1.0) * amp for c in range(3)], axis=1)


def simple_cubic(nx, ny, nz, a, jitter, seed):
    """Jittered simple-cubic lattice in a centred periodic box. Returns (xyz, L)."""
    ix, iy, iz = blocked_order(nx, ny, nz)
    L = np.array([nx * a, ny * a, nz * a])
    xyz = (np.stack([ix, iy, iz], axis=1) + 0.5) * a - 0.5 * L
    xyz = xyz + _jitter(seed, xyz.shape[0], jitter)
    return wrap(xyz, L), L


def fcc(ncell, a, jitter, seed):
    """Jittered FCC lattice (4 sites per cubic cell); ``ncell`` is an int or a
    (nx, ny, nz) tuple. Returns (xyz, L)."""
    nx, ny, nz = (ncell, ncell, ncell) if np.isscalar(ncell) else ncell
    ix, iy, iz = blocked_order(nx, ny, nz)
    basis = np.array([[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.5, 0.0, 0.5], [0.0, 0.5, 0.5]])
    cells = np.stack([ix, iy, iz], axis=1).astype(np.float64)
    xyz = ((cells[:, None, :] + basis[None, :, :] + 0.25) * a).reshape(-1, 3)
    L = np.array([nx * a, ny * a, nz * a])
    xyz = xyz - 0.5 * L + _jitter(seed, xyz.shape[0], jitter)
    return wrap(xyz, L), L


def uniform_random(n, L, seed, sort_cell=None):
    """Uniform random positions; optionally sorted by cells of width sort_cell."""
    L = np.asarray([L] * 3 if np.isscalar(L) else L, dtype=np.float64)
    tag = np.arange(n, dtype=np.uint64)
    xyz = np.stack([(u01(seed, tag, c) - 0.5) * L[c] for c in range(3)], axis=1)
    order = np.arange(n)
    if sort_cell:
        dim = np.maximum((L / sort_cell).astype(np.int64), 1)
        c = np.minimum(((xyz + 0.5 * L) / (L / dim)).astype(np.int64), dim - 1)
        b = 2
        key = ((c[:, 2] // b) * ((dim[1] + b - 1) // b) + (c[:, 1] // b)) * ((dim[0] + b - 1) // b) + (c[:, 0] // b)
        inner = ((c[:, 2] % b) * b + (c[:, 1] % b)) * b + (c[:, 0] % b)
        order = np.lexsort((inner, key))
        xyz = xyz[order]
    return xyz, L, order


def random_quaternions(n, seed):
    tag = np.arange(n, dtype=np.uint64)
    q = np.stack([normal(seed, tag, c) for c in range(4)], axis=1)
    return q / np.linalg.norm(q, axis=1, keepdims=True)


# ---------------------------------------------------------------------------
# named configurations
# ---------------------------------------------------------------------------
def config_c1():
    """Hertz: N=4,096 random soft spheres, 1 type, r_cut=1.0 (BASELINE.json configs[0])."""
    xyz, L, _ = uniform_random(4096, 16.0, seed=1, sort_cell=1.4)
    return dict(name="C1", xyz=xyz, L=L, potential="Hertz", params=dict(epsilon=1.0), r_cut=1.0, r_buff=0.4)


def config_plj_sc(n_side=64, seed=2):
    """PerturbedLJ on a jittered simple-cubic lattice at rho*=0.8 (configs[1] when n_side=64)."""
    a = 0.8 ** (-1.0 / 3.0)
    xyz, L = simple_cubic(n_side, n_side, n_side, a, 0.1 * a, seed)
    return dict(name="C2" if n_side == 64 else "C2-%d" % n_side, xyz=xyz, L=L, potential="PerturbedLennardJones",
                params=dict(epsilon=1.0, sigma=1.0, attraction_scale_factor=0.5), r_cut=3.0, r_buff=0.4)


def config_north_star(ncell=64, seed=3):
    """PerturbedLJ north star: FCC ncell^3 x 4 at rho*=0.8 (N=1,048,576 for ncell=64)."""
    a = (4.0 / 0.8) ** (1.0 / 3.0)
    xyz, L = fcc(ncell, a, 0.05 * a, seed)
    name = "NS" if ncell == 64 else ("NS-%d" % ncell if np.isscalar(ncell) else "NS-%dx%dx%d" % tuple(ncell))
    return dict(name=name, xyz=xyz, L=L, potential="PerturbedLennardJones",
                params=dict(epsilon=1.0, sigma=1.0, attraction_scale_factor=0.5), r_cut=3.0, r_buff=0.4)


def config_chains(nx=128, ny=128, nz=64, chain_len=32, seed=4):
    """Linear chains laid along x on a simple-cubic lattice (configs[2] at the
    default size: 32,768 chains of 32 beads), particles in blocked spatial order."""
    a = 0.8 ** (-1.0 / 3.0)
    assert nx % chain_len == 0
    L = np.array([nx * a, ny * a, nz * a])
    ix, iy, iz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    # x fastest
    order = np.lexsort((ix.ravel(), iy.ravel(), iz.ravel()))
    ix, iy, iz = ix.ravel()[order], iy.ravel()[order], iz.ravel()[order]
    xyz = (np.stack([ix, iy, iz], axis=1) + 0.5) * a - 0.5 * L
    n = xyz.shape[0]
    xyz = wrap(xyz + _jitter(seed, n, 0.05 * a), L)
    idx = np.arange(n)
    first = idx[(ix % chain_len) != (chain_len - 1)]
    bonds = np.stack([first, first + 1], axis=1)
    # HOOMD's SFC sorter reorders particles in memory regardless of chain
    # membership (bonds follow through the tag lookup): emit the particles in the
    # same blocked spatial order as the other configurations and remap the bonds
    key = (((iz // 4) * ((ny + 3) // 4) + (iy // 4)) * ((nx + 3) // 4) + (ix // 4)).astype(np.int64)
    inner = ((iz % 4) * 4 + (iy % 4)) * 4 + (ix % 4)
    perm = np.lexsort((inner, key))          # new position -> old index
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    xyz = xyz[perm]
    bonds = inv[bonds].astype(np.uint32)
    return dict(name="C3", xyz=xyz, L=L, bonds=bonds, potential="PerturbedLennardJones",
                params=dict(epsilon=1.0, sigma=1.0, attraction_scale_factor=0.5), r_cut=3.0, r_buff=0.4,
                bond_potential="DoubleWell", bond_params=dict(r_0=1.0, r_1=1.5, U_1=1.0, U_tilt=0.5))


def config_dpd(n=2097152, rho=3.0, kT=1.0, seed_pos=5, seed_vel=6):
    """DPD fluid: uniform random positions, Maxwell-Boltzmann velocities (configs[3])."""
    L = (n / rho) ** (1.0 / 3.0)
    xyz, Lv, order = uniform_random(n, L, seed_pos, sort_cell=1.4)
    tag = order.astype(np.uint64)
    vel = np.stack([normal(seed_vel, tag, c) * np.sqrt(kT) for c in range(3)], axis=1)
    return dict(name="C4", xyz=xyz, L=Lv, vel=vel, tag=order.astype(np.uint32), potential="DPDGeneralWeight",
                params=dict(A=25.0, gamma=4.5, s=0.5), r_cut=1.0, r_buff=0.4, kT=kT, dt=0.01, seed=7)


def config_tpm(nx=64, ny=64, nz=128, seed_pos=8, seed_q=9):
    """TwoPatchMorse patchy colloids on a simple-cubic lattice a=1.2 (configs[4])."""
    a = 1.2
    xyz, L = simple_cubic(nx, ny, nz, a, 0.05 * a, seed_pos)
    q = random_quaternions(xyz.shape[0], seed_q)
    return dict(name="C5", xyz=xyz, L=L, orientation=q, potential="TwoPatchMorse",
                params=dict(M_d=1.8341, M_r=0.0302, r_eq=1.0043, omega=5.0, alpha=0.40, repulsion=False),
                r_cut=1.6, r_buff=0.4)
