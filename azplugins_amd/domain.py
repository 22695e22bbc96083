"""Run-time side of the spatial domain decomposition: particle
migration and ghost re-selection at every neighbor-list rebuild, and the per-step halo
exchange of all the arrays a potential reads in ONE collective.

HOOMD's ``Communicator`` does this for the reference (the pair potentials only
participate); the MI355X-native form here keeps every step on the device:

* ``DeviceDomain.rebuild()`` -- collective, called when the neighbor list is rebuilt:
  owners are recomputed from the current positions (torch ops on the device), particles
  that left the sub-box travel to their new owner with everything they carry (position,
  velocity, orientation, tag, image) in one ``all_to_all_single``, the local particles are
  re-ordered *interior | boundary* along a blocked cell curve (compact 256-particle tiles
  for the tile plan), and every rank selects, per peer, the boundary particles inside that
  peer's ghost shell (one count exchange). Sender and receiver agree on the order of the
  ghost rows by construction: a peer's ghosts arrive in the sender's send order.
* ``DeviceDomain.exchange(names)`` -- per step: the rows the peers need are packed
  field after field into one send buffer (libazp ``azp_halo_pack_fields``), ONE
  ``all_to_all_single`` over RCCL / xGMI moves them, and they are unpacked into the ghost
  rows. Positions alone go straight into the ghost rows of the position array.

Everything is written against ``torch.distributed`` and plain tensors, so the same code
runs under gloo with CPU tensors (tests/test_domain.py) and under RCCL with device
tensors (bench.py --gpus N, Simulation.run with ``sim.domain`` set).
"""

import numpy as np

from .decomposition import Decomposition

_ROW = {"pos": 4, "vel": 4, "orientation": 4, "tag": 1, "image": 3, "angmom": 4, "inertia": 3}
_ALL = ("pos", "vel", "orientation", "tag", "image", "angmom", "inertia")


class DeviceDomain:
    def __init__(self, decomp, rank, arrays, group=None, density=None):
        """``arrays``: this rank's LOCAL particles, no ghosts -- ``pos`` (n, 4) float64 is
        required; ``vel``, ``orientation``, ``angmom`` (n, 4) float64, ``inertia`` (n, 3) float64, ``tag`` (n,)
        int32, ``image`` (n, 3) int32 travel with the particles when present. Call ``rebuild()`` next."""
        import torch

        assert isinstance(decomp, Decomposition)
        self.decomp = decomp
        self.rank = int(rank)
        self.world = decomp.world
        self.group = group
        self.names = [n for n in _ALL if n in arrays]
        assert "pos" in self.names
        self.arrays = {n: arrays[n] for n in self.names}
        self.device = self.arrays["pos"].device
        self.N_local = int(self.arrays["pos"].shape[0])
        self.n_ghost = 0
        self.n_interior = 0
        f64 = torch.float64
        grid = np.asarray(decomp.grid)
        self._L = torch.tensor(decomp.L, dtype=f64, device=self.device)
        self._width = torch.tensor(decomp.width, dtype=f64, device=self.device)
        self._grid = torch.tensor(grid, dtype=torch.int64, device=self.device)
        self._decomposed = [k for k in range(3) if grid[k] > 1]
        lo = np.stack([decomp.bounds(r)[0] for r in range(self.world)])
        self._lo = torch.tensor(lo, dtype=f64, device=self.device)
        self._density = density
        self.send_idx = torch.zeros(0, dtype=torch.int64, device=self.device)
        self.send_splits = [0] * self.world
        self.recv_splits = [0] * self.world
        self.num_rebuilds = 0
        self.num_migrated = 0
        self.bytes_sent_per_step = 0
        self._bufs = {}

    # -- geometry on the device --------------------------------------------
    def _wrap(self, xyz):
        import torch

        L = self._L
        return xyz - L * torch.floor(xyz / L + 0.5)

    def _owner(self, xyz):
        import torch

        c = torch.floor((xyz + 0.5 * self._L) / self._width).to(torch.int64)
        c = torch.minimum(torch.clamp(c, min=0), self._grid - 1)
        nx, ny = int(self._grid[0]), int(self._grid[1])
        return (c[:, 2] * ny + c[:, 1]) * nx + c[:, 0]

    def _depth(self, xyz):
        """Distance to the nearest decomposed face of this rank's sub-box."""
        import torch

        d = torch.full((xyz.shape[0],), float("inf"), dtype=torch.float64, device=xyz.device)
        lo = self._lo[self.rank]
        for k in self._decomposed:
            d = torch.minimum(d, torch.minimum(xyz[:, k] - lo[k], lo[k] + self._width[k] - xyz[:, k]))
        return d

    def _in_ghost_shell(self, xyz, peer):
        import torch

        mask = torch.ones(xyz.shape[0], dtype=torch.bool, device=xyz.device)
        lo = self._lo[peer]
        for k in self._decomposed:
            d = xyz[:, k] - (lo[k] + 0.5 * self._width[k])
            d = d - self._L[k] * torch.round(d / self._L[k])
            mask &= d.abs() <= 0.5 * self._width[k] + self.decomp.r_ghost
        return mask

    def _blocked_key(self, xyz, lo, extent, block=4, particles_per_block=256):
        """Sort key of a blocked cell curve over [lo, lo + extent) (cf. decomposition._blocked_key)."""
        import torch

        density = self._density if self._density else max(xyz.shape[0], 1) / float(np.prod(np.maximum(extent, 1e-12)))
        w = (particles_per_block / density) ** (1.0 / 3.0) / block
        dims = np.maximum(np.ceil(np.asarray(extent, dtype=np.float64) / w).astype(np.int64), 1)
        dims_t = torch.tensor(dims, dtype=torch.int64, device=xyz.device)
        lo_t = torch.tensor(np.asarray(lo, dtype=np.float64), dtype=torch.float64, device=xyz.device)
        c = torch.floor((xyz - lo_t) / w).to(torch.int64)
        c = torch.minimum(torch.clamp(c, min=0), dims_t - 1)
        nb = (dims + block - 1) // block
        key = ((c[:, 2] // block) * int(nb[1]) + (c[:, 1] // block)) * int(nb[0]) + (c[:, 0] // block)
        inner = ((c[:, 2] % block) * block + (c[:, 1] % block)) * block + (c[:, 0] % block)
        return key * block ** 3 + inner

    # -- collectives ---------------------------------------------------------
    def _a2a_counts(self, counts):
        """counts: python list of ints per destination rank -> list of ints per source rank."""
        import torch
        import torch.distributed as dist

        if self.world == 1 or not dist.is_initialized():
            return list(counts)
        dev = "cpu" if dist.get_backend(self.group) == "gloo" else self.device
        send = torch.tensor(counts, dtype=torch.int64, device=dev)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        return [int(c) for c in recv.tolist()]

    def all_reduce_flag(self, flag):
        """True on every rank if it is True on any (the collective rebuild decision, HOOMD's
        MPI_Allreduce on the distance-check result)."""
        import torch
        import torch.distributed as dist

        if self.world == 1 or not dist.is_initialized():
            return bool(flag)
        dev = "cpu" if dist.get_backend(self.group) == "gloo" else self.device
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item())

    def _a2a_rows(self, send, send_splits, recv_splits, out=None):
        import torch
        import torch.distributed as dist

        if out is None:
            out = torch.empty((sum(recv_splits),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if self.world == 1 or not dist.is_initialized():
            out.copy_(send)
            return out
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal only (several ranks on one GPU): gloo has no device all-to-all
            tmp = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(tmp, send.cpu(), output_split_sizes=recv_splits, input_split_sizes=send_splits, group=self.group)
            out.copy_(tmp)
        else:
            dist.all_to_all_single(out, send, output_split_sizes=recv_splits, input_split_sizes=send_splits, group=self.group)
        return out

    # -- migration + ghost selection ------------------------------------------
    def rebuild(self):
        """Collective. After it: ``arrays[name]`` has N_local + n_ghost rows (ghost rows
        filled), ordered interior | boundary | ghosts grouped by owner."""
        import torch

        N = self.N_local
        loc = {n: self.arrays[n][:N] for n in self.names}
        pos = loc["pos"]
        xyz = self._wrap(pos[:, :3])
        pos = torch.cat([xyz, pos[:, 3:4]], dim=1)
        loc["pos"] = pos
        # 1. migration: every particle goes to the rank that owns its position
        owner = self._owner(xyz)
        order = torch.sort(owner, stable=True).indices
        counts = torch.bincount(owner, minlength=self.world).tolist()
        self.num_migrated = N - int(counts[self.rank])
        width = sum(_ROW[n] for n in self.names)
        payload = torch.cat([loc[n].reshape(N, -1).to(torch.float64) if n in ("tag", "image") else loc[n] for n in self.names], dim=1)
        payload = payload.index_select(0, order).contiguous()
        recv_counts = self._a2a_counts(counts)
        got = self._a2a_rows(payload, counts, recv_counts)
        assert got.shape[1] == width
        n_new = got.shape[0]
        new = {}
        col = 0
        for n in self.names:
            w = _ROW[n]
            block = got[:, col:col + w]
            col += w
            if n == "tag":
                new[n] = block.reshape(-1).to(torch.int32)
            elif n == "image":
                new[n] = block.to(torch.int32).contiguous()
            else:
                new[n] = block.contiguous()
        # 2. order: interior (deeper than the ghost width: lists no ghost) | boundary, each along a
        #    blocked cell curve anchored at the group's own corner
        xyz = new["pos"][:, :3]
        lo, hi = (np.asarray(b, dtype=np.float64) for b in self.decomp.bounds(self.rank))
        depth = self._depth(xyz)
        shell = depth <= self.decomp.r_ghost  # (<=: a particle exactly r_ghost deep can still be a peer's ghost)
        lo_int = lo + np.where(np.asarray(self.decomp.grid) > 1, self.decomp.r_ghost, 0.0)
        key_int = self._blocked_key(xyz, lo_int, np.maximum(hi - lo - 2 * (lo_int - lo), 1e-9))
        key_bnd = self._blocked_key(xyz, lo, hi - lo)
        big = int(max(int(key_int.max()) if n_new else 0, int(key_bnd.max()) if n_new else 0)) + 1
        key = torch.where(shell, key_bnd + big, key_int)
        perm = torch.sort(key, stable=True).indices
        for n in self.names:
            new[n] = new[n].index_select(0, perm)
        # whole tiles of 256 (cf. decomposition.RankDomain): the tile that mixes interior and boundary particles, with
        # its large staged set, belongs to the boundary launch
        self.n_interior = int((~shell).sum().item()) // 256 * 256
        self.N_local = n_new
        # 3. ghosts: per peer, my boundary particles inside its ghost shell (ascending local index)
        xyz_b = new["pos"][self.n_interior:, :3]
        idx, send_splits = [], []
        for peer in range(self.world):
            if peer == self.rank or not self._decomposed:
                send_splits.append(0)
                continue
            sel = torch.nonzero(self._in_ghost_shell(xyz_b, peer)).reshape(-1) + self.n_interior
            idx.append(sel)
            send_splits.append(int(sel.numel()))
        self.send_idx = torch.cat(idx) if idx else torch.zeros(0, dtype=torch.int64, device=self.device)
        self.send_splits = send_splits
        self.recv_splits = self._a2a_counts(send_splits)
        self.n_ghost = sum(self.recv_splits)
        # 4. arrays with ghost rows, then the first exchange of everything
        for n in self.names:
            a = new[n]
            ghost = torch.zeros((self.n_ghost,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
            self.arrays[n] = torch.cat([a, ghost], dim=0).contiguous()
        self._bufs = {}
        self.num_rebuilds += 1
        # (image flags and the rotational state are the owner's business: they migrate, ghosts do not need them)
        self.exchange([n for n in self.names if n not in ("image", "angmom", "inertia")])
        return self.arrays

    # -- per-step exchange -------------------------------------------------------
    def _packed_width(self, names):
        return sum(((self._row_bytes(n) + 7) // 8) * 8 for n in names)

    def _row_bytes(self, n):
        a = self.arrays[n]
        return a.element_size() * (a.shape[1] if a.dim() == 2 else 1)

    def pack(self, names=("pos",)):
        """Gather the rows the peers need into one send buffer (current stream)."""
        import ctypes as C

        import torch

        from . import _lib

        names = list(names)
        n_send = int(self.send_idx.numel())
        wbytes = self._packed_width(names)
        key = ("send", tuple(names))
        if key not in self._bufs:
            self._bufs[key] = torch.empty((n_send, wbytes // 8), dtype=torch.float64, device=self.device)
        buf = self._bufs[key]
        if n_send == 0:
            return names, buf
        if buf.is_cuda:
            fkey = ("pack_fields", tuple(names))
            if fkey not in self._bufs:  # (cleared at every rebuild, when the arrays are replaced)
                fields = (_lib.HaloField * len(names))()
                for c, n in enumerate(names):
                    fields[c].d_data = self.arrays[n].data_ptr()
                    fields[c].row_bytes = self._row_bytes(n)
                self._bufs[fkey] = fields
            fields = self._bufs[fkey]
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(_lib.lib().azp_halo_pack_fields(n_send, len(names), fields, self.send_idx.data_ptr(), buf.data_ptr(), wbytes, stream),
                       "azp_halo_pack_fields")
        else:
            raw = buf.view(torch.uint8).reshape(n_send, wbytes)
            off = 0
            for n in names:
                rb = self._row_bytes(n)
                rows = self.arrays[n].index_select(0, self.send_idx).contiguous().view(torch.uint8).reshape(n_send, rb)
                raw[:, off:off + rb] = rows
                off += ((rb + 7) // 8) * 8
        return names, buf

    def transfer(self, packed):
        """ONE all_to_all_single, then the ghost rows of every packed array are overwritten."""
        import ctypes as C

        import torch

        from . import _lib

        names, buf = packed
        N = self.N_local
        wbytes = self._packed_width(names)
        self.bytes_sent_per_step = buf.numel() * buf.element_size()
        if names == ["pos"]:
            # the receive buffer IS the ghost region of the position array: no unpack pass
            self._a2a_rows(buf, self.send_splits, self.recv_splits, out=self.arrays["pos"][N:])
            return
        key = ("recv", tuple(names))
        if key not in self._bufs:
            self._bufs[key] = torch.empty((self.n_ghost, wbytes // 8), dtype=torch.float64, device=self.device)
        recv = self._a2a_rows(buf, self.send_splits, self.recv_splits, out=self._bufs[key])
        if self.n_ghost == 0:
            return
        if recv.is_cuda:
            fields = (_lib.HaloField * len(names))()
            for c, n in enumerate(names):
                a = self.arrays[n]
                fields[c].d_data = a.data_ptr() + N * self._row_bytes(n)
                fields[c].row_bytes = self._row_bytes(n)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(_lib.lib().azp_halo_unpack_fields(self.n_ghost, len(names), fields, recv.data_ptr(), wbytes, stream),
                       "azp_halo_unpack_fields")
        else:
            raw = recv.view(torch.uint8).reshape(self.n_ghost, wbytes)
            off = 0
            for n in names:
                rb = self._row_bytes(n)
                a = self.arrays[n]
                a[N:] = raw[:, off:off + rb].contiguous().view(a.dtype).reshape((self.n_ghost,) + tuple(a.shape[1:]))
                off += ((rb + 7) // 8) * 8

    def exchange(self, names=("pos",)):
        self.transfer(self.pack(names))

    # -- state plumbing ------------------------------------------------------------
    def attach_state(self, state):
        """Point a ``State`` at the current arrays (after ``rebuild``)."""
        state.N = self.N_local
        state.n_ghost = self.n_ghost
        for n in self.names:
            setattr(state, n, self.arrays[n])
        # (Simulation.attach_domain insists on every array the integrator and the forces use being in self.names:
        # an array that does not migrate would keep its old size and order)
        if getattr(state, "bond_tags", None) is not None:
            state.relocalize_bonds()  # bond table and exclusions by local index: every index changed
        state.position_generation += 1
        state.order_generation = getattr(state, "order_generation", 0) + 1  # every index changed: the list must be rebuilt
