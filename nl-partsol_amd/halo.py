"""Ghost-node exchange for a slab partition of the particles (multi-GPU hook of include/nlps_gpu.h).

Particles are range-partitioned into slabs along the slowest grid axis; every rank keeps the full
nodal arrays in GRID numbering (x fastest, slab axis slowest), so the node layers a rank shares with
a neighbouring rank are one contiguous slice.  After each nodal scatter the overlapping layers are
summed (doubles) or OR-ed (bytes) with the two neighbouring ranks only: xGMI is point-to-point, two
concurrent neighbour transfers per rank use two distinct links, a ring all-reduce of the whole grid
would be per-link bound.  `mode="allreduce"` keeps the simple all-reduce of the whole array
(BASELINE.json's wording) for comparison.  Works on any torch tensor (RCCL on GPU, gloo on CPU)."""


class SlabHalo:
    def __init__(self, torch, dist, rank, world, plane_nodes, nlayers, lo, hi, mode="p2p"):
        """lo[r], hi[r]: inclusive range of node layers rank r's particles may touch."""
        self.torch, self.dist, self.rank, self.world, self.mode = torch, dist, rank, world, mode
        self.plane, self.nz = int(plane_nodes), int(nlayers)
        self.lo, self.hi = list(lo), list(hi)
        self.plans, self.tensors, self.pending, self.events, self.side = {}, {}, {}, {}, None
        for r in range(world - 2):
            if self.hi[r] >= self.lo[r + 2]:
                raise ValueError("slabs too thin: rank %d overlaps rank %d" % (r, r + 2))

    @staticmethod
    def layer_ranges(world, cells_per_rank, margin, nlayers, reach=4):
        """Rank r owns cells [margin + r*c, margin + (r+1)*c): its closest nodes lie on the node planes
        of those cells, the 5^d stencil reaches 2 planes further, `reach` >= 2 adds room for drift.
        cells_per_rank may be a list (slabs of unequal thickness: a cube whose layers do not divide evenly)."""
        cz = list(cells_per_rank) if hasattr(cells_per_rank, "__len__") else [int(cells_per_rank)] * world
        start = [margin + sum(cz[:r]) for r in range(world + 1)]
        lo = [max(0, start[r] - reach) for r in range(world)]
        hi = [min(nlayers - 1, start[r + 1] + reach) for r in range(world)]
        return lo, hi

    def overlap(self, a, b):
        lo, hi = max(self.lo[a], self.lo[b]), min(self.hi[a], self.hi[b])
        return (lo, hi) if lo <= hi else None

    def exchange(self, arr, nfield, kind):
        """arr: flat tensor [nlayers*plane*nfield]; kind 0 = sum, 1 = OR (bytes 0/1)."""
        torch, dist = self.torch, self.dist
        if self.world == 1:
            return 0
        if self.mode == "allreduce":
            dist.all_reduce(arr, op=dist.ReduceOp.SUM if kind == 0 else dist.ReduceOp.MAX)
            return 0
        if arr.is_cuda and dist.get_backend() == "gloo":
            return self._exchange_staged(arr, nfield, kind)
        plan = self.plans.get((arr.data_ptr(), nfield, kind))
        if plan is None:  # slices, receive buffers and P2P descriptors are built once per nodal array
            ops, recv = [], []
            for nb in (self.rank - 1, self.rank + 1):
                if nb < 0 or nb >= self.world:
                    continue
                ov = self.overlap(self.rank, nb)
                if ov is None:
                    continue
                sl = arr[ov[0] * self.plane * nfield:(ov[1] + 1) * self.plane * nfield]
                rbuf = torch.empty_like(sl)
                ops.append(dist.P2POp(dist.isend, sl, nb))  # the slice is contiguous: sent in place
                ops.append(dist.P2POp(dist.irecv, rbuf, nb))
                recv.append((sl, rbuf))
            plan = (ops, recv, arr)
            self.plans[(arr.data_ptr(), nfield, kind)] = plan
        ops, recv, _ = plan
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for sl, rbuf in recv:  # after the sends completed (stream order), so the neighbour got the unsummed slice
            if kind == 0:
                sl.add_(rbuf)
            else:
                torch.maximum(sl, rbuf, out=sl)
        return 0

    def _exchange_staged(self, arr, nfield, kind):
        """Device arrays over a CPU-only backend (gloo): the halo slices go through host memory.  Only for
        rehearsing the N > 1 path on a box without one GPU per rank; RCCL never takes this branch."""
        torch, dist = self.torch, self.dist
        ops, recv = [], []
        for nb in (self.rank - 1, self.rank + 1):
            if nb < 0 or nb >= self.world:
                continue
            ov = self.overlap(self.rank, nb)
            if ov is None:
                continue
            sl = arr[ov[0] * self.plane * nfield:(ov[1] + 1) * self.plane * nfield]
            s_host = sl.cpu()
            r_host = torch.empty_like(s_host)
            ops.append(dist.P2POp(dist.isend, s_host, nb))
            ops.append(dist.P2POp(dist.irecv, r_host, nb))
            recv.append((sl, r_host))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for sl, r_host in recv:
            r = r_host.to(sl.device)
            if kind == 0:
                sl.add_(r)
            else:
                torch.maximum(sl, r, out=sl)
        return 0

    def migrate(self, S, keep_lo, keep_hi):
        """Moves the particles whose closest node left [keep_lo, keep_hi] (node layers) to the neighbouring rank:
        select + pack on the device, counts and rows exchanged with the two neighbours, commit.  Returns
        (sent_down, sent_up, received)."""
        torch, dist = self.torch, self.dist
        # an edge rank has no neighbour on its outer side: its keep range is clamped to the grid there, so that no
        # particle is ever selected towards a rank that does not exist (it would be dropped at the commit)
        if self.rank == 0:
            keep_lo = 0
        if self.rank == self.world - 1:
            keep_hi = self.nz - 1
        n_down, n_up, rw, dptr_down, dptr_up = S.migration_select(keep_lo, keep_hi)
        if (n_down and self.rank == 0) or (n_up and self.rank == self.world - 1):
            raise RuntimeError("migrate: %d / %d particles selected towards a non-existent neighbour" % (n_down, n_up))
        if self.world == 1:
            S.migration_commit(0, 0, 0, 0)
            return n_down, n_up, 0
        nbrs = [(self.rank - 1, n_down, dptr_down), (self.rank + 1, n_up, dptr_up)]
        nbrs = [(r, n, p) for r, n, p in nbrs if 0 <= r < self.world]
        staged = dist.get_backend() == "gloo"
        dev = "cpu" if staged else "cuda"
        # 1. how many rows come from each neighbour
        send_n = {r: torch.tensor([n], dtype=torch.int64, device=dev) for r, n, _ in nbrs}
        recv_n = {r: torch.zeros(1, dtype=torch.int64, device=dev) for r, _, _ in nbrs}
        ops = []
        for r, _, _ in nbrs:
            ops += [dist.P2POp(dist.isend, send_n[r], r), dist.P2POp(dist.irecv, recv_n[r], r)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        # 2. the rows
        ops, keep, recv = [], [], {}
        for r, n, p in nbrs:
            if n > 0:
                t = device_tensor(torch, p, n * rw, 8)
                t = t.cpu() if staged else t
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, r))
            m = int(recv_n[r].item())
            if m > 0:
                recv[r] = torch.empty(m * rw, dtype=torch.float64, device=dev)
                ops.append(dist.P2POp(dist.irecv, recv[r], r))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if not staged:
            torch.cuda.current_stream().synchronize()
        got = [(recv[r].data_ptr(), recv[r].numel() // rw) for r in sorted(recv)] + [(0, 0), (0, 0)]
        S.migration_commit(got[0][0], got[0][1], got[1][0], got[1][1])
        return n_down, n_up, got[0][1] + got[1][1]

    def ghost_bands(self, rank):
        """(band_lo, band_hi): node layers <= band_lo are shared with rank-1, layers >= band_hi with rank+1."""
        lo_ov = self.overlap(rank, rank - 1) if rank > 0 else None
        hi_ov = self.overlap(rank, rank + 1) if rank + 1 < self.world else None
        return (lo_ov[1] if lo_ov else -1), (hi_ov[0] if hi_ov else self.nz)

    def exchange_ptr(self, dptr, nelem, nfield, elem_bytes, kind, phase=0):
        """Entry point for the C callback: wraps the raw device pointer once and reuses the tensor.
        phase 0: exchange in the order of the current stream.  phase 1: start the exchange on a side stream that
        first waits for everything queued on the current stream; phase 2: the current stream waits for it.  The
        library runs the tiles that do not touch a ghost band between the two (nlps_gpu_set_ghost_bands)."""
        torch = self.torch
        t = self.tensors.get((dptr, nelem, elem_bytes))
        if t is None:
            t = device_tensor(torch, dptr, nelem, elem_bytes)
            self.tensors[(dptr, nelem, elem_bytes)] = t
        if phase == 0 or self.world == 1:
            return self.exchange(t, nfield, kind) if phase != 2 else 0
        if phase == 1:
            if self.side is None:
                self.side = torch.cuda.Stream()
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                st = self.exchange(t, nfield, kind)
                ev = self.events.get((dptr, nfield, kind))  # one event per nodal array, re-recorded every step
                if ev is None:
                    ev = self.events[(dptr, nfield, kind)] = torch.cuda.Event()
                ev.record(self.side)
            self.pending[(dptr, nfield, kind)] = ev
            return st
        ev = self.pending.pop((dptr, nfield, kind), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        return 0


def device_tensor(torch, dptr, n, elem_bytes):
    """Wraps a raw device pointer owned by the library as a torch tensor (no copy)."""

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8" if elem_bytes == 8 else "|u1",
                                  "data": (int(dptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda")
