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
        self.bufs = {}
        for r in range(world - 2):
            if self.hi[r] >= self.lo[r + 2]:
                raise ValueError("slabs too thin: rank %d overlaps rank %d" % (r, r + 2))

    @staticmethod
    def layer_ranges(world, cells_per_rank, margin, nlayers, reach=4):
        """Rank r owns cells [margin + r*c, margin + (r+1)*c): its closest nodes lie on the node planes
        of those cells, the 5^d stencil reaches 2 planes further, `reach` >= 2 adds room for drift."""
        lo = [max(0, margin + r * cells_per_rank - reach) for r in range(world)]
        hi = [min(nlayers - 1, margin + (r + 1) * cells_per_rank + reach) for r in range(world)]
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
        ops, recv = [], []
        for nb in (self.rank - 1, self.rank + 1):
            if nb < 0 or nb >= self.world:
                continue
            ov = self.overlap(self.rank, nb)
            if ov is None:
                continue
            sl = arr[ov[0] * self.plane * nfield:(ov[1] + 1) * self.plane * nfield]
            key = (nb, nfield, arr.dtype)
            if key not in self.bufs:
                self.bufs[key] = (torch.empty_like(sl), torch.empty_like(sl))
            sbuf, rbuf = self.bufs[key]
            sbuf.copy_(sl)
            ops.append(dist.P2POp(dist.isend, sbuf, nb))
            ops.append(dist.P2POp(dist.irecv, rbuf, nb))
            recv.append((sl, rbuf))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for sl, rbuf in recv:
            if kind == 0:
                sl.add_(rbuf)
            else:
                torch.maximum(sl, rbuf, out=sl)
        return 0


def device_tensor(torch, dptr, n, elem_bytes):
    """Wraps a raw device pointer owned by the library as a torch tensor (no copy)."""

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8" if elem_bytes == 8 else "|u1",
                                  "data": (int(dptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda")
