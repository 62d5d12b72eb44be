"""Row-block sharded SpMV across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The reference has no distributed code at all; this is the MI355X-first design
of section 8(e): rank g owns the contiguous rows [lo_g, hi_g) of A (with GLOBAL column indices), the
matching slices of x and y, and a full-length x buffer.  Before each multiply the x entries its rows
reference are brought into that buffer by one of two exchanges:

  * "allgather": every rank contributes its slice, ncclAllGather -- the north-star exchange; volume
    per rank = (world-1)/world * N values, per-link bound on the point-to-point xGMI fabric.
  * "halo": at setup each rank reduces its column indices to a span [col_min, col_max]; only the
    parts of other ranks' slices inside that span are exchanged, with batched point-to-point
    send/recv straight into the x buffer.  For a banded matrix (5-pt Poisson: span = own rows +- m)
    that is 2*m values per rank instead of N.  A matrix whose rows reference every column
    degenerates to the all-gather volume.
  * "auto" (default): halo when it moves less than half of the all-gather volume.

The local multiply is the single-GPU C-ABI call (cmi_spmv_*), so N ranks = N independent hot paths
joined by exactly one exchange step.
"""
from dataclasses import dataclass


def partition_rows(num_rows, world):
    """Equal-count row blocks (the all-gather needs equal counts): count = ceil(num_rows/world);
    rank r owns [r*count, min((r+1)*count, num_rows)).  Returns world+1 offsets."""
    count = -(-num_rows // world) if world > 0 else 0
    return [min(r * count, num_rows) for r in range(world + 1)]


@dataclass
class ExchangePlan:
    mode: str                 # "allgather" | "halo"
    recv: list                # per peer: (lo, hi) range of the x buffer this rank receives (may be empty)
    send: list                # per peer: (lo, hi) range of this rank's slice it sends
    count: int                # padded slice length (all-gather)
    recv_values: int          # values received per exchange
    allgather_values: int     # what the all-gather would receive


class ShardedVectorExchange:
    """Owns the full-length x buffer of one rank and fills it before a multiply."""

    def __init__(self, num_cols, rank, world, col_min, col_max, dtype, device, mode="auto", group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.rank, self.world, self.group = rank, world, group
        self.num_cols = num_cols
        self.offsets = partition_rows(num_cols, world)
        self.count = self.offsets[1] - self.offsets[0] if world > 0 else 0
        self.lo, self.hi = self.offsets[rank], self.offsets[rank + 1]
        # buffer padded to world*count so the all-gather can write equal-sized pieces
        self.x_full = torch.zeros(max(world * self.count, 1), dtype=dtype, device=device)
        self.x_local = self.x_full[self.lo:self.hi]  # this rank's slice lives inside the buffer

        # every rank learns every rank's column span (setup-time, tiny)
        span = torch.tensor([col_min, col_max], dtype=torch.int64, device=device)
        spans = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
        if world > 1:
            dist.all_gather(spans, span, group=group)
        else:
            spans = [span]
        spans = [tuple(int(v) for v in s.tolist()) for s in spans]

        def overlap(span_, owner):
            lo = max(span_[0], self.offsets[owner])
            hi = min(span_[1] + 1, self.offsets[owner + 1])
            return (lo, hi) if hi > lo else (0, 0)

        recv = [overlap(spans[rank], p) if p != rank else (0, 0) for p in range(world)]
        send = [overlap(spans[p], rank) if p != rank else (0, 0) for p in range(world)]
        recv_values = sum(h - l for l, h in recv)
        allgather_values = (world - 1) * self.count
        if mode == "auto":
            # the same decision on every rank: compare the WORST rank's halo volume
            worst = torch.tensor([recv_values], dtype=torch.int64, device=device)
            if world > 1:
                dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=group)
            mode = "halo" if 2 * int(worst.item()) < allgather_values else "allgather"
        if mode not in ("halo", "allgather"):
            raise ValueError(f"unknown exchange mode {mode!r}")
        self.plan = ExchangePlan(mode, recv, send, self.count, recv_values, allgather_values)
        self._gather_in = None
        if mode == "allgather" and world > 1:
            self._gather_in = torch.zeros(self.count, dtype=dtype, device=device)

    def exchange(self):
        """Fill x_full with what this rank's rows need.  x_local must already hold the fresh slice."""
        if self.world == 1:
            return
        dist, plan = self.dist, self.plan
        if plan.mode == "allgather":
            n = self.hi - self.lo
            self._gather_in[:n].copy_(self.x_local)
            dist.all_gather_into_tensor(self.x_full, self._gather_in, group=self.group)
            return
        ops = []
        for p in range(self.world):
            l, h = plan.send[p]
            if h > l:
                ops.append(dist.P2POp(dist.isend, self.x_full[l:h], p, group=self.group))
        for p in range(self.world):
            l, h = plan.recv[p]
            if h > l:
                ops.append(dist.P2POp(dist.irecv, self.x_full[l:h], p, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()


class ShardedCsr:
    """Row block [lo, hi) of a CSR matrix with global column indices + the x exchange.

    `local_multiply(x_full, y_local)` performs the single-GPU SpMV; the default calls the C-ABI
    through matrices.multiply.  (The CPU gloo tests inject a host stand-in to check the sharding and
    exchange logic without a GPU.)"""

    def __init__(self, A_local, num_cols, rank, world, mode="auto", group=None, local_multiply=None,
                 col_span=None):
        import torch
        self.A = A_local
        self.rank, self.world = rank, world
        dev = A_local.values.device
        if col_span is None:
            if A_local.num_entries > 0:
                col_span = (int(A_local.column_indices.min().item()), int(A_local.column_indices.max().item()))
            else:
                col_span = (0, -1)
        self.vec = ShardedVectorExchange(num_cols, rank, world, col_span[0], col_span[1], A_local.values.dtype, dev,
                                         mode=mode, group=group)
        if A_local.num_rows != self.vec.hi - self.vec.lo:
            raise ValueError(f"rank {rank}: local block has {A_local.num_rows} rows, partition expects "
                             f"{self.vec.hi - self.vec.lo}")
        self.x_view = self.vec.x_full[:num_cols]
        if local_multiply is None:
            from .matrices import multiply as _mul
            local_multiply = lambda x_full, y: _mul(self.A, x_full, y)  # noqa: E731
        self._mul = local_multiply
        self.torch = torch

    @property
    def x_local(self):
        return self.vec.x_local

    def multiply(self, y_local, exchange=True):
        """y_local = A[lo:hi, :] * x, with x's slices taken from every rank's x_local."""
        if exchange:
            self.vec.exchange()
        self._mul(self.x_view, y_local)
        return y_local
