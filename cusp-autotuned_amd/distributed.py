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
  * "peer": the halo plan, ONE-SIDED.  xGMI is a load/store fabric, so each rank maps its
    neighbours' x buffers once (IPC handles of cmi_malloc'ed buffers) and before a multiply PULLS
    the ranges it needs with one small copy kernel on its own stream (cmi_copy_ranges): no
    collective launch, no second stream, no proxy thread.  Measured on one MI355X with a 1-rank RCCL
    communicator standing in for the wire (tools/shard_step_probe.py): the grouped ncclSend/ncclRecv
    step costs ~65 us of device time around a 134 us SpMV and does not overlap with it; the pull
    costs ~3 us.  Visibility is at kernel boundaries, so the CALLER orders the peers' producing
    kernels before the pull: a barrier (bench.py: x is static between its barriers) or the
    all-reduces of the algorithm (krylov.cg).  `fence()` is that barrier for anyone else.
  * "auto" (default): halo when it moves less than half of the all-gather volume -- one-sided
    ("peer") when the buffers are in HBM and every rank could map and verify its neighbours'
    buffers, two-sided ("halo") otherwise.

The local multiply is the single-GPU C-ABI call (cmi_spmv_*), so N ranks = N independent hot paths
joined by exactly one exchange step.
"""
from dataclasses import dataclass


def partition_rows(num_rows, world):
    """Equal-count row blocks (the all-gather needs equal counts): count = ceil(num_rows/world);
    rank r owns [r*count, min((r+1)*count, num_rows)).  Returns world+1 offsets."""
    count = -(-num_rows // world) if world > 0 else 0
    return [min(r * count, num_rows) for r in range(world + 1)]


def partition_by_entries(row_offsets, world):
    """Row blocks balanced by ENTRIES ("balanced by nnz"): block r ends at the first row whose
    cumulative entry count reaches (r+1)/world of the total.  row_offsets: the GLOBAL CSR row offsets (host
    sequence or tensor, length num_rows + 1).  Returns world+1 row offsets; slices may differ in length (the
    halo and one-sided exchanges take any partition; the all-gather pads to the longest slice)."""
    import bisect
    ro = row_offsets.tolist() if hasattr(row_offsets, "tolist") else list(row_offsets)
    num_rows, nnz = len(ro) - 1, ro[-1]
    cuts = [0]
    for r in range(1, world):
        target = (nnz * r + world - 1) // world
        row = bisect.bisect_left(ro, target, lo=cuts[-1], hi=num_rows)  # first row offset >= target
        cuts.append(min(max(row, cuts[-1]), num_rows))
    cuts.append(num_rows)
    return cuts


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

    def __init__(self, num_cols, rank, world, col_min, col_max, dtype, device, mode="auto", group=None, offsets=None, comm=None):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.rank, self.world, self.group = rank, world, group
        # DATA PATH: the product's communicator (binding.Comm = cmi_comm: RCCL behind the C-ABI) whenever the vectors are in HBM and
        # the job really has one process per GPU (torch.distributed's "nccl" backend, or a communicator handed in); torch.distributed
        # then only carries set-up records.  The gloo rehearsals (ranks sharing a GPU, CPU tests) keep the injected host transport.
        self.comm = comm
        if comm is None and world > 1 and torch.device(device).type == "cuda" and dist.is_initialized() and dist.get_backend(group) == "nccl":
            from . import binding as B
            import os
            import sys
            # Making the communicator is COLLECTIVE (ncclCommInitRank inside cmi_comm_create): a rank that cannot enter it must say so BEFORE
            # the others do, or they wait for it for ever.  So the ranks first agree that every one of them can bind RCCL -- a
            # local, non-collective check -- and only then call it.
            ok, why = 1, ""
            if os.environ.get("CMI_PYTHON_COMM", "1") == "0":
                ok, why = 0, "CMI_PYTHON_COMM=0"
            else:
                try:
                    B.comm_library_version()
                except Exception as e:  # noqa: BLE001 -- librccl not loadable on this rank
                    ok, why = 0, f"{type(e).__name__}: {e}"
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 1:
                try:
                    self.comm = B.Comm.from_torch_distributed(group)
                except Exception as e:  # noqa: BLE001 -- an error every rank sees alike (argument check) or one the init itself reports
                    ok, why = 0, f"{type(e).__name__}: {e}"
                flag = torch.tensor([ok], dtype=torch.int32, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0:
                if rank == 0 or why:
                    print(f"[cusp-autotuned_amd] rank {rank}: the C-ABI communicator is not used ({why or 'another rank could not make it'}); "
                          "the exchange falls back to torch.distributed collectives", file=sys.stderr)
                self.comm = None
        self.num_cols = num_cols
        self._span = (col_min, col_max)
        # the partition of x (= of the rows): equal counts by default, any monotone cut list on request
        # (partition_by_entries); `count` = the longest slice = the all-gather's padded piece
        self.offsets = list(offsets) if offsets is not None else partition_rows(num_cols, world)
        if len(self.offsets) != world + 1 or self.offsets[0] != 0 or self.offsets[-1] != num_cols or \
                any(b < a for a, b in zip(self.offsets, self.offsets[1:])):
            raise ValueError(f"offsets must be {world + 1} non-decreasing cuts from 0 to {num_cols}")
        self.count = max((b - a for a, b in zip(self.offsets, self.offsets[1:])), default=0)
        self.uniform = all(self.offsets[r] == min(r * self.count, num_cols) for r in range(world + 1))
        self.lo, self.hi = self.offsets[rank], self.offsets[rank + 1]
        # buffer padded to world*count so the all-gather of a uniform partition can write straight into it; in
        # HBM it is an allocation of its own (not a slice of torch's caching allocator) so that peers can map it
        numel = max(world * self.count, num_cols, 1)
        self._buffer = None
        if torch.device(device).type == "cuda":
            from . import binding as B
            self._buffer = B.DeviceBuffer(numel * torch.empty(0, dtype=dtype).element_size(), device)
            self.x_full = self._buffer.tensor(dtype)
        else:
            self.x_full = torch.zeros(numel, dtype=dtype, device=device)
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
        import os
        try_peer = mode == "peer"
        if mode == "auto":
            # the same decision on every rank: compare the WORST rank's halo volume
            worst = torch.tensor([recv_values], dtype=torch.int64, device=device)
            if world > 1:
                dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=group)
            mode = "halo" if 2 * int(worst.item()) < allgather_values else "allgather"
            try_peer = mode == "halo" and os.environ.get("CMI_EXCHANGE_PEER", "1") != "0"
        if mode not in ("halo", "allgather", "peer"):
            raise ValueError(f"unknown exchange mode {mode!r}")
        self.plan = ExchangePlan("halo" if mode == "peer" else mode, recv, send, self.count, recv_values, allgather_values)
        self._pulls, self._peer_ptrs = [], {}
        if try_peer and world > 1 and (self._buffer is not None or mode == "peer"):
            if self._setup_peer():
                self.plan.mode = "peer"
            elif mode == "peer":
                raise RuntimeError("exchange mode 'peer' requested but a rank could not map or verify its neighbours' buffers")
        self._gather_in = None
        self._gather_out = None  # non-uniform partition: padded landing zone of the all-gather
        self._unpad = None
        self._ops = None
        # RCCL orders its transfers after the work already queued on the compute stream.  A host
        # transport (gloo in the one-GPU tests) reads the send buffer from the CPU with no such
        # ordering, so the stream is drained first -- the callers queue kernels ahead of the host
        # (krylov.cg launches the next SpMV before it reads the residual).
        self._host_transport = world > 1 and dist.get_backend(group) != "nccl" and self.x_full.is_cuda
        if mode == "allgather" and world > 1:
            self._gather_in = torch.zeros(self.count, dtype=dtype, device=device)

    def _setup_peer(self):
        """Map the neighbours' buffers and verify one pull end to end; True only if EVERY rank succeeded
        (collective: all ranks call it).  Any failure leaves the two-sided plan in force."""
        dist, torch = self.dist, self.torch
        from . import binding as B
        ok, why = 1, ""
        handles = [None] * self.world
        my_device = self.x_full.device.index if self.x_full.is_cuda else None
        try:
            mine = (self._buffer.ipc_handle(), my_device) if self._buffer is not None else None
        except Exception as e:  # noqa: BLE001
            mine, ok, why = None, 0, f"ipc_get_handle: {e}"
        dist.all_gather_object(handles, mine, group=self.group)
        devices = [h[1] if h is not None else None for h in handles]
        handles = [h[0] if h is not None else None for h in handles]
        item = self.x_full.element_size()
        ranges = []
        if ok and all(h is not None for h in handles):
            try:
                for p, (l, h) in enumerate(self.plan.recv):
                    if h > l:
                        # the driver's answer to "may kernels on my GPU touch that GPU's memory?" comes first
                        # (ranks of one node see the same device numbering)
                        if not B.device_can_access_peer(my_device, devices[p]):
                            raise RuntimeError(f"GPU {my_device} has no peer access to GPU {devices[p]} (rank {p})")
                        if p not in self._peer_ptrs:
                            self._peer_ptrs[p] = B.ipc_open(handles[p])
                        # every rank's buffer is indexed by GLOBAL column: the same offset on both sides
                        ranges.append((self._peer_ptrs[p] + l * item, self._buffer.ptr + l * item, (h - l) * item))
                self._pulls = [B.CopyRanges(ranges[i:i + B.MAX_COPY_RANGES]) for i in range(0, len(ranges), B.MAX_COPY_RANGES)]
            except Exception as e:  # noqa: BLE001
                ok, why = 0, f"ipc_open_handle: {e}"
        else:
            ok = 0
            why = why or "a rank has no exportable buffer"
        # end-to-end check with a per-rank signature (collective steps run on every rank, ok or not)
        self.x_local.fill_(float(self.rank + 1))
        self.fence()
        if ok:
            try:
                for c in self._pulls:
                    c.launch()
                torch.cuda.current_stream(self.x_full.device).synchronize()
                for p, (l, h) in enumerate(self.plan.recv):
                    if h > l and not bool((self.x_full[l:h] == float(p + 1)).all()):
                        ok, why = 0, f"pull from rank {p} returned wrong data"
            except Exception as e:  # noqa: BLE001
                ok, why = 0, f"pull: {e}"
        flag = torch.tensor([ok], dtype=torch.int32, device=self.x_full.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        self.fence()
        self.x_full.zero_()
        self.fence()
        if int(flag.item()) == 1:
            return True
        if why:
            import sys
            print(f"[cusp-autotuned_amd] rank {self.rank}: one-sided exchange unavailable ({why}); using two-sided halo exchange",
                  file=sys.stderr)
        self._release_peers()
        return False

    def _release_peers(self):
        self._pulls = []
        if self._peer_ptrs:
            from . import binding as B
            for ptr in self._peer_ptrs.values():
                try:
                    B.ipc_close(ptr)
                except Exception:  # noqa: BLE001
                    pass
            self._peer_ptrs = {}

    def fence(self):
        """Everything queued on this rank's stream has completed AND every rank has reached this point:
        the ordering the one-sided pull needs between the peers' producers and itself when the caller's
        algorithm does not already provide it."""
        if self.x_full.is_cuda:
            self.torch.cuda.current_stream(self.x_full.device).synchronize()
        if self.world > 1:
            if self.comm is not None:
                self.comm.barrier()
            else:
                self.dist.barrier(group=self.group)

    def close(self):
        """Unmap the peers' buffers (collective by convention: call it on every rank before the buffers die)."""
        if self.world > 1 and self._peer_ptrs:
            self.fence()
        self._release_peers()

    def __del__(self):
        try:
            self._release_peers()
        except Exception:  # noqa: BLE001
            pass

    def _build_ops(self):
        dist, plan = self.dist, self.plan
        ops = []
        for p in range(self.world):
            l, h = plan.send[p]
            if h > l:
                ops.append(dist.P2POp(dist.isend, self.x_full[l:h], p, group=self.group))
        for p in range(self.world):
            l, h = plan.recv[p]
            if h > l:
                ops.append(dist.P2POp(dist.irecv, self.x_full[l:h], p, group=self.group))
        return ops

    def start(self):
        """Begin filling x_full (asynchronous on the communication stream); returns the work handles.
        x_local must already hold the fresh slice."""
        if self.world == 1 and self.comm is None:
            return []
        dist, plan = self.dist, self.plan
        if self.comm is not None and plan.mode in ("allgather", "halo"):
            return self._start_comm()
        if self.world == 1:
            return []
        if plan.mode == "peer":
            for c in self._pulls:  # one launch (16 ranges each) on the caller's stream; nothing to wait for
                c.launch()
            return []
        if self._host_transport:
            self.torch.cuda.current_stream(self.x_full.device).synchronize()
        if plan.mode == "allgather":
            n = self.hi - self.lo
            self._gather_in[:n].copy_(self.x_local)
            if self.uniform:
                return [dist.all_gather_into_tensor(self.x_full, self._gather_in, group=self.group, async_op=True)]
            # slices of different lengths: gather the padded pieces, then move each to its global position
            if self._gather_out is None:
                self._gather_out = self.torch.zeros(self.world * self.count, dtype=self.x_full.dtype, device=self.x_full.device)
            dist.all_gather_into_tensor(self._gather_out, self._gather_in, group=self.group)
            self._unpad_gathered()
            return []
        if self._ops is None:
            self._ops = self._build_ops()  # built once: the ranges never change
        if not self._ops:
            return []
        try:
            return dist.batch_isend_irecv(self._ops)
        except RuntimeError as e:  # a transport without point-to-point support: use the collective
            import sys
            print(f"[cusp-autotuned_amd] rank {self.rank}: point-to-point halo exchange failed ({e}); "
                  "falling back to all-gather", file=sys.stderr)
            self.plan.mode = "allgather"
            if self._gather_in is None:
                self._gather_in = self.torch.zeros(self.count, dtype=self.x_full.dtype, device=self.x_full.device)
            return self.start()

    def _start_comm(self):
        """The exchange through the product's communicator (cmi_allgather_* / cmi_allgatherv_* / cmi_halo_exchange_*): enqueued on the
        compute stream, nothing to wait for on the host."""
        plan = self.plan
        if plan.mode == "allgather":
            if self.uniform:  # in place: this rank's slice already sits at its place in the buffer
                self.comm.allgather(self.x_full[self.lo:], self.x_full, self.count)
            else:
                counts = [b - a for a, b in zip(self.offsets, self.offsets[1:])]
                self.comm.allgatherv(self.x_full[self.lo:], self.x_full, counts, self.offsets[:-1])
            return []
        if getattr(self, "_halo_args", None) is None:
            peers = [p for p in range(self.world) if p != self.rank and (plan.recv[p][1] > plan.recv[p][0] or plan.send[p][1] > plan.send[p][0])]
            self._halo_args = (peers, [plan.send[p][0] for p in peers], [plan.send[p][1] - plan.send[p][0] for p in peers],
                               [plan.recv[p][0] for p in peers], [plan.recv[p][1] - plan.recv[p][0] for p in peers])
        self.comm.halo_exchange(self.x_full, *self._halo_args)
        return []

    def _unpad_gathered(self):
        item = self.x_full.element_size()
        pieces = [(p * self.count, self.offsets[p], self.offsets[p + 1] - self.offsets[p]) for p in range(self.world) if p != self.rank]
        if self.x_full.is_cuda:
            if self._unpad is None:
                from . import binding as B
                src, dst = self._gather_out.data_ptr(), self.x_full.data_ptr()
                rs = [(src + a * item, dst + b * item, n * item) for a, b, n in pieces if n > 0]
                self._unpad = [B.CopyRanges(rs[i:i + B.MAX_COPY_RANGES]) for i in range(0, len(rs), B.MAX_COPY_RANGES)]
            for c in self._unpad:
                c.launch()
        else:
            for a, b, n in pieces:
                self.x_full[b:b + n].copy_(self._gather_out[a:a + n])

    @staticmethod
    def finish(works):
        for w in works:
            w.wait()  # RCCL: the compute stream waits for the communication stream, the host does not block

    def exchange(self):
        """Fill x_full with what this rank's rows need."""
        self.finish(self.start())


class ShardedCsr:
    """Row block [lo, hi) of a CSR matrix with global column indices + the x exchange.

    `local_multiply(x_full, y_local)` performs the single-GPU SpMV; the default calls the C-ABI
    through matrices.multiply.  (The CPU gloo tests inject a host stand-in to check the sharding and
    exchange logic without a GPU.)"""

    def __init__(self, A_local, num_cols, rank, world, mode="auto", group=None, local_multiply=None,
                 col_span=None, overlap=True, interior=None, offsets=None, comm=None, local_format=None):
        """local_format ("ell" | "dia" | "coo" | "hyb", round 4): the partition, the column window and the exchange are derived from
        the CSR block as always -- they are properties of its entries -- and the block is then converted ONCE into that format; the
        multiply is exchange + the format's single-GPU multiply on the rectangular block (cusp/distributed/matrix.h is the C++ twin)."""
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
                                         mode=mode, group=group, offsets=offsets, comm=comm)
        if A_local.num_rows != self.vec.hi - self.vec.lo:
            raise ValueError(f"rank {rank}: local block has {A_local.num_rows} rows, partition expects "
                             f"{self.vec.hi - self.vec.lo}")
        self.x_view = self.vec.x_full[:num_cols]
        self._custom = local_multiply is not None
        if local_multiply is None:
            from .matrices import multiply as _mul
            local_multiply = lambda x_full, y: _mul(self.A, x_full, y)  # noqa: E731
        self._mul = local_multiply
        self.torch = torch
        # Overlap (halo mode): rows [a, b) reference only this rank's own slice of x, so they are
        # multiplied WHILE the halo is in flight; the boundary rows [0, a) and [b, n) follow once it
        # has landed.  A row range of a CSR matrix is itself a CSR matrix (row offsets are absolute
        # positions in the shared column/value arrays), so the split needs no copy and no new kernel.
        self.interior = None
        self._cfg = None
        if overlap and world > 1 and self.vec.plan.mode == "halo" and not self._custom and self.vec.comm is None:
            self.interior = interior if interior is not None else self._interior_rows()
            from . import binding as B
            self._cfg = B.tuning_select(B.FORMAT_CSR, B.F64 if A_local.values.dtype == torch.float64 else B.F32,
                                        A_local.num_rows, A_local.num_cols, A_local.num_entries)
            # ... or what the block's plan made of the table entry, when that is a row-tile kernel that needs no plan to run
            # (stencil rows: the wave-tile kernel)
            if A_local.num_entries > 0 and A_local.values.is_cuda:
                planned = A_local.plan().config()
                if planned.kernel in (B.CSR_STREAM, B.CSR_STREAM_WAVE):
                    self._cfg = planned

        if local_format not in (None, "csr"):
            from .matrices import convert
            self.A = convert(A_local, local_format)  # rows x num_cols in the format, global column indices
            self.interior = None                      # (the interior / boundary split is a row range of CSR arrays)
            self.local_format = local_format
        else:
            self.local_format = "csr"

    def _interior_rows(self):
        """Largest middle block [a, b) of rows whose columns all lie in [lo, hi) (setup-time)."""
        torch, A = self.torch, self.A
        n = A.num_rows
        if n == 0 or A.num_entries == 0:
            return None
        lens = (A.row_offsets[1:] - A.row_offsets[:-1]).long()
        cols = A.column_indices.double()
        cmin = torch.segment_reduce(cols, "min", lengths=lens, unsafe=True)
        cmax = torch.segment_reduce(cols, "max", lengths=lens, unsafe=True)
        empty = lens == 0
        local = ((cmin >= self.vec.lo) & (cmax < self.vec.hi)) | empty
        bad = torch.nonzero(~local).flatten()
        if bad.numel() == 0:
            return (0, n)
        # boundary rows cluster at the two ends of a banded block: take the widest run between them
        first_bad, last_bad = int(bad[0]), int(bad[-1])
        mid = n // 2
        below = bad[bad < mid]
        above = bad[bad >= mid]
        a = int(below[-1]) + 1 if below.numel() else 0
        b = int(above[0]) if above.numel() else n
        del first_bad, last_bad
        return (a, b) if b - a > n // 2 else None

    @property
    def x_local(self):
        return self.vec.x_local

    def _rows(self, a, b, y_local):
        """y_local[a:b] = A[a:b, :] * x (a row range of the local block)."""
        from . import binding as B
        A = self.A
        if b <= a:
            return
        # launch shape chosen for the WHOLE local block (a row range passes the full column/value
        # arrays, so its own entry count is not what the tuning table should see)
        B.spmv_csr(b - a, A.num_cols, A.row_offsets[a:b + 1], A.column_indices, A.values, self.x_view, y_local[a:b],
                   cfg=self._cfg)

    def new_exchanged_vector(self):
        """Another full-length vector with this matrix's exchange plan and transport (collective: every
        rank calls it).  krylov.cg keeps its residual in one so that peers can pull its boundary values."""
        v = self.vec
        return ShardedVectorExchange(v.num_cols, v.rank, v.world, v._span[0], v._span[1], v.x_full.dtype, v.x_full.device,
                                     mode=v.plan.mode, group=v.group, offsets=v.offsets, comm=v.comm)

    def halo_ranges(self):
        """[lo, hi) of this rank's slice merged with the ranges it receives: the contiguous pieces of the
        full-length buffers this rank reads (one piece for a banded matrix)."""
        pieces = sorted([(self.vec.lo, self.vec.hi)] + [(l, h) for l, h in self.vec.plan.recv if h > l])
        out = [list(pieces[0])]
        for l, h in pieces[1:]:
            if l <= out[-1][1]:
                out[-1][1] = max(out[-1][1], h)
            else:
                out.append([l, h])
        return [tuple(r) for r in out]

    def multiply_dot(self, y_local, result, workspace, exchange=True):
        """y_local = A[lo:hi, :] * x and result[0] = <y_local, x_local> (this rank's part of <A p, p>:
        the caller all-reduces it).  One fused launch when the block is multiplied whole; the
        overlapped (interior / boundary) schedule keeps its three launches and adds a dot."""
        from . import binding as B
        A = self.A
        if (self.interior is None or not exchange) and not self._custom and self.vec.x_full.is_cuda and self.local_format == "csr":
            if exchange:
                self.vec.exchange()
            B.spmv_csr_dot(A.num_rows, A.num_cols, A.row_offsets, A.column_indices, A.values, self.x_view, y_local,
                           self.vec.x_local, result, workspace, plan=A.plan() if A.num_entries > 0 else None)
            return y_local
        self.multiply(y_local, exchange=exchange)
        B.blas_dotd(y_local, self.vec.x_local, result, workspace)  # result is a double whatever the vectors' type
        return y_local

    def multiply(self, y_local, exchange=True):
        """y_local = A[lo:hi, :] * x, with x's slices taken from every rank's x_local."""
        if not exchange:
            self._mul(self.x_view, y_local)
            return y_local
        if self.interior is None:
            self.vec.exchange()
            self._mul(self.x_view, y_local)
            return y_local
        a, b = self.interior
        works = self.vec.start()          # halo in flight ...
        self._rows(a, b, y_local)         # ... while the interior rows are multiplied
        self.vec.finish(works)
        self._rows(0, a, y_local)
        self._rows(b, self.A.num_rows, y_local)
        return y_local
