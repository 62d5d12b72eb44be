"""Conjugate gradients on device vectors, single GPU or row-block sharded -- the caller of the SpMV
hot path (reference cusp/krylov/detail/cg.inl:41-107, cusp/detail/monitor.inl).  Same operations in
the same order as the reference, so the residual history matches (docs/quickstart.md:72-87).

Every vector operation is a C-ABI call (cmi_blas_*_f64, cmi_spmv_*); Python only sequences them.
Sharded (ShardedCsr): every rank holds its slice of x, b and of the four work vectors; the search
direction p lives INSIDE the rank's x-exchange buffer, so `y <- A p` is exchange + local SpMV with no
extra copy; the two dot products and the norm per iteration are 8-byte all-reduces (RCCL), as
SURVEY.md section 8(e) lays out.
"""
import math


class Monitor:
    """cusp::monitor (reference cusp/monitor.h:118-250): stop when ||r|| <= abs + rel*||b|| or at the
    iteration limit; keeps the residual history."""

    def __init__(self, b_norm, iteration_limit=500, relative_tolerance=1e-5, absolute_tolerance=0.0, verbose=False):
        self.b_norm = float(b_norm)
        self.r_norm = float("inf")
        self.iteration_limit = int(iteration_limit)
        self.iteration_count = 0
        self.relative_tolerance = float(relative_tolerance)
        self.absolute_tolerance = float(absolute_tolerance)
        self.verbose = verbose
        self.residuals = []

    def tolerance(self):
        return self.absolute_tolerance + self.relative_tolerance * self.b_norm

    def converged(self):
        return self.r_norm <= self.tolerance()

    def finished(self, r_norm):
        self.r_norm = float(r_norm)
        self.residuals.append(self.r_norm)
        if self.verbose:
            print(f"       {self.iteration_count:10d}       {self.r_norm:10.6e}")
        return self.converged() or self.iteration_count >= self.iteration_limit

    def increment(self):
        self.iteration_count += 1


class _Ops:
    """dot / norm with the cross-rank reduction folded in."""

    def __init__(self, device, group, world, comm=None):
        import torch
        from . import binding as B
        self.B, self.torch = B, torch
        self.ws = B.blas_workspace(device)
        self.res = torch.zeros(1, dtype=torch.float64, device=device)
        self.group, self.world, self.comm = group, world, comm

    def reduce(self, t):
        """sum of a float64 device scalar over the ranks, in place: cmi_allreduce_f64 through the product's communicator when the
        sharded operator carries one (one process per GPU), torch.distributed in the rehearsals"""
        if self.comm is not None:
            self.comm.allreduce(t)
        elif self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, group=self.group)

    def dot(self, x, y):
        self.B.blas_dotd(x, y, self.res, self.ws)  # a double in device memory, f64 or f32 vectors
        self.reduce(self.res)
        return float(self.res.item())

    def nrm2(self, x):
        return math.sqrt(self.dot(x, x))


def cg(A, x, b, monitor=None, iteration_limit=500, relative_tolerance=1e-5, absolute_tolerance=0.0, group=None,
       fused=True):
    """Solve A x = b (A symmetric positive definite).  A: a matrices.* container (one GPU) or a
    distributed.ShardedCsr (x, b = this rank's slices).  x holds the initial guess and the result.
    Returns the Monitor (residual history in .residuals).

    fused=True (default; f64 and f32): the unpreconditioned iteration runs as SpMV(+dot) + cmi_cg_update +
    cmi_cg_direction_x with alpha / beta kept in device memory (as doubles) -- 3-4 launches and ONE host read per
    iteration (the convergence check, hidden behind the next SpMV) instead of the reference's 7 passes
    and 3 host syncs; the per-element arithmetic is unchanged.  fused=False replays cg.inl operation
    by operation."""
    import torch
    from . import binding as B
    from .distributed import ShardedCsr
    from .matrices import multiply

    sharded = isinstance(A, ShardedCsr)
    world = A.world if sharded else 1
    n = x.numel()
    dev = x.device
    ops = _Ops(dev, group, world, comm=A.vec.comm if sharded else None)
    if monitor is None:
        monitor = Monitor(ops.nrm2(b), iteration_limit, relative_tolerance, absolute_tolerance)

    if x.dtype not in (torch.float64, torch.float32) or b.dtype != x.dtype:
        raise TypeError(f"cg: x and b must both be float64 or float32 device vectors, got {x.dtype} / {b.dtype}")
    y = torch.empty(n, dtype=x.dtype, device=dev)
    z = torch.empty_like(y)
    r = torch.empty_like(y)
    # the search direction lives in the exchange buffer when sharded: A p needs no staging copy
    p = A.x_local if sharded else torch.empty_like(y)

    one_sided = sharded and A.vec.plan.mode == "peer"

    def spmv(v, out):  # out <- A v
        if sharded:
            if one_sided:
                A.vec.fence()   # the peers have finished pulling the previous contents ...
            if v.data_ptr() != A.x_local.data_ptr():
                B.blas_copy(v, A.x_local)
            if one_sided:
                A.vec.fence()   # ... and every slice is complete before anyone pulls (correct, not fast:
            A.multiply(out)     #     the fused path below needs no fence inside its loop)
        else:
            multiply(A, v, out)

    if fused:
        return _cg_fused(A, x, b, monitor, ops, spmv, y, r, p, world, group)

    spmv(x, y)                                   # y <- A x            (cg.inl:63)
    B.blas_axpby(1.0, b, -1.0, y, r)             # r <- b - A x        (:66)
    B.blas_copy(r, z)                            # z <- M r, M = I     (:69, linear_operator.h:204-208)
    B.blas_copy(z, p)                            # p <- z              (:72)
    rz = ops.dot(r, z)                           # rz = <r, z>         (:75)
    while not monitor.finished(ops.nrm2(r)):     # monitor.inl:181-207
        spmv(p, y)                               # y <- A p            (:80)  THE HOT PATH
        alpha = rz / ops.dot(y, p)               # (:83)
        B.blas_axpy(alpha, p, x)                 # x <- x + alpha p    (:86)
        B.blas_axpy(-alpha, y, r)                # r <- r - alpha y    (:89)
        B.blas_copy(r, z)                        # z <- M r            (:92)
        rz_old = rz
        rz = ops.dot(r, z)                       # (:97)
        beta = rz / rz_old
        B.blas_axpby(1.0, z, beta, p, p)         # p <- z + beta p     (:103)
        monitor.increment()                      # (:105)
    return monitor


class _SyncScalar:
    """Host transport stand-in for binding.HostScalar when the vectors are not in HBM (CPU tests)."""

    def fetch(self, t):
        self.src = t

    def wait(self):
        return float(self.src.item())


def _cg_fused(A, x, b, monitor, ops, spmv, y, r, p, world, group):
    """Unpreconditioned CG with z == r folded away and the scalars resident on the device.

    Per iteration: y <- A p together with <y, p> (one launch for CSR: cmi_spmv_csr_dot_f64), then
    cmi_cg_update (r, <r,r>), then cmi_cg_direction_x (x, p).  The single host read (the monitor's residual
    norm) lands in page-locked memory (written by the reduction itself on one GPU, an asynchronous
    copy behind the all-reduce when sharded) that the host waits for only AFTER queueing the next
    iteration's SpMV: that SpMV reads p and writes the scratch y -- no solver state -- so it
    is harmless if the monitor then stops, and the device never idles on the host round trip."""
    import torch
    from . import binding as B
    from .distributed import ShardedCsr
    from .matrices import CooMatrix, CsrMatrix, DiaMatrix, EllMatrix, HybMatrix
    dev = x.device
    rr = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(2)]  # <r,r> ping-pong
    yp = torch.zeros(1, dtype=torch.float64, device=dev)
    rr_host = B.HostScalar() if dev.type == "cuda" else _SyncScalar()
    mirror = rr_host if world == 1 and dev.type == "cuda" else None  # sharded: <r,r> is all-reduced first

    reduce_ = ops.reduce

    # fold-ahead (single GPU, CSR through its plan): the partials of <y,p> stay in the workspace and cmi_cg_update_fold_* folds
    # them itself, <r,r> likewise at the front of the direction kernel: three launches per iteration.  Opt-in
    # ($CMI_CG_FOLD_AHEAD=1): measured -1.4 % per iteration, inside the box-to-box spread
    import os
    fold_ahead = (world == 1 and isinstance(A, CsrMatrix) and p.is_cuda and A.num_entries > 0 and os.environ.get("CMI_CG_FOLD_AHEAD", "0") == "1")

    def spmv_partials():                         # y <- A p; returns the number of <y,p> partials left in the workspace (0: none)
        return B.spmv_csr_dot_partials(A.plan(), A.row_offsets, A.column_indices, A.values, p, y, p, ops.ws)

    def spmv_dot():                              # y <- A p, yp <- <y, p>
        if isinstance(A, ShardedCsr):            # CSR / ELL / DIA fuse the dot for f64 and f32 (the scalar is a double either way);
                                                 # everything else: SpMV, then a dot into a double
            A.multiply_dot(y, yp, ops.ws)        # p IS A.x_local
        elif isinstance(A, CsrMatrix) and p.is_cuda:
            B.spmv_csr_dot(A.num_rows, A.num_cols, A.row_offsets, A.column_indices, A.values, p, y, p, yp, ops.ws,
                           plan=A.plan() if A.num_entries > 0 else None)
        elif isinstance(A, CooMatrix) and p.is_cuda and A.num_entries > 0:   # sorted entries: the CSR kernel's fused dot on the plan's offsets
            B.spmv_coo_dot_plan(A.plan(), A.row_indices, A.column_indices, A.values, p, y, p, yp, ops.ws)
        elif isinstance(A, HybMatrix) and A.coo.num_entries == 0 and p.is_cuda:   # everything in the ELL part: its fused dot
            e = A.ell
            B.spmv_ell_dot(A.num_rows, A.num_cols, e.num_entries_per_row, e.pitch, e.column_indices, e.values, p, y, p, yp, ops.ws)
        elif isinstance(A, HybMatrix) and p.is_cuda and A.plan() is not None:   # one launch through the plan: its fused dot
            B.spmv_hyb_dot_plan_args(A._plan_args, p, y, p, yp, ops.ws)
        elif isinstance(A, EllMatrix) and p.is_cuda:
            B.spmv_ell_dot(A.num_rows, A.num_cols, A.num_entries_per_row, A.pitch, A.column_indices, A.values, p, y, p, yp, ops.ws,
                           row_lengths=A.row_lengths)
        elif isinstance(A, DiaMatrix) and p.is_cuda:
            B.spmv_dia_dot(A.num_rows, A.num_cols, A.diagonal_offsets.numel(), A.pitch, A.diagonal_offsets, A.values, p, y, p, yp, ops.ws)
        else:
            spmv(p, y)
            B.blas_dotd(y, p, yp, ops.ws)
        reduce_(yp)

    # One-sided sharding ("peer" exchange): p is exchanged ONCE.  Afterwards every rank keeps the p
    # entries of its halo itself: p_halo <- r_halo + beta p_halo, the same arithmetic with the same
    # all-reduced beta as the owner, so the same bits -- and r_halo is PULLED from the owner right
    # after the all-reduce of <r,r>, which completes only when every rank's cg_update (the kernel that
    # wrote that r) has; the owner overwrites r again only after the next all-reduce of <y,p>, which
    # needs this rank's contribution, queued behind this pull.  The algorithm's own all-reduces are
    # all the ordering the one-sided pulls need: no collective is added, no exchange per iteration.
    one_sided = isinstance(A, ShardedCsr) and A.vec.plan.mode == "peer"
    if one_sided:
        r_ex = A.new_exchanged_vector()
        r = r_ex.x_local
        p_full, r_full = A.vec.x_full, r_ex.x_full
        p = A.x_local                                               # (a view of p_full; r likewise of r_full)
        halo_only = []                                              # the pieces of the full-length buffers this rank reads
        for l, h in A.halo_ranges():                                # ... minus its own slice
            if l < A.vec.lo:
                halo_only.append((l, min(h, A.vec.lo)))
            if h > A.vec.hi:
                halo_only.append((max(l, A.vec.hi), h))

    spmv(x, y)                                   # y <- A x
    B.blas_axpby(1.0, b, -1.0, y, r)             # r <- b - A x
    if one_sided:
        A.vec.fence()                            # (setup) the peers have pulled x out of the buffer p reuses
    B.blas_copy(r, p)                            # p <- z = r
    B.blas_dotd(r, r, rr[0], ops.ws)             # rz = <r, r>
    reduce_(rr[0])
    if one_sided:
        A.vec.exchange()                         # p halos, ordered behind every rank's copy by the all-reduce
    rr_host.fetch(rr[0])
    cur = 0
    while True:
        if one_sided:
            A.multiply_dot(y, yp, ops.ws, exchange=False)           # halos of p are already in place
            reduce_(yp)
        elif fold_ahead:
            np_yp = spmv_partials()              # THE HOT PATH (queued before the host waits)
            if np_yp == 0:                       # this plan's kernel cannot fuse the dot: y is there, add the dot
                B.blas_dotd(y, p, yp, ops.ws)
        else:
            spmv_dot()                           # THE HOT PATH (queued before the host waits)
        if monitor.finished(math.sqrt(rr_host.wait())):             # the one host read per iteration
            break
        if fold_ahead and np_yp > 0:
            np_rr = B.cg_update_fold(rr[cur], yp, np_yp, y, r, ops.ws)
            B.cg_direction_x_fold(rr[cur ^ 1], np_rr, rr[cur], yp, r, p, x, ops.ws, mirror=mirror)
            cur ^= 1
            monitor.increment()
            continue
        B.cg_update(rr[cur], yp, None, y, None, r, rr[cur ^ 1], ops.ws, mirror=mirror)   # r, <r,r> in one pass
        if mirror is None:
            reduce_(rr[cur ^ 1])
            rr_host.fetch(rr[cur ^ 1])
        # x <- x + alpha p rides with the direction pass (which reads p anyway): 8 vector passes per iteration, not 9
        if one_sided:
            r_ex.exchange()                                         # pull the peers' boundary r
            for l, h in halo_only:                                  # halo: p <- r + beta p (the owner's arithmetic)
                B.cg_direction(rr[cur ^ 1], rr[cur], r_full[l:h], p_full[l:h])
        B.cg_direction_x(rr[cur ^ 1], rr[cur], yp, r, p, x)         # own slice: x <- x + alpha p; p <- r + beta p
        cur ^= 1
        monitor.increment()
    if one_sided:
        A.vec.fence()                                               # nobody is still pulling from r_ex
        r_ex.close()
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()                # the discarded SpMV must not outlive y
    return monitor
