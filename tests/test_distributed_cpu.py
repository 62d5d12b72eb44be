"""CPU tests (-m "not gpu") of the N>1 path: two gloo ranks, row-block shards with global column
indices, the x exchange (all-gather and halo), and reassembly of y.  The local SpMV is injected (the
CPU oracle stands in for the HIP kernel, which needs a GPU) so what is checked here is the sharding
and exchange logic that bench.py --gpus N and the sharded CG run on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, m, n, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cusp_autotuned_amd as cmi
        import oracle
        orc = oracle.Oracle()
        N = m * n
        Ap, Aj, Ax = orc.poisson5pt_csr(m, n)
        offs = cmi.distributed.partition_rows(N, world)
        lo, hi = offs[rank], offs[rank + 1]
        lAp = (Ap[lo:hi + 1] - Ap[lo]).astype(np.int32)
        lAj, lAx = Aj[Ap[lo]:Ap[hi]], Ax[Ap[lo]:Ap[hi]]
        A = cmi.CsrMatrix(hi - lo, N, len(lAx), torch.from_numpy(lAp), torch.from_numpy(lAj.copy()),
                          torch.from_numpy(lAx.copy()))

        def local_multiply(x_full, y):  # host stand-in for cmi_spmv_csr_f64
            y.copy_(torch.from_numpy(orc.spmv_csr(lAp, lAj, lAx, x_full.numpy())))

        sh = cmi.distributed.ShardedCsr(A, N, rank, world, mode=mode, local_multiply=local_multiply)
        x = oracle.fill_x(N)
        sh.x_local.copy_(torch.from_numpy(x[lo:hi]))
        y = torch.full((hi - lo,), 10.0, dtype=torch.float64)
        sh.multiply(y)
        # a second multiply with a new x: the exchange must refresh every needed remote entry
        sh.x_local.mul_(-2.0)
        y2 = torch.empty_like(y)
        sh.multiply(y2)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), y=y.numpy(), y2=y2.numpy(), lo=lo, hi=hi,
                 mode=sh.vec.plan.mode, recv=sh.vec.plan.recv_values, ag=sh.vec.plan.allgather_values,
                 pieces=np.array(sh.halo_ranges()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world,m,n", [("allgather", 2, 12, 9), ("halo", 2, 12, 9), ("auto", 2, 37, 40),
                                            ("auto", 3, 5, 7), ("halo", 4, 6, 11)])
def test_sharded_spmv_gloo(tmp_path, orc, mode, world, m, n):
    import oracle
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, m, n, str(tmp_path)), nprocs=world, join=True)
    N = m * n
    Ap, Aj, Ax = orc.poisson5pt_csr(m, n)
    x = oracle.fill_x(N)
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    want2 = orc.spmv_csr(Ap, Aj, Ax, -2.0 * x)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert parts[0]["lo"] == 0 and parts[-1]["hi"] == N
    assert np.array_equal(np.concatenate([p["y"] for p in parts]), want)
    assert np.array_equal(np.concatenate([p["y2"] for p in parts]), want2)
    for r, p in enumerate(parts):
        # 5-pt Poisson: own slice + the halo on either side is ONE contiguous piece of the full-length buffer
        lo, hi = int(p["lo"]), int(p["hi"])
        assert p["pieces"].tolist() == [[max(lo - m, 0), min(hi + m, N)]], (r, p["pieces"])
        if mode != "auto":
            assert str(p["mode"]) == mode
        if str(p["mode"]) == "halo":
            assert int(p["recv"]) <= 2 * m  # 5-pt Poisson: at most m values from each neighbour
    if mode == "auto" and world == 2 and m == 37:
        assert all(str(p["mode"]) == "halo" for p in parts)  # 37 values vs 740 for the all-gather


def _irregular(seed=3, n=900):
    """Square matrix with a heavy tail of row lengths and a band + a few far entries (global CSR, numpy)."""
    rng = np.random.default_rng(seed)
    lens = np.minimum((rng.pareto(1.1, size=n) * 2).astype(np.int64) + 1, 200)
    lens[n // 3] = 0
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    cols = []
    for i, l in enumerate(lens):
        near = np.clip(i + rng.integers(-40, 41, size=int(l)), 0, n - 1)
        cols.append(np.sort(np.unique(near))[:l] if l else near)
        lens[i] = len(cols[-1])
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    Aj = np.concatenate(cols).astype(np.int32)
    Ax = rng.standard_normal(len(Aj))
    return n, Ap, Aj, Ax


def _worker_by_entries(rank, world, port, mode, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cusp_autotuned_amd as cmi
        import oracle
        orc = oracle.Oracle()
        N, Ap, Aj, Ax = _irregular()
        offs = cmi.distributed.partition_by_entries(Ap, world)
        lo, hi = offs[rank], offs[rank + 1]
        lAp = (Ap[lo:hi + 1] - Ap[lo]).astype(np.int32)
        lAj, lAx = Aj[Ap[lo]:Ap[hi]].copy(), Ax[Ap[lo]:Ap[hi]].copy()
        A = cmi.CsrMatrix(hi - lo, N, len(lAx), torch.from_numpy(lAp), torch.from_numpy(lAj), torch.from_numpy(lAx))

        def local_multiply(x_full, y):
            y.copy_(torch.from_numpy(orc.spmv_csr(lAp, lAj, lAx, x_full.numpy())))

        sh = cmi.distributed.ShardedCsr(A, N, rank, world, mode=mode, local_multiply=local_multiply, offsets=offs)
        x = oracle.fill_x(N)
        sh.x_local.copy_(torch.from_numpy(x[lo:hi]))
        y = torch.empty(hi - lo, dtype=torch.float64)
        sh.multiply(y)
        sh.x_local.mul_(3.0)
        y2 = torch.empty_like(y)
        sh.multiply(y2)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), y=y.numpy(), y2=y2.numpy(), lo=lo, hi=hi, nnz=len(lAx), mode=sh.vec.plan.mode,
                 uniform=sh.vec.uniform)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("halo", 2), ("allgather", 2), ("allgather", 3), ("auto", 3), ("halo", 4)])
def test_sharded_spmv_partitioned_by_entries(tmp_path, orc, mode, world):
    """Row blocks balanced by entries have different lengths: the halo plan takes any cut list, the all-gather
    pads to the longest slice and moves the pieces to their global positions."""
    import oracle
    mp.spawn(_worker_by_entries, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    N, Ap, Aj, Ax = _irregular()
    x = oracle.fill_x(N)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert np.array_equal(np.concatenate([p["y"] for p in parts]), orc.spmv_csr(Ap, Aj, Ax, x))
    assert np.array_equal(np.concatenate([p["y2"] for p in parts]), orc.spmv_csr(Ap, Aj, Ax, 3.0 * x))
    nnz = [int(p["nnz"]) for p in parts]
    rows = [int(p["hi"]) - int(p["lo"]) for p in parts]
    assert max(nnz) - min(nnz) <= 2 * 200 and len(set(rows)) > 1, (nnz, rows)   # entries balanced (to a row), rows not
    assert not any(bool(p["uniform"]) for p in parts)


def test_partition_by_entries():
    import cusp_autotuned_amd as cmi
    f = cmi.distributed.partition_by_entries
    assert f([0, 1, 2, 3, 4], 2) == [0, 2, 4]
    assert f([0, 10, 10, 10, 11], 2) == [0, 1, 4]                 # one heavy row: it gets a block of its own
    assert f([0, 0, 0, 0, 0], 3) == [0, 0, 0, 4]                  # empty matrix: everything in the last block
    assert f([0, 5], 4) == [0, 1, 1, 1, 1]                        # fewer rows than ranks
    assert f(np.array([0, 2, 4, 6, 8, 10]), 5) == [0, 1, 2, 3, 4, 5]


def test_partition_rows():
    import cusp_autotuned_amd as cmi
    assert cmi.distributed.partition_rows(10, 4) == [0, 3, 6, 9, 10]
    assert cmi.distributed.partition_rows(8, 8) == list(range(9))
    assert cmi.distributed.partition_rows(3, 4) == [0, 1, 2, 3, 3]
    assert cmi.distributed.partition_rows(9998244 * 8, 8)[1] == 9998244
