#!/usr/bin/env python3
"""BASELINE.json configs[4]'s PER-RANK shape on ONE GPU: what one of the eight ranks of
`poisson5pt 1e8 rows, row-block sharded across 8 x MI355X, RCCL all-gather of x, inside cusp::krylov::cg` holds and runs --

  * the rank's block of poisson5pt(10000, 10000): 1250 grid lines = 1.25e7 rows, GLOBAL column indices, multiplied from a full-length x
    buffer of 1e8 entries (800 MB): the local SpMV alone, for the first, a middle and the last rank (time, GB/s on the block's
    algorithmic bytes 12 nnz + 20 rows + 4, bit-exact against the stencil's closed form);
  * the all-gather INTO that buffer through the product's communicator (cmi_comm = RCCL behind the C-ABI) with ONE rank: the call the
    8-GPU run makes with count = 1.25e7, in place -- on one rank it moves nothing, so this is RCCL's per-call cost, the floor under the
    0.65 ms the xGMI links need for the real thing;
  * a CG iteration at the rank's size: poisson5pt(10000, 1250) (the rank's diagonal block: 1.25e7 rows, the same band of 10000) solved
    by cmi.krylov.cg through a ShardedCsr on that communicator -- every dot product an RCCL all-reduce, the exchange an RCCL all-gather.

Nothing here is an 8-GPU measurement: no byte crosses a link.  It pins the per-rank compute and the per-call software floor, so that
the driver's 8-GPU run has one unknown left (the links)."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (stencil_expected: the closed form the sharded legs are validated against)


def events_us(fn, reps, batches=5):
    lib = cmi.lib()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    cmi.check(lib.cmi_event_create(ctypes.byref(e0)))
    cmi.check(lib.cmi_event_create(ctypes.byref(e1)))
    t_end = time.time() + 0.05
    while time.time() < t_end:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    out = []
    for _ in range(batches):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        cmi.check(lib.cmi_event_record(e0, s))
        for _ in range(reps):
            fn()
        cmi.check(lib.cmi_event_record(e1, s))
        ms = ctypes.c_float()
        cmi.check(lib.cmi_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        out.append(ms.value / reps * 1e3)
    return float(np.median(out)), float(min(out))


def main():
    g = int(os.environ.get("CONFIGS4_GRID", "10000"))
    world = 8
    lines = -(-g // world)
    N = g * g
    dev = torch.device("cuda", 0)
    out = {"grid": g, "rows_global": N, "ranks_emulated": world, "rows_per_rank": lines * g, "device": torch.cuda.get_device_name(0)}
    x_full = cmi.fill_x(N, torch.float64, "cpu").to(dev)  # the rank's full-length x buffer (every rank holds one)
    print(f"# poisson5pt({g},{g}): {N} rows; rank blocks of {lines} grid lines = {lines * g} rows; x buffer {N * 8 / 1e6:.0f} MB", flush=True)
    legs = []
    for rank in (0, world // 2, world - 1):
        l0, l1 = min(rank * lines, g), min((rank + 1) * lines, g)
        lo, hi = l0 * g, l1 * g
        A = cmi.poisson5pt(g, g, "csr", dtype=torch.float64, device=dev, row_begin=lo, row_end=hi)
        y = torch.full((hi - lo,), 10.0, dtype=torch.float64, device=dev)
        cmi.multiply(A, x_full, y)
        want = bench.stencil_expected(torch, cmi, g, g, lo, hi, dev)
        exact = bool(torch.equal(y, want))
        del want
        med, best = events_us(lambda: cmi.multiply(A, x_full, y), 20)
        alg = cmi.csr_bytes(A.num_rows, A.num_entries)
        rec = {"rank": rank, "rows": A.num_rows, "entries": A.num_entries, "kernel": A.plan().config().as_dict(), "spmv_us": round(med, 2), "spmv_us_min": round(best, 2),
               "algorithmic_bytes": alg, "gbps": round(alg / med / 1e3, 1), "frac_of_8TBps": round(alg / med / 8e6, 4), "gflops": round(2.0 * A.num_entries / med / 1e3, 1),
               "bit_exact_vs_stencil_closed_form": exact}
        legs.append(rec)
        print(f"rank {rank} of {world}: block {A.num_rows} rows x {N} columns, {A.num_entries} entries: local SpMV {med:8.1f} us (fastest batch {best:.1f}) = "
              f"{rec['gbps']} GB/s = {rec['frac_of_8TBps']} of 8 TB/s, {rec['gflops']} GFLOP/s; bit-exact {exact}; {A.plan().config()}", flush=True)
        del A, y
        torch.cuda.empty_cache()
    out["local_spmv"] = legs
    # ---- the all-gather into the buffer through the product's communicator, ONE rank ----
    comm = cmi.binding.Comm(0, 1)
    count = lines * g
    xl = x_full[:count]
    med, best = events_us(lambda: comm.allgather(xl, x_full, count), 20)
    out["allgather_one_rank"] = {"count": count, "us": round(med, 2), "us_min": round(best, 2), "rccl_version": comm.library_version(),
                                 "note": "in place, world 1: no byte moves -- RCCL's per-call cost through cmi_allgather_f64"}
    print(f"cmi_allgather_f64 (RCCL {comm.library_version()}, 1 rank, in place, count {count}): {med:.1f} us per call (fastest batch {best:.1f})", flush=True)
    s = torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev)
    med, best = events_us(lambda: comm.allreduce(s), 50)
    out["allreduce_one_rank"] = {"doubles": 2, "us": round(med, 2), "us_min": round(best, 2)}
    print(f"cmi_allreduce_f64 (1 rank, 2 doubles): {med:.1f} us per call", flush=True)
    del x_full
    torch.cuda.empty_cache()
    # ---- CG at the rank's size through that communicator ----
    A = cmi.poisson5pt(g, lines, "csr", dtype=torch.float64, device=dev)  # the diagonal block of a middle rank: same rows, same band
    n = A.num_rows
    sh = cmi.distributed.ShardedCsr(A, n, 0, 1, mode="allgather", comm=comm)
    b = cmi.fill_x(n, torch.float64, "cpu").to(dev)
    x0 = torch.zeros(n, dtype=torch.float64, device=dev)
    cmi.krylov.cg(sh, x0.clone(), b, iteration_limit=3, relative_tolerance=0.0)
    res = {}
    for its in (50, 100):
        xs = x0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mon = cmi.krylov.cg(sh, xs, b, iteration_limit=its, relative_tolerance=0.0)
        torch.cuda.synchronize()
        res[its] = (time.perf_counter() - t0, mon.iteration_count, mon.residuals[-1])
    marg = (res[100][0] - res[50][0]) / 50 * 1e6
    # the same solve on the plain single-GPU path (no communicator): what the RCCL calls add per iteration
    xs = x0.clone()
    cmi.krylov.cg(A, xs.clone(), b, iteration_limit=3, relative_tolerance=0.0)
    plain = {}
    for its in (50, 100):
        xs = x0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mon2 = cmi.krylov.cg(A, xs, b, iteration_limit=its, relative_tolerance=0.0)
        torch.cuda.synchronize()
        plain[its] = (time.perf_counter() - t0, mon2.iteration_count, mon2.residuals[-1])
    marg_plain = (plain[100][0] - plain[50][0]) / 50 * 1e6
    out["cg_at_rank_size"] = {"matrix": f"poisson5pt({g},{lines}): {n} rows, {A.num_entries} entries", "through_comm_us_per_marginal_iteration": round(marg, 2),
                              "through_comm_us_per_iteration_100": round(res[100][0] / 100 * 1e6, 2), "single_gpu_path_us_per_marginal_iteration": round(marg_plain, 2),
                              "same_residual_after_100": bool(abs(res[100][2] - plain[100][2]) <= 1e-9 * abs(plain[100][2]))}
    print(f"CG at the rank's size, poisson5pt({g},{lines}) = {n} rows: through the 1-rank communicator {marg:.1f} us per marginal iteration "
          f"({res[100][0] / 100 * 1e6:.1f} whole solve / 100); single-GPU path {marg_plain:.1f}; residual after 100: {res[100][2]:.6e} vs {plain[100][2]:.6e}", flush=True)
    sh.vec.close()
    comm.close()
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
