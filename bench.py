#!/usr/bin/env python3
"""bench.py -- SpMV GFLOP/s + achieved HBM GB/s (fp64) on 5-point Poisson, the metric of BASELINE.json.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic input = one y = A*x through the
C-ABI (cmi_spmv_csr_f64) with A, x, y resident in HBM.

N = 1: BASELINE.json configs[1], poisson5pt 3162x3162 (9 998 244 rows, 49 978 572 entries), CSR,
       int32/f64, kernel + launch shape from the persisted tuning table.
N > 1: weak scaling -- every rank owns a 3162x3162-point row block of the global
       poisson5pt(3162, 3162*N) (N=8: 8.0e7 rows; BASELINE.json configs[4] shape), global column
       indices; a step = exchange of x over xGMI + the local SpMV.  Default exchange ("auto"): the
       one-sided halo pull -- each rank copies the 2*3162 boundary values its rows reference straight
       out of its neighbours' mapped buffers with one small kernel on its own stream ("peer"); if the
       buffers cannot be mapped, the two-sided RCCL halo exchange ("halo"); "allgather" on request
       (cusp-autotuned_amd/distributed.py).  value = 2*global_nnz / max-over-ranks time.  The north-star's
       literal exchange (RCCL all-gather of the whole x before each multiply) is timed beside it on up
       to 20 steps and reported as `allgather_exchange` (never as `value`).

Timing protocol (reference performance/spmv/benchmark.h:84-120): W untimed warm-up steps, then
exactly K steps between a barrier + device synchronise on both sides; MAX over ranks.
The dominant kernel's average launch duration is measured live with HIP events on the stream the
kernel is launched on (cmi_event_*), for the roofline object.

The `cpu_baseline` leg (rank 0, N = 1 only) times the REFERENCE's own sequential host kernel
(oracle/_ref, kind "reference") -- or the C restatement (kind "port") when that library is absent --
on the same matrix for ~10-20 s, and also checks the GPU result against it.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC (needed by RCCL and by the one-sided exchange's buffer mapping);
# normally exported already -- set before the HIP runtime is loaded in case it is not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

M = 3162  # grid side: 3162^2 = 9 998 244 rows per GPU
# --format hyb: ELL width of the split.  4 leaves the fifth entry of every interior row (9 985 596 entries) to the COO part,
# so both kernels of the HYB multiply are timed (5 = the longest row would be ELL alone; the tuned width rule is
# cmi_hyb_entries_per_row, tools/autotune_hyb.py)
HYB_WIDTH = 4
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cg-iterations", type=int, default=100, help="iterations of the secondary CG leg (0 = skip)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "peer", "halo", "allgather"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--format", default="csr", choices=["csr", "ell", "dia", "coo", "hyb"],
                    help="N=1 only: bench another format of the same matrix (BASELINE.json configs[2])")
    return ap.parse_args()


def pmc_traffic(kernel_substr):
    """(HBM-side bytes per launch of the dominant kernel, source file) from the newest committed rocprofv3 --pmc summary
    that has this kernel (profiles/*pmc*.json, produced by tools/pmc_summary.py from separate --pmc passes over the same
    matrix and tuning-table config), or (None, None).  PMC collection needs the profiler around the process, so the
    figure is a committed measurement of this kernel, not one taken in this run: the line says so (`traffic_source`)."""
    pdir = os.path.join(ROOT, "profiles")
    best, src = None, None
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith(".json") and "pmc" in f:
                try:
                    doc = json.load(open(os.path.join(pdir, f)))
                    for k in doc.get("kernels", []):
                        if kernel_substr in k.get("kernel", "") and k.get("hbm_bytes_per_launch"):
                            best, src = float(k["hbm_bytes_per_launch"]), "profiles/" + f
                except Exception:
                    pass
    return best, src


def cpu_baseline(cmi, A, x_host, y_gpu_host, seconds):
    """Times the reference's sequential host SpMV on the same matrix; checks the GPU result."""
    import numpy as np
    import oracle
    Ap, Aj, Ax = (t.cpu().numpy() for t in (A.row_offsets, A.column_indices, A.values))
    nnz = A.num_entries
    orc = oracle.Oracle()
    kind = "reference" if oracle.have_reference() else "port"
    if kind == "reference":
        refl = oracle.Reference()
        run = lambda: refl.spmv_csr(A.num_cols, Ap, Aj, Ax, x_host)  # noqa: E731
    else:
        run = lambda: orc.spmv_csr(Ap, Aj, Ax, x_host)  # noqa: E731
    y = run()  # warm-up (first touch)
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t0 < seconds and reps < 500):
        y = run()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    exact = bool(np.array_equal(y, y_gpu_host))
    max_rel = float(np.max(np.abs(y - y_gpu_host)) / max(float(np.max(np.abs(y))), 1e-300))
    out = {"value": round(2.0 * nnz / dt / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": kind,
           "sample": f"{reps} full SpMVs of the same poisson5pt {M}x{M} CSR fp64 matrix ({dt * 1e3:.1f} ms each), "
                     "single thread, -O3 -ffp-contract=off",
           "ms_per_spmv": round(dt * 1e3, 3), "gpu_result_bit_exact": exact, "gpu_max_rel_err": max_rel}
    # The multi-core baseline: the OpenMP row-parallel kernel (reference omp/detail/multiply/csr_spmv.h:51-86, restated --
    # that header cannot be compiled here, DESIGN.md section 5 -- so kind "port"), on copies first-touched in parallel by the
    # threads that use them, one pinned thread per allowed CPU (at most 16: the CPU share of a one-GPU box), timed in C.
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    threads = max(1, min(allowed, 16))
    try:
        numa_nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        numa_nodes = None
    r = orc.bench_csr_omp(A.num_cols, Ap, Aj, Ax, x_host, threads=threads, seconds=max(seconds / 3, 2.0))
    dt2 = r["seconds_per_spmv"]
    omp_exact = bool(np.array_equal(r["y"], y))
    omp = {"value": round(2.0 * nnz / dt2 / 1e9, 4), "unit": "GFLOP/s", "cores": r["threads"], "kind": "port",
           "sample": f"{r['reps']} full SpMVs of the same matrix ({dt2 * 1e3:.2f} ms each), OpenMP static row split, "
                     "gcc -O3 -ffp-contract=off",
           "ms_per_spmv": round(dt2 * 1e3, 3), "threads_pinned": r["pinned"], "cpus": r["cpus"],
           "nproc": os.cpu_count(), "cpus_allowed": allowed, "numa_nodes": numa_nodes,
           "numa_policy": "default (local) allocation; every array first-touched in parallel by the thread that reads or "
                          "writes it (static row blocks; x by slices)",
           "identical_to_sequential_reference": omp_exact}
    if not (exact or max_rel <= 1e-6):
        raise SystemExit(f"parity gate failed: GPU y differs from the CPU reference (max rel {max_rel})")
    return out, omp


def stencil_expected(torch, cmi, m, n, lo, hi, dev, scale=1.0):
    """Rows [lo, hi) of y = poisson5pt(m, n) * x for the bench's x (cmi.fill_x), from the stencil itself:
    no matrix, no exchange, no kernel of the library.  Same arithmetic as the reference host loop
    (sequential/multiply/csr_spmv.h:60-72) on the gallery layout (columns ascending: i-m, i-1, i, i+1, i+m;
    values -1,-1,4,-1,-1): sum = 0, then sum = sum + a*x in storage order -- bit-exact in fp64 because -1*x
    and 4*x are exact.  Used to validate an N>1 exchange before it is timed and by the 1e8-row test."""
    N = m * n
    e0, e1 = max(lo - m, 0), min(hi + m, N)
    xe = (cmi.fill_x(e1 - e0, start=e0) * scale).to(dev)          # this rank's rows +- one grid line
    i = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    ix = i % m
    zero = torch.zeros((), dtype=torch.float64, device=dev)

    def term(off, mask, coef):
        j = (i + off - e0).clamp_(0, e1 - e0 - 1)
        return torch.where(mask, coef * xe[j], zero)

    s = torch.zeros(hi - lo, dtype=torch.float64, device=dev)
    s = s + term(-m, i >= m, -1.0)
    s = s + term(-1, ix > 0, -1.0)
    s = s + term(0, i >= 0, 4.0)
    s = s + term(1, ix < m - 1, -1.0)
    s = s + term(m, i + m < N, -1.0)
    return s


def main():
    args = parse()
    import torch
    import cusp_autotuned_amd as cmi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                             f"python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # CMI_BENCH_REHEARSAL=1 (tests only): all ranks share GPU 0 and talk over gloo, so that the N>1
    # code path can be exercised end to end on a one-GPU box.  RCCL refuses two ranks on one device.
    rehearsal = os.environ.get("CMI_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    lib = cmi.lib()
    stream = torch.cuda.current_stream()
    sptr = ctypes.c_void_p(stream.cuda_stream)

    # ---- workload --------------------------------------------------------------------------
    m, n = M, M * world
    rows_per_rank = M * M
    lo, hi = rank * rows_per_rank, (rank + 1) * rows_per_rank
    A = cmi.poisson5pt(m, n, "csr", dtype=torch.float64, device=dev, row_begin=lo, row_end=hi)
    N_global = m * n
    nnz_global = cmi.poisson5pt_num_entries(m, n)
    fmt = args.format if world == 1 else "csr"
    Afmt = A if fmt == "csr" else (cmi.poisson5pt(m, n, "dia", device=dev) if fmt == "dia" else
                                   cmi.convert(A, fmt, num_entries_per_row=HYB_WIDTH if fmt == "hyb" else None))
    # deterministic, RNG-free input; each rank generates only its own slice
    x_host = cmi.fill_x(rows_per_rank, start=lo).numpy()
    y = torch.full((rows_per_rank,), 10.0, dtype=torch.float64, device=dev)

    if world == 1:
        x = torch.from_numpy(x_host).to(dev)
        step = lambda: cmi.multiply(Afmt, x, y)  # noqa: E731
        exchange_info = None
    else:
        # The exchange is validated BEFORE anything is timed: one sharded multiply against the stencil's closed
        # form (bit-exact, no other exchange involved).  A transport that maps its neighbours' buffers but
        # delivers wrong halos is dropped for the next one (peer -> halo -> allgather), on every rank alike.
        span = (max(lo - m, 0), min(hi + m, N_global) - 1)
        want = stencil_expected(torch, cmi, m, n, lo, hi, dev)
        rejected = []
        sh = None
        for mode_try in {"auto": ["auto", "halo", "allgather"], "peer": ["peer", "halo", "allgather"],
                         "halo": ["halo", "allgather"], "allgather": ["allgather"]}[args.exchange]:
            if sh is not None:
                sh.vec.close()
            sh = cmi.distributed.ShardedCsr(A, N_global, rank, world, mode=mode_try, col_span=span)
            sh.x_local.copy_(torch.from_numpy(x_host).to(dev))
            sh.vec.fence()
            y.fill_(10.0)
            sh.multiply(y)
            good = torch.tensor([int(torch.equal(y, want))], dtype=torch.int32, device=dev)
            if sh.vec.plan.mode in os.environ.get("CMI_BENCH_REJECT", "").split(","):
                good.zero_()  # tests only: pretend this exchange delivered wrong halos, to exercise the hand-over
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            if int(good.item()) == 1:
                break
            rejected.append(sh.vec.plan.mode)
        else:
            raise SystemExit(f"parity gate failed: sharded y differs from the stencil's closed form with every exchange {rejected}")
        del want
        step = lambda: sh.multiply(y)  # noqa: E731
        p = sh.vec.plan
        exchange_info = {"mode": p.mode, "values_received_per_rank": p.allgather_values if p.mode == "allgather" else p.recv_values,
                         "allgather_values": p.allgather_values, "validated_against": "stencil closed form (bit-exact)",
                         "rejected_exchanges": rejected}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- settle, warm-up, then exactly K timed steps ----------------------------------------------
    # Settle phase (disclosed as `settle_launches`, not part of W or K): the device comes out of setup at idle clocks
    # and a 20-step timed region lasts < 3 ms, so the clocks are brought up first.  Then the contract's W warm-up steps.
    SETTLE = 100
    for _ in range(SETTLE):
        step()
    barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N>1: the north-star's literal exchange (RCCL all-gather of the whole x before each multiply) timed
    #      beside the default one, same protocol, fewer steps (it moves (N-1)/N of x per rank per step) ------
    allgather_leg = None
    if dist is not None and exchange_info["mode"] != "allgather":
        try:
            sh_ag = cmi.distributed.ShardedCsr(A, N_global, rank, world, mode="allgather")
            sh_ag.x_local.copy_(sh.x_local)
            y_ag = torch.empty_like(y)
            ag_steps = max(1, min(args.steps, 20))
            for _ in range(2):
                sh_ag.multiply(y_ag)
            barrier()
            t0 = time.perf_counter()
            for _ in range(ag_steps):
                sh_ag.multiply(y_ag)
            barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            same = torch.tensor([int(torch.equal(y_ag, y))], dtype=torch.int32, device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            ag_s = float(t.item()) / ag_steps

            def exchange_only(vec):  # the exchange alone, same protocol (make the xGMI bound visible)
                for _ in range(2):
                    vec.exchange()
                barrier()
                t0 = time.perf_counter()
                for _ in range(ag_steps):
                    vec.exchange()
                barrier()
                tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return round(float(tt.item()) / ag_steps * 1e3, 5)

            exchange_info["exchange_only_ms"] = exchange_only(sh.vec)
            allgather_leg = {"ms_per_step": round(ag_s * 1e3, 5), "value": round(2.0 * nnz_global / ag_s / 1e9, 3), "unit": "GFLOP/s",
                             "steps": ag_steps, "values_received_per_rank": sh_ag.vec.plan.allgather_values,
                             "exchange_only_ms": exchange_only(sh_ag.vec),
                             "y_identical_to_default_exchange": bool(int(same.item()))}
            del sh_ag, y_ag
        except Exception as e:  # noqa: BLE001 -- the secondary leg must never take the main line down
            allgather_leg = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- dominant kernel: launch duration with HIP events on ITS stream ---------------------------------------
    # >= 200 launches whatever --steps says, as KERNEL_BATCHES batches of launches each bracketed by its own event pair
    # (a batch, not a single launch: an event pair around one launch also times the ~2 us launch gap).  Reported:
    # the mean over all batches (= the average launch duration, what roofline.achieved is computed from), the median
    # batch and the fastest batch.  Launch-to-launch spread on this pool is 130-161 us (profiles/r01_bench_kernel_stats.csv),
    # so 20 launches after a cold start -- round 1's protocol -- could land 5 % off the average.
    KERNEL_BATCHES = 10
    per_batch = max(20, -(-max(args.steps, 200) // KERNEL_BATCHES))
    evs = []
    for _ in range(2 * KERNEL_BATCHES):
        e = ctypes.c_void_p()
        cmi.check(lib.cmi_event_create(ctypes.byref(e)))
        evs.append(e)
    x_kernel = x if world == 1 else sh.x_view
    kernel_only = (lambda: cmi.multiply(Afmt, x_kernel, y))
    for _ in range(10):  # re-warm: the secondary legs above ran other kernels
        kernel_only()
    torch.cuda.synchronize()
    for bi in range(KERNEL_BATCHES):
        cmi.check(lib.cmi_event_record(evs[2 * bi], sptr))
        for _ in range(per_batch):
            kernel_only()
        cmi.check(lib.cmi_event_record(evs[2 * bi + 1], sptr))
    batch_ms = []
    for bi in range(KERNEL_BATCHES):
        ms = ctypes.c_float()
        cmi.check(lib.cmi_event_elapsed_ms(evs[2 * bi], evs[2 * bi + 1], ctypes.byref(ms)))
        batch_ms.append(ms.value / per_batch)
    for e in evs:
        cmi.check(lib.cmi_event_destroy(e))
    kernel_ms = sum(batch_ms) / len(batch_ms)
    kernel_ms_median = sorted(batch_ms)[len(batch_ms) // 2]
    kernel_ms_min = min(batch_ms)
    if dist is not None:
        t = torch.tensor([kernel_ms, kernel_ms_median, kernel_ms_min], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernel_ms, kernel_ms_median, kernel_ms_min = (float(v) for v in t.tolist())

    # ---- the caller: CG iterations/s on the same matrix (cusp::krylov::cg's loop, cg.inl:80-105; fused device
    #      path, sharded when N>1), with the recurrence residual checked against b - A x at the end -----------
    cg_leg = None
    if fmt == "csr" and args.cg_iterations > 0:
        try:
            b_vec = torch.from_numpy(x_host).to(dev)
            x_sol = torch.zeros(rows_per_rank, dtype=torch.float64, device=dev)
            op = A if world == 1 else sh
            cmi.krylov.cg(op, x_sol.clone(), b_vec, iteration_limit=3, relative_tolerance=0.0)  # warm-up
            barrier()
            t0 = time.perf_counter()
            mon = cmi.krylov.cg(op, x_sol, b_vec, iteration_limit=args.cg_iterations, relative_tolerance=0.0)
            barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if dist is not None:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            # true residual of the returned x
            r_true = torch.empty_like(b_vec)
            if world == 1:
                cmi.multiply(A, x_sol, r_true)
            else:
                sh.vec.fence()
                sh.x_local.copy_(x_sol)
                sh.vec.fence()
                sh.multiply(r_true)
            rr = ((b_vec - r_true) ** 2).sum().reshape(1)
            if dist is not None:
                dist.all_reduce(rr)
            true_norm = float(rr.item()) ** 0.5
            cg_s = float(t.item())
            cg_leg = {"iterations": mon.iteration_count, "iterations_per_s": round(mon.iteration_count / cg_s, 1),
                      "us_per_iteration": round(cg_s / max(mon.iteration_count, 1) * 1e6, 2),
                      "final_residual_norm": mon.residuals[-1], "true_residual_norm": true_norm,
                      "residual_consistent": bool(abs(true_norm - mon.residuals[-1]) <= 1e-6 * mon.residuals[0]),
                      "exchange": None if world == 1 else sh.vec.plan.mode}
        except Exception as e:  # noqa: BLE001 -- a secondary leg
            cg_leg = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- numbers -------------------------------------------------------------------------------
    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * nnz_global / (elapsed / args.steps) / 1e9
    local_nnz, local_rows = A.num_entries, A.num_rows
    if fmt == "csr":
        alg_bytes = cmi.csr_bytes(local_rows, local_nnz)
        kname = "csr"
    elif fmt == "ell":
        alg_bytes, kname = cmi.ell_bytes(local_rows, Afmt.num_entries_per_row, Afmt.pitch), "ell"
    elif fmt == "dia":
        alg_bytes, kname = cmi.dia_bytes(local_rows, 5, Afmt.pitch), "dia"
    elif fmt == "coo":
        alg_bytes, kname = cmi.coo_bytes(local_rows, local_nnz), "coo"
    else:  # hyb: the ELL part's bytes (x and y once) + the COO part's three streams; dominant kernel = the ELL launch
        alg_bytes, kname = cmi.ell_bytes(local_rows, HYB_WIDTH, Afmt.ell.pitch) + 16 * Afmt.coo.num_entries, "ell"
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(kname)
    cfg = cmi.tuning_select({"csr": 0, "ell": 1, "dia": 2, "coo": 3, "hyb": 1}[fmt], cmi.F64, local_rows, N_global,
                            local_nnz if fmt in ("csr", "coo") else local_rows * (HYB_WIDTH if fmt == "hyb" else 5))

    if rank == 0:
        line = {
            "metric": "spmv_gflops_fp64_poisson5pt",
            "value": round(gflops, 3),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: ranks share one GPU over gloo; not a measurement)",
            "hbm_gbps": round(alg_bytes * world / (elapsed / args.steps) / 1e9, 2),
            "config": {"workload": f"poisson5pt {m}x{n} {fmt.upper()} int32/f64, y = A*x "
                                   f"({N_global} rows, {nnz_global} entries; {M}x{M} grid points per GPU)",
                       "format": fmt, **({"hyb_ell_width": HYB_WIDTH, "hyb_coo_entries": Afmt.coo.num_entries} if fmt == "hyb" else {}),
                       "rows_per_gpu": local_rows, "entries_per_gpu": local_nnz,
                       "kernel_config": cfg.as_dict(), "parallelism": f"row-block x{world}",
                       "x_exchange": exchange_info},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": None if traffic is None else f"{traffic_src}: committed rocprofv3 --pmc summary of this kernel on this matrix and config (separate passes, FETCH_SIZE x2 correction); not collected in this run",
                         "kernel_avg_ms": round(kernel_ms, 6), "kernel_median_ms": round(kernel_ms_median, 6),
                         "kernel_min_ms": round(kernel_ms_min, 6),
                         "kernel_timing": f"{KERNEL_BATCHES} batches x {per_batch} launches, one HIP event pair per batch on the launch stream",
                         "kernel_avg_over_ms_per_step": round(kernel_ms / ms_per_step, 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_gflops": round(2.0 * local_nnz / (kernel_ms * 1e-3) / 1e9, 2)},
            "settle_launches": SETTLE,
        }
        if allgather_leg is not None:
            line["allgather_exchange"] = allgather_leg
        if cg_leg is not None:
            line["cg"] = cg_leg
        if world == 1 and not args.no_cpu_baseline and fmt == "csr":
            base, omp = cpu_baseline(cmi, A, x_host, y.cpu().numpy(), args.cpu_seconds)
            line["cpu_baseline"] = base
            line["cpu_baseline_omp"] = omp
        print(json.dumps(line), flush=True)
    if dist is not None:
        sh.vec.close()  # barrier + unmap the peers' buffers before anyone frees them
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
