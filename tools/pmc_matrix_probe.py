#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes on the configs[3] matrices (no counter file existed for the
25-80 entries/row class).  One process, one profiler start-up: per matrix (ldoor / nlpkkt120 / thermal2 stand-ins or the real
files, tools/suitesparse_like.py) every VARIANT is launched K times back to back; the manifest (stdout, `MANIFEST` lines) lets
tools/pmc_matrix_table.py map dispatches back to variants.  A calibration kernel of known byte count (axpby, 16 B per lane)
runs last, as MI355X_MICROARCH.md's HBM section prescribes for FETCH_SIZE / WRITE_SIZE.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir>/fetch -- python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120
    python3 tools/pmc_matrix_probe.py ldoor --time       # no profiler: HIP-event timing of the same variants

Variants (each validated against csr_scalar before it is launched for the counters):
    plan      the plan's choice (what cusp::multiply runs)
    stream    csr_stream, the table's shape for this row length (no plan)
    wavev     CMI_CSR_STREAM_WAVEV through a plan, when the library has it
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
import suitesparse_like as ssl  # noqa: E402

K = int(os.environ.get("PMC_LAUNCHES", "6"))


SCALE = float(os.environ.get("PMC_SCALE", "1.0"))  # stand-in size (1.0: the published one)
DT = torch.float32 if os.environ.get("PMC_DTYPE", "f64") == "f32" else torch.float64  # value type of the matrices
VB = 4 if DT == torch.float32 else 8


def cached(name):
    """stand-in at SCALE, generated once per box (the passes are separate processes)"""
    path = f"/tmp/cmi_{name}_{SCALE}.npz"
    if os.path.exists(path):
        z = np.load(path)
        return z["Ap"], z["Aj"], z["Ax"], str(z["src"])
    Ap, Aj, Ax, src = ssl.load(name, SCALE)
    np.savez(path, Ap=Ap, Aj=Aj, Ax=Ax, src=src)
    return Ap, Aj, Ax, src


def variants_for(A):
    out = [("plan", None)]
    out.append(("stream", "table"))
    for spec in (t for t in os.environ.get("PMC_PIPE", "").split(",") if t):  # csr_stream_pipe: block:rows_per_block
        blk, rpb = (int(v) for v in spec.split(":"))
        out.append((f"pipe{blk}x{rpb}", ("cfg", cmi.Config(kernel=cmi.CSR_STREAM_PIPE, block_size=blk, rows_per_block=rpb, nontemporal=3))))
    for spec in (t for t in os.environ.get("PMC_STREAM", "").split(",") if t):  # explicit csr_stream shapes: block:rows_per_block:vectors:policy
        blk, rpb, ipt, pol = (int(v) for v in spec.split(":"))
        out.append((f"stream{blk}x{rpb}x{ipt}/pol{pol}", ("cfg", cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, rows_per_block=rpb, items_per_thread=ipt,
                                                                              threads_per_row=1, nontemporal=pol))))
    for k in (int(t) for t in os.environ.get("PMC_WAVEP", "").split(",") if t):  # csr_wave's LANE-STRIDED body on a plan-built partition, k entries per lane
        for pol in (int(s) for s in os.environ.get("PMC_WAVEP_POL", "3").split(",") if s):
            out.append((f"wavep{k}/pol{pol}", cmi.Config(kernel=cmi.CSR_STREAM_WAVE, rows_per_block=-1, items_per_thread=k, nontemporal=pol)))
    if hasattr(cmi, "CSR_STREAM_WAVEV"):
        for v in (int(s) for s in os.environ.get("PMC_WAVEV", "2,4").split(",") if s):
            for pol in (int(s) for s in os.environ.get("PMC_WAVEV_POL", "0").split(",") if s):  # 0: the plan's own policy choice
                for swz in (int(s) for s in os.environ.get("PMC_WAVEV_SWZ", "0").split(",") if s):
                    out.append((f"wavev{v}" + (f"/pol{pol}" if pol else "") + (f"/swz{swz}" if swz else ""),
                                cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=v, nontemporal=pol, xcd_swizzle=swz)))
    # round 4: the run-compressed column copy on wave tiles (and its packed twin: ("values", cfg) -> a plan made with the values), the 16-bit copy
    if hasattr(cmi, "CSR_STREAM_WAVER"):
        for v in (int(s) for s in os.environ.get("PMC_WAVER", "4").split(",") if s):
            for pol in (int(s) for s in os.environ.get("PMC_WAVER_POL", "0").split(",") if s):
                for cap in (int(s) for s in os.environ.get("PMC_WAVER_CAP", "0").split(",") if s):  # 0: the rule; 3 / 4: entries per piece at most
                    for swz in (int(s) for s in os.environ.get("PMC_WAVER_SWZ", "0").split(",") if s):  # 0: the table's XCD dealing; -1: launch order; C: chunks of C tiles
                        out.append((f"waver{v}" + (f"/pol{pol}" if pol else "") + (f"/cap{cap}" if cap else "") + (f"/swz{swz}" if swz else ""),
                                    cmi.Config(kernel=cmi.CSR_STREAM_WAVER, items_per_thread=v, nontemporal=pol, threads_per_row=cap, xcd_swizzle=swz)))
                if os.environ.get("PMC_PACKED", "1") != "0":
                    out.append((f"packed{v}" + (f"/pol{pol}" if pol else ""), ("values", cmi.Config(kernel=cmi.CSR_STREAM_PACKED, items_per_thread=v, nontemporal=pol))))
    if os.environ.get("PMC_C16", "0") != "0":
        out.append(("c16", cmi.Config(kernel=cmi.CSR_STREAM_C16)))
    if os.environ.get("PMC_PLAN_AGAIN", "0") != "0":
        out.append(("plan-again", None))
    return out


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "ldoor,nlpkkt120").split(",")
    timing = "--time" in sys.argv
    lib = cmi.lib()
    import ctypes
    for name in names:
        t0 = time.time()
        Ap, Aj, Ax, src = cached(name)
        rows, nnz = len(Ap) - 1, int(Ap[-1])
        # PMC_COLS: what the x gathers cost -- the same matrix with every column index replaced by 0 (`zero`: one cache line serves every
        # gather) or by its row's index (`row`: a gather instruction touches as few lines as y's store does); sums differ, the streams do not
        cols_mode = os.environ.get("PMC_COLS", "")
        if cols_mode == "zero":
            Aj = np.zeros_like(Aj)
        elif cols_mode == "row":
            Aj = np.repeat(np.arange(rows, dtype=np.int32), np.diff(Ap))
        elif cols_mode == "seq":      # entry e gathers x[e mod rows]: a gather instruction's lanes sit in the fewest lines a stride allows
            Aj = (np.arange(nnz, dtype=np.int64) % rows).astype(np.int32)
        elif cols_mode.startswith("win"):  # win512: a pseudo-random column inside the 512-column window its row sits in: the lines are cache-resident,
            w = int(cols_mode[3:] or 512)  # but the lanes of one gather instruction are spread over w / 16 of them
            e = np.arange(nnz, dtype=np.int64)
            r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(Ap))
            Aj = np.minimum((r // w) * w + ((e * 2654435761) >> 7) % w, rows - 1).astype(np.int32)
        if cols_mode:
            name = f"{name}[cols={cols_mode}]"
        A = cmi.CsrMatrix(rows, rows, nnz, torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax).cuda().to(DT))
        x = cmi.fill_x(rows, DT, "cuda")
        y = torch.empty(rows, dtype=DT, device="cuda")
        cmi.multiply(A, x, y, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
        want = y.clone()
        alg = cmi.csr_bytes(rows, nnz, VB)
        print(f"# {name}: {src}; rows {rows} entries {nnz} algorithmic bytes {alg}; set-up {time.time() - t0:.1f} s", flush=True)
        for label, cfg in variants_for(A):
            plan = None
            explicit = None
            with_values = isinstance(cfg, tuple) and cfg[0] == "values"
            if with_values:
                cfg = cfg[1]
            if isinstance(cfg, tuple):
                explicit, cfg = cfg[1], "explicit"
            if isinstance(cfg, cmi.Config):
                try:
                    plan = (cmi.Plan.csr_values(rows, rows, A.row_offsets, A.column_indices, A.values, cfg) if with_values else
                            cmi.Plan.csr(DT, rows, rows, A.row_offsets, A.column_indices, cfg=cfg))
                except Exception as e:  # noqa: BLE001
                    print(f"# {name} {label}: no plan ({e})")
                    continue

            def go():
                if cfg is None:
                    cmi.multiply(A, x, y)
                elif cfg == "table":
                    cmi.spmv_csr(rows, rows, A.row_offsets, A.column_indices, A.values, x, y)
                elif cfg == "explicit":
                    cmi.spmv_csr(rows, rows, A.row_offsets, A.column_indices, A.values, x, y, cfg=explicit)
                else:
                    cmi.spmv_csr_plan(plan, A.row_offsets, A.column_indices, A.values, x, y)
            y.fill_(7.0)
            go()
            exact = bool(torch.equal(y, want))
            desc = (A.plan().config() if cfg is None else explicit if explicit is not None else plan.config() if plan is not None else
                    cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64 if VB == 8 else cmi.F32, rows, rows, nnz))
            if timing:
                e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
                cmi.check(lib.cmi_event_create(ctypes.byref(e0)))
                cmi.check(lib.cmi_event_create(ctypes.byref(e1)))
                ts = []
                # settle: the first variant of a matrix used to be timed while the clocks were still coming up behind the set-up (r3: `plan` 95.3 us
                # against `stream` 89.5 us with the SAME printed config; r4 session 2: 88.4 against 77.7) -- every variant now runs SETTLE
                # untimed launches first
                t_settle = time.time()
                while time.time() - t_settle < float(os.environ.get("PMC_SETTLE_S", "0.05")):  # (by time: 60 launches of a 80 us kernel were not enough)
                    for _ in range(20):
                        go()
                    torch.cuda.synchronize()
                for _ in range(7):
                    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                    cmi.check(lib.cmi_event_record(e0, s))
                    for _ in range(20):
                        go()
                    cmi.check(lib.cmi_event_record(e1, s))
                    ms = ctypes.c_float()
                    cmi.check(lib.cmi_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
                    ts.append(ms.value / 20 * 1e3)
                t = float(np.median(ts))
                owns = f"\tplan owns {plan.device_bytes() / 1e6:.1f} MB" if plan is not None and hasattr(plan, "device_bytes") else ""
                print(f"TIME\t{name}:{label}\t{t:.1f} us\t{alg / t / 1e3:.0f} GB/s\tfrac {alg / t / 1e3 / 8000:.3f}\tbit-exact {exact}\t{desc}{owns}", flush=True)
            else:
                torch.cuda.synchronize()
                for _ in range(K):
                    go()
                torch.cuda.synchronize()
                print(f"MANIFEST\t{name}:{label}\t{K + 1}\t{alg}\t{exact}\t{desc}", flush=True)
        del A, x, y, want
        torch.cuda.empty_cache()
    n_cal = 1 << 25
    a = torch.ones(n_cal, dtype=torch.float64, device="cuda")
    b = torch.ones(n_cal, dtype=torch.float64, device="cuda")
    c = torch.empty(n_cal, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6):
        cmi.blas_axpby(2.0, a, 3.0, b, c)
    torch.cuda.synchronize()
    print(json.dumps({"calibration_kernel": "axpby_kernel", "calibration_read_bytes": 16 * n_cal, "calibration_write_bytes": 8 * n_cal}))


if __name__ == "__main__":
    main()
