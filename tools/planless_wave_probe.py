#!/usr/bin/env python3
"""Plan-less cmi_spmv_csr_f64 (NULL config) with and without the round-4 rule that gives stencil-like matrices the wave-tile kernel
(adopt only if no bucket regresses > 2 %).  One process per setting of $CMI_PLANLESS_WAVE (the library reads it once):

    python tools/planless_wave_probe.py            # runs itself twice (0 / 1) as child processes and prints the table

Matrices: stencils (5-point 3162^2 = the headline, 5-point 1000^2, 7-point 3-D, 9-point 2-D, tridiagonal), matrices that only LOOK like
stencils from their sizes (random row lengths whose mean happens to sit just below an integer), thermal2-like, and matrices the rule must
leave alone.  Every result is checked against csr_scalar bit for bit."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def matrices():
    import numpy as np
    import scipy.sparse as sp
    rng = np.random.default_rng(3)

    def stencil(offs, n):
        A = sp.diags([np.ones(n - abs(o)) for o in offs], offs, shape=(n, n), format="csr")
        A.sort_indices()
        return A.indptr.astype(np.int32), A.indices.astype(np.int32), rng.standard_normal(A.nnz)

    def random_lens(rows, lo, hi, fix_mean=None):
        lens = rng.integers(lo, hi + 1, size=rows)
        if fix_mean is not None:  # nudge single rows (a random selection, all at once) until the mean sits just below the integer
            diff = int(lens.sum()) - int(fix_mean * rows)
            order = rng.permutation(rows)
            if diff > 0:
                lens[order[lens[order] > lo][:diff]] -= 1
            elif diff < 0:
                lens[order[lens[order] < hi][:-diff]] += 1
        Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
        ri = np.repeat(np.arange(rows, dtype=np.int64), lens)
        Aj = np.clip(ri + rng.integers(-40, 41, size=len(ri)), 0, rows - 1).astype(np.int32)
        return Ap, Aj, rng.standard_normal(len(Aj))

    import suitesparse_like as ssl
    yield "poisson5pt 3162^2 (headline)", None
    yield "poisson5pt 1000^2", stencil([-1000, -1, 0, 1, 1000], 1000 * 1000)
    g = 160
    yield "7-point 160^3", stencil([-g * g, -g, -1, 0, 1, g, g * g], g ** 3)
    yield "9-point 2000^2", stencil([-2001, -2000, -1999, -1, 0, 1, 1999, 2000, 2001], 2000 * 2000)
    yield "tridiagonal 1e7", stencil([-1, 0, 1], 10 ** 7)
    yield "random lengths 1..9, mean 4.99 (looks like a stencil)", random_lens(3000000, 1, 9, 4.99)
    yield "random lengths 4..6, mean 4.95", random_lens(3000000, 4, 6, 4.95)
    yield "random lengths 1..13, mean 6.99", random_lens(2000000, 1, 13, 6.99)
    yield "random lengths 1..9, mean ~5.0 (also looks like one)", random_lens(3000000, 1, 9)
    yield "thermal2-like (rule: no)", ssl.GENERATORS["thermal2"](1.0)
    # session 30 of round 4: matrices beyond the Infinity Cache too -- stencils, and look-alikes whose tiles need passes
    g = 215
    yield "7-point 215^3 (beyond the cache)", stencil([-g * g, -g, -1, 0, 1, g, g * g], g ** 3)
    yield "random lengths 1..9, mean 4.99, 7 M rows (looks like a stencil, beyond the cache)", random_lens(7000000, 1, 9, 4.99)
    yield "random lengths 4..6, mean 4.95, 7 M rows", random_lens(7000000, 4, 6, 4.95)
    yield "random lengths 1..13, mean 6.99, 5 M rows (beyond the cache)", random_lens(5000000, 1, 13, 6.99)


def child():
    import ctypes
    import numpy as np
    import torch
    import cusp_autotuned_amd as cmi
    lib = cmi.lib()
    out = {}
    for name, m in matrices():
        if m is None:
            A = cmi.poisson5pt(3162, 3162, "csr", dtype=torch.float64, device="cuda")
            dAp, dAj, dAx = A.row_offsets, A.column_indices, A.values
        else:
            dAp, dAj, dAx = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in m)
        rows = dAp.numel() - 1
        nnz = dAj.numel()
        x = cmi.fill_x(rows, torch.float64, "cuda")
        y = torch.empty(rows, dtype=torch.float64, device="cuda")
        cmi.spmv_csr(rows, rows, dAp, dAj, dAx, x, y, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
        want = y.clone()
        y.fill_(7.0)
        go = lambda: cmi.spmv_csr(rows, rows, dAp, dAj, dAx, x, y)  # noqa: E731 -- NULL config, no plan
        go()
        exact = bool(torch.equal(y, want))
        for _ in range(60):
            go()
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        cmi.check(lib.cmi_event_create(ctypes.byref(e0)))
        cmi.check(lib.cmi_event_create(ctypes.byref(e1)))
        ts = []
        for _ in range(7):
            s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            cmi.check(lib.cmi_event_record(e0, s))
            for _ in range(20):
                go()
            cmi.check(lib.cmi_event_record(e1, s))
            ms = ctypes.c_float()
            cmi.check(lib.cmi_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
            ts.append(ms.value / 20 * 1e3)
        out[name] = {"us": float(np.median(ts)), "bit_exact": exact, "mean": nnz / rows, "alg_bytes": cmi.csr_bytes(rows, nnz)}
        print(f"  {name}: {out[name]['us']:.1f} us", file=sys.stderr, flush=True)
        del dAp, dAj, dAx, x, y, want
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child()
    res = {}
    # "0": the table's row-tile kernel; "1": + the csr_wave rule (what the library does); "2": was, in session 31 of round 4, a third setting --
    # 16-byte-vector wave tiles on fixed row ranges for f64 stencil sizes beyond the cache -- tried and dropped
    # (profiles/r04_planless_vector_tiles_not_kept.txt); the column is kept so that the table reads the same: it repeats setting "1"
    for setting, extra in (("0", {"CMI_PLANLESS_WAVE": "0"}), ("1", {"CMI_PLANLESS_WAVE": "1"}), ("2", {"CMI_PLANLESS_WAVE": "1"})):
        env = dict(os.environ, **extra)
        print(f"# child: {extra}", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, stdout=subprocess.PIPE, text=True, timeout=900)  # (stderr: progress, passed through)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(r.stdout[-2000:])
            raise SystemExit(f"child with {extra} failed")
        res[setting] = json.loads(line[0][7:])
    print(f"{'matrix':84s} {'mean':>6s} {'table (us)':>11s} {'csr_wave rule':>14s} {'(again)':>15s} {'ratio':>6s}  frac of 8 TB/s  bit-exact")
    for name in res["0"]:
        a, b, c = res["0"][name], res["1"][name], res["2"][name]
        print(f"{name:84s} {a['mean']:6.2f} {a['us']:11.1f} {b['us']:14.1f} {c['us']:15.1f} {c['us'] / a['us']:6.3f}  {a['alg_bytes'] / a['us'] / 8e6:.3f} -> {c['alg_bytes'] / c['us'] / 8e6:.3f}  {a['bit_exact'] and b['bit_exact'] and c['bit_exact']}")


if __name__ == "__main__":
    main()
