#!/usr/bin/env python3
"""What the OFFLINE selection gives away against a per-matrix search (the reference's KTT tuner searches per matrix at run time,
cusp/system/cuda/ktt/kernel.h:37-62; this library ships one table + plan rules and never searches on the caller's time).

Per matrix and value type: the AUTO plan made with the columns (what cusp::multiply on a container runs), the plan-less call (NULL
config), and every CSR kernel the library has under an explicit config -- csr_scalar, csr_vector T = 2..64 (the reference's whole
selector space, csr_vector_spmv.h:225-258), the table's csr_stream, csr_wave (equal-length tiles), wave tiles V = 1 / 2 / 4 with and
without the LDS x window, the run-compressed copy V = 2 / 4, csr_balanced.  Each candidate is validated before it is timed (bit-exact
ones against csr_scalar's bits, csr_vector / csr_balanced -- re-associated sums -- to 1e-10 / 1e-5 relative).  Opt-in plans that own
a copy of the values or 16-bit columns are not candidates (they are never selected automatically).

    regret = time(AUTO plan) / min(time over all candidates)          1.00: no search would have found anything faster

    python3 tools/auto_regret.py [--dtypes f64,f32] [--only name,name] [--log out.jsonl]
"""
import argparse
import json
import math
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
import autotune as at  # noqa: E402


def lens_csr(lens, seed, spread=3):
    """row lengths -> CSR with strictly increasing columns, stride `spread` around the diagonal (tools/autotune_hyb.py's layout, on the host)"""
    lens = np.asarray(lens, np.int64)
    rows = len(lens)
    Ap = np.zeros(rows + 1, np.int64)
    Ap[1:] = np.cumsum(lens)
    row = np.repeat(np.arange(rows, dtype=np.int64), lens)
    j = np.arange(int(Ap[-1]), dtype=np.int64) - Ap[:-1][row]
    col = np.clip(row + (j - lens[row] // 2) * spread, 0, rows - 1)
    edge = (col == 0) | (col == rows - 1)
    col[edge] = np.clip(row + j, 0, rows - 1)[edge]
    Ax = np.random.default_rng(seed).standard_normal(int(Ap[-1]))
    return Ap.astype(np.int32), col.astype(np.int32), Ax


def scattered_csr(rows, mean, seed):
    rng = np.random.default_rng(seed)
    lens = rng.poisson(mean, size=rows).astype(np.int64)
    Ap = np.zeros(rows + 1, np.int64)
    Ap[1:] = np.cumsum(lens)
    nnz = int(Ap[-1])
    Aj = rng.integers(0, rows, size=nnz).astype(np.int32)
    row = np.repeat(np.arange(rows, dtype=np.int64), lens)
    Aj = Aj[np.lexsort((Aj, row))]
    return Ap.astype(np.int32), Aj, rng.standard_normal(nnz)


def matrices():
    """(name, builder) -- seeded, every one beyond the 256 MiB Infinity Cache in f64 unless its name says otherwise"""
    rng = np.random.default_rng(404)
    out = []
    out.append(("poisson5pt 3162^2 (headline)", lambda: at.stencil_csr(3162, 3162, 1, [(0, -1, 0, -1.0), (-1, 0, 0, -1.0), (0, 0, 0, 4.0), (1, 0, 0, -1.0), (0, 1, 0, -1.0)], np.float64)))
    out.append(("7-point 215^3", lambda: at.stencil_csr(215, 215, 215, at.stencil_points(7), np.float64)))
    out.append(("9-point 3000^2", lambda: at.stencil_csr(3000, 3000, 1, at.stencil_points(9), np.float64)))
    out.append(("27-point 140^3", lambda: at.stencil_csr(140, 140, 140, at.stencil_points(27), np.float64)))
    out.append(("27-point 66^3 x 3 dof (81 per row)", lambda: at.block_expand(*at.stencil_csr(66, 66, 66, at.stencil_points(27), np.float64), 3, np.float64)))
    out.append(("7-point 100^3 x 3 dof (21 per row)", lambda: at.block_expand(*at.stencil_csr(100, 100, 100, at.stencil_points(7), np.float64), 3, np.float64)))
    for nm in ("thermal2", "ldoor", "nlpkkt120"):
        out.append((f"{nm}-like x1.0", lambda nm=nm: ssl.load(nm, 1.0)[:3]))
    out.append(("thermal2-like x3.0", lambda: ssl.load("thermal2", 3.0)[:3]))
    for mean in (8, 16, 32):
        out.append((f"poisson({mean}) lengths, columns anywhere in +-2000", lambda mean=mean: at.synthetic_csr(48_000_000 // mean, 48_000_000 // mean, mean, 7 + mean, np.float64)))
    out.append(("uniform 1..16, stride-3 columns", lambda: lens_csr(rng.integers(1, 17, size=4_000_000), 1)))
    b = np.full(4_000_000, 4)
    b[rng.random(4_000_000) < 0.10] = 40
    out.append(("bimodal 4 / 40 (10 %), stride-3 columns", lambda b=b: lens_csr(b, 2)))
    p = np.minimum(4 + np.floor(rng.pareto(1.3, size=2_000_000) * 3), 3000).astype(np.int64)
    out.append(("power-law tail (4 .. 3000), stride-3 columns", lambda p=p: lens_csr(p, 3)))
    sk = np.full(2_000_000, 6)
    sk[rng.integers(0, 2_000_000, size=8)] = 1_000_000
    out.append(("8 rows of 10^6 in 2 M rows of 6", lambda sk=sk: lens_csr(sk, 4, spread=1)))
    out.append(("poisson(10) lengths, columns anywhere (scattered)", lambda: scattered_csr(3_000_000, 10, 5)))
    def fixed_len(rows, k, span, seed):
        r = np.random.default_rng(seed)
        Ap = (np.arange(rows + 1, dtype=np.int64) * k).astype(np.int32)
        base = np.repeat(np.arange(rows, dtype=np.int64), k)
        Aj = (r.integers(0, rows, size=rows * k) if span is None else np.clip(base + r.integers(-span, span + 1, size=rows * k), 0, rows - 1))
        Aj = np.sort(Aj.reshape(rows, k), axis=1).reshape(-1).astype(np.int32)
        return Ap, Aj, r.standard_normal(rows * k)
    # equal row lengths WITHOUT a stencil's columns: the stencil rules see only the lengths
    out.append(("6 per row exactly, columns anywhere (scattered)", lambda: fixed_len(6_000_000, 6, None, 61)))
    out.append(("6 per row exactly, columns anywhere in +-2000", lambda: fixed_len(6_000_000, 6, 2000, 62)))
    out.append(("9 per row exactly, columns anywhere in +-2000", lambda: fixed_len(4_000_000, 9, 2000, 63)))
    out.append(("poisson5pt 1000^2 (cache-resident)", lambda: at.stencil_csr(1000, 1000, 1, [(0, -1, 0, -1.0), (-1, 0, 0, -1.0), (0, 0, 0, 4.0), (1, 0, 0, -1.0), (0, 1, 0, -1.0)], np.float64)))
    return out


def matrices2():
    """the second set (session 34): sizes INSIDE the Infinity Cache (the rules have size gates), very short rows, rows of 12..44 without runs,
    long rows, a rectangular matrix"""
    rng = np.random.default_rng(505)
    out = []
    out.append(("7-point 100^3 (84 MB: cache-resident)", lambda: at.stencil_csr(100, 100, 100, at.stencil_points(7), np.float64)))
    out.append(("27-point 80^3 (166 MB: cache-resident)", lambda: at.stencil_csr(80, 80, 80, at.stencil_points(27), np.float64)))
    out.append(("ldoor-like x0.3 (cache-resident)", lambda: ssl.load("ldoor", 0.3)[:3]))
    out.append(("nlpkkt120-like x0.15 (cache-resident)", lambda: ssl.load("nlpkkt120", 0.15)[:3]))
    out.append(("poisson(16) lengths, columns anywhere in +-2000, 1 M rows (cache-resident)", lambda: at.synthetic_csr(1_000_000, 1_000_000, 16, 23, np.float64)))
    out.append(("bidiagonal 2 x 10^7", lambda: at.stencil_csr(20_000_000, 1, 1, [(0, 0, 0, 2.0), (1, 0, 0, -1.0)], np.float64)))
    out.append(("random lengths 1..4, stride-3 columns, 12 M rows", lambda: lens_csr(rng.integers(1, 5, size=12_000_000), 6)))
    out.append(("uniform 10..40, stride-3 columns, 1.5 M rows", lambda: lens_csr(rng.integers(10, 41, size=1_500_000), 7)))
    out.append(("uniform 20..60, stride-1 columns (one run per row), 1 M rows", lambda: lens_csr(rng.integers(20, 61, size=1_000_000), 8, spread=1)))
    out.append(("uniform 100..300, stride-3 columns, 200 k rows", lambda: lens_csr(rng.integers(100, 301, size=200_000), 9)))
    out.append(("uniform 400..1200, stride-1 columns, 50 k rows", lambda: lens_csr(rng.integers(400, 1201, size=50_000), 10, spread=1)))

    def rectangular():
        rows, cols, k = 8_000_000, 500_000, 5
        r = np.random.default_rng(11)
        Ap = (np.arange(rows + 1, dtype=np.int64) * k).astype(np.int32)
        centre = np.repeat(np.arange(rows, dtype=np.int64) * cols // rows, k)
        Aj = np.sort(np.clip(centre + r.integers(-30, 31, size=rows * k), 0, cols - 1).reshape(rows, k), axis=1).reshape(-1).astype(np.int32)
        return Ap, Aj, r.standard_normal(rows * k)
    out.append(("rectangular 8 M x 500 k, 5 per row near the scaled diagonal", rectangular))
    return out


def matrices3():
    """the third set (session 35): where is the size gate of the wave tiles on GATHER-BOUND band matrices?  (set 2 had them winning by 12-40 %
    at 16 M entries, inside the cache, where the rule of round 3 -- streams beyond 0.75 x the cache -- does not admit them)"""
    out = []
    for mean in (8, 16, 32):
        for entries in (2_000_000, 4_000_000, 8_000_000, 12_000_000):
            rows = entries // mean
            out.append((f"poisson({mean}) lengths, columns anywhere in +-2000, {entries // 1_000_000} M entries", lambda rows=rows, mean=mean: at.synthetic_csr(rows, rows, mean, 100 + mean, np.float64)))
    return out


def matrices4():
    """the fourth set (session 37): column runs of other lengths (2 / 6 dof per node, pairs, dense blocks), 50..90 entries per row with and
    without runs, an arrow matrix, a stencil with holes, a small FEM mesh"""
    rng = np.random.default_rng(707)
    out = []
    out.append(("9-point 1500^2 x 2 dof (18 per row, runs of 6)", lambda: at.block_expand(*at.stencil_csr(1500, 1500, 1, at.stencil_points(9), np.float64), 2, np.float64)))
    out.append(("5-point 2200^2 x 2 dof (10 per row, runs of 2 / 6)", lambda: at.block_expand(*at.stencil_csr(2200, 2200, 1, [(0, -1, 0, -1.0), (-1, 0, 0, -1.0), (0, 0, 0, 4.0), (1, 0, 0, -1.0), (0, 1, 0, -1.0)], np.float64), 2, np.float64)))
    out.append(("27-point 40^3 x 6 dof (162 per row, runs of 18)", lambda: at.block_expand(*at.stencil_csr(40, 40, 40, at.stencil_points(27), np.float64), 6, np.float64)))
    out.append(("uniform 50..90, stride-3 columns, 600 k rows", lambda: lens_csr(rng.integers(50, 91, size=600_000), 21)))
    out.append(("uniform 50..90, stride-1 columns (one run per row), 600 k rows", lambda: lens_csr(rng.integers(50, 91, size=600_000), 22, spread=1)))

    def pairs():
        rows, k = 2_500_000, 16
        Ap = (np.arange(rows + 1, dtype=np.int64) * k).astype(np.int32)
        base = np.repeat(np.arange(rows, dtype=np.int64), k) + np.tile((np.arange(k) // 2 - 4) * 50 + np.arange(k) % 2, rows)
        return Ap, np.clip(base, 0, rows - 1).astype(np.int32), np.random.default_rng(23).standard_normal(rows * k)
    out.append(("16 per row as 8 pairs of neighbours 50 apart (runs of 2)", pairs))

    def dense_blocks():
        b, nb = 32, 40_000
        rows = b * nb
        Ap = (np.arange(rows + 1, dtype=np.int64) * b).astype(np.int32)
        Aj = (np.repeat(np.arange(nb, dtype=np.int64) * b, b * b).reshape(nb, b, b) + np.arange(b)[None, None, :]).reshape(-1).astype(np.int32)
        return Ap, Aj, np.random.default_rng(24).standard_normal(rows * b)
    out.append(("block-diagonal, dense 32 x 32 blocks (1.28 M rows)", dense_blocks))

    def arrow():
        n = 4_000_000
        lens = np.full(n, 5, np.int64)
        lens[0] = n
        Ap = np.zeros(n + 1, np.int64)
        Ap[1:] = np.cumsum(lens)
        Aj = np.empty(int(Ap[-1]), np.int32)
        Aj[:n] = np.arange(n)
        r = np.arange(1, n, dtype=np.int64)
        body = np.stack([np.zeros(n - 1, np.int64), np.clip(r - 1, 0, n - 1), r, np.clip(r + 1, 0, n - 1), np.clip(r + 2, 0, n - 1)], axis=1)
        Aj[n:] = body.reshape(-1)
        return Ap.astype(np.int32), Aj, np.random.default_rng(25).standard_normal(int(Ap[-1]))
    out.append(("arrow: 4 M rows of 5 under one dense row", arrow))

    def holes():
        Ap, Aj, Ax = at.stencil_csr(200, 200, 200, at.stencil_points(7), np.float64)
        keep = np.random.default_rng(26).random(len(Aj)) > 0.12
        row = np.repeat(np.arange(len(Ap) - 1), np.diff(Ap))
        keep |= Aj == row                                   # the diagonal stays
        lens = np.bincount(row[keep], minlength=len(Ap) - 1)
        Ap2 = np.zeros(len(Ap), np.int64)
        Ap2[1:] = np.cumsum(lens)
        return Ap2.astype(np.int32), Aj[keep], Ax[keep]
    out.append(("7-point 200^3 with 12 % of the off-diagonal entries removed", holes))
    out.append(("thermal2-like x0.3 (37 MB)", lambda: ssl.load("thermal2", 0.3)[:3]))
    return out


def time_us(go, settle_s=0.05, budget_s=0.25):
    """median of 5 groups; group size from a first probe so that slow candidates (csr_scalar on skewed rows: 100 ms+) stay bounded"""
    go()
    torch.cuda.synchronize()
    t0 = time.time()
    go()
    torch.cuda.synchronize()
    once = max(time.time() - t0, 2e-6)
    if once > 0.05:
        return once * 1e6
    n = 0
    while time.time() - t0 < settle_s:  # (synchronising now and then: launches of a slow candidate must not pile up behind the host's clock)
        go()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    launches = int(max(3, min(60, budget_s / 5 / once)))
    out = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            go()
        b.record()
        b.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / launches)
    return sorted(out)[2]


def candidates(mean):
    c = [("csr_scalar", cmi.Config(kernel=cmi.CSR_SCALAR), "explicit")]
    for t in (2, 4, 8, 16, 32, 64):
        c.append((f"csr_vector T={t}", cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=t), "explicit"))
    c.append(("csr_stream (table)", cmi.Config(kernel=cmi.CSR_STREAM), "explicit"))
    c.append(("csr_wave (equal-length tiles)", cmi.Config(kernel=cmi.CSR_STREAM_WAVE), "explicit"))
    c.append(("csr_balanced", cmi.Config(kernel=cmi.CSR_BALANCED), "plan"))
    for v in (1, 2, 4):
        c.append((f"wave tiles V={v}", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=v), "plan"))
    c.append(("wave tiles V=4 + x window", cmi.Config(kernel=cmi.CSR_STREAM_WAVEX, items_per_thread=4), "plan"))
    for v in (2, 4):
        c.append((f"run-compressed copy V={v}", cmi.Config(kernel=cmi.CSR_STREAM_WAVER, items_per_thread=v), "plan"))
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="f64,f32")
    ap.add_argument("--only", default="")
    ap.add_argument("--log", default="")
    ap.add_argument("--set", type=int, default=1)
    args = ap.parse_args()
    log = open(args.log, "w") if args.log else None
    only = [s for s in args.only.split(",") if s]
    summary = []
    for name, build in (matrices() if args.set == 1 else matrices2() if args.set == 2 else matrices3() if args.set == 3 else matrices4()):
        if only and not any(o in name for o in only):
            continue
        t0 = time.time()
        Ap, Aj, Ax = build()
        Ap, Aj = np.asarray(Ap, np.int32), np.asarray(Aj, np.int32)
        rows, nnz = len(Ap) - 1, int(Ap[-1])
        cols = max(rows, int(Aj.max()) + 1 if nnz else 1)
        mean = nnz / max(rows, 1)
        max_len = int(np.diff(Ap).max()) if rows else 0
        for tag in args.dtypes.split(","):
            dt = torch.float64 if tag == "f64" else torch.float32
            vb = 8 if tag == "f64" else 4
            dAp, dAj = torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda()
            dAx = torch.from_numpy(np.asarray(Ax)).cuda().to(dt)
            A = cmi.CsrMatrix(rows, cols, nnz, dAp, dAj, dAx)
            x = cmi.fill_x(cols, dt, "cuda")
            y = torch.empty(rows, dtype=dt, device="cuda")
            cmi.multiply(A, x, y, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
            want = y.clone()
            scale = float(want.abs().max().item()) or 1.0
            alg = cmi.csr_bytes(rows, nnz, vb)
            print(f"# {name} [{tag}]: rows {rows} entries {nnz} mean {mean:.1f}; algorithmic bytes {alg / 1e6:.0f} MB; set-up {time.time() - t0:.1f} s", flush=True)
            res = {}

            def check(label, exact):
                if exact:
                    return torch.equal(y, want)
                tol = (1e-10 if tag == "f64" else 1e-5) * scale * max(1.0, math.sqrt(max_len))
                return bool(((y - want).abs().max() <= tol).item())

            # what the library runs by itself
            plan = cmi.Plan.csr(dt, rows, cols, dAp, dAj)
            kcfg = plan.config()
            y.fill_(float("nan"))
            cmi.spmv_csr_plan(plan, dAp, dAj, dAx, x, y)
            exact_auto = torch.equal(y, want)
            ok = exact_auto or check("auto", False)
            auto_us = time_us(lambda: cmi.spmv_csr_plan(plan, dAp, dAj, dAx, x, y)) if ok else None
            y.fill_(float("nan"))
            cmi.spmv_csr(rows, cols, dAp, dAj, dAx, x, y)
            planless_us = time_us(lambda: cmi.spmv_csr(rows, cols, dAp, dAj, dAx, x, y)) if (torch.equal(y, want) or check("planless", False)) else None
            for label, cfg, how in candidates(mean):
                try:
                    if how == "plan":
                        p = cmi.Plan.csr(dt, rows, cols, dAp, dAj, cfg=cfg)
                        go = lambda p=p: cmi.spmv_csr_plan(p, dAp, dAj, dAx, x, y)  # noqa: E731
                    else:
                        go = lambda cfg=cfg: cmi.spmv_csr(rows, cols, dAp, dAj, dAx, x, y, cfg=cfg)  # noqa: E731
                    y.fill_(float("nan"))
                    go()
                except Exception as e:  # noqa: BLE001
                    res[label] = (None, f"refused ({str(e)[:60]})")
                    continue
                exact = not (label.startswith("csr_vector") or label == "csr_balanced")
                if not check(label, exact):
                    res[label] = (None, "WRONG RESULT")
                    continue
                res[label] = (time_us(go), "")
            timed = {k: v[0] for k, v in res.items() if v[0] is not None}
            best = min(timed, key=timed.get)
            ref_t = 32 if mean > 16 else 16 if mean > 8 else 8 if mean > 4 else 4 if mean > 2 else 2  # the reference's selector (csr_vector_spmv.h:225-258)
            rec = {"matrix": name, "dtype": tag, "rows": rows, "entries": nnz, "auto_kernel": kcfg.kernel, "auto_config": kcfg.as_dict(), "auto_us": auto_us, "auto_bit_exact": exact_auto,
                   "planless_us": planless_us, "best": best, "best_us": timed[best], "regret": auto_us / timed[best] if auto_us else None,
                   "reference_selector_us": timed.get(f"csr_vector T={ref_t}"), "frac_auto": alg / (auto_us * 1e-6) / 8e12 if auto_us else None,
                   "candidates": {k: (v[0], v[1]) for k, v in res.items()}}
            if log:
                log.write(json.dumps(rec) + "\n")
                log.flush()
            summary.append(rec)
            print(f"  AUTO plan: kernel {kcfg.kernel} {auto_us:.1f} us ({rec['frac_auto']:.2f} of peak on CSR's bytes; bit-exact {exact_auto}); plan-less {planless_us:.1f} us; "
                  f"best candidate {best} {timed[best]:.1f} us; regret {rec['regret']:.3f}; the reference's selector (T={ref_t}) {rec['reference_selector_us']:.1f} us", flush=True)
            for k, v in sorted(res.items(), key=lambda kv: kv[1][0] if kv[1][0] is not None else 1e18):
                print(f"      {k:32s} " + (f"{v[0]:10.1f} us" if v[0] is not None else f"{v[1]}"))
            del A, dAp, dAj, dAx, x, y, want, plan
            torch.cuda.empty_cache()
    print("\n== regret of the AUTO plan (time / best candidate's time)")
    for tag in args.dtypes.split(","):
        rs = [r for r in summary if r["dtype"] == tag and r["regret"]]
        if not rs:
            continue
        g = math.exp(sum(math.log(r["regret"]) for r in rs) / len(rs))
        worst = max(rs, key=lambda r: r["regret"])
        print(f"{tag}: {len(rs)} matrices, geometric mean {g:.3f}, worst {worst['regret']:.3f} ({worst['matrix']}: AUTO {worst['auto_us']:.1f} us, {worst['best']} {worst['best_us']:.1f} us)")
        for r in rs:
            print(f"   {r['matrix']:55s} AUTO k{r['auto_kernel']:<3d} {r['auto_us']:9.1f} us   best {r['best']:30s} {r['best_us']:9.1f} us   regret {r['regret']:.3f}   vs the reference's selector x{r['reference_selector_us'] / r['auto_us']:.2f}")


if __name__ == "__main__":
    main()
