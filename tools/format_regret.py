#!/usr/bin/env python3
"""ELL and DIA: does the per-bucket tuning table generalise?  tools/autotune.py --per-bucket tuned each (format, value type, width bucket)
entry on ONE matrix of that width; here the table's choice (what cusp::multiply on an ell_matrix / dia_matrix runs: cmi.multiply with no
config) is timed against the WHOLE search space of tools/autotune.py (ell_space / dia_space: block size x rows per lane x cache policy x
XCD dealing, plus the lanes-per-row shapes for wide ELL rows) on OTHER matrices -- every configuration validated against csr_scalar
before it is timed, as the reference's KTT tuner validates each configuration (cusp/system/cuda/ktt/kernel.h:37-62).

    regret = time(table's choice) / min(time over the space)

    python3 tools/format_regret.py [--dtypes f64,f32] [--formats ell,dia] [--log out.jsonl]
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

P5 = [(0, -1, 0, -1.0), (-1, 0, 0, -1.0), (0, 0, 0, 4.0), (1, 0, 0, -1.0), (0, 1, 0, -1.0)]


def matrices():
    rng = np.random.default_rng(808)
    out = []  # (name, builder, formats)
    out.append(("poisson5pt 3162^2 (the tuning matrix of its bucket)", lambda: at.stencil_csr(3162, 3162, 1, P5, np.float64), ("ell", "dia")))
    out.append(("poisson5pt 10000 x 1250 (configs[4]'s rank block, square)", lambda: at.stencil_csr(10000, 1250, 1, P5, np.float64), ("ell", "dia")))
    out.append(("tridiagonal 10^7", lambda: at.stencil_csr(10_000_000, 1, 1, [(-1, 0, 0, -1.0), (0, 0, 0, 2.0), (1, 0, 0, -1.0)], np.float64), ("ell", "dia")))
    out.append(("7-point 215^3", lambda: at.stencil_csr(215, 215, 215, at.stencil_points(7), np.float64), ("ell", "dia")))
    out.append(("9-point 3000^2", lambda: at.stencil_csr(3000, 3000, 1, at.stencil_points(9), np.float64), ("ell", "dia")))
    out.append(("27-point 140^3", lambda: at.stencil_csr(140, 140, 140, at.stencil_points(27), np.float64), ("ell", "dia")))
    out.append(("thermal2-like x3 (width 11)", lambda: ssl.load("thermal2", 3.0)[:3], ("ell",)))
    out.append(("nlpkkt120-like x0.5 (width 28)", lambda: ssl.load("nlpkkt120", 0.5)[:3], ("ell",)))
    out.append(("ldoor-like x0.5 (width 77)", lambda: ssl.load("ldoor", 0.5)[:3], ("ell",)))

    def uniform():
        import auto_regret as ar
        return ar.lens_csr(rng.integers(1, 17, size=4_000_000), 1)
    out.append(("uniform 1..16, stride-3 columns (width 16, half of it padding)", uniform, ("ell",)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="f64,f32")
    ap.add_argument("--formats", default="ell,dia")
    ap.add_argument("--log", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    logf = open(args.log, "w") if args.log else None

    def log(rec):
        if logf and rec.get("status") != "Ok":
            logf.write(json.dumps(rec) + "\n")

    timer = at.Timer(cmi, torch)
    scalar = cmi.Config(kernel=cmi.CSR_SCALAR)
    summary = []
    for name, build, fmts in matrices():
        t0 = time.time()
        Ap, Aj, Ax = build()
        rows, nnz = len(Ap) - 1, int(Ap[-1])
        for tag in args.dtypes.split(","):
            tdt = torch.float64 if tag == "f64" else torch.float32
            vb = 8 if tag == "f64" else 4
            A = cmi.CsrMatrix(rows, rows, nnz, torch.from_numpy(np.asarray(Ap, np.int32)).cuda(), torch.from_numpy(np.asarray(Aj, np.int32)).cuda(), torch.from_numpy(np.asarray(Ax)).cuda().to(tdt))
            x = cmi.fill_x(rows, tdt, "cuda")
            y = torch.empty(rows, dtype=tdt, device="cuda")
            cmi.multiply(A, x, y, cfg=scalar)
            want = y.clone()
            scale = float(want.abs().max().item()) or 1.0
            for fmt in fmts:
                if fmt not in args.formats.split(","):
                    continue
                M = cmi.convert(A, fmt)
                if fmt == "ell":
                    width = M.num_entries_per_row
                    space = at.ell_space(cmi, False, width)
                    alg = cmi.ell_bytes(rows, width, M.pitch, vb)
                    exact = (cmi.ELL_ROW,)
                else:
                    width = M.diagonal_offsets.numel()
                    space = at.dia_space(cmi, False)
                    alg = cmi.dia_bytes(rows, width, M.pitch, vb)
                    exact = (cmi.DIA_ROW,)

                def check(cfg):
                    y.fill_(10.0)
                    cmi.multiply(M, x, y, cfg=cfg)
                    if cfg.kernel in exact and cfg.threads_per_row <= 1:
                        return bool(torch.equal(y, want)), "bit-exact required"
                    return bool(((y - want).abs().max() <= (1e-6 if tag == "f64" else 1e-5) * scale).item()), "tolerance"

                y.fill_(10.0)
                cmi.multiply(M, x, y)  # the table's choice
                table_ok = bool(torch.equal(y, want)) or bool(((y - want).abs().max() <= (1e-6 if tag == "f64" else 1e-5) * scale).item())
                label = f"{fmt}/{tag}/{name}"
                best, ms, res = at.tune_one(cmi, torch, timer, label, space, lambda cfg: cmi.multiply(M, x, y, cfg=cfg), check, args.iters, args.rounds, log, alg)
                # the table's choice, timed the same way, interleaved with the best three of the space
                top = [r[1] for r in res[:3]]
                seen = {"table": [], **{i: [] for i in range(len(top))}}
                for _ in range(5):
                    seen["table"].append(timer.time(lambda: cmi.multiply(M, x, y), args.iters * 2))
                    for i, cfg in enumerate(top):
                        seen[i].append(timer.time(lambda cfg=cfg: cmi.multiply(M, x, y, cfg=cfg), args.iters * 2))
                t_table = float(np.median(seen["table"]))
                t_best = min(float(np.median(seen[i])) for i in range(len(top)))
                i_best = min(range(len(top)), key=lambda i: float(np.median(seen[i])))
                sel = cmi.tuning_select(cmi.FORMAT_ELL if fmt == "ell" else cmi.FORMAT_DIA, cmi.F64 if tag == "f64" else cmi.F32, rows, rows, rows * width)
                rec = {"matrix": name, "format": fmt, "dtype": tag, "rows": rows, "width": int(width), "table_config": sel.as_dict(), "table_us": t_table * 1e3, "table_ok": table_ok,
                       "best_config": top[i_best].as_dict(), "best_us": t_best * 1e3, "regret": t_table / t_best, "frac_table": alg / (t_table * 1e-3) / 8e12, "space": len(space), "valid": len(res)}
                summary.append(rec)
                if logf:
                    logf.write(json.dumps(rec) + "\n")
                    logf.flush()
                print(f"{label}: width {width}, {len(res)}/{len(space)} valid; table {t_table * 1e3:.1f} us ({rec['frac_table']:.3f} of peak, ok {table_ok}) {sel}; "
                      f"best of the space {t_best * 1e3:.1f} us {top[i_best]}; regret {rec['regret']:.3f}   [{time.time() - t0:.0f} s]", flush=True)
                del M
                torch.cuda.empty_cache()
            del A, x, y, want
            torch.cuda.empty_cache()
    print("\n== regret of the table's choice (time / best of the whole space)")
    for fmt in args.formats.split(","):
        for tag in args.dtypes.split(","):
            rs = [r for r in summary if r["format"] == fmt and r["dtype"] == tag]
            if rs:
                g = math.exp(sum(math.log(r["regret"]) for r in rs) / len(rs))
                w = max(rs, key=lambda r: r["regret"])
                print(f"{fmt} {tag}: {len(rs)} matrices, geometric mean {g:.3f}, worst {w['regret']:.3f} ({w['matrix']})")
                for r in rs:
                    print(f"   {r['matrix'][:62]:62s} width {r['width']:3d}  table {r['table_us']:8.1f} us ({r['frac_table']:.3f})  best {r['best_us']:8.1f} us  regret {r['regret']:.3f}")


if __name__ == "__main__":
    main()
