#!/usr/bin/env python3
"""Lanes per row for wide ELL matrices (the THREADS_PER_ROW parameter of the reference's KTT ELL kernel,
cusp/system/cuda/ktt/kernels/ell_kernel.h:102-109,165-173): times cmi_spmv_ell_f64/f32 with 1, 2, 4, 8, 16 lanes per row and
with the NULL config (auto rule, csrc/common.h kEllSlice*) on few-row / wide-row shapes, and prints one table row per shape.
Every lane count is checked against the one-lane result (1e-6 of sum |a_ij x_j|).

    python tools/ell_wide_probe.py [f64|f32]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "f64"
dt = torch.float64 if tag == "f64" else torch.float32
vb = 8 if tag == "f64" else 4
SHAPES = [(2000, 512), (10000, 256), (30000, 128), (60000, 128), (100000, 64), (200000, 64), (120000, 32), (400000, 32),
          (1000000, 32), (50000, 24), (3000000, 16)]
g = torch.Generator(device="cuda").manual_seed(5)


def build(rows, width):
    """column-major ELL arrays on the device: row i holds `len_i` in [width/2, width] columns from a band around i."""
    pitch = (rows + 31) // 32 * 32
    lens = torch.randint(width // 2, width + 1, (rows,), device="cuda", generator=g)
    slot = torch.arange(width, device="cuda").view(width, 1)
    band = 8 * width
    base = torch.arange(rows, device="cuda").view(1, rows)
    cols = (base + slot * (band // width) + torch.randint(0, band // width, (width, rows), device="cuda", generator=g)) % rows
    valid = slot < lens.view(1, rows)
    Aj = torch.full((width, pitch), -1, dtype=torch.int32, device="cuda")
    Aj[:, :rows] = torch.where(valid, cols, torch.full_like(cols, -1)).to(torch.int32)
    Ax = torch.zeros((width, pitch), dtype=dt, device="cuda")
    Ax[:, :rows] = torch.where(valid, torch.randn((width, rows), dtype=dt, device="cuda", generator=g), torch.zeros((), dtype=dt, device="cuda"))
    return pitch, Aj.reshape(-1), Ax.reshape(-1), lens


def time_us(fn, iters=50, rounds=5):
    for _ in range(5):
        fn()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(best)[len(best) // 2]


print(f"{tag}: us per multiply (GB/s algorithmic); * = what the NULL config runs")
print(f"{'rows':>8} {'width':>5} | " + " ".join(f"{'lanes ' + str(l):>14}" for l in (1, 2, 4, 8, 16)) + " |        auto")
for rows, width in SHAPES:
    pitch, Aj, Ax, lens = build(rows, width)
    x = torch.randn(rows, dtype=dt, device="cuda", generator=g)
    y = torch.empty(rows, dtype=dt, device="cuda")
    nbytes = width * pitch * (4 + vb) + 2 * vb * rows
    ref = None
    cells = []
    times = {}
    for lanes in (1, 2, 4, 8, 16):
        cfg = cmi.Config(kernel=cmi.ELL_ROW, threads_per_row=lanes, nontemporal=3)
        cmi.spmv_ell(rows, rows, width, pitch, Aj, Ax, x, y, cfg=cfg)
        if ref is None:
            ref = y.clone()
            mag = torch.zeros(rows, dtype=dt, device="cuda")
            cmi.spmv_ell(rows, rows, width, pitch, Aj, Ax.abs(), x.abs(), mag, cfg=cfg)
        else:
            tol = (1e-6 if tag == "f64" else 1e-4) * mag + 1e-30
            assert bool(((y - ref).abs() <= tol).all()), (rows, width, lanes)
        us = time_us(lambda: cmi.spmv_ell(rows, rows, width, pitch, Aj, Ax, x, y, cfg=cfg))
        times[lanes] = us
        cells.append(f"{us:7.1f} ({nbytes / us / 1e3:5.0f})")
    auto = time_us(lambda: cmi.spmv_ell(rows, rows, width, pitch, Aj, Ax, x, y))
    plan = cmi.Plan(cmi.FORMAT_ELL, dt, rows, rows, rows * width, None)
    print(f"{rows:8d} {width:5d} | " + " ".join(cells) + f" | {auto:7.1f} {'slices' if not plan.info()['storage_order_sums'] else 'row'}"
          f"  best lanes {min(times, key=times.get)}", flush=True)
    del Aj, Ax
