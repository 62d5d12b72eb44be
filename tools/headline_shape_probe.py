"""Headline matrix (poisson5pt 3162^2, CSR, f64): csr_stream launch shapes around the tuned one, timed in interleaved
rounds (HIP events, 30 launches per sample, median of 7): rows per tile x XCD chunk x block size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
import autotune as at  # noqa: E402

A = cmi.poisson5pt(3162, 3162, "csr")
x = cmi.fill_x(A.num_rows, device="cuda")
y = torch.empty(A.num_rows, dtype=torch.float64, device="cuda")
ref = torch.empty_like(y)
cmi.multiply(A, x, ref, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
timer = at.Timer(cmi, torch)
cfgs = []
for blk, rpbs in ((256, (128, 144, 160, 176, 192, 204)), (128, (64, 80, 96)), (512, (352, 384, 400))):
    for rpb in rpbs:
        for swz in (0, 16, 32, 48, 64, 96):
            cfgs.append(cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, rows_per_block=rpb, items_per_thread=1, nontemporal=2, xcd_swizzle=swz))
ok = []
for c in cfgs:
    y.fill_(10.0)
    cmi.multiply(A, x, y, cfg=c)
    if torch.equal(y, ref):
        ok.append(c)
times = {id(c): [] for c in ok}
for _ in range(7):
    for c in ok:
        times[id(c)].append(timer.time(lambda: cmi.multiply(A, x, y, cfg=c), 30))
rows = sorted((float(np.median(times[id(c)])), min(times[id(c)]), c) for c in ok)
alg = cmi.csr_bytes(A.num_rows, A.num_entries)
for med, mn, c in rows[:25]:
    print(f"{med * 1e3:7.1f} us (min {mn * 1e3:6.1f})  {alg / med / 1e9:5.2f} TB/s  block {c.block_size} rows/tile {c.rows_per_block} xcd_swizzle {c.xcd_swizzle}", flush=True)
print("...")
for med, mn, c in rows[-3:]:
    print(f"{med * 1e3:7.1f} us (min {mn * 1e3:6.1f})  {alg / med / 1e9:5.2f} TB/s  block {c.block_size} rows/tile {c.rows_per_block} xcd_swizzle {c.xcd_swizzle}", flush=True)
