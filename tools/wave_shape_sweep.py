#!/usr/bin/env python3
"""Launch shape of the wave-tile kernel (CMI_CSR_STREAM_WAVE) on the headline matrix: waves per workgroup x rows per wave x cache
policy x XCD dealing, each checked bit for bit against the plan's result.    python tools/wave_shape_sweep.py [f64|f32]"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
from stream_shape_ab import time_us  # noqa: E402

dt = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else torch.float64
A = cmi.poisson5pt(3162, 3162, "csr", dtype=dt)
N = A.num_rows
x = cmi.fill_x(N, dt, "cuda")
want = torch.empty(N, dtype=dt, device="cuda")
base = A.plan()
cmi.multiply(A, x, want)
print("plan:", base.config())
cands = []
for blk, rpw, nt, swz in itertools.product((64, 128, 256, 512), (64, 32), (2, 3), (0, 16, 32, 64, 128, 256)):
    cfg = cmi.Config(kernel=cmi.CSR_STREAM_WAVE, block_size=blk, items_per_thread=5, rows_per_block=rpw * (blk // 64), nontemporal=nt, xcd_swizzle=swz)
    p = cmi.Plan.csr(dt, N, N, A.row_offsets, A.column_indices, cfg=cfg)
    y = torch.full((N,), 7.0, dtype=dt, device="cuda")
    cmi.spmv_csr_plan(p, A.row_offsets, A.column_indices, A.values, x, y)
    assert torch.equal(y, want), cfg
    cands.append((cfg, p, y))
fns = [(lambda p=p, y=y: cmi.spmv_csr_plan(p, A.row_offsets, A.column_indices, A.values, x, y)) for _, p, y in cands]
fns.append(lambda: cmi.multiply(A, x, want))
t = time_us(fns, 20, 3)
rank = sorted(zip(t[:-1], [c for c, _, _ in cands]), key=lambda r: r[0])
for us, c in rank[:12]:
    print(f"  {us:7.1f} us  block {c.block_size} rows/wave {c.rows_per_block // (c.block_size // 64)} policy {c.nontemporal} swizzle {c.xcd_swizzle}")
print(f"  ... {len(rank)} shapes, slowest {rank[-1][0]:.1f} us; the plan's shape: {t[-1]:.1f} us")
