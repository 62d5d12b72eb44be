#!/usr/bin/env python3
"""Launch shape of the fused SpMV + <y,p> instance INSIDE the CG iteration (not stand-alone: the vector kernels around it change
what is in the caches): times 100 iterations of spmv_csr_dot -> cg_update -> cg_direction_x on the headline matrix per shape
(block, rows per tile, XCD dealing), HIP events around the loop, no host read in between.

    python tools/cg_dot_shape_probe.py
"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402

A = cmi.poisson5pt(3162, 3162, "csr")
N = A.num_rows
ws = cmi.blas_workspace()
g = torch.Generator(device="cuda").manual_seed(3)
p = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
r = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
x = torch.zeros(N, dtype=torch.float64, device="cuda")
y = torch.empty(N, dtype=torch.float64, device="cuda")
rr = [torch.ones(1, dtype=torch.float64, device="cuda"), torch.ones(1, dtype=torch.float64, device="cuda")]
yp = torch.ones(1, dtype=torch.float64, device="cuda")


def iteration(plan, cur):
    cmi.spmv_csr_dot(N, N, A.row_offsets, A.column_indices, A.values, p, y, p, yp, ws, plan=plan)
    cmi.cg_update(rr[cur], yp, None, y, None, r, rr[cur ^ 1], ws)
    cmi.cg_direction_x(rr[cur ^ 1], rr[cur], yp, r, p, x)


def time_us(plan, iters=100, rounds=3):
    out = []
    for _ in range(rounds):
        p.normal_(generator=g); r.normal_(generator=g); x.zero_(); rr[0].fill_(1.0); rr[1].fill_(1.0)
        for i in range(5):
            iteration(plan, i & 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            iteration(plan, i & 1)
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(out)[len(out) // 2]


base = time_us(A.plan())
print(f"table plan {A.plan().config()}: {base:.1f} us per iteration (GPU time, 3 kernels + 2 folds)")
res = []
shapes = [(blk, 1, rpb, swz) for blk, rpb, swz in itertools.product((256, 512), (128, 176, 192, 256, 384), (0, 16, 32, 64))]
shapes += [(blk, ipt, rpb, 0) for blk, ipt, rpb in ((128, 1, 96), (128, 2, 128), (256, 2, 192), (256, 2, 256), (512, 2, 384), (512, 2, 512), (1024, 1, 768), (256, 4, 256))]
if len(sys.argv) > 1 and sys.argv[1] == "--wide":
    shapes = [sh for sh in shapes if sh[3] == 0]
for blk, ipt, rpb, swz in shapes:
    nt = 2
    if rpb > blk or rpb * 5 + 3 > blk * ipt * 4:
        continue
    cfg = cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, rows_per_block=rpb, items_per_thread=ipt, nontemporal=nt, xcd_swizzle=swz)
    plan = cmi.Plan(cmi.FORMAT_CSR, torch.float64, N, N, A.num_entries, A.row_offsets, cfg=cfg)
    t = time_us(plan)
    res.append((t, blk, ipt, rpb, swz))
    print(f"  block {blk} vectors/lane {ipt} rows/tile {rpb} swizzle {swz}: {t:.1f} us", flush=True)
res.sort()
print("best:", res[:3], "table:", round(base, 1))
