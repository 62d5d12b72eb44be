import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import cusp_autotuned_amd as cmi, autotune as at, unstructured_probe as u
timer = at.Timer(cmi, torch)
def run(label, Ap, Aj, Ax):
    rows, nnz = len(Ap) - 1, len(Aj)
    dAp, dAj, dAx = (torch.from_numpy(a).cuda() for a in (Ap.astype(np.int32), Aj.astype(np.int32), Ax.astype(np.float64)))
    x = cmi.fill_x(rows, device="cuda"); y = torch.empty(rows, dtype=torch.float64, device="cuda")
    alg = cmi.csr_bytes(rows, nnz)
    cfgs = [None] + [cmi.Config(kernel=cmi.CSR_STREAM, block_size=b, rows_per_block=r, items_per_thread=1, nontemporal=2, xcd_swizzle=s)
                     for b, rs in ((256, (176, 144, 128, 96)), (128, (80, 64, 48)), (64, (32, 24))) for r in rs for s in (0, 8, 32)]
    t = {i: [] for i in range(len(cfgs))}
    for _ in range(5):
        for i, c in enumerate(cfgs):
            t[i].append(timer.time(lambda: cmi.spmv_csr(rows, rows, dAp, dAj, dAx, x, y, cfg=c), 30))
    res = sorted((float(np.median(t[i])), i) for i in range(len(cfgs)))
    print(f"{label}: {rows} rows, {nnz} entries, {alg/6.2e12*1e6:.1f} us at 6.2 TB/s")
    for med, i in res[:5] + [r for r in res if r[1] == 0]:
        c = cfgs[i]
        print(f"   {med*1e3:7.1f} us {alg/med/1e9:5.2f} TB/s  " + ("table" if c is None else f"block {c.block_size} rows/tile {c.rows_per_block} swz {c.xcd_swizzle}"), flush=True)
for n in (300_000, 1_228_045):
    A = u.rcm(u.delaunay_laplacian(n))
    run(f"Delaunay RCM {n}", A.indptr, A.indices, A.data)
for g in (500, 1000, 2000):
    Ap, Aj, Ax = at.stencil_csr(g, g, 1, at.stencil_points(9), np.float64)
    run(f"stencil9 {g}^2", Ap, Aj, Ax)
for g in (700, 1500):
    A = cmi.poisson5pt(g, g, "csr")
    run(f"poisson5pt {g}^2", A.row_offsets.cpu().numpy(), A.column_indices.cpu().numpy(), A.values.cpu().numpy())
