"""3-D stencil matrices (27-point on 150^3, and x3 dof blocks): does dealing LARGE chunks of tiles to each XCD
help?  The x strips a row gathers from the z-1 / z+1 planes lie nx*ny rows away, so only a chunk that spans more
than a plane lets one L2 see a strip twice.  Prints us per SpMV and algorithmic TB/s for xcd_swizzle values."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
import autotune as at  # noqa: E402

timer = at.Timer(cmi, torch)
for label, g, dof in (("stencil27 150^3", 150, 1), ("27-point x3 dof 70^3", 70, 3), ("stencil7 200^3", 200, 0)):
    if dof == 0:
        Ap, Aj, Ax = at.stencil_csr(g, g, g, at.stencil_points(7), np.float64)
    else:
        Ap, Aj, Ax = at.stencil_csr(g, g, g, at.stencil_points(27), np.float64)
        if dof > 1:
            Ap, Aj, Ax = at.block_expand(Ap, Aj, Ax, dof, np.float64)
    n, nnz = len(Ap) - 1, len(Aj)
    dAp, dAj, dAx = (torch.from_numpy(a).cuda() for a in (Ap, Aj, Ax))
    x = torch.rand(n, dtype=torch.float64, device="cuda")
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    ref = torch.empty_like(y)
    base = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, n, n, nnz)
    cmi.spmv_csr(n, n, dAp, dAj, dAx, x, ref, cfg=base)
    alg = cmi.csr_bytes(n, nnz)
    print(f"{label}: rows {n}, entries {nnz}, mean {nnz / n:.1f}; table config {base.as_dict()}", flush=True)
    tiles = -(-n // base.rows_per_block)
    for swz in (0, 1, 8, 32, 128, 512, 2048, 8192):
        cfg = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, n, n, nnz)
        cfg.xcd_swizzle = swz
        fn = lambda: cmi.spmv_csr(n, n, dAp, dAj, dAx, x, y, cfg=cfg)  # noqa: E731
        fn()
        assert torch.equal(y, ref) or float((y - ref).abs().max()) < 1e-9
        best = min(timer.time(fn, 30) for _ in range(3))
        print(f"    xcd_swizzle {swz:5d} ({'chunk rows ' + str(swz * base.rows_per_block) if swz > 1 else 'launch order' if swz == 0 else 'eighths'}; {tiles} tiles): "
              f"{best * 1e3:7.1f} us  {alg / best / 1e9:6.2f} TB/s", flush=True)
