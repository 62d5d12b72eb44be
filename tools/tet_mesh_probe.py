"""3-D unstructured case: P1 Laplacian on a Delaunay tetrahedralisation of random points (~15.5 entries per row: the
upper end of the table's [8, 16) bucket, whose entry was tuned on a 9-point stencil), after reverse Cuthill-McKee.
Table selection vs explicit csr_stream shapes (items per thread x rows per tile x XCD chunk)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
from scipy.spatial import Delaunay  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
import autotune as at  # noqa: E402
import unstructured_probe as u  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
t0 = time.time()
pts = np.random.default_rng(4).random((n, 3))
tet = Delaunay(pts).simplices
pairs = [(a, b) for a in range(4) for b in range(4) if a != b]
i = np.concatenate([tet[:, a] for a, b in pairs])
j = np.concatenate([tet[:, b] for a, b in pairs])
G = sp.coo_matrix((np.ones(len(i)), (i, j)), shape=(n, n)).tocsr()
G.data[:] = -1.0
A = (G + sp.diags(np.asarray(-G.sum(axis=1)).ravel() + 1e-3)).tocsr()
A = u.rcm(A)
rows, nnz = A.shape[0], A.nnz
print(f"Delaunay 3-D: {rows} rows, {nnz} entries ({nnz / rows:.2f} per row, max {np.diff(A.indptr).max()}), built + RCM in {time.time() - t0:.0f} s", flush=True)
Ap, Aj, Ax = (torch.from_numpy(a).cuda() for a in (A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)))
x = cmi.fill_x(rows, device="cuda")
y = torch.empty(rows, dtype=torch.float64, device="cuda")
ref = torch.empty_like(y)
cmi.spmv_csr(rows, rows, Ap, Aj, Ax, x, ref, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
alg = cmi.csr_bytes(rows, nnz)
timer = at.Timer(cmi, torch)
mean = nnz / rows
cfgs = [("table", None)]
for blk in (128, 256, 512):
    for ipt in (1, 2, 4):
        fit = int((blk * ipt * 4 - 3) / mean)
        for rpb in sorted({max(1, fit // 16 * 16), max(1, (fit * 7 // 8) // 16 * 16)}):
            for swz in (0, 8, 32):
                cfgs.append((f"block {blk} ipt {ipt} rows/tile {rpb} swz {swz}", cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, items_per_thread=ipt, rows_per_block=rpb, nontemporal=2, xcd_swizzle=swz)))
t = {k: [] for k, _ in cfgs}
for k, c in cfgs:
    y.fill_(1.0)
    cmi.spmv_csr(rows, rows, Ap, Aj, Ax, x, y, cfg=c)
    assert torch.equal(y, ref), k
for _ in range(5):
    for k, c in cfgs:
        t[k].append(timer.time(lambda: cmi.spmv_csr(rows, rows, Ap, Aj, Ax, x, y, cfg=c), 30))
res = sorted((float(np.median(v)), k) for k, v in t.items())
print(f"{alg / 1e6:.0f} MB = {alg / 6.2e12 * 1e6:.1f} us at 6.2 TB/s; table config {cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, rows, rows, nnz).as_dict()}")
for med, k in res[:8] + [r for r in res if r[1] == "table"]:
    print(f"   {med * 1e3:7.1f} us  {alg / med / 1e9:5.2f} TB/s  {k}", flush=True)
