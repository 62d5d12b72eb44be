"""Does the tuning table generalise between the mean row lengths it was tuned at?  Synthetic matrices (Poisson-distributed
row lengths, columns clustered within +-2000 of the diagonal) at means BETWEEN the tuning points; the table's selection
against a small search over csr_stream shapes (items per thread x rows per tile x lanes per row).  Reports the gap."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
import autotune as at  # noqa: E402

timer = at.Timer(cmi, torch)
rows = 1_000_000
for mean in (2.2, 4.0, 6.5, 7.5, 12.0, 15.0, 20.0, 40.0, 64.0, 100.0, 140.0):
    r = rows if mean < 30 else rows // 2
    Ap, Aj, Ax = at.synthetic_csr(r, r, mean, int(mean * 10), np.float64)
    nnz = len(Aj)
    dAp, dAj, dAx = (torch.from_numpy(a).cuda() for a in (Ap, Aj, Ax))
    x = cmi.fill_x(r, device="cuda")
    y = torch.empty(r, dtype=torch.float64, device="cuda")
    ref = torch.empty_like(y)
    cmi.spmv_csr(r, r, dAp, dAj, dAx, x, ref, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
    m = nnz / r
    cfgs = [("table", None)]
    for blk in (256, 512):
        for ipt in (1, 2, 4):
            for tpr in (0, 4, 16, 32):
                if tpr and (tpr > 4 * m or tpr * 32 < m):
                    continue
                fit = int((blk * ipt * 4 - 3) / m)
                fit = min(fit, 4 * (blk // max(tpr, 1)))
                if fit < 1:
                    continue
                for frac in (1.0, 0.85):
                    rpb = max(1, int(fit * frac))
                    if rpb >= 32:
                        rpb = rpb // 16 * 16
                    cfgs.append((f"b{blk} ipt{ipt} tpr{tpr} rows{rpb}", cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, items_per_thread=ipt,
                                                                                   rows_per_block=rpb, threads_per_row=tpr, nontemporal=2, xcd_swizzle=8)))
    good = []
    for k, c in cfgs:
        y.fill_(1.0)
        try:
            cmi.spmv_csr(r, r, dAp, dAj, dAx, x, y, cfg=c)
        except cmi.CmiError:
            continue
        if float((y - ref).abs().max()) <= 1e-9 * float(ref.abs().max()):
            good.append((k, c))
    t = {k: [] for k, _ in good}
    for _ in range(3):
        for k, c in good:
            t[k].append(timer.time(lambda: cmi.spmv_csr(r, r, dAp, dAj, dAx, x, y, cfg=c), 20))
    res = sorted((float(np.median(v)), k) for k, v in t.items())
    tab = [m_ for m_, k in res if k == "table"][0]
    sel = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, r, r, nnz)
    print(f"mean {m:6.2f}: table {tab * 1e3:7.1f} us (b{sel.block_size} ipt{sel.items_per_thread} tpr{sel.threads_per_row} rows{sel.rows_per_block})   "
          f"best {res[0][0] * 1e3:7.1f} us ({res[0][1]})   gap {100 * (tab / res[0][0] - 1):5.1f} %", flush=True)
