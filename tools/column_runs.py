"""Run-length histogram of CONSECUTIVE column indices inside the rows of the configs[3] matrices: how long are the runs a
run-compressed column copy (CMI_CSR_STREAM_WAVER, csrc/spmv_csr_runs.hip) would store, and how many pieces per entry remain at each cap.
Host only (numpy); output committed as profiles/r04_column_runs.txt."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import suitesparse_like as sl
for name in ("thermal2", "ldoor", "nlpkkt120"):
    t = time.time()
    Ap, Aj, Ax = sl.GENERATORS[name](1.0)
    nnz = len(Aj); rows = len(Ap) - 1
    # run = maximal stretch of consecutive columns inside a row
    brk = np.ones(nnz, bool)
    brk[1:] = (Aj[1:] != Aj[:-1] + 1)
    brk[Ap[:-1][Ap[:-1] < nnz]] = True
    starts = np.nonzero(brk)[0]
    lens = np.diff(np.append(starts, nnz))
    h = np.bincount(lens)
    print(name, "rows", rows, "nnz", nnz, "runs", len(lens), "mean run %.3f" % (nnz / len(lens)), "gen %.0fs" % (time.time() - t))
    tot = 0
    for L in range(1, len(h)):
        if h[L]: print("   len %2d: %9d runs  %5.1f%% of entries" % (L, h[L], 100.0 * L * h[L] / nnz))
    for cap in (2, 3, 4, 6, 8, 16):
        nr = np.sum((lens + cap - 1) // cap)
        print("   cap %2d: %.3f entries per run-piece; pieces/entry %.3f" % (cap, nnz / nr, nr / nnz))
