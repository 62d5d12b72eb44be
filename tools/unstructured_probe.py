"""A thermal2-like matrix without the file: P1 finite-element Laplacian on a Delaunay triangulation of random points
(the SuiteSparse thermal2: 1 228 045 rows, 8 580 313 entries, ~7 per row, unstructured steady-state thermal FEM -- figures
quoted from the collection's page from memory; the file itself is not available here).  Rows in the generator's RANDOM
numbering (no locality at all: every x gather misses L1/L2) and after reverse Cuthill-McKee (what a user would do before
solving).  Every CSR kernel family with its threads-per-row sweep -- BASELINE.json configs[3]'s 'CSR-vector threads-per-row
autotune' -- checked against csr_scalar before it is timed.

    python tools/unstructured_probe.py [points=1228045]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402


def delaunay_laplacian(npoints, seed=2):
    """CSR (int32 / f64) of the P1 stiffness-like graph Laplacian of a Delaunay triangulation of random points in
    the unit square: A_ij = -1 for mesh neighbours, A_ii = degree + 1e-3 (SPD), columns ascending in every row."""
    import scipy.sparse as sp
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = rng.random((npoints, 2))
    tri = Delaunay(pts).simplices
    i = np.concatenate([tri[:, 0], tri[:, 1], tri[:, 2], tri[:, 1], tri[:, 2], tri[:, 0]])
    j = np.concatenate([tri[:, 1], tri[:, 2], tri[:, 0], tri[:, 0], tri[:, 1], tri[:, 2]])
    G = sp.coo_matrix((np.ones(len(i)), (i, j)), shape=(npoints, npoints)).tocsr()
    G.data[:] = -1.0                                   # duplicates (shared edges) were summed: back to -1
    deg = np.asarray(-G.sum(axis=1)).ravel()
    A = (G + sp.diags(deg + 1e-3)).tocsr()
    A.sort_indices()
    return A


def rcm(A):
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    perm = reverse_cuthill_mckee(A, symmetric_mode=True)
    B = A[perm][:, perm].tocsr()
    B.sort_indices()
    return B


def main():
    import torch
    import cusp_autotuned_amd as cmi
    import autotune as at
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_228_045
    t0 = time.time()
    A = delaunay_laplacian(n)
    print(f"Delaunay P1 Laplacian: {A.shape[0]} rows, {A.nnz} entries ({A.nnz / A.shape[0]:.2f} per row, max {np.diff(A.indptr).max()}); "
          f"built in {time.time() - t0:.0f} s", flush=True)
    timer = at.Timer(cmi, torch)
    for label, M in (("random numbering", A), ("reverse Cuthill-McKee", rcm(A))):
        Ap, Aj, Ax = (torch.from_numpy(a).cuda() for a in (M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)))
        rows, nnz = M.shape[0], M.nnz
        band = int(np.abs(M.indices - np.repeat(np.arange(rows), np.diff(M.indptr))).max())
        x = cmi.fill_x(rows, device="cuda")
        y = torch.empty(rows, dtype=torch.float64, device="cuda")
        ref = torch.empty_like(y)
        cmi.spmv_csr(rows, rows, Ap, Aj, Ax, x, ref, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
        alg = cmi.csr_bytes(rows, nnz)
        print(f"{label}: bandwidth {band}; {alg / 1e6:.0f} MB = {alg / 6.2e12 * 1e6:.1f} us at 6.2 TB/s", flush=True)
        variants = [("table (cfg NULL)", None), ("csr_scalar", cmi.Config(kernel=cmi.CSR_SCALAR))]
        variants += [(f"csr_vector T={t}", cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=t)) for t in (2, 4, 8, 16, 32)]
        variants += [("csr_stream strict", cmi.Config(kernel=cmi.CSR_STREAM, threads_per_row=1)),
                     ("csr_stream 128 rows/tile, chunks of 8", cmi.Config(kernel=cmi.CSR_STREAM, rows_per_block=128, xcd_swizzle=8, nontemporal=2)),
                     ("csr_stream_pipe", cmi.Config(kernel=cmi.CSR_STREAM_PIPE)), ("csr_balanced", cmi.Config(kernel=cmi.CSR_BALANCED))]
        for name, cfg in variants:
            y.fill_(3.0)
            cmi.spmv_csr(rows, rows, Ap, Aj, Ax, x, y, cfg=cfg)
            err = float((y - ref).abs().max() / ref.abs().max())
            assert err <= 1e-12, (name, err)
            t = min(timer.time(lambda: cmi.spmv_csr(rows, rows, Ap, Aj, Ax, x, y, cfg=cfg), 30) for _ in range(3))
            print(f"    {name:42s} {t * 1e3:8.1f} us  {alg / t / 1e9:5.2f} TB/s  {2 * nnz / t / 1e6:7.1f} GFLOP/s   max rel diff vs scalar {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
