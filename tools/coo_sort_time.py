"""Time cmi_coo_sort_by_row on the headline matrix's entries in a random order (the one-off step in front of the plan path) and the
multiplies either side of it: atomics kernels on the unsorted entries vs the plan path on the sorted ones.
usage: python3 tools/coo_sort_time.py [--grid 3162]"""
import argparse
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import cusp_autotuned_amd as cmi  # noqa: E402


def timed(f, reps):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=3162)
    a = ap.parse_args()
    A = cmi.poisson5pt(a.grid, a.grid, fmt="coo", dtype=torch.float64)
    n = A.num_entries
    x = torch.rand(A.num_cols, dtype=torch.float64, device="cuda")
    y = torch.empty(A.num_rows, dtype=torch.float64, device="cuda")
    y_sorted = torch.empty_like(y)
    cmi.multiply(A, x, y_sorted)
    t_plan_sorted = timed(lambda: cmi.multiply(A, x, y_sorted), 20)
    for name, perm in (("random order", torch.randperm(n, device="cuda")),
                       ("blocks of 4096 entries shuffled", (torch.randperm((n + 4095) // 4096, device="cuda")[:, None] * 4096 + torch.arange(4096, device="cuda")[None, :]).reshape(-1))):
        perm = perm[perm < n]
        U = cmi.CooMatrix(A.num_rows, A.num_cols, n, A.row_indices[perm].contiguous(), A.column_indices[perm].contiguous(), A.values[perm].contiguous())
        cmi.multiply(U, x, y)
        err = float((y - y_sorted).abs().max())
        t_unsorted = timed(lambda: cmi.multiply(U, x, y), 5)
        for and_column in (False, True):
            W = cmi.CooMatrix(U.num_rows, U.num_cols, n, U.row_indices.clone(), U.column_indices.clone(), U.values.clone())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            (W.sort_by_row_and_column if and_column else W.sort_by_row)()
            torch.cuda.synchronize()
            t_sort = (time.perf_counter() - t0) * 1e3
            cmi.multiply(W, x, y)
            t_after = timed(lambda: cmi.multiply(W, x, y), 20)
            print(f"{name:34s} {n} entries: multiply as given (atomics kernels) {t_unsorted:8.1f} us, max |diff| vs sorted {err:.2e}; "
                  f"{'sort_by_row_and_column' if and_column else 'sort_by_row':22s} once {t_sort:7.2f} ms = {t_sort * 1e3 / t_unsorted:5.1f} such multiplies; "
                  f"multiply afterwards (plan: {cmi.kernel_name(W.plan().config().kernel) if hasattr(cmi, 'kernel_name') else W.plan().config().kernel}) {t_after:7.1f} us; "
                  f"sorted original {t_plan_sorted:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
