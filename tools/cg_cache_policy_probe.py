"""Does keeping the matrix streams out of the caches (nt loads) pay inside CG, where four 80 MB vectors compete
with the 640 MB of matrix streams for the 256 MiB Infinity Cache?  Times the three kernels of a fused CG
iteration back to back (scalars are whatever they are: timing only) for each SpMV cache policy (tools)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cusp_autotuned_amd as cmi
    from cusp_autotuned_amd import binding as B
    A = cmi.poisson5pt(3162, 3162, "csr")
    n = A.num_rows
    base = B.tuning_select(B.FORMAT_CSR, B.F64, n, n, A.num_entries)
    ws = cmi.blas_workspace()
    for nt in (2, 3, 0, 1):
        cfg = cmi.Config(kernel=base.kernel, block_size=base.block_size, rows_per_block=base.rows_per_block,
                         items_per_thread=base.items_per_thread, nontemporal=nt, xcd_swizzle=base.xcd_swizzle)
        p = torch.full((n,), 1e-3, dtype=torch.float64, device="cuda")
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        r = torch.full((n,), 1e-3, dtype=torch.float64, device="cuda")
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        rr = [torch.ones(1, dtype=torch.float64, device="cuda") for _ in range(2)]
        yp = torch.ones(1, dtype=torch.float64, device="cuda")

        def iteration(cur):
            B.spmv_csr_dot(n, n, A.row_offsets, A.column_indices, A.values, p, y, p, yp, ws, cfg=cfg)
            B.cg_update(rr[cur], yp, p, y, x, r, rr[cur ^ 1], ws)
            B.cg_direction(rr[cur ^ 1], rr[cur], r, p)

        for i in range(5):
            iteration(i & 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 200
        for i in range(iters):
            iteration(i & 1)
        torch.cuda.synchronize()
        print(f"nontemporal={nt} (bit0: nt loads of Aj/Ax, bit1: nt stores of y): {(time.perf_counter() - t0) / iters * 1e6:7.1f} us per iteration", flush=True)


if __name__ == "__main__":
    main()
