"""Irregular row-length distributions through the table-selected CSR kernel and the explicit variants
(tools, not the product): uniform short rows + a few very long rows, and a power-law tail."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(torch, lens, ncols, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Ap = torch.zeros(lens.numel() + 1, dtype=torch.int64, device="cuda")
    Ap[1:] = torch.cumsum(lens, 0)
    nnz = int(Ap[-1])
    Aj = torch.randint(0, ncols, (nnz,), generator=g, device="cuda", dtype=torch.int32)
    Ax = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64)
    return Ap.to(torch.int32), Aj, Ax, nnz


def sweep():
    """Where does the merge-path kernel overtake the row-tile kernel?  2M rows of 3..11 entries with
    columns near the diagonal (the row-tile kernel at its best) + 64 rows of L entries."""
    import torch
    import cusp_autotuned_amd as cmi
    n = 2_000_000
    g = torch.Generator(device="cuda").manual_seed(3)
    for L in (0, 256, 1024, 4096, 16384, 65536, 262144, 1048576):
        lens = torch.randint(3, 12, (n,), generator=g, device="cuda")
        if L:
            lens[torch.randint(0, n, (64,), generator=g, device="cuda")] = L
        Ap = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        Ap[1:] = torch.cumsum(lens, 0)
        nnz = int(Ap[-1])
        rows = torch.repeat_interleave(torch.arange(n, device="cuda"), lens)
        Aj = ((rows + torch.randint(-2000, 2000, (nnz,), generator=g, device="cuda")) % n).to(torch.int32)
        del rows
        Ax = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64)
        Ap = Ap.to(torch.int32)
        x = torch.rand(n, device="cuda", dtype=torch.float64)
        y = torch.empty(n, device="cuda", dtype=torch.float64)
        out = []
        # 'table' = cfg NULL (table + row-length profile); 'stream' = the table's row-tile config forced (long rows
        # streamed by their whole workgroup); 'strict' = the same with threads_per_row = 1 (storage order for every row)
        tcfg = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, n, n, nnz)
        strict = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, n, n, nnz)
        strict.threads_per_row = 1
        variants = [("table", None), ("stream", tcfg), ("strict", strict)] + \
            [(f"bal/{per}", cmi.Config(kernel=cmi.CSR_BALANCED, items_per_thread=per)) for per in (4, 8)] + \
            [("bal/persistent8", cmi.Config(kernel=cmi.CSR_BALANCED, blocks_per_cu=8))]
        for vn, cfg in variants:
            reps = 3 if (vn == "strict" and L > 65536) else 10
            for _ in range(2):
                cmi.spmv_csr(n, n, Ap, Aj, Ax, x, y, cfg=cfg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                cmi.spmv_csr(n, n, Ap, Aj, Ax, x, y, cfg=cfg)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / reps * 1e6)
        print(f"64 rows of {L:7d}: nnz {nnz:9d}  " + "  ".join(f"{v[0]} {o:7.1f}" for v, o in zip(variants, out)), flush=True)
    A = cmi.poisson5pt(3162, 3162, "csr")
    x = cmi.fill_x(A.num_rows).cuda()
    y = torch.empty(A.num_rows, dtype=torch.float64, device="cuda")
    variants = [v for v in variants if v[0] not in ("stream", "strict")]  # those configs were shaped for the last matrix
    for vn, cfg in variants:
        for _ in range(3):
            cmi.multiply(A, x, y, cfg=cfg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            cmi.multiply(A, x, y, cfg=cfg)
        torch.cuda.synchronize()
        print(f"poisson5pt 3162^2 {vn}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us", flush=True)


def main():
    import torch
    import cusp_autotuned_amd as cmi
    if "--sweep" in sys.argv:
        return sweep()
    n = 2_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    cases = {}
    base = torch.randint(3, 12, (n,), generator=g, device="cuda")
    cases["uniform 3..11"] = base.clone()
    a = base.clone(); a[torch.randint(0, n, (8,), generator=g, device="cuda")] = 1_000_000
    cases["+ 8 rows of 1e6"] = a
    b = base.clone(); b[torch.randint(0, n, (2000,), generator=g, device="cuda")] = 20_000
    cases["+ 2000 rows of 2e4"] = b
    u = torch.rand(n, generator=g, device="cuda")
    cases["power law (alpha 1.8, max 2e5)"] = torch.clamp((3.0 * u.pow(-1 / 0.8)).long(), max=200_000)
    for name, lens in cases.items():
        Ap, Aj, Ax, nnz = build(torch, lens, n, 7)
        x = torch.rand(n, device="cuda", dtype=torch.float64)
        y = torch.empty(n, device="cuda", dtype=torch.float64)
        ref = torch.empty_like(y)
        cmi.spmv_csr(n, n, Ap, Aj, Ax, x, ref, cfg=cmi.Config(kernel=cmi.CSR_SCALAR, block_size=256))
        bytes_ = 12 * nnz + 20 * n
        print(f"{name}: nnz {nnz}, mean {nnz / n:.1f}, max {int(lens.max())}; {bytes_ / 1e6:.0f} MB = {bytes_ / 6e12 * 1e6:.0f} us at 6 TB/s")
        variants = [("table", None), ("scalar", cmi.Config(kernel=cmi.CSR_SCALAR, block_size=256)),
                    ("vector8", cmi.Config(kernel=cmi.CSR_VECTOR, block_size=256, threads_per_row=8)),
                    ("vector64", cmi.Config(kernel=cmi.CSR_VECTOR, block_size=256, threads_per_row=64))]
        if hasattr(cmi, "CSR_BALANCED"):
            variants.append(("balanced", cmi.Config(kernel=cmi.CSR_BALANCED, block_size=256)))
        for vn, cfg in variants:
            for _ in range(2):
                cmi.spmv_csr(n, n, Ap, Aj, Ax, x, y, cfg=cfg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                cmi.spmv_csr(n, n, Ap, Aj, Ax, x, y, cfg=cfg)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / reps * 1e6
            err = float(((y - ref).abs() / (ref.abs() + 1e-300)).max())
            print(f"    {vn:10s} {us:10.1f} us   {bytes_ / us / 1e6:7.2f} TB/s   max rel diff vs scalar {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
