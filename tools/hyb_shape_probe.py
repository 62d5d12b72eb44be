"""Cache policy x XCD dealing of the one-launch HYB kernel (hyb_tile_kernel) on the headline matrix split at K = 4 and 3:
checks that the ELL table key the kernel inherits (nontemporal, xcd_swizzle) is the right one.  python tools/hyb_shape_probe.py"""
import itertools, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cusp_autotuned_amd as cmi
A = cmi.poisson5pt(3162, 3162, "csr"); N = A.num_rows
x = cmi.fill_x(N, device="cuda"); y = torch.empty(N, dtype=torch.float64, device="cuda")
def t_us(fn, iters=50, rounds=3):
    for _ in range(5): fn()
    out=[]
    for _ in range(rounds):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize(); out.append(e0.elapsed_time(e1)*1e3/iters)
    return sorted(out)[len(out)//2]
for K in (4, 3):
    H = cmi.convert(A, "hyb", num_entries_per_row=K); e, c = H.ell, H.coo
    print("K", K, "default plan:", round(t_us(lambda: cmi.multiply(H, x, y)), 1))
    for nt, swz in itertools.product((0, 1, 2, 3), (0, 16, 32, 64, 128)):
        pl = cmi.Plan.hyb(torch.float64, N, N, K, c.row_indices, cfg_ell=cmi.Config(kernel=cmi.ELL_ROW, threads_per_row=1, nontemporal=nt, xcd_swizzle=swz))
        t = t_us(lambda: cmi.spmv_hyb_plan(pl, e.pitch, e.column_indices, e.values, c.row_indices, c.column_indices, c.values, x, y))
        print(f"  nt {nt} swizzle {swz}: {t:.1f} us", flush=True)
