"""Repeat every kernel thousands of times on the headline matrix and check the results never change (tools)."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, cusp_autotuned_amd as cmi
A = cmi.poisson5pt(3162, 3162, "csr")
n = A.num_rows
x = cmi.fill_x(n).cuda()
y0 = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty_like(y0)
ws = cmi.blas_workspace()
res = torch.zeros(1, dtype=torch.float64, device="cuda")
cmi.multiply(A, x, y0)
cmi.spmv_csr_dot(n, n, A.row_offsets, A.column_indices, A.values, x, y, x, res, ws)
d0 = float(res)
t0 = time.time()
bad = 0
for it in range(200):
    for _ in range(100):
        cmi.multiply(A, x, y)
    cmi.spmv_csr_dot(n, n, A.row_offsets, A.column_indices, A.values, x, y, x, res, ws)
    if not torch.equal(y, y0) or float(res) != d0:
        bad += 1
print("20000 SpMV + 200 fused dots in %.1f s, mismatching checks: %d" % (time.time() - t0, bad))
# balanced kernel + ELL/DIA/COO loops
for fmt in ("ell", "dia", "coo", "hyb"):
    M = cmi.poisson5pt(3162, 3162, "dia") if fmt == "dia" else cmi.convert(A, fmt, num_entries_per_row=5 if fmt == "hyb" else None)
    cmi.multiply(M, x, y0)
    ok = True
    for _ in range(2000):
        cmi.multiply(M, x, y)
    ok = torch.allclose(y, y0, rtol=1e-12, atol=1e-12)
    print(fmt, "2000 multiplies consistent:", bool(ok))
bal = cmi.Config(kernel=cmi.CSR_BALANCED)
cmi.multiply(A, x, y0, cfg=bal)
for _ in range(2000):
    cmi.multiply(A, x, y, cfg=bal)
print("balanced 2000 multiplies consistent:", bool(torch.allclose(y, y0, rtol=1e-12, atol=1e-12)))
