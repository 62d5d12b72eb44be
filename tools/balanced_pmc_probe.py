"""csr_stream vs csr_balanced on the headline matrix, for rocprofv3 --kernel-trace / --pmc passes (tools).
Launch order: 3 x [stream, balanced (zero fill + kernel)], then 3 x balanced with accumulate (no fill)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402

A = cmi.poisson5pt(3162, 3162, "csr")
x = cmi.fill_x(A.num_rows).cuda()
y = torch.empty(A.num_rows, dtype=torch.float64, device="cuda")
bal = cmi.Config(kernel=cmi.CSR_BALANCED, items_per_thread=8)
for _ in range(3):
    cmi.multiply(A, x, y)
    cmi.multiply(A, x, y, cfg=bal)
torch.cuda.synchronize()
for _ in range(3):
    cmi.multiply(A, x, y, accumulate=True, cfg=bal)
torch.cuda.synchronize()
