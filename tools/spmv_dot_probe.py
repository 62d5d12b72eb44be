"""Where do the ~17 us between the plain SpMV (130 us) and the SpMV+<Ap,p> kernel inside CG (147 us) go?  Runs, back to
back on the headline matrix: the plain kernel, the fused kernel with w == x (CG's call), the fused kernel with another w,
and the fused kernel preceded by a 720 MB vector sweep (what CG's update/direction passes do to the caches).
Run under  rocprofv3 --kernel-trace --stats  to read the kernels' own durations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402

A = cmi.poisson5pt(3162, 3162, "csr")
N = A.num_rows
x = cmi.fill_x(N, device="cuda")
w = torch.rand(N, dtype=torch.float64, device="cuda")
y = torch.empty(N, dtype=torch.float64, device="cuda")
res = torch.zeros(1, dtype=torch.float64, device="cuda")
ws = cmi.blas_workspace()
a, b, c = (torch.rand(N, dtype=torch.float64, device="cuda") for _ in range(3))
args = (N, N, A.row_offsets, A.column_indices, A.values)


def timed(label, fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:58s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us per call", flush=True)


timed("plain SpMV", lambda: cmi.spmv_csr(*args, x, y))
timed("SpMV + <y, x> (w == x) + fold", lambda: cmi.spmv_csr_dot(*args, x, y, x, res, ws))
timed("SpMV + <y, w> (w != x) + fold", lambda: cmi.spmv_csr_dot(*args, x, y, w, res, ws))


def with_sweep():
    cmi.blas_axpby(1.0, a, 2.0, b, c)   # 240 MB
    cmi.blas_axpby(1.0, c, 2.0, a, b)   # 240 MB
    cmi.blas_axpby(1.0, b, 2.0, c, a)   # 240 MB
    cmi.spmv_csr_dot(*args, x, y, x, res, ws)


timed("3 vector sweeps (720 MB) + SpMV + <y, x> + fold", with_sweep)
timed("3 vector sweeps alone", lambda: (cmi.blas_axpby(1.0, a, 2.0, b, c), cmi.blas_axpby(1.0, c, 2.0, a, b), cmi.blas_axpby(1.0, b, 2.0, c, a)))
