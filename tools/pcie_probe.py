"""Host -> HBM transfer time of the headline matrix (CSR arrays + x) from pageable and from page-locked host
memory: what a caller who hands over HOST buffers would pay before the first SpMV (tools)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

N, NNZ = 9_998_244, 49_978_572
host = {"Ap": torch.zeros(N + 1, dtype=torch.int32), "Aj": torch.zeros(NNZ, dtype=torch.int32),
        "Ax": torch.zeros(NNZ, dtype=torch.float64), "x": torch.zeros(N, dtype=torch.float64)}
total = sum(t.numel() * t.element_size() for t in host.values())
for label, src in (("pageable", host), ("page-locked", {k: v.pin_memory() for k, v in host.items()})):
    dst = {k: torch.empty_like(v, device="cuda") for k, v in src.items()}
    for _ in range(2):
        for k in src:
            dst[k].copy_(src[k], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        for k in src:
            dst[k].copy_(src[k], non_blocking=True)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:12s}: {total / 1e6:.0f} MB in {dt * 1e3:.1f} ms = {total / dt / 1e9:.1f} GB/s  (= {dt / 134e-6:.0f} SpMVs of 134 us)")
