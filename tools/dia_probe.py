"""DIA on the headline matrix: every launch shape (block x rows per lane x cache policy), interleaved rounds."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
import autotune as at  # noqa: E402

for dt in (torch.float64, torch.float32):
    D = cmi.poisson5pt(3162, 3162, "dia", dtype=dt)
    A = cmi.poisson5pt(3162, 3162, "csr", dtype=dt)
    x = cmi.fill_x(D.num_rows, dt, "cuda")
    y = torch.empty(D.num_rows, dtype=dt, device="cuda")
    ref = torch.empty_like(y)
    cmi.multiply(A, x, ref, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
    timer = at.Timer(cmi, torch)
    cfgs = [cmi.Config(kernel=cmi.DIA_ROW, block_size=b, items_per_thread=r, nontemporal=nt) for b in (64, 128, 256, 512, 1024) for r in (1, 2) for nt in (0, 1, 2, 3)]
    for c in cfgs:
        y.fill_(7.0)
        cmi.multiply(D, x, y, cfg=c)
        assert torch.equal(y, ref), c
    t = {id(c): [] for c in cfgs}
    for _ in range(7):
        for c in cfgs:
            t[id(c)].append(timer.time(lambda: cmi.multiply(D, x, y, cfg=c), 30))
    rows = sorted((float(np.median(t[id(c)])), c) for c in cfgs)
    alg = cmi.dia_bytes(D.num_rows, 5, D.pitch, 8 if dt == torch.float64 else 4)
    print(dt, "table:", cmi.tuning_select(cmi.FORMAT_DIA, cmi.F64 if dt == torch.float64 else cmi.F32, D.num_rows, D.num_rows, 5 * D.num_rows).as_dict())
    for med, c in rows[:6] + rows[-2:]:
        print(f"  {med * 1e3:7.1f} us  {alg / med / 1e9:5.2f} TB/s  block {c.block_size} rows/lane {c.items_per_thread} nt {c.nontemporal}", flush=True)
