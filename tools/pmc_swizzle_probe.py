#!/usr/bin/env python3
"""FETCH_SIZE of csr_stream on the headline matrix as a function of how tiles are dealt to the XCDs
(xcd_swizzle = 0: launch order, C >= 2: chunks of C tiles per XCD, 1: one contiguous eighth per XCD).
Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv`; 10 launches per setting, in
the order printed."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402

M = 3162
A = cmi.poisson5pt(M, M, "csr")
x = cmi.fill_x(M * M, device="cuda")
y = torch.empty(M * M, dtype=torch.float64, device="cuda")
order = [0, 8, 16, 32, 64, 128, 1]
for swz in order:
    cfg = cmi.Config(kernel=cmi.CSR_STREAM, block_size=256, rows_per_block=192, items_per_thread=1, nontemporal=2, xcd_swizzle=swz)
    for _ in range(10):
        cmi.multiply(A, x, y, cfg=cfg)
    torch.cuda.synchronize()
print(json.dumps({"order": order, "launches_each": 10, "compulsory_read_bytes": cmi.csr_bytes(M * M, A.num_entries) - 8 * M * M}))
