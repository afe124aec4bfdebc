#!/usr/bin/env python3
"""Which plan serves graphs whose rows mostly outgrow the packed table?  F = 32 aggregation, kernel-only, on random
graphs of n rows with average in-degree k: table-only plan (long rows taken by the whole wave), skew plan (rows beyond 8
edges through the segment kernels) + table, no plan (row-group kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import gae_dgl_amd as G
from gae_dgl_amd import ops

dev = "cuda:0"


def timed(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


rng = np.random.default_rng(0)
for n in (3000, 20000, 100000):
    for k in (4, 12, 24, 48, 96):
        e = n * k
        src = rng.integers(0, n, e); dst = rng.integers(0, n, e)
        ip, ix = ops.csr_from_coo(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, n)
        H = ops.pad_rows(torch.randn(n, 32, device=dev))
        deg = (ip[1:] - ip[:-1])
        plans = {"table-only": ops.spmm_plan(ip, indices=ix, threshold=10 ** 6, ell=True),
                 "skew+table": ops.spmm_plan(ip, indices=ix, threshold=8),
                 "none": None}
        row = f"n {n:6d} avg degree {k:3d} (rows > 16: {100.0 * float((deg > 16).float().mean()):5.1f} %, longest {int(deg.max()):4d}):"
        for name, pl in plans.items():
            t = timed(lambda: ops.spmm_raw(ip, ix, H, n, plan=pl))
            row += f"  {name} {t:8.1f} us"
        print(row, flush=True)
