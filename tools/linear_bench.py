#!/usr/bin/env python3
"""A/B of the Linear forward/backward kernels on the Pubmed layer shapes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gae_dgl_amd import _lib, ops
dev = torch.device("cuda:0")
def t_(fn, iters=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
n = 19717
for (fin, fout) in ((500, 32), (32, 16)):
    M = torch.randn(n, fin, device=dev); W = torch.randn(fout, fin, device=dev) / fin ** 0.5; b = torch.randn(fout, device=dev)
    Y = ops.linear_fwd_raw(M, W, b, 1); dY = torch.randn(n, fout, device=dev)
    print(f"linear {fin}->{fout}: fwd {t_(lambda: ops.linear_fwd_raw(M, W, b, 1)):.1f} us")
    for rows in (64, 128, 256, 512):
        _lib.call("gae_tuning_set", b"atb_rows", rows)
        print(f"   atb_rows={rows}: bwd(dW,db) {t_(lambda: ops.linear_bwd_raw(dY, Y, 1, M, W, True, True, False)):.1f} us   "
              f"bwd(dW,db,dM) {t_(lambda: ops.linear_bwd_raw(dY, Y, 1, M, W, True, True, True)):.1f} us")
