#!/usr/bin/env python3
"""A/B of the Linear forward/backward kernels on the layer shapes of the BASELINE configs.  Launches are replayed
from one HIP graph (eager Python launches cost ~10 us of host time each and would hide the kernels)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gae_dgl_amd import _lib, ops
dev = torch.device("cuda:0")


def t_(fn, iters=50):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="0,128,256,512")
ap.add_argument("--knobs", default="")
ap.add_argument("--only", default="")
a = ap.parse_args()
for kv in filter(None, a.knobs.split(",")):
    k, v = kv.split("=")
    _lib.call("gae_tuning_set", k.encode(), int(v))
shapes = (("pubmed L1", 19717, 500, 32), ("pubmed L2", 19717, 32, 16), ("cora L1", 2708, 1433, 32),
          ("citeseer L1", 3327, 3703, 32), ("zinc4096 L1", 94752, 39, 32), ("zinc4096 L2", 94752, 32, 16))
for (name, n, fin, fout) in shapes:
    if a.only and a.only not in name:
        continue
    M = ops.pad_rows(torch.randn(n, fin, device=dev)); W = torch.randn(fout, fin, device=dev) / fin ** 0.5
    b = torch.randn(fout, device=dev)
    Y = ops.linear_fwd_raw(M, W, b, 1); dY = torch.randn(n, fout, device=dev)
    mb = n * fin * 4 / 1e6
    tf = t_(lambda: ops.linear_fwd_raw(M, W, b, 1))
    print(f"{name}: {n} x {fin} -> {fout}  (M = {mb:.1f} MB)   fwd {tf:.1f} us ({mb / tf:.2f} TB/s of M)")
    for rows in (int(r) for r in a.rows.split(",")):
        _lib.call("gae_tuning_set", b"atb_rows", rows)
        t1 = t_(lambda: ops.linear_bwd_raw(dY, Y, 1, M, W, True, True, False))
        t2 = t_(lambda: ops.linear_bwd_raw(dY, Y, 1, M, W, True, True, True))
        print(f"   atb_rows={rows}: bwd(dW,db) {t1:.1f} us ({mb / t1:.2f} TB/s)   bwd(dW,db,dM) {t2:.1f} us")
    _lib.call("gae_tuning_set", b"atb_rows", 0)
