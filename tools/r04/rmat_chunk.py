#!/usr/bin/env python3
"""[historical: ran on commit e8f2a2f with the torch plan builder; ops.HOMED_SEGMENT / HOMED_HOT_COLUMNS are gone]
Experiment (round 4): chunk length / row threshold of the XCD-pinned part of the RMAT skew plan.
Equal-count chunks of the (row, home) column lists, launched in the order of their first column, are aligned in
column space across rows (R-MAT's column marginal barely depends on the row): the shorter the chunks, the narrower
the column window the concurrently running waves of an XCD gather from.
  python tools/r04/rmat_chunk.py [scale]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import _lib, ops, workloads as W

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
src, dst = W.rmat_edges(scale, 16, device=dev)
n = 1 << scale
ip, ix = ops.csr_from_coo(dst, src, n, n)
del src, dst
H = torch.rand(n, 32, device=dev)
out = torch.empty(n, 32, device=dev)


def timeit(fn, iters=4):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ref = None
configs = [(256, 512, 1 << 20), (256, 256, 1 << 20), (256, 128, 1 << 20), (256, 64, 1 << 20), (128, 64, 1 << 20),
           (128, 128, 1 << 20), (64, 64, 1 << 20), (256, 128, 1 << 18), (256, 64, 1 << 18), (256, 128, 1 << 22)]
for t2, seg, hot in configs:
    ops.HOMED_MIN_DEGREE, ops.HOMED_SEGMENT, ops.HOMED_HOT_COLUMNS = t2, seg, hot
    torch.cuda.synchronize(); t0 = time.time()
    plan = ops.spmm_plan(ip, threshold=ops.SKEW_THRESHOLD, indices=ix, ell=False, n_cols=n)
    torch.cuda.synchronize(); tb = time.time() - t0
    res = {}
    for name, parts in (("all", 7), ("light", 1), ("mid", 2), ("pinned", 4)):
        _lib.call("gae_tuning_set", b"spmm_parts", parts)
        res[name] = timeit(lambda: ops.spmm_raw(ip, ix, H, n, out=out, plan=plan))
    _lib.call("gae_tuning_set", b"spmm_parts", 7)
    ops.spmm_raw(ip, ix, H, n, out=out, plan=plan)
    if ref is None:
        ref = out.clone()
    err = float((out - ref).abs().max() / ref.abs().max())
    nv = plan.homed["identity"].numel() if plan.homed else 0
    print(f"T2={t2:4d} seg={seg:4d} hot={hot:8d}: all {res['all']:.3f} ms  light {res['light']:.3f}  mid {res['mid']:.3f}  "
          f"pinned {res['pinned']:.3f}  virtual rows {nv}  plan build {tb:.2f} s  err vs first {err:.1e}", flush=True)
    del plan
    torch.cuda.empty_cache()
