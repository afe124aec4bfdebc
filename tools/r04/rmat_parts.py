#!/usr/bin/env python3
"""RMAT s24 product by parts (round 4): light rows through the plan's list + fill stream vs the all-row sweep, F = 32 / 16.
  python tools/r04/rmat_parts.py [scale]"""
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
torch.cuda.synchronize(); t0 = time.time()
plan = ops.spmm_plan(ip, threshold=ops.SKEW_THRESHOLD, indices=ix, ell=False, n_cols=n)
torch.cuda.synchronize()
print(f"plan build {time.time() - t0:.2f} s, light rows {plan.n_light}, heavy {plan.n_heavy}, bytes "
      f"{sum(t.numel() * t.element_size() for t in plan.tensors if t is not None) / 1e9:.2f} GB")


def timeit(fn, iters=4):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for F in (32, 16):
    H = torch.rand(n, F, device=dev)
    out = torch.empty(n, F, device=dev)
    ref = None
    for light in (0, 1):
        _lib.call("gae_tuning_set", b"spmm_light", light)
        res = {}
        for name, parts in (("all", 7), ("light", 1), ("mid", 2), ("pinned", 4)):
            _lib.call("gae_tuning_set", b"spmm_parts", parts)
            res[name] = timeit(lambda: ops.spmm_raw(ip, ix, H, n, out=out, plan=plan))
        _lib.call("gae_tuning_set", b"spmm_parts", 7)
        ops.spmm_raw(ip, ix, H, n, out=out, plan=plan)
        same = True if ref is None else bool(torch.equal(out, ref))
        ref = out.clone()
        print(f"F={F} spmm_light={light}: all {res['all']:.3f} ms  light {res['light']:.3f}  mid {res['mid']:.3f}  "
              f"pinned {res['pinned']:.3f}  identical to previous: {same}", flush=True)
_lib.call("gae_tuning_set", b"spmm_light", 1)
# cache hints of the heavy-row kernels: hot-column tags (streaming loads for the other columns) vs plain loads
H = torch.rand(n, 32, device=dev); out = torch.empty(n, 32, device=dev)
for hot in (1, 0):
    _lib.call("gae_tuning_set", b"spmm_hot", hot)
    res = {}
    for name, parts in (("all", 7), ("mid", 2), ("pinned", 4)):
        _lib.call("gae_tuning_set", b"spmm_parts", parts)
        res[name] = timeit(lambda: ops.spmm_raw(ip, ix, H, n, out=out, plan=plan))
    _lib.call("gae_tuning_set", b"spmm_parts", 7)
    print(f"F=32 spmm_hot={hot}: all {res['all']:.3f} ms  mid {res['mid']:.3f}  pinned {res['pinned']:.3f}", flush=True)
_lib.call("gae_tuning_set", b"spmm_hot", 1)
