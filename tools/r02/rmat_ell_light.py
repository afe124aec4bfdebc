#!/usr/bin/env python3
"""RMAT s24: the light rows (<= 8 edges) through the packed neighbour table (one load in front of the gather instead
of indptr -> indices) against the row-group kernel; parts of the launch timed with knob spmm_parts."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import ops, _lib, workloads as W
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
n = 1 << scale
src, dst = W.rmat_edges(scale, 16, seed=0, device=dev)
ip, ix = ops.csr_from_coo(dst, src, n, n)
del src, dst
H = torch.rand(n, F, device=dev); out = torch.empty(n, F, device=dev)


def timed(fn, rounds=6):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rounds


if len(sys.argv) > 3:
    _lib.call("gae_tuning_set", b"spmm_ell_rpg", int(sys.argv[3]))
ref = None
for label, kw in (("row groups", dict(ell=False)), ("table, 8 slots", dict(ell=True, ell_width=8)),
                  ("table, 16 slots", dict(ell=True, ell_width=16)), ("table, 4 slots", dict(ell=True, ell_width=4))):
    plan = ops.spmm_plan(ip, indices=ix, n_cols=n, **kw)
    fn = lambda: ops.spmm_raw(ip, ix, H, n, out=out, plan=plan)
    _lib.call("gae_tuning_set", b"spmm_parts", 7)
    t_all = timed(fn)
    res = out.clone()
    if ref is None:
        ref = res
    _lib.call("gae_tuning_set", b"spmm_parts", 1)
    t_light = timed(fn)
    _lib.call("gae_tuning_set", b"spmm_parts", 7)
    print("%-16s whole launch %.3f ms, light rows alone %.3f ms, identical to row groups: %s"
          % (label, t_all, t_light, bool(torch.equal(res, ref))), flush=True)
    del plan
