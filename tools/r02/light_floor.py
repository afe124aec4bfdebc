#!/usr/bin/env python3
"""Floor of the light-row pass of the RMAT product: the row-group kernel on an EMPTY graph of 2^24 rows (stores only),
on the RMAT graph restricted to its rows of <= 8 edges, next to a fill and a copy of the same output bytes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import ops, _lib, workloads as W
dev = torch.device("cuda:0")
scale, F = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 1 << scale
H = torch.rand(n, F, device=dev); out = torch.empty(n, F, device=dev)


def timed(fn, rounds=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rounds


print("fill   %.3f ms" % timed(lambda: out.zero_()))
print("copy   %.3f ms" % timed(lambda: out.copy_(H)))
ip0 = torch.zeros(n + 1, dtype=torch.int32, device=dev); ix0 = torch.zeros(1, dtype=torch.int32, device=dev)[:0]
for rpg in (1, 2, 4):
    _lib.call("gae_tuning_set", b"spmm_rpg", rpg)
    print("empty graph, rpg %d   %.3f ms" % (rpg, timed(lambda: ops.spmm_raw(ip0, ix0, H, n, out=out))))
src, dst = W.rmat_edges(scale, 16, seed=0, device=dev)
deg = torch.bincount(dst, minlength=n)
keep = deg[dst] <= 8
ipl, ixl = ops.csr_from_coo(dst[keep], src[keep], n, n)
print("light rows: %d edges, %d rows with edges" % (int(keep.sum()), int(((deg > 0) & (deg <= 8)).sum())))
del src, dst
for rpg in (1, 2, 4):
    _lib.call("gae_tuning_set", b"spmm_rpg", rpg)
    print("light-row graph, rpg %d   %.3f ms" % (rpg, timed(lambda: ops.spmm_raw(ipl, ixl, H, n, out=out))))
for nt in (0, 1, 2):
    _lib.call("gae_tuning_set", b"spmm_rpg", 2)
    _lib.call("gae_tuning_set", b"spmm_nt", nt)
    print("light-row graph, store policy %d   %.3f ms" % (nt, timed(lambda: ops.spmm_raw(ipl, ixl, H, n, out=out))))
