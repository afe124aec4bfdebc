#!/usr/bin/env python3
"""How fast can 39.4 MB be written?  torch fill / zero / copy vs the SpMM kernels on an edge-less graph
(stores only) with each store policy -- the floor under every SpMM launch that materialises M."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmm_bench import time_once, knob

dev = torch.device("cuda:0")
n, F, ld = 19717, 500, 512
M = torch.empty(n, ld, device=dev); H = torch.rand(n, ld, device=dev)
M2 = torch.empty(n * ld, device=dev)
for name, fn, b in (("zero_", lambda: M2.zero_(), 4 * n * ld), ("fill_", lambda: M2.fill_(1.5), 4 * n * ld),
                    ("copy_ (R+W)", lambda: M.copy_(H), 8 * n * ld), ("mul_ in place (R+W same)", lambda: M.mul_(1.0001), 8 * n * ld),
                    ("sum (R only)", lambda: H.sum(), 4 * n * ld)):
    t = min(time_once(fn, 50) for _ in range(3))
    print(f"{name:28s} {t*1e6:7.2f} us  {b/t/1e12:5.2f} TB/s")
ip = torch.zeros(n + 1, dtype=torch.int32, device=dev); ix = torch.zeros(0, dtype=torch.int32, device=dev)
plan = ops.spmm_plan(ip, indices=ix, ell=True, ell_width=16)
out = M[:, :F]
for nt in (0, 1, 2):
    for tv, sc in ((0, False), (16, True), (8, True)):
        for use_plan in (False, True):
            knob("spmm_nt", nt); knob("spmm_tile_vecs", tv)
            fn = lambda: ops.spmm_raw(ip, ix, H[:, :F], n, out=out, plan=plan if use_plan else None, out_padded=True, scattered=sc)
            t = min(time_once(fn, 50) for _ in range(3))
            print(f"spmm no-edge store={nt} tile_vecs={tv:2d} table={use_plan!s:5s} {t*1e6:7.2f} us  {4*n*ld/t/1e12:5.2f} TB/s (writes only)")
