#!/usr/bin/env python3
"""Run one SpMM shape a few times (target of rocprofv3 --pmc / --kernel-trace passes)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gae_dgl_amd import _lib, ops, workloads as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="pubmed500")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--knobs", default="")
ap.add_argument("--rmat-scale", type=int, default=22)
a = ap.parse_args()
dev = torch.device("cuda:0")
for kv in filter(None, a.knobs.split(",")):
    k, v = kv.split("=")
    _lib.call("gae_tuning_set", k.encode(), int(v))
if a.shape.startswith("pubmed"):
    n, src, dst, _ = W.citation_graph("pubmed")
    s, d = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
    al = a.shape.endswith("a")
    F = int(a.shape[6:].rstrip("a")); ld = (F + 31) // 32 * 32 if al else F
elif a.shape.startswith("zinc"):
    gp, src, dst, _ = W.zinc_like(249455)
    n = int(gp[-1])
    s, d = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
    F = int(a.shape[4:]); ld = (F + 3) // 4 * 4
else:
    s, d = W.rmat_edges(a.rmat_scale, 16, device=dev)
    n = 1 << a.rmat_scale
    F = int(a.shape[4:]); ld = F
ip, ix = ops.csr_from_coo(d, s, n, n)
H = torch.rand(n, ld, device=dev)[:, :F]
out = torch.empty(n, ld, device=dev)[:, :F]
for _ in range(a.iters):
    ops.spmm_raw(ip, ix, H, n, out=out)
torch.cuda.synchronize()
print("done", a.shape, n, int(ix.numel()), F)
