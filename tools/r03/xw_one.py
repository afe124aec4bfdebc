#!/usr/bin/env python3
"""Run one layer-1 dense pass a few times exactly as the package launches it (target of rocprofv3 --pmc):
  python tools/r03/xw_one.py --shape pubmed|cora|citeseer [--dtype bf16] [--op fwd|wgrad] [--iters N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import ops, workloads as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="pubmed")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--op", default="fwd")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
n, src, dst, X = W.citation_graph(a.shape)
K, J = X.shape[1], 32
Xd = torch.from_numpy(X).to(dev)
if a.dtype == "bf16":
    Xd = Xd.to(torch.bfloat16)
Xd = ops.pad_rows(Xd)
Wt = torch.randn(J, K, device=dev) / K ** 0.5
G = torch.randn(n, J, device=dev)
for _ in range(a.iters):
    if a.op == "fwd":
        ops.xw_fwd_raw(Xd, Wt, None, 0)
    else:
        ops.xw_wgrad_raw(Xd, G, None, G, G, J)
torch.cuda.synchronize()
print("done", a.shape, n, K, J, a.dtype, a.op, "ld", Xd.stride(0))
