#!/usr/bin/env python3
"""Run one SpMM shape a few times, exactly as the package launches it (target of rocprofv3 --pmc / --kernel-trace)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gae_dgl_amd import _lib, ops, workloads as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="pubmed500", help="pubmedF | coraF | citeseerF | zincF (whole set) | zincbF (batch) | rmatF")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--knobs", default="")
ap.add_argument("--rmat-scale", type=int, default=22)
ap.add_argument("--plain", action="store_true", help="no plan, no feature tiles, no block-diagonal kernel")
a = ap.parse_args()
dev = torch.device("cuda:0")
for kv in filter(None, a.knobs.split(",")):
    k, v = kv.split("=")
    _lib.call("gae_tuning_set", k.encode(), int(v))
bd = None
name = a.shape.rstrip("0123456789")
F = int(a.shape[len(name):])
if name in ("pubmed", "cora", "citeseer"):
    n, src, dst, _ = W.citation_graph(name)
    s, d = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
elif name in ("zinc", "zincb"):              # whole set / the first 4096 molecules (one training batch)
    gp, src, dst, _ = W.zinc_like(249455)
    if name == "zincb":
        gp = gp[:4097]
        keep = dst < gp[-1]
        src, dst = src[keep], dst[keep]
    n = int(gp[-1])
    s, d = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
    bd = None if a.plain else ops.BlockDiag(gp, dev)
else:
    # (few large chunks: rocprofv3 --pmc segfaults on the ~10^4 generator dispatches of the default chunking; the
    #  graph has the same parameters, not the same edges, as the bench's)
    s, d = W.rmat_edges(a.rmat_scale, 16, device=dev, chunk=1 << 26)
    n = 1 << a.rmat_scale
ip, ix = ops.csr_from_coo(d, s, n, n)
H = ops.pad_rows(torch.rand(n, F, device=dev))
out = ops.pad_rows(torch.empty(n, F, device=dev))
if a.plain:
    plan = None
elif name == "zincb":      # a training batch carries the packed table its gather wrote (dataset.DeviceGraphDataset)
    plan = ops.spmm_plan(ip, indices=ix, ell=True, ell_width=ops.ell_width_for_degrees(ip[1:] - ip[:-1]))
else:
    plan = ops.spmm_plan(ip, indices=ix if bd is None else None, n_cols=n)
scattered = (not a.plain) and bd is None and F > ops.TILE_MIN_F and ops.gather_scattered(ip, ix, F * 4)
for _ in range(a.iters):
    ops.spmm_raw(ip, ix, H, n, out=out, plan=plan, blockdiag=bd, out_padded=True, scattered=scattered)
torch.cuda.synchronize()
print("done", a.shape, n, int(ix.numel()), F, "ld", H.stride(0), "scattered", scattered, "plan", plan is not None,
      "blockdiag", bd is not None)
