"""does the ISSUE ORDER of the mid rows (9..256 edges, one segment each) and of the light rows (1..8) of
the R-MAT s24 product matter?  Idea: order them by the 64-MiB column block of their median neighbour so that the waves
running together gather from one Infinity-Cache-sized window of H.  The descriptors are permuted in place (every mid
row has ONE segment whose sum goes straight to M, every light-list entry writes its own row: any order is correct --
checked against the default order's output).  Whole product, graph-replay timing, F = 32 and 16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gae_dgl_amd import ops, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << scale
src, dst = W.rmat_edges(scale, 16, seed=0, device=dev)
ip, ix = ops.csr_from_coo(dst, src, n, n)
del src, dst
plan = ops.spmm_plan(ip, indices=ix, ell=False, n_cols=n)
assert plan.seg_desc is not None and plan.light_desc is not None and plan.mid_ids is not None
sd0, ld0 = plan.seg_desc.clone(), plan.light_desc.clone()
mid = plan.mid_ids
MASK = 0x7fffffff
g = torch.Generator(device=dev).manual_seed(0)


def keys(desc, ids):
    e0, e1 = desc[:, 1].long(), desc[:, 2].long()
    med = (ids[(e0 + e1) // 2].long() & MASK)
    first = (ids[e0].long() & MASK)
    return med, first, (e1 - e0)


def orders(desc, ids):
    med, first, deg = keys(desc, ids)
    m = desc.shape[0]
    yield "ascending row (default)", torch.arange(m, device=dev)
    yield "random", torch.randperm(m, device=dev, generator=g)
    yield "median neighbour's 64-MiB block (stable)", torch.sort(med >> 19, stable=True).indices
    yield "median neighbour's 8-MiB block (stable)", torch.sort(med >> 16, stable=True).indices
    yield "median neighbour", torch.sort(med, stable=True).indices
    yield "first neighbour", torch.sort(first, stable=True).indices
    yield "longest first", torch.sort(deg, descending=True, stable=True).indices


for F in (32, 16):
    H = torch.rand(n, F, device=dev, generator=g)
    out = torch.empty(n, F, device=dev)
    ref = ops.spmm_raw(ip, ix, H, n, plan=plan).clone()
    print(f"== R-MAT s{scale}, F = {F}: whole product (light + mid + pinned + combine), us per launch", flush=True)
    for part, d0, live, ids in (("mid rows", sd0, plan.seg_desc, mid), ("light rows", ld0, plan.light_desc, ix)):
        for name, perm in orders(d0, ids):
            live.copy_(d0[perm])
            t = bench.time_launches(lambda: ops.spmm_raw(ip, ix, H, n, out=out, plan=plan), iters=10, warmup=3)
            ok = torch.equal(out, ref)
            print(f"  {part:10s} in order of {name:44s} {t * 1e6:9.1f} us   {'same bits' if ok else 'DIFFERENT'}", flush=True)
        live.copy_(d0)
