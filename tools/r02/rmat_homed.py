#!/usr/bin/env python3
"""[historical: round-2 experiment; ops.hot_indices_for / the torch plan builder are gone -- the device builder of csrc/plan_build.hip writes the tags]
Experiment: does pinning the gathers of the very long rows of a power-law graph to the L2 that 'owns' the column
(home = column % 8 = XCD of the block that gathers it) raise the hit rate?  Uses only the existing C ABI: the edges
of rows with more than T2 in-edges are regrouped into VIRTUAL rows (row, home, chunk of <= SEG edges) and the plan
arrays are written so that segment s is handled by block s / 4, i.e. XCD (s / 4) % 8 == home of the segment.
Compares (A) the same edges as ordinary heavy rows with (B) the homed virtual rows.
  python tools/r02/rmat_homed.py [scale] [T2]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import _lib, ops, workloads as W

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T2 = int(sys.argv[2]) if len(sys.argv) > 2 else 256
SEG = 512
dev = torch.device("cuda:0")
src, dst = W.rmat_edges(scale, 16, device=dev)
n = 1 << scale
ip, ix = ops.csr_from_coo(dst, src, n, n)
del src, dst
deg = (ip[1:] - ip[:-1]).to(torch.int64)
vh = torch.nonzero(deg > T2).flatten()
print(f"rows with more than {T2} edges: {vh.numel()}, their edges: {int(deg[vh].sum())} of {ix.numel()}")
# ---- (A) sub-CSR of the very heavy rows, ordinary plan
R = vh.numel()
dA = deg[vh]
ipA = torch.zeros(R + 1, dtype=torch.int32, device=dev); ipA[1:] = torch.cumsum(dA, 0).to(torch.int32)
starts = ip[vh].to(torch.int64)
eidx = torch.repeat_interleave(starts - ipA[:-1].to(torch.int64), dA) + torch.arange(int(dA.sum()), device=dev)
ixA = ix[eidx].contiguous()
H = torch.rand(n, 32, device=dev)


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


planA = ops.spmm_plan(ipA, threshold=8, segment=SEG, indices=ixA, ell=False, hot=True, n_cols=n)
outA = torch.empty(R, 32, device=dev)
tA = timeit(lambda: ops.spmm_raw(ipA, ixA, H, R, out=outA, plan=planA))
print(f"(A) ordinary heavy rows: {tA:.3f} ms")
# ---- (B) virtual rows (row slot, home, chunk), segments interleaved by home in groups of 4
rowslot = torch.repeat_interleave(torch.arange(R, device=dev), dA)
home = ((ixA.to(torch.int64) * 2654435761) >> 13) & 7      # a hash: the low bits of RMAT's hub ids are all zero
key = (rowslot * 8 + home) * (1 << 25) + ixA.to(torch.int64)
order = torch.argsort(key)
ixB = ixA[order].contiguous()
grp = (rowslot * 8 + home)[order]                      # sorted
cnt = torch.bincount(grp, minlength=R * 8)             # edges per (row, home)
nchunk = (cnt + SEG - 1) // SEG
gstart = torch.cumsum(cnt, 0) - cnt
# virtual rows: one per (group, chunk)
vg = torch.repeat_interleave(torch.arange(R * 8, device=dev), nchunk)
first = torch.cumsum(nchunk, 0) - nchunk
vk = torch.arange(vg.numel(), device=dev) - first[vg]
v_e0 = gstart[vg] + vk * SEG
v_e1 = torch.minimum(v_e0 + SEG, gstart[vg] + cnt[vg])
v_home = vg % 8
# order the virtual rows so that position p has home (p // 4) % 8: take per-home lists, pad to equal length L
lists = [torch.nonzero(v_home == h).flatten() for h in range(8)]
L = max(int(l.numel()) for l in lists)
L = (L + 3) // 4 * 4
V = 8 * L
pos_of = torch.full((V,), -1, dtype=torch.int64, device=dev)
for h in range(8):
    q = torch.arange(lists[h].numel(), device=dev)
    p = (q // 4) * 32 + h * 4 + (q % 4)                # block = p // 4, block % 8 == h
    pos_of[p] = lists[h]
valid = pos_of >= 0
e0 = torch.zeros(V, dtype=torch.int64, device=dev); e1 = torch.zeros(V, dtype=torch.int64, device=dev)
e0[valid] = v_e0[pos_of[valid]]; e1[valid] = v_e1[pos_of[valid]]
# a CSR needs monotone indptr: virtual row p reads [e0, e1) of ixB -> build a gathered index array in position order
lens = (e1 - e0)
ipB = torch.zeros(V + 1, dtype=torch.int32, device=dev); ipB[1:] = torch.cumsum(lens, 0).to(torch.int32)
src_off = torch.repeat_interleave(e0 - ipB[:-1].to(torch.int64), lens) + torch.arange(int(lens.sum()), device=dev)
ixP = ixB[src_off].contiguous()
tags = ops.hot_indices_for(ixP, n, hot_columns=int(os.environ.get("HOT", 262144)))
hr = torch.arange(V, dtype=torch.int32, device=dev)    # slot p = virtual row p, one segment each, in position order
plan = ops.SpmmPlan(8, SEG, V, V, hr, hr.clone(), hr.clone(), None, None, tags)
outB = torch.empty(V, 32, device=dev)
wsb = _lib.load().gae_spmm_workspace_bytes
tB = timeit(lambda: ops.spmm_raw(ipB, ixP, H, V, out=outB, plan=plan))
print(f"(B) homed virtual rows ({V} virtual rows, {int(valid.sum())} real): {tB:.3f} ms")
# check: sums of virtual rows per real row == (A)
vrow_real = torch.full((V,), 0, dtype=torch.int64, device=dev)
vrow_real[valid] = (vg[pos_of[valid]] // 8)
chk = torch.zeros(R, 32, device=dev, dtype=torch.float64).index_add_(0, vrow_real[valid], outB[valid].double())
print("max rel diff vs (A):", float((chk - outA.double()).abs().max() / outA.abs().max()))
