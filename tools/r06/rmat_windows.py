#!/usr/bin/env python3
"""VERDICT r05 #4, measured before built: what would the R-MAT s24 product cost by COLUMN WINDOW?

The product M = A H gathers one 128-byte row of H per edge; H is 2 GiB, the Infinity Cache 256 MB.  If the edges are
processed one column window at a time (all edges whose column lies in a 2^w-row slice of H, the slice then sits in the
Infinity Cache), every gather that misses its L2 is served by the cache instead of DRAM.  Here the existing kernels and
plans run on the sub-graph of each window (rows = all 2^24, columns restricted to the window; accumulate into M from the
second window on: GAE_SPMM_ACCUMULATE, the form a windowed product would take), so the table holds the REAL cost of every
window including the read-modify-write of M, against the one-launch product.
  python tools/r06/rmat_windows.py [--scale 24] [--window-rows-log2 19,20,21]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from gae_dgl_amd import ops, workloads as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=24)
ap.add_argument("--window-rows-log2", default="19,20,21")
ap.add_argument("--F", type=int, default=32)
a = ap.parse_args()
dev = torch.device("cuda:0")
n, F = 1 << a.scale, a.F
src, dst = W.rmat_edges(a.scale, 16, seed=0, device=dev)
E = int(src.numel())
H = torch.rand(n, F, device=dev)
out = torch.empty(n, F, device=dev)
ip, ix = ops.csr_from_coo(dst, src, n, n)
plan = ops.spmm_plan(ip, indices=ix, ell=False, n_cols=n)
t_full = B.time_launches(lambda: ops.spmm_raw(ip, ix, H, n, out=out, plan=plan), iters=5, warmup=2)
ref = out.clone()
print(f"R-MAT s{a.scale}: {E} edges, F = {F}; one launch over all columns: {t_full * 1e3:.3f} ms = {t_full / E * 1e12:.2f} ps per edge")
del ip, ix, plan
torch.cuda.empty_cache()
for wl in (int(x) for x in a.window_rows_log2.split(",")):
    wrows = 1 << wl
    nwin = n // wrows
    print(f"\n-- windows of 2^{wl} rows of H = {wrows * F * 4 >> 20} MiB, {nwin} windows")
    tot, tot_edges = 0.0, 0
    win = src >> wl
    order = torch.argsort(win, stable=True)
    counts = torch.bincount(win, minlength=nwin).tolist()
    s_sorted, d_sorted = src[order], dst[order]
    del order, win
    pos = 0
    acc = torch.zeros(n, F, device=dev)
    rows = []
    for w in range(nwin):
        e = counts[w]
        s_w, d_w = s_sorted[pos:pos + e], d_sorted[pos:pos + e]
        pos += e
        if e == 0:
            continue
        ipw, ixw = ops.csr_from_coo(d_w, s_w, n, n)
        pw = ops.spmm_plan(ipw, indices=ixw, ell=False, n_cols=n)
        dead = (ipw[1:] == ipw[:-1]).to(torch.uint8).contiguous()      # rows without an edge in this window: not touched
        pw.set_skip_rows(dead, covers_all_empty=True)
        fn = lambda: ops.spmm_raw(ipw, ixw, H, n, out=acc, plan=pw, accumulate=True, skip_dead=True)
        t = B.time_launches(fn, iters=3, warmup=1)
        rows.append((w, e, t))
        tot += t; tot_edges += e
        del ipw, ixw, pw
    torch.cuda.empty_cache()
    for w, e, t in rows[:6] + rows[-2:]:
        print(f"   window {w:3d}: {e:10d} edges ({100.0 * e / E:5.2f} %)  {t * 1e3:7.3f} ms  {t / e * 1e12:7.2f} ps per edge")
    print(f"   SUM over {len(rows)} windows: {tot * 1e3:.3f} ms for {tot_edges} edges = {tot / tot_edges * 1e12:.2f} ps per edge "
          f"(one launch: {t_full * 1e3:.3f} ms); every window launch re-reads and re-writes the rows of M it touches")
    del s_sorted, d_sorted, acc
    torch.cuda.empty_cache()
