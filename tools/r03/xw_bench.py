"""layer-1 chains, kernel by kernel (graph-replay timing, bench.time_launches):
reference order  (A X) W^T : spmm F_in | linear fwd | (dW, db)          -- old dense kernels (knob xw=0) and new
transform first  A (X W^T) : xw_fwd | spmm F_out (+ b, act) | gated spmm F_out | xw_wgrad
  python tools/r03/xw_bench.py [pubmed cora citeseer]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gae_dgl_amd as G  # noqa: E402
from gae_dgl_amd import _lib, ops, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
names = sys.argv[1:] or ["pubmed", "cora", "citeseer"]
for name in names:
    n, src, dst, X = W.citation_graph(name, seed=0)
    K, J = X.shape[1], 32
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    ip, ix = g.csr(); tp, tx = g.csc()
    Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
    Wt = torch.randn(J, K, device=dev) / K ** 0.5
    b = torch.randn(J, device=dev)
    P, n_split = ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True)      # as the layer launches it: split partials kept
    Y = ops.spmm_epilogue_raw(ip, ix, P, n, g.spmm_plan(False), b, 1)
    dY = torch.randn(n, J, device=dev)
    Gt = ops.spmm_epilogue_raw(tp, tx, dY, n, g.spmm_plan(True), None, 0, Y)
    M = ops.spmm_raw(ip, ix, Xd, n, plan=g.spmm_plan(False), scattered=g.scattered(K * 4))
    x_mb = n * K * 4 / 1e6
    rows = []

    def t(label, fn, mb):
        us = bench.time_launches(fn, iters=50, warmup=10) * 1e6
        rows.append((label, us, mb / us if mb else 0.0))

    sc = K > ops.TILE_MIN_F and g.scattered(K * 4)
    out = ops.pad_rows(torch.empty_like(Xd))
    t("spmm F_in (A X)", lambda: ops.spmm_raw(ip, ix, Xd, n, out=out, plan=g.spmm_plan(False), out_padded=True, scattered=sc), 2 * x_mb)
    for knob, tag in ((0, "old"), (1, "new")):
        _lib.call("gae_tuning_set", b"xw", knob)
        t(f"linear fwd on M [{tag}]", lambda: ops.linear_fwd_raw(M, Wt, b, 1), x_mb)
        t(f"linear bwd dW, db on M [{tag}]", lambda: ops.linear_bwd_raw(dY, Y, 1, M, Wt, True, True, False), x_mb)
    t(f"xw_fwd (X W^T, {n_split} split partials kept)", lambda: ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True), x_mb)
    t("spmm F_out (+ split sum) + b, relu", lambda: ops.spmm_epilogue_raw(ip, ix, P, n, g.spmm_plan(False), b, 1), 0)
    t("gated spmm F_out on A^T", lambda: ops.spmm_epilogue_raw(tp, tx, dY, n, g.spmm_plan(True), None, 0, Y), 0)
    t("xw_wgrad (G^T X, db)", lambda: ops.xw_wgrad_raw(Xd, Gt, None, dY, Y, J), x_mb)
    print(f"== {name}: n = {n}, f_in = {K} (X {x_mb:.1f} MB)")
    for label, us, tbs in rows:
        print(f"  {label:34s} {us:8.2f} us" + (f"   {tbs:5.2f} TB/s of X" if tbs else ""))
    ref = rows[0][1] + rows[1][1] + rows[2][1]
    ref_new = rows[0][1] + rows[3][1] + rows[4][1]
    tf = sum(r[1] for r in rows[5:9])
    print(f"  chain, reference order, old dense kernels : {ref:7.2f} us")
    print(f"  chain, reference order, stream kernels    : {ref_new:7.2f} us")
    print(f"  chain, transform first                    : {tf:7.2f} us")
