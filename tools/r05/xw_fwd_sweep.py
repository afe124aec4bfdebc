"""gae_xw_fwd on the Pubmed shape: ring depth x reduction chunk x blocks per CU (+ the no-MFMA / no-load forms)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gae_dgl_amd import _lib, ops, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "pubmed"
n, src, dst, X = W.citation_graph(name, seed=0)
K, J = X.shape[1], 32
Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
Wt = torch.randn(J, K, device=dev) / K ** 0.5
ref = Xd[:, :K].double() @ Wt.double().t()
x_mb = n * K * 4 / 1e6
setk = lambda k, v: _lib.call("gae_tuning_set", k.encode(), int(v))


def run(label, **kn):
    for k, v in kn.items():
        setk(k, v)
    P, ns = ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True)
    Pf = P.reshape(ns, n, J).sum(0) if ns > 1 else P[:, :J]
    err = float((Pf.double() - ref).abs().max() / ref.abs().max())
    ts = sorted(bench.time_launches(lambda: ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True), iters=50, warmup=10) * 1e6
                for _ in range(3))
    print(f"  {label:44s} {ts[0]:7.2f} {ts[1]:7.2f} {ts[2]:7.2f} us   {x_mb / ts[1]:5.2f} TB/s  err {err:.1e}", flush=True)
    for k in kn:
        setk(k, {"xw_p3": 1, "xw_bpc": 1}.get(k, 0))


print(f"== {name}: n = {n}, f_in = {K} (X {x_mb:.1f} MB)")
run("exact fp32 (xw_p3 = 0)", xw_p3=0)
run("p3 (default)")
run("p3 no MFMA", xw_dbg=1)
run("p3 no X loads", xw_dbg=2)
sys.exit(0)
run("p3 default (depth 2, tc 6, 1 block / CU)")
for depth in (2, 3, 4):
    for tc in (6, 3):
        for bpc in (1, 2):
            if bpc == 2 and tc == 6:
                continue
            run(f"p3 depth {depth} tc {tc} bpc {bpc}", xw_depth=depth, xw_tc=tc, xw_bpc=bpc)
run("p3 no MFMA (dbg 1)", xw_dbg=1)
run("p3 no X loads (dbg 2)", xw_dbg=2)
run("p3 depth 4 tc 3 no MFMA", xw_dbg=1, xw_depth=4, xw_tc=3)
run("p3 depth 4 tc 3 no X loads", xw_dbg=2, xw_depth=4, xw_tc=3)
for rows in (32, 48, 64, 96):
    run(f"p3 depth 3 tc 3 rows/block {rows}", xw_depth=3, xw_tc=3, xw_rows=rows)
