"""layer-1 dense pair A/B: gae_xw_fwd / gae_xw_wgrad under knob settings, time (graph replay) and error against fp64
  python tools/r05/xw_ab.py [pubmed cora citeseer] [--knobs xw_p3=0,1 ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gae_dgl_amd import _lib, ops, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
names = [a for a in sys.argv[1:] if not a.startswith("--") and "=" not in a] or ["pubmed", "cora", "citeseer"]
sweeps = [a for a in sys.argv[1:] if "=" in a] or ["xw_p3=0,1"]


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


for name in names:
    n, src, dst, X = W.citation_graph(name, seed=0)
    K, J = X.shape[1], 32
    Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
    g = torch.Generator(device=dev).manual_seed(0)
    Wt = torch.randn(J, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(J, device=dev, generator=g)
    G = torch.randn(n, J, device=dev, generator=g)
    ref_f = Xd[:, :K].double() @ Wt.double().t()
    ref_b = G.double().t() @ Xd[:, :K].double()
    x_mb = n * K * 4 / 1e6
    print(f"== {name}: n = {n}, f_in = {K} (X {x_mb:.1f} MB)")
    for sw in sweeps:
        knob, vals = sw.split("=")
        for v in vals.split(","):
            _lib.call("gae_tuning_set", knob.encode(), int(v))
            P, ns = ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True)
            Pf = P.reshape(ns, n, J).sum(0) if ns > 1 else P[:, :J]
            ef = rel(Pf, ref_f)
            tf = bench.time_launches(lambda: ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True), iters=50, warmup=10) * 1e6
            dW, _ = ops.xw_wgrad_raw(Xd, G, None, G, None, J)
            eb = rel(dW, ref_b)
            tb = bench.time_launches(lambda: ops.xw_wgrad_raw(Xd, G, None, G, None, J), iters=50, warmup=10) * 1e6
            print(f"  {knob}={v}: xw_fwd {tf:7.2f} us ({x_mb / tf:5.2f} TB/s, err {ef:.1e})   xw_wgrad {tb:7.2f} us "
                  f"({x_mb / tb:5.2f} TB/s, err {eb:.1e})")
        _lib.call("gae_tuning_set", knob.encode(), int(vals.split(",")[-1]))
