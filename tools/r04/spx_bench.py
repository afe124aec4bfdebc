"""layer-1 dense passes on compressed input features (csrc/spfeat.hip) against the dense pair (xw.hip), kernel only
  python tools/r04/spx_bench.py [pubmed cora citeseer]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import gae_dgl_amd as G
from gae_dgl_amd import ops, workloads as W
dev = torch.device("cuda:0")
t = lambda fn: bench.time_launches(fn, iters=50, warmup=10) * 1e6
for name in (sys.argv[1:] or ["pubmed", "cora", "citeseer"]):
    n, src, dst, X = W.citation_graph(name, seed=0)
    K, J = X.shape[1], 32
    Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
    sf = G.SparseFeatures.from_dense(Xd)
    Wt = torch.randn(J, K, device=dev) / K ** 0.5
    Gm = torch.randn(n, J, device=dev); Y = torch.randn(n, J, device=dev)
    print(f"== {name}: {sf}, max segments per feature {sf.max_segments}, segments {sf.seg_feat.numel()}")
    with ops.deferred_grad_reductions() as step:
        print(f"  dense  xw_fwd {t(lambda: ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True)):6.2f}   xw_wgrad (dW, db) {t(lambda: ops.xw_wgrad_raw(Xd, Gm, None, Gm, Y, J)):6.2f} us")
        print(f"  sparse spx_fwd {t(lambda: ops.spx_fwd_raw(sf, Wt)):6.2f}   spx_wgrad dW only {t(lambda: ops.spx_wgrad_raw(sf, Gm, None, None, J, need_db=False)):6.2f}   dW, db {t(lambda: ops.spx_wgrad_raw(sf, Gm, Gm, Y, J)):6.2f} us")
        step.partials.clear()
