#!/usr/bin/env python3
"""Kernel time of the fused layer's backward launch with and without its weight-gradient side work
(gae_gcn_layer_fused_wgrad vs gae_gcn_layer_fused on A^T), graph-replay timing, per variant of the side work."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gae_dgl_amd as G
from gae_dgl_amd import _lib, ops, workloads as W
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import time_launches

dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["pubmed", "cora"]:
    n, src, dst, _ = W.citation_graph(name)
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    g.csr(); g.csc(); plan_t = g.spmm_plan(True)
    dY = torch.randn(n, 16, device=dev); Wt = torch.randn(16, 32, device=dev) * 0.1
    M = torch.randn(n, 32, device=dev)
    ti, tx = g.csc()
    base = time_launches(lambda: ops.gcn_layer_fused_raw(ti, tx, dY, n, plan_t, Wt, None, ops.ACT_IDENTITY, w_transposed=True, want_m=False))
    print(f"{name}: n {n}   backward gather alone {base*1e6:6.2f} us")
    for v in (2, 6, 14):
        _lib.call("gae_tuning_set", b"ell_side", v)
        with ops.deferred_grad_reductions():
            t = time_launches(lambda: ops.gcn_layer_fused_wgrad_raw(ti, tx, dY, n, plan_t, Wt, M, None))
            ops.current_step().partials.clear()
        print(f"   ell_side {v}: with dW / db side work {t*1e6:6.2f} us")
    _lib.call("gae_tuning_set", b"ell_side", 14)
