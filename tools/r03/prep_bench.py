#!/usr/bin/env python3
"""Kernel time of the last encoder launch with and without the loss's prepare step in its epilogue
(gae_gcn_layer_fused vs gae_gcn_layer_fused_prep; dropout drawn / no dropout), graph-replay timing."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gae_dgl_amd as G
from gae_dgl_amd import ops, workloads as W
from bench import time_launches

dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["pubmed", "cora"]:
    n, src, dst, _ = W.citation_graph(name)
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    ip, ix = g.csr(); plan = g.spmm_plan(False)
    H = torch.randn(n, 32, device=dev); Wt = torch.randn(16, 32, device=dev) * 0.1; b = torch.zeros(16, device=dev)
    base = time_launches(lambda: ops.gcn_layer_fused_raw(ip, ix, H, n, plan, Wt, b, ops.ACT_IDENTITY, want_m=True))
    print(f"{name}: n {n}   fused layer 32 -> 16 alone {base*1e6:6.2f} us")
    draws = torch.zeros(1, dtype=torch.int64, device=dev)
    for label, drop in (("no dropout", None), ("dropout drawn", (0.1, 7, 0, draws))):
        req = ops.loss_prepare_request(g, 16, None, drop)
        t = time_launches(lambda: ops.gcn_layer_fused_prep_raw(ip, ix, H, n, plan, Wt, b, None, req, want_m=True))
        print(f"   + prepare epilogue, {label:14s} {t*1e6:6.2f} us")
