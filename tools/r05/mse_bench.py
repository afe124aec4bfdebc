#!/usr/bin/env python3
"""ops.decoder_mse (optuna_gae.py:16,21 without the N x N matrices: O(N d^2 + E d)) next to the fused BCE loss and,
where it fits, the reference-shaped chain (N x N logits + torch's MSELoss): loss + dZ, HIP-event time per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import gae_dgl_amd as G
from gae_dgl_amd import ops, workloads as W

dev = "cuda:0"


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def case(name, n, src, dst, d=16):
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    g.csr(); g.csc(); g.spmm_plan(False); g.spmm_plan(True); g.adjacency_sq_sum()
    torch.manual_seed(0)
    Z = (torch.randn(n, d, device=dev) * 0.5).requires_grad_(True)
    mask = (torch.rand(n, d, device=dev) >= 0.1).float() / 0.9

    def run(loss_fn):
        def f():
            Z.grad = None
            loss_fn().backward()
        return f
    t_mse = timed(run(lambda: ops.decoder_mse(Z, mask, g)))
    t_bce = timed(run(lambda: ops.decoder_bce(Z, mask, g)))
    line = f"{name:28s} N {n:7d} E {len(src):8d} d {d}: decoder_mse {t_mse:9.1f} us   decoder_bce (fused) {t_bce:9.1f} us"
    if n <= 20000:
        adj = g.dense_adjacency()
        t_dense = timed(run(lambda: torch.nn.MSELoss()(ops.decoder_dense(Z, mask), adj)), iters=10, warm=2)
        a = float(ops.decoder_mse(Z, mask, g)); b = float(torch.nn.MSELoss()(ops.decoder_dense(Z, mask), adj))
        line += f"   N x N chain + MSELoss {t_dense:9.1f} us   (values {a:.7g} / {b:.7g})"
    print(line, flush=True)


for nm in ("cora", "pubmed"):
    n, src, dst, _ = W.citation_graph(nm, seed=0)
    case(nm, n, src, dst)
gp, src, dst, X = W.zinc_like(4096, seed=0)
case("zinc batch of 4096 molecules", int(gp[-1]), src, dst)
