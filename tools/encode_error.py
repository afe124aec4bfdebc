#!/usr/bin/env python3
"""end-to-end error of GAE.encode and of the parameter gradients against fp64 (torch CPU), bf16 x 3 vs fp32 Linear"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gae_dgl_amd as G
from gae_dgl_amd import _lib, ops, workloads as W
dev = torch.device("cuda:0")
for name in ("cora", "pubmed"):
    n, src, dst, X = W.citation_graph(name, seed=0)
    A = torch.sparse_coo_tensor(np.stack([dst, src]), np.ones(len(src)), (n, n), dtype=torch.float64).coalesce()
    torch.manual_seed(0)
    m = G.GAE(X.shape[1], [32, 16]).to(dev); m.decoder.dropout = 0.0
    Ws = [l.apply_mod.linear.weight.detach().double().cpu() for l in m.layers]
    bs = [l.apply_mod.linear.bias.detach().double().cpu() for l in m.layers]
    Xr = torch.tensor(X, dtype=torch.float64)
    Wr = [w.clone().requires_grad_(True) for w in Ws]; br = [b.clone().requires_grad_(True) for b in bs]
    h = torch.relu(torch.sparse.mm(A, Xr) @ Wr[0].t() + br[0])
    Zr = torch.sparse.mm(A, h) @ Wr[1].t() + br[1]
    adj = A.to_dense(); pw = (n * n - adj.sum()) / adj.sum()
    lr = torch.nn.functional.binary_cross_entropy_with_logits(Zr @ Zr.t(), adj, pos_weight=pw)
    lr.backward()
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
    for mode in (1, 0):
        _lib.call("gae_tuning_set", b"linear_bf16", mode); _lib.call("gae_tuning_set", b"atb_bf16", mode)
        m.zero_grad()
        g.ndata['h'] = Xd
        Z = m.encode(g)
        g.ndata['h'] = Xd
        loss = m.reconstruction_loss(g); loss.backward()
        ez = float((Z.double().cpu() - Zr.detach()).abs().max() / Zr.detach().abs().max())
        el = abs(float(loss) - float(lr)) / abs(float(lr))
        eg = max(float((l.apply_mod.linear.weight.grad.double().cpu() - w.grad).abs().max() / w.grad.abs().max())
                 for l, w in zip(m.layers, Wr))
        print(f"{name}: {'bf16x3' if mode else 'fp32  '} Linear  encode err {ez:.1e}  loss err {el:.1e}  dW err {eg:.1e}")
_lib.call("gae_tuning_set", b"linear_bf16", 0); _lib.call("gae_tuning_set", b"atb_bf16", 1)
