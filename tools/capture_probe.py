#!/usr/bin/env python3
"""find which configuration breaks HIP-graph capture of the training step"""
import os, sys, faulthandler
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gae_dgl_amd as G
from gae_dgl_amd import ops
from gae_dgl_amd.capture import CapturedTrainStep
from gae_dgl_amd.optim import Adam
# faulthandler.enable()
n, F, drop, warm = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
eager_first = (sys.argv[5] if len(sys.argv) > 5 else "1") == "1"
gen = sys.argv[6] if len(sys.argv) > 6 else "rng"
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
if gen == "rng":
    src = rng.integers(0, n, 5 * n); dst = rng.integers(0, n, 5 * n)
else:
    from gae_dgl_amd import workloads as W
    n, src, dst, _ = W.citation_graph(gen)
X = torch.randn(n, F, device=dev)
g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
m = G.GAE(F, [32, 16]).to(dev); m.decoder.dropout = drop
opt = Adam(m.parameters(), lr=1e-2)
Xd = ops.pad_rows(X)
g.ndata['h'] = Xd
if eager_first:
    loss = m.reconstruction_loss(g); opt.zero_grad(); ops.backward(loss); opt.step()
    torch.cuda.synchronize(); print("eager ok", float(loss.detach()), flush=True)
    del loss      # a live loss keeps AccumulateGrad nodes of the default stream alive and breaks the capture
step = CapturedTrainStep(m, opt, g, Xd, warmup=warm)
print("captured", flush=True)
for _ in range(3):
    l = step()
torch.cuda.synchronize(); print("replayed", float(l), flush=True)
