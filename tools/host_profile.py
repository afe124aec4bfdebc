#!/usr/bin/env python3
"""cProfile of the host side of inductive training steps at a small batch size (where the step is host-bound)"""
import cProfile, os, pstats, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gae_dgl_amd as G
from gae_dgl_amd import ops
from gae_dgl_amd.dataset import DeviceGraphDataset
from gae_dgl_amd.optim import Adam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
ds = DeviceGraphDataset.synthetic_zinc(20000, seed=0, device=dev)
m = G.GAE(39, [32, 16]).to(dev)
opt = Adam(m.parameters(), lr=1e-3)
perm = np.random.default_rng(0).permutation(20000)
d_perm = torch.from_numpy(perm).to(dev)


def step(k):
    bg = ds._assemble(d_perm[k * B:(k + 1) * B], perm[k * B:(k + 1) * B])      # what ds.epoch() does per batch
    loss = m.reconstruction_loss(bg)
    opt.zero_grad(); ops.backward(loss); opt.step()


for k in range(20):
    step(k)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for k in range(20, 120):
    step(k)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
