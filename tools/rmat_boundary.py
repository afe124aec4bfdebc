#!/usr/bin/env python3
"""How many distinct remote columns does each row block of the RMAT graph reference? (boundary vs all-gather volume)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gae_dgl_amd import workloads as W
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
src, dst = W.rmat_edges(scale, 16, device=dev)
n = 1 << scale
for world in (2, 4, 8):
    b = (n + world - 1) // world
    fr = []
    for r in range(world):
        m = (dst >= r * b) & (dst < (r + 1) * b)
        cols = src[m]
        remote = cols[(cols < r * b) | (cols >= (r + 1) * b)]
        need = torch.unique(remote).numel()
        fr.append((int(m.sum()), need, need / ((world - 1) * b)))
    print(f"world {world}: per rank (edges, distinct remote cols, fraction of all-gather rows):",
          [(e, k, round(f, 3)) for e, k, f in fr])
