#!/usr/bin/env python3
"""How long does one dependent kernel node of a replayed HIP graph take when the kernel does (almost) nothing?"""
import torch
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
for n_nodes in (50, 200):
    for _ in range(3):
        x.add_(1.0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_nodes):
            x.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n_nodes)
    print(f"{n_nodes} dependent trivial kernels in one graph: {best:.2f} us per node")
