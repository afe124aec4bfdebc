#!/usr/bin/env python3
"""Compute-only time of ONE rank's encoder step of the 8-way row-sharded RMAT s24 (bench.py --gpus 8 without the
collectives: the exchanges return buffers of the right shape filled once), for several row costs of the block
balance.  One GPU plays the ranks one after the other.

    python tools/r02/rmat_rank_step.py [scale] [world] [row costs, comma separated]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gae_dgl_amd as G
from gae_dgl_amd import ops, parallel, workloads as W
from gae_dgl_amd.optim import Adam
from gae_dgl_amd.parallel import ShardedGraph, LocalGroup, sharded_encode

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
costs = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 8, 16, 32]
dev = torch.device("cuda:0")
n = 1 << scale
src, dst = W.rmat_edges(scale, 16, seed=0, device=dev)


class NoCommGraph(ShardedGraph):
    """exchanges replaced by resident buffers (timing only: the values are not the other ranks' rows)"""
    def exchange_start(self, h_local, which="fwd"):
        key = (which, h_local.shape[1])
        self._buf = getattr(self, "_buf", {})
        if key not in self._buf:
            rows = self.part.split[which]["n_remote_cols"]
            self._buf[key] = torch.rand(rows, h_local.shape[1], device=h_local.device)
        return self._buf[key], (lambda: None)


for cost in costs:
    parallel.ROW_COST = cost
    line = []
    for r in range(world):
        sg = NoCommGraph(n, src, dst, rank=r, group=LocalGroup(world), mode="boundary", device=dev, balance="nnz",
                         overlap=True)
        p = sg.part
        for w in ("fwd", "bwd"):
            for part in ("own", "remote"):
                sg.csr(w, part); sg.plan(w, part)
        X = torch.rand(p.n_local, 32, device=dev)
        dZ = torch.randn(p.n_local, 16, device=dev) / n
        torch.manual_seed(0)
        model = G.GAE(32, [32, 16]).to(dev)
        opt = Adam(model.parameters(), lr=1e-2)

        def step():
            z = sharded_encode(model, sg, X, transform_first=True)
            opt.zero_grad(); z.backward(dZ); opt.step()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        recv = sum(sg.exchange_bytes(f, which=w) for f, w in ((32, "fwd"), (16, "fwd"), (16, "bwd")))
        line.append((r, p.n_local, sg.n_edges("fwd"), round(ms, 3), round(recv / 1e6, 1)))
        del sg, model, opt, X, dZ
        torch.cuda.empty_cache()
    print(f"row cost {cost}: max {max(l[3] for l in line):.3f} ms, sum {sum(l[3] for l in line):.3f} ms | "
          f"(rank, rows, in-edges, ms, MB received/step): {line}", flush=True)
