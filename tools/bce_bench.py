#!/usr/bin/env python3
"""A/B of the fused decoder+BCE kernel variants (interleaved rounds, median/min)."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gae_dgl_amd as G
from gae_dgl_amd import _lib, ops, workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("--graph", default="pubmed")
ap.add_argument("--d", type=int, default=16)
ap.add_argument("--variants", default="sym=1;sym=0;sym=0,sb=0")
ap.add_argument("--n", type=int, default=0, help="random graph with this many nodes instead of --graph")
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.n:
    rng = np.random.default_rng(0)
    n = a.n; src = rng.integers(0, n, 5 * n); dst = rng.integers(0, n, 5 * n)
else:
    n, src, dst, _ = W.citation_graph(a.graph)
g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
g.csr(); g.csc()
Z = torch.randn(n, a.d, device=dev) * 0.5
mask = ops.dropout_mask((n, a.d), 0.1, 1, device=dev)
pw = (n * n - g.number_of_edges()) / g.number_of_edges()
res = {}
ref = None
for rnd in range(a.rounds + 1):
    for v in a.variants.split(";"):
        kv = dict(x.split("=") for x in v.split(","))
        _lib.call("gae_tuning_set", b"bce_ri", int(kv.get("ri", 2)))
        _lib.call("gae_tuning_set", b"bce_s_bf16", int(kv.get("sb", 3)))
        _lib.call("gae_tuning_set", b"bce_pv_bf16", int(kv.get("pb", 1)))
        _lib.call("gae_tuning_set", b"bce_sym", int(kv.get("sym", 1)))
        _lib.call("gae_tuning_set", b"bce_sym_grid", int(kv.get("grid", 16384)))
        _lib.call("gae_tuning_set", b"bce_sym_ri", int(kv.get("sri", 0)))
        _lib.call("gae_tuning_set", b"bce_grid", int(kv.get("fgrid", 2048)))
        _lib.call("gae_tuning_set", b"bce_sym_tiles", int(kv.get("tiles", 0)))
        _lib.call("gae_tuning_set", b"bce_strip_store", int(kv.get("strip", -1)))
        _lib.call("gae_tuning_set", b"bce_fold_mirror", int(kv.get("fold", 1)))
        _lib.call("gae_tuning_set", b"bce_sym_tr", int(kv.get("tr", 1)))
        _lib.call("gae_tuning_set", b"bce_sym_bal", int(kv.get("bal", 1)))
        fn = lambda: ops.decoder_bce_raw(Z, mask, g.csr(), g.csc(), pw, True)
        loss, dz = fn(); torch.cuda.synchronize()
        if rnd == 0:
            if ref is None: ref = (loss.clone(), dz.clone())
            else:
                print(f"   {v}: loss rel diff {abs(float(loss) - float(ref[0])) / abs(float(ref[0])):.2e}, "
                      f"dZ rel diff {float((dz - ref[1]).abs().max()) / float(ref[1].abs().max()):.2e} (vs first variant)")
            continue
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(v, []).append(e0.elapsed_time(e1) * 1e-3 / 10)
for v, ts in res.items():
    print(f"{v:16s} median {np.median(ts)*1e6:8.1f} us  min {np.min(ts)*1e6:8.1f} us   {n*n/np.median(ts)/1e12:.3f} T logits/s")
