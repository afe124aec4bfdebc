#!/usr/bin/env python3
"""the layer-1 aggregation A X at the input width (Pubmed F = 500, also Cora 1433 / Citeseer 3703) under
every XCD feature-tile width and store policy: warm (one operand set, Infinity-Cache resident), MALL-cold (rotation over
operand sets adding up to >= 512 MB), bit-identity with the untiled launch.
  python tools/r06/spmm_tile_sweep.py [--shape pubmed] [--tiles 0,8,16,24,32,-1] [--stores -1,0,1,2]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
import gae_dgl_amd as G  # noqa: E402
from gae_dgl_amd import _lib, ops, workloads as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="pubmed")
ap.add_argument("--tiles", default="0,8,16,24,32,40,64,-1")
ap.add_argument("--stores", default="-1,0,1,2")
ap.add_argument("--knobs", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
for kv in filter(None, a.knobs.split(",")):
    k, v = kv.split("=")
    _lib.call("gae_tuning_set", k.encode(), int(v))
n, src, dst, X = W.citation_graph(a.shape, seed=0)
F, E = X.shape[1], int(src.size)
b = W.spmm_alg_bytes(n, n, E, F, 4)


def make():
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    ip, ix = g.csr()
    Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
    out = ops.pad_rows(torch.empty(Xd.shape, device=dev))
    plan = g.spmm_plan(False)
    sc = F > ops.TILE_MIN_F and g.scattered(F * 4)
    return (lambda: ops.spmm_raw(ip, ix, Xd, n, out=out, plan=plan, out_padded=True, scattered=sc)), out


K = B.cold_copies(b)
sets = [make() for _ in range(K)]
_lib.call("gae_tuning_set", b"spmm_tile_vecs", -1)
ref = sets[0][0]().clone()
print(f"{a.shape}: n {n} E {E} F {F} ld {sets[0][1].stride(0)}  B_alg {b / 1e6:.1f} MB  cold rotation over {K} operand sets")
print(f"{'tile_vecs':>9s} {'store':>5s} {'warm us':>8s} {'frac':>6s} {'cold us':>8s} {'frac_cold':>9s}  bit-identical")
for tv in (int(t) for t in a.tiles.split(",")):
    for st in (int(s) for s in a.stores.split(",")):
        _lib.call("gae_tuning_set", b"spmm_tile_vecs", tv)
        _lib.call("gae_tuning_set", b"spmm_nt", st)
        try:
            got = sets[0][0]()
            same = bool(torch.equal(got, ref))
            tw = B.time_launches(sets[0][0], iters=50, warmup=20)
            tc = B.time_rotation([s[0] for s in sets])
            print(f"{tv:9d} {st:5d} {tw * 1e6:8.2f} {b / tw / 8e12:6.3f} {tc * 1e6:8.2f} {b / tc / 8e12:9.3f}  {same}", flush=True)
        except Exception as ex:  # noqa: BLE001
            print(f"{tv:9d} {st:5d}  failed: {ex}", flush=True)
_lib.call("gae_tuning_set", b"spmm_tile_vecs", 0)
_lib.call("gae_tuning_set", b"spmm_nt", -1)
