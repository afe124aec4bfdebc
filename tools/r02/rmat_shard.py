#!/usr/bin/env python3
"""One GPU plays rank r of an 8-way row-sharded RMAT s24: the rank's own-column and remote-column SpMM launches with
and without the XCD-pinned (homed) part of the skew plan, checked against the rows the single-GPU product gives.

    python tools/r02/rmat_shard.py [scale] [world] [ranks, comma separated]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import ops, workloads as W
from gae_dgl_amd.parallel import ShardedGraph, LocalGroup

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ranks = [int(r) for r in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, world // 2, world - 1]
dev = torch.device("cuda:0")
n = 1 << scale
src, dst = W.rmat_edges(scale, 16, seed=0, device=dev)
F = 32
H = torch.rand(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
ip, ix = ops.csr_from_coo(dst, src, n, n)
ref = ops.spmm_raw(ip, ix, H, n, plan=ops.spmm_plan(ip, threshold=ops.SKEW_THRESHOLD, indices=ix, ell=False,
                                                    n_cols=n, homed=False))
scale_ref = float(ref.abs().max())
del ip, ix


def timed(fn, rounds=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rounds


for mode, overlap in (("boundary", True), ("boundary", False), ("allgather", False)):
    for r in ranks:
        row = [f"{mode:9s} overlap={int(overlap)} rank {r}"]
        for homed_edges in (1 << 62, ops.HOMED_MIN_EDGES):
            saved = ops.HOMED_MIN_EDGES
            ops.HOMED_MIN_EDGES = homed_edges
            g = LocalGroup(world)
            sg = ShardedGraph(n, src, dst, rank=r, group=g, mode=mode, device=dev, balance="nnz", overlap=overlap)
            g.publish(H)
            p = sg.part
            h_local = H[p.r0:p.r1].contiguous()
            from gae_dgl_amd.parallel import ShardedSpMMFunction
            out = ShardedSpMMFunction._product(sg, h_local, "fwd")
            err = float((out - ref[p.r0:p.r1]).abs().max()) / scale_ref
            ms = timed(lambda: ShardedSpMMFunction._product(sg, h_local, "fwd"))
            # the products alone (no index_select / cat of the stand-in exchange)
            if overlap:
                recv, _ = sg.exchange_start(h_local, "fwd")
                oip, oix = sg.csr("fwd", "own"); rip, rix = sg.csr("fwd", "remote")
                o = torch.empty(p.n_local, F, device=dev)
                t_own = timed(lambda: ops.spmm_raw(oip, oix, h_local, p.n_local, out=o, plan=sg.plan("fwd", "own")))
                t_rem = timed(lambda: ops.spmm_raw(rip, rix, recv, p.n_local, out=o, accumulate=True,
                                                   plan=sg.plan("fwd", "remote")))
                parts = f"own {t_own:.3f} + remote {t_rem:.3f}"
                nv = sg.plan("fwd", "remote").homed
            else:
                full = sg.exchange(h_local, "fwd")
                lip, lix = sg.csr("fwd")
                o = torch.empty(p.n_local, F, device=dev)
                t_all = timed(lambda: ops.spmm_raw(lip, lix, full, p.n_local, out=o, plan=sg.plan("fwd")))
                parts = f"spmm {t_all:.3f}"
                nv = sg.plan("fwd").homed
            row.append(f"{'homed' if homed_edges < (1 << 62) else 'plain'}{'*' if nv is not None else ''}: "
                       f"{parts} ms (product+exchange stand-in {ms:.3f}), rel err {err:.1e}")
            ops.HOMED_MIN_EDGES = saved
            del sg, g
            torch.cuda.empty_cache()
        print(" | ".join(row), f"| rows {p.n_local} edges {int((dst >= p.r0).sum() - (dst >= p.r1).sum())}", flush=True)
