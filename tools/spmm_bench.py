#!/usr/bin/env python3
"""Within-process interleaved A/B of SpMM kernel variants on the BASELINE
shapes (cdna guide rule 24: interleave variants in one process, report median
and min).  Usage: python tools/spmm_bench.py [--rmat-scale 22] [--shapes ...]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gae_dgl_amd import _lib, ops, workloads as W  # noqa: E402


def knob(name, v):
    _lib.call("gae_tuning_set", name.encode(), int(v))


def time_once(fn, iters):
    """per-launch time of `iters` launches replayed from one HIP graph (eager Python launches cost ~10 us of host
    time each and would hide any kernel shorter than that)"""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rmat-scale", type=int, default=22)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--no-plan", action="store_true")
    ap.add_argument("--blockdiag", type=int, default=0, help="member graphs per block (0 = off) for the zinc shapes")
    ap.add_argument("--hot-cols", type=int, default=0, help="ops.HOT_COLUMNS for the rmat plan (0 = default)")
    ap.add_argument("--homed-deg", type=int, default=0, help="ops.HOMED_MIN_DEGREE (0 = default)")
    ap.add_argument("--homed-hot", type=int, default=0, help="ops.HOMED_HOT_COLUMNS (0 = default)")
    ap.add_argument("--homed-sweep", type=int, default=-1, help="ops.HOMED_COLUMN_SWEEP (0 / 1; -1 = default)")
    ap.add_argument("--thr", type=int, default=64)
    ap.add_argument("--seg", type=int, default=512)
    ap.add_argument("--variants", default="v1:1:0,v2:1:0,v2:2:0,v2:1:1,v2:2:1")
    ap.add_argument("--shapes", default="pubmed500,pubmed32,zincb39,zincb32,zinc39,zinc32,rmat32")
    ap.add_argument("--knobs", default="", help="comma-separated gae_tuning_set name=value pairs applied to every variant")
    args = ap.parse_args()
    if args.hot_cols:
        ops.HOT_COLUMNS = args.hot_cols
    if args.homed_deg:
        ops.HOMED_MIN_DEGREE = args.homed_deg
    if args.homed_hot:
        pass    # (round 4: the pinned part carries no hot tags)
    if args.homed_sweep >= 0:
        pass    # (round 4: chunks are always launched in first-column order)
    for kv in filter(None, args.knobs.split(",")):
        k, v = kv.split("=")
        knob(k, int(v))
    dev = torch.device("cuda:0")
    shapes = {}
    want = args.shapes.split(",")
    if any(s.startswith("pubmed") for s in want):
        n, src, dst, _ = W.citation_graph("pubmed")
        ip, ix = ops.csr_from_coo(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, n)
        shapes["pubmed500"] = (ip, ix, n, 500, 500)
        shapes["pubmed32"] = (ip, ix, n, 32, 32)
        shapes["pubmed500a"] = (ip, ix, n, 500, 512)   # rows padded to whole 128-byte lines
        shapes["pubmed500b"] = (ip, ix, n, 500, 544)   # ... and an odd number of lines per row
        shapes["pubmed500c"] = (ip, ix, n, 500, 576)   # 18 lines
    if any(s.startswith("pdiag") for s in want):   # diagnostics: Pubmed rows / degrees with controlled gather targets
        n, src, dst, _ = W.citation_graph("pubmed")
        rng = np.random.default_rng(3)
        d = torch.from_numpy(dst).to(dev)
        for nm, srcx in (("pdiag_col2k", rng.integers(0, 2000, dst.size)),          # H slice fits every L2
                         ("pdiag_self", dst.copy()),                                 # every neighbour = the row itself
                         ("pdiag_near", np.clip(dst + rng.integers(-8, 9, dst.size), 0, n - 1))):
            ipx, ixx = ops.csr_from_coo(d, torch.from_numpy(srcx).to(dev), n, n)
            shapes[nm] = (ipx, ixx, n, 500, 512)
        ipz = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        shapes["pdiag_zero"] = (ipz, torch.zeros(0, dtype=torch.int32, device=dev), n, 500, 512)   # no edges: stores only
    if any(s.startswith("preg") for s in want):     # random graph, every row exactly 4 / 8 in-edges
        n = 19717
        rng = np.random.default_rng(2)
        for dg in (4, 8):
            dstr = np.repeat(np.arange(n, dtype=np.int64), dg)
            srcr = rng.integers(0, n, dstr.size)
            ipr, ixr = ops.csr_from_coo(torch.from_numpy(dstr).to(dev), torch.from_numpy(srcr).to(dev), n, n)
            shapes[f"preg{dg}_500a"] = (ipr, ixr, n, 500, 512)
    for nm, Fd in (("cora", 1433), ("citeseer", 3703)):
        if any(s.startswith(nm) for s in want):
            n, src, dst, _ = W.citation_graph(nm)
            ip, ix = ops.csr_from_coo(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, n)
            shapes[f"{nm}{Fd}"] = (ip, ix, n, Fd, (Fd + 3) // 4 * 4)
            shapes[f"{nm}{Fd}a"] = (ip, ix, n, Fd, (Fd + 31) // 32 * 32)
    if any(s.startswith("pband") for s in want) or any(s.startswith("pself") for s in want):
        n, src, dst, _ = W.citation_graph("pubmed")
        rng = np.random.default_rng(1)
        srcb = np.clip(dst + rng.integers(-32, 33, dst.size), 0, n - 1)
        ipb, ixb = ops.csr_from_coo(torch.from_numpy(dst).to(dev), torch.from_numpy(srcb).to(dev), n, n)
        ips, ixs = ops.csr_from_coo(torch.from_numpy(dst).to(dev), torch.from_numpy(dst).to(dev), n, n)
        shapes["pband500"] = (ipb, ixb, n, 500, 500); shapes["pband32"] = (ipb, ixb, n, 32, 32)
        shapes["pself500"] = (ips, ixs, n, 500, 500); shapes["pself32"] = (ips, ixs, n, 32, 32)
    if any(s.startswith("zinc") for s in want):
        gp, src, dst, _ = W.zinc_like(249455)
        N = int(gp[-1])
        ip, ix = ops.csr_from_coo(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), N, N)
        nb = int(gp[4096]); eb = int(ip[nb])
        shapes["zincb39"] = (ip[:nb + 1].clone(), ix[:eb].clone(), nb, 39, 40)
        shapes["zincb32"] = (shapes["zincb39"][0], shapes["zincb39"][1], nb, 32, 32)
        shapes["zinc39"] = (ip, ix, N, 39, 40)
        shapes["zinc32"] = (ip, ix, N, 32, 32)
    if "rmat32" in want:
        src, dst = W.rmat_edges(args.rmat_scale, 16, device=dev)
        n = 1 << args.rmat_scale
        ip, ix = ops.csr_from_coo(dst, src, n, n)
        del src, dst
        shapes["rmat32"] = (ip, ix, n, 32, 32)
    bdiag = {}
    if any(s.startswith("zinc") for s in want) and args.blockdiag:
        bdiag["zinc39"] = bdiag["zinc32"] = ops.BlockDiag(gp, dev, graphs_per_block=args.blockdiag)
        bdiag["zincb39"] = bdiag["zincb32"] = ops.BlockDiag(gp[:4097], dev, graphs_per_block=args.blockdiag)
        for b in bdiag.values():
            b.min_blocks = 0
    plans = {}
    if "rmat32" in shapes and not args.no_plan:
        plans["rmat32"] = ops.spmm_plan(shapes["rmat32"][0], args.thr, args.seg, indices=shapes["rmat32"][1], ell=False,
                                        n_cols=shapes["rmat32"][2])
        pl = plans["rmat32"]
        print(f"rmat plan: thr={args.thr} seg={args.seg} heavy rows={pl.n_heavy} segments={pl.n_segments}")
    variants = []
    for v in args.variants.split(","):
        # name:rpg:nt:tile_vecs:opts   opts = letters: e = packed neighbour table + spmm_ell.hip kernels (w4 / w8 =
        #                              table width, default 16), E = table + row-group kernel, t = GAE_SPMM_TILE,
        #                              p = store pad, b = block-diagonal kernel
        parts = v.split(":")
        name, rpg, nt = parts[:3]
        tv = int(parts[3]) if len(parts) > 3 else 0
        opts = parts[4] if len(parts) > 4 else ""
        variants.append((v, 1 if name == "v1" else 2, int(rpg), int(nt), tv, opts))
    ell_plans = {}
    for sname in want:
        ip, ix, n, F, ld = shapes[sname]
        H = torch.rand(n, ld, device=dev)[:, :F]
        out = torch.empty(n, ld, device=dev)[:, :F]
        nnz = int(ix.numel())
        alg = W.spmm_alg_bytes(n, n, nnz, F)
        iters = max(3, min(100, int(2e-2 / max(alg / 3e12, 1e-6))))
        res = {v[0]: [] for v in variants}
        ref = None
        for rnd in range(args.rounds + 1):
            for (label, var, rpg, nt, tv, opts) in variants:
                knob("spmm_variant", var); knob("spmm_rpg", rpg); knob("spmm_nt", nt); knob("spmm_tile_vecs", tv)
                knob("spmm_ell", 2 if "E" in opts else 1); knob("spmm_ell_rpg", rpg)
                plan = plans.get(sname)
                if ("e" in opts or "E" in opts) and plan is None:
                    w = 4 if "w4" in opts else 8 if "w8" in opts else 16
                    if (id(ip), w) not in ell_plans:
                        ell_plans[(id(ip), w)] = ops.spmm_plan(ip, indices=ix, ell=True, ell_width=w)
                    plan = ell_plans[(id(ip), w)]
                fn = lambda: ops.spmm_raw(ip, ix, H, n, out=out, plan=plan, out_padded="p" in opts,
                                          scattered="t" in opts,
                                          blockdiag=bdiag.get(sname) if "b" in opts else None)
                fn(); torch.cuda.synchronize()
                if rnd == 0:
                    if ref is None:
                        ref = out.clone()
                    elif plans.get(sname) is None:
                        assert torch.equal(out[:, :F], ref[:, :F]), f"variant {label} differs on {sname}"
                    continue
                res[label].append(time_once(fn, iters))
        print(f"== {sname}: n={n} nnz={nnz} F={F} ld={ld} alg={alg/1e6:.1f} MB iters={iters}")
        Hc = torch.rand(n, F, device=dev); oc = torch.empty_like(Hc)
        tc = min(time_once(lambda: oc.copy_(Hc), iters) for _ in range(3))
        print(f"   [copy H->M  {tc*1e6:9.1f} us  {2*Hc.numel()*4/tc/1e9:8.1f} GB/s : device copy of the same H/M bytes]")
        del Hc, oc
        for label, ts in res.items():
            med, mn = float(np.median(ts)), float(np.min(ts))
            print(f"   {label:10s} median {med*1e6:9.1f} us  min {mn*1e6:9.1f} us  "
                  f"{alg/med/1e9:8.1f} GB/s ({alg/med/8e12*100:5.1f}% of 8 TB/s)  {nnz/med/1e9:7.2f} Gedge/s")


if __name__ == "__main__":
    main()
