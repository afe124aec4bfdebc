#!/usr/bin/env python3
"""Benchmark of the GCN-encoder hot path on MI355X (contract: see the task
statement; metric and configs: BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W [--workload pubmed|cora|citeseer|zinc|rmat]

A "step" is one pass of the hot path over one batch of synthetic input:
  * citation workloads (pubmed = BASELINE configs[1], default at N=1): one
    full-graph training step = encoder forward (2 HIP SpMM + 2 fused Linear),
    inner-product decoder + weighted BCE, backward (HIP SpMM on A^T, Linear
    backward), Adam -- i.e. one transductive epoch.
  * value = SpMM edges aggregated per step (fwd + bwd launches) / step time,
    whole job, inputs resident in HBM.
The JSON line also carries `roofline` (dominant SpMM kernel, live HIP-event
timing inside the timed region), `cpu_baseline` (the CPU oracle's step timed on
this host) and `extra` (kernel-only SpMM numbers for the other BASELINE shapes).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--loss", choices=["fused", "dense"], default="fused",
                    help="fused = HIP decoder+BCE kernel (never materialises N x N); dense = reference-shaped "
                         "N x N logits/labels with torch BCE")
    return ap.parse_args()


def time_launches(fn, iters=50, warmup=5):
    """average duration of `fn`'s launches between one HIP event pair on the launch stream"""
    for _ in range(warmup):
        fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def spmm_probe(indptr, indices, n, F, dtype=torch.float32, label=""):
    """kernel-only SpMM throughput for one shape (fwd structure only)"""
    from gae_dgl_amd import ops, workloads as W
    H = torch.rand(n, F, device=indptr.device).to(dtype)
    out = torch.empty_like(H)
    t = time_launches(lambda: ops.spmm_raw(indptr, indices, H, n, out=out))
    nnz = int(indices.numel())
    b = W.spmm_alg_bytes(n, n, nnz, F, H.element_size())
    return {"shape": label, "n": n, "nnz": nnz, "F": F, "dtype": str(dtype).replace("torch.", ""),
            "us_per_launch": t * 1e6, "edges_per_s": nnz / t, "alg_bytes": b, "achieved_GBs": b / t / 1e9,
            "frac_hbm_peak": b / t / 1e9 / HBM_PEAK_GBS}


def citation_workload(name, args, dev):
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    n, src, dst, X = W.citation_graph(name, seed=0)
    F_in = X.shape[1]
    hidden = [32, 16]
    torch.manual_seed(0)
    model = G.GAE(F_in, hidden).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)       # train_transductive.py:43
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Xd = torch.from_numpy(X).to(dev)
    g.csr(); g.csc()                                          # structure is static across epochs
    E = g.number_of_edges()
    bce = torch.nn.functional.binary_cross_entropy_with_logits

    def step_dense():
        g.ndata['h'] = Xd
        adj = g.dense_adjacency()                             # train_transductive.py:59
        pw = (n * n - adj.sum()) / adj.sum()                  # :60
        logits = model(g)
        loss = bce(logits, adj, pos_weight=pw)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss

    def step_fused():
        g.ndata['h'] = Xd
        loss = model.reconstruction_loss(g)                   # same quantity, fused HIP kernel
        opt.zero_grad(); loss.backward(); opt.step()
        return loss

    step = step_fused if args.loss == "fused" else step_dense

    edges_per_step = E * (len(hidden) + len(hidden) - 1)      # L fwd + (L-1) bwd SpMM launches
    meta = {"workload": f"{name}-transductive-gae", "n_nodes": n, "n_edges": E, "in_dim": F_in,
            "hidden_dims": hidden, "norm": "none", "loss": args.loss + "-bce", "optimizer": "adam lr=1e-2"}
    dominant = ("spmm", n, n, F_in, "torch.float32")
    alg = W.spmm_alg_bytes(n, n, E, F_in, 4)
    return step, edges_per_step, meta, dominant, alg, (n, src, dst, X, F_in, hidden)


def cpu_baseline(cpu_args, edges_per_step, seconds):
    from oracle import gae_oracle as O
    n, src, dst, X, F_in, hidden = cpu_args
    ref = O.CpuReferenceStep(src, dst, n, X, F_in, hidden, lr=1e-2, seed=0)
    ref.step()  # warm-up
    t0 = time.perf_counter(); k = 0
    while True:
        ref.step(); k += 1
        el = time.perf_counter() - t0
        if el >= seconds or k >= 50:
            break
    return {"value": edges_per_step * k / el, "unit": "edges/s", "cores": torch.get_num_threads(),
            "kind": "port", "ms_per_step": el / k * 1e3,
            "sample": f"{k} full training steps of the same workload by oracle/gae_oracle.py:CpuReferenceStep "
                      f"(torch CPU, {torch.get_num_threads()} threads, host cpu_count={os.cpu_count()})"}


def extras(dev):
    """kernel-only SpMM numbers on the other BASELINE shapes (F quoted with each)"""
    from gae_dgl_amd import ops, workloads as W
    out = []
    # ZINC: batch of 4096 graphs and the whole set as one block-diagonal launch
    gptr, src, dst, X = W.zinc_like(249455, seed=0)
    N = int(gptr[-1])
    s = torch.from_numpy(src).to(dev); d = torch.from_numpy(dst).to(dev)
    ip, ix = ops.csr_from_coo(d, s, N, N)
    nb = int(gptr[4096]); eb = int(ip[nb])
    ipb, ixb = ip[:nb + 1].clone(), ix[:eb].clone()
    for F in (39, 32):
        out.append(spmm_probe(ipb, ixb, nb, F, label="zinc-batch4096"))
    Xp = torch.zeros(N, 40, device=dev); Xp[:, :39] = torch.from_numpy(X).to(dev)
    Hv = Xp[:, :39]
    o = torch.empty(N, 40, device=dev)[:, :39]
    t = time_launches(lambda: ops.spmm_raw(ip, ix, Hv, N, out=o), iters=20)
    b = W.spmm_alg_bytes(N, N, int(ix.numel()), 39, 4)
    out.append({"shape": "zinc-whole-set(ld=40)", "n": N, "nnz": int(ix.numel()), "F": 39, "us_per_launch": t * 1e6,
                "edges_per_s": ix.numel() / t, "alg_bytes": b, "achieved_GBs": b / t / 1e9,
                "frac_hbm_peak": b / t / 1e9 / HBM_PEAK_GBS})
    out.append(spmm_probe(ip, ix, N, 32, label="zinc-whole-set"))
    del s, d, ip, ix, Xp, o
    # RMAT scale 24
    src, dst = W.rmat_edges(24, 16, device=dev)
    n = 1 << 24
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    del src, dst
    out.append(spmm_probe(ip, ix, n, 32, label="rmat-s24-ef16"))
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    from gae_dgl_amd import ops

    workload = args.workload or "pubmed"
    step, edges_per_step, meta, dominant, alg_bytes, cpu_args = citation_workload(workload, args, dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = ops.EventProfiler()
    ops.profiler = prof
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ops.profiler = None
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    if rank != 0:
        return
    times = prof.summary()
    dom = times.get(dominant, [])
    t_dom = float(np.mean(dom)) if dom else float("nan")
    spmm_t = sum(sum(v) for k, v in times.items() if k[0] == "spmm")
    value = edges_per_step * args.steps / elapsed
    line = {
        "metric": "edges aggregated/sec (SpMM fwd+bwd)", "value": value, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": meta,
        "epoch_time_s": elapsed / args.steps,
        "spmm_only": {"edges_per_s": edges_per_step * args.steps / spmm_t if spmm_t else None,
                      "sum_event_time_s": spmm_t, "launches": sum(len(v) for k, v in times.items() if k[0] == "spmm"),
                      "note": "HIP-event time of the SpMM launches inside the timed steps"},
        "roofline": {"bound": "hbm", "kernel": f"spmm_rowgroup F={dominant[3]} (layer-1 aggregation)",
                     "achieved": alg_bytes / t_dom / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / t_dom / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "alg_bytes_per_launch": alg_bytes, "avg_launch_us": t_dom * 1e6, "launches_timed": len(dom)},
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cpu_args, edges_per_step, args.cpu_seconds)
    if not args.no_extra:
        line["extra"] = {"spmm_kernel_only": extras(dev)}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
