#!/usr/bin/env python3
"""Plan construction on RMAT s24 (device builder, csrc/plan_build.hip): wall time and bytes per direction.
  python tools/r04/plan_build_time.py [scale]        (run under rocprofv3 --kernel-trace --stats to list its kernels)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import ops, workloads as W

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
src, dst = W.rmat_edges(scale, 16, device=dev)
n = 1 << scale
csr = {"fwd": ops.csr_from_coo(dst, src, n, n), "bwd": ops.csr_from_coo(src, dst, n, n)}
del src, dst
for rep in range(2):
    for k, (ip, ix) in csr.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan = ops.spmm_plan(ip, threshold=ops.SKEW_THRESHOLD, indices=ix, ell=False, n_cols=n)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"rep {rep} {k}: plan build {dt * 1e3:.1f} ms, {plan.nbytes() / 1e9:.3f} GB; light {plan.n_light}, mid {plan.n_heavy} "
              f"({plan.n_segments} segments), pinned {0 if plan.homed is None else plan.homed['rows'].numel()} rows / "
              f"{0 if plan.homed is None else plan.homed['n_edges']} edges / {0 if plan.homed is None else plan.homed['n_virtual']} "
              f"positions, tagged {plan.hot_indices is not None}", flush=True)
        del plan
