#!/usr/bin/env python3
"""gather-only / store-only / both: do the phases of the SpMM overlap or add?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import _lib, ops, workloads as W
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmm_bench import time_once, knob
dev = torch.device("cuda:0")
n, src, dst, _ = W.citation_graph("pubmed")
F, ld = 500, 512
rng = np.random.default_rng(3)
graphs = {"pubmed": src, "col2k": rng.integers(0, 2000, dst.size)}
H = torch.rand(n, ld, device=dev)[:, :F]; out = torch.empty(n, ld, device=dev)[:, :F]
d = torch.from_numpy(dst).to(dev)
for gname, s_ in graphs.items():
    ip, ix = ops.csr_from_coo(d, torch.from_numpy(s_).to(dev), n, n)
    for w in (16, 8):
        plan = ops.spmm_plan(ip, indices=ix, ell=True, ell_width=w)
        for tv in (16, 8):
            for rpg in (1, 2):
                for st in (1, 2):
                    row = []
                    for nostore in (0, 1):
                        knob("spmm_tile_vecs", tv); knob("spmm_ell_rpg", rpg); knob("spmm_nt", st); knob("spmm_ell_nostore", nostore)
                        fn = lambda: ops.spmm_raw(ip, ix, H, n, out=out, plan=plan, out_padded=True, scattered=True)
                        row.append(min(time_once(fn, 50) for _ in range(3)) * 1e6)
                    print(f"{gname:7s} W={w:2d} tile_vecs={tv:2d} rpg={rpg} store={st}: full {row[0]:6.2f} us   gather-only {row[1]:6.2f} us")
