"""knob sweeps of the stream kernels on one citation shape (graph-replay timing)
  python tools/r03/xw_sweep.py [pubmed]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gae_dgl_amd import _lib, ops, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "pubmed"
n, src, dst, X = W.citation_graph(name, seed=0)
K, J = X.shape[1], 32
Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
Wt = torch.randn(J, K, device=dev) / K ** 0.5
G = torch.randn(n, J, device=dev)
Y = torch.randn(n, J, device=dev)


def knob(k, v):
    _lib.call("gae_tuning_set", k.encode(), int(v))


def t(fn):
    return bench.time_launches(fn, iters=50, warmup=10) * 1e6


print(f"{name}: n {n} K {K}")
for rows in (0, 16, 32, 48, 64, 80, 96, 160, 320):
    knob("xw_rows", rows)
    line = f"  fwd xw_rows {rows:4d}:"
    for dbg in (0, 1, 2):
        knob("xw_dbg", dbg)
        line += f"  dbg{dbg} {t(lambda: ops.xw_fwd_raw(Xd, Wt, None, 0)):7.2f}"
    print(line + " us")
knob("xw_rows", 0); knob("xw_dbg", 0)
for parts in (0, 7, 14, 21, 28, 56):
    knob("xw_parts", parts)
    print(f"  wgrad xw_parts {parts:3d}: {t(lambda: ops.xw_wgrad_raw(Xd, G, None, G, Y, J)):7.2f} us (kernel + reduce)"
          f"   dW only {t(lambda: ops.xw_wgrad_raw(Xd, G, None, None, None, J)):7.2f}")
knob("xw_parts", 0)
for dbg in (0, 1, 2):
    knob("xw_dbg", dbg)
    print(f"  wgrad dbg{dbg}: dW only {t(lambda: ops.xw_wgrad_raw(Xd, G, None, None, None, J)):7.2f}   with db {t(lambda: ops.xw_wgrad_raw(Xd, G, None, G, Y, J)):7.2f}   masked {t(lambda: ops.xw_wgrad_raw(Xd, G, Y, G, Y, J)):7.2f}")
knob("xw_dbg", 0)
for d in (0, 2, 4, 5, 6):
    knob("xw_depth", d)
    print(f"  xw_depth {d}: fwd {t(lambda: ops.xw_fwd_raw(Xd, Wt, None, 0)):7.2f}   wgrad dW only {t(lambda: ops.xw_wgrad_raw(Xd, G, None, None, None, J)):7.2f}   with db {t(lambda: ops.xw_wgrad_raw(Xd, G, None, G, Y, J)):7.2f}")
knob("xw_depth", 0)
for x in (1, 0):
    knob("xw_xcd", x)
    for parts in (0, 24, 32, 40):
        knob("xw_parts", parts)
        print(f"  xw_xcd {x} parts {parts}: wgrad dW only {t(lambda: ops.xw_wgrad_raw(Xd, G, None, None, None, J)):7.2f}   with db {t(lambda: ops.xw_wgrad_raw(Xd, G, None, G, Y, J)):7.2f}")
knob("xw_xcd", 1); knob("xw_parts", 0)
