"""xtg_kernel (dW1 = G^T X) on Pubmed with parts of the kernel switched off (knob xw_dbg: 1 no MFMA, 2 no X loads,
3 no G loads, 4 no main loop) -- where do its 14 us go?  (Measured, round 4: 2.5 us launch + epilogue, 7.4 us for the
loads alone, 6.4 us for G loads + MFMAs alone, 12.9 us together; an interleaved row assignment -- one contiguous window
of X in flight across the launch instead of 32 windows 1.3 MB apart -- changed nothing: not a memory-channel effect.)  kernel-only (the partial-sum form, no reduction launch)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from gae_dgl_amd import _lib, ops, workloads as W
dev = torch.device("cuda:0")
n, src, dst, X = W.citation_graph(sys.argv[1] if len(sys.argv) > 1 else "pubmed", seed=0)
K, J = X.shape[1], 32
Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
G = torch.randn(n, J, device=dev); Y = torch.randn(n, J, device=dev)
def t(fn): return bench.time_launches(fn, iters=100, warmup=20) * 1e6
Wt = torch.randn(J, K, device=dev) / K ** 0.5
with ops.deferred_grad_reductions() as step:
  for il in (0,):
    print(f"xw_fwd {t(lambda: ops.xw_fwd_raw(Xd, Wt, None, 0)):6.2f} us")
    for glds in (1, 0):          # G staged in LDS once per block | loaded per row group
        _lib.call("gae_tuning_set", b"xw_glds", glds)
        a = t(lambda: ops.xw_wgrad_raw(Xd, G, None, None, None, J))
        b = t(lambda: ops.xw_wgrad_raw(Xd, G, None, G, Y, J))
        c = t(lambda: ops.xw_wgrad_raw(Xd, G, Y, G, Y, J))
        print(f"  xw_glds {glds}: dW only {a:6.2f}  with db {b:6.2f}  masked {c:6.2f} us", flush=True)
    _lib.call("gae_tuning_set", b"xw_glds", 0)
    for dbg in (0, 1, 2, 3, 4):
        _lib.call("gae_tuning_set", b"xw_dbg", dbg)
        a = t(lambda: ops.xw_wgrad_raw(Xd, G, None, None, None, J))
        b = t(lambda: ops.xw_wgrad_raw(Xd, G, None, G, Y, J))
        c = t(lambda: ops.xw_wgrad_raw(Xd, G, Y, G, Y, J))
        print(f"  dbg{dbg}: dW only {a:6.2f}  with db {b:6.2f}  masked {c:6.2f} us", flush=True)
    _lib.call("gae_tuning_set", b"xw_dbg", 0)
    _lib.call("gae_tuning_set", b"xw_glds", 1)
  step.partials.clear()
