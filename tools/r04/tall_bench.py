"""the dense passes of the R-MAT rank step (csrc/tall.hip) on 2^24 rows: all rows | a contiguous 30 % | list mode with 30 % live rows
  python tools/r04/tall_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gae_dgl_amd import ops
dev = torch.device("cuda:0")
n = 1 << 24
g = torch.Generator(device=dev).manual_seed(0)
M1 = torch.randn(n, 32, device=dev, generator=g); G = torch.randn(n, 16, device=dev, generator=g); dZ = torch.randn(n, 16, device=dev, generator=g)
W1 = torch.randn(32, 32, device=dev, generator=g) / 6; b1 = torch.randn(32, device=dev, generator=g); W2 = torch.randn(16, 32, device=dev, generator=g) / 6
m1_dead = (torch.rand(n, device=dev, generator=g) < 0.7); g_dead = (torch.rand(n, device=dev, generator=g) < 0.7)
md, gd = m1_dead.to(torch.uint8), g_dead.to(torch.uint8)
rows = torch.nonzero(~m1_dead).reshape(-1).to(torch.int32); gdl = gd[rows.long()].contiguous()
k = int(rows.numel())
t = lambda fn: bench.time_launches(fn, iters=10, warmup=3) * 1e3
print(f"n {n}, live {k}")
print(f"linear2 all rows          {t(lambda: ops.linear2_fwd_raw(M1, W1, b1, 1, W2, want_y1=False)):.3f} ms")
print(f"linear2 first {k} rows   {t(lambda: ops.linear2_fwd_raw(M1[:k], W1, b1, 1, W2, want_y1=False)):.3f} ms")
print(f"linear2 list mode         {t(lambda: ops.linear2_fwd_raw(M1, W1, b1, 1, W2, want_y1=False, a_dead=md, rows=rows)):.3f} ms")
print(f"gcn2 all rows             {t(lambda: ops.gcn2_bwd_dense_raw(G, dZ, None, 1, M1, W2, W1=W1, b1=b1)):.3f} ms")
print(f"gcn2 first {k} rows      {t(lambda: ops.gcn2_bwd_dense_raw(G[:k], dZ[:k], None, 1, M1[:k], W2, W1=W1, b1=b1)):.3f} ms")
print(f"gcn2 list mode            {t(lambda: ops.gcn2_bwd_dense_raw(G, dZ, None, 1, M1, W2, W1=W1, b1=b1, m1_dead=md, g_dead=gd, rows=rows, g_dead_listed=gdl)):.3f} ms")
