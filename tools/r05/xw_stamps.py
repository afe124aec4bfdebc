"""where does gae_xw_fwd's time go?  s_memtime stamps of every (block, wave): start, prologue issued, after each tile,
after the reduction.  Prints the distribution over blocks relative to the earliest start of the launch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gae_dgl_amd import _lib, ops, workloads as W  # noqa: E402

dev = torch.device("cuda:0")
n, src, dst, X = W.citation_graph("pubmed", seed=0)
K, J = X.shape[1], 32
Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
Wt = torch.randn(J, K, device=dev) / K ** 0.5
stamps = torch.zeros(256, 8, 16, dtype=torch.int64, device=dev)
setk = lambda k, v: _lib.call("gae_tuning_set", k.encode(), int(v))
for _ in range(3):
    ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True)
lo = stamps.data_ptr() & 0xffffffff
setk("xw_stamps", lo - (1 << 32) if lo >= (1 << 31) else lo); setk("xw_stamps_hi", stamps.data_ptr() >> 32); setk("xw_dbg", 3)
for rep in range(3):
    stamps.zero_()
    ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64)
    live = s[:, 0, 0] > 0
    s = s[live]
    # the counters of different XCDs are not synchronised: every block is measured from ITS earliest wave start
    rel = s - s[:, :, 0].min(axis=1)[:, None, None]
    print(f"rep {rep}: {int(live.sum())} blocks; shader cycles after the block's first wave started")
    names = ["start", "prologue issued", "tile 0", "tile 1", "tile 2", "tile 3", "tile 4", "", "", "", "", "", "reduced"]
    for k, nm in enumerate(names):
        if not nm:
            continue
        v = rel[:, :, k][s[:, :, k] > 0]
        if v.size:
            print(f"  {nm:16s} min {v.min():8.0f}  p10 {np.percentile(v, 10):8.0f}  median {np.median(v):8.0f}  "
                  f"p90 {np.percentile(v, 90):8.0f}  max {v.max():8.0f}")
    last_tile = np.where(s[:, :, 6] > 0, rel[:, :, 6], rel[:, :, 5])
    print(f"  per block: slowest wave's last tile  median {np.median(last_tile.max(axis=1)):8.0f}   fastest wave's  "
          f"{np.median(last_tile.min(axis=1)):8.0f}   block end  median {np.median(rel[:, :, 12].max(axis=1)):8.0f}  "
          f"max {rel[:, :, 12].max():8.0f}")
    for w in range(8):
        print(f"    wave {w}: start {np.median(rel[:, w, 0]):6.0f}  prologue {np.median(rel[:, w, 1]):6.0f}  tile0 {np.median(rel[:, w, 2]):6.0f}  "
              f"last tile {np.median(last_tile[:, w]):6.0f}  reduced {np.median(rel[:, w, 12]):6.0f}")
np.save(os.path.join(ROOT, "gpurun_out", "r05_xw_stamps.npy"), stamps.cpu().numpy())
setk("xw_dbg", 0); setk("xw_stamps", 0); setk("xw_stamps_hi", 0)
