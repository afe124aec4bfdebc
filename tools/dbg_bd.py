import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from gae_dgl_amd import ops, workloads as W, _lib
dev = torch.device("cuda:0")
for ng in (2000, 20000, 70000, 249455):
    gp, src, dst, _ = W.zinc_like(ng)
    n = int(gp[-1])
    ip, ix = ops.csr_from_coo(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, n)
    H = torch.rand(n, 32, device=dev)
    ref = ops.spmm_raw(ip, ix, H, n)
    bd = ops.BlockDiag(gp, dev, graphs_per_block=4)
    bp = bd.block_ptr.cpu().numpy()
    out = torch.full((n, 32), -1.0, device=dev)
    ops.spmm_raw(ip, ix, H, n, out=out, blockdiag=bd)
    torch.cuda.synchronize()
    print(ng, "blocks", bd.n_blocks, "max_rows", bd.max_rows, "bp ok", bp[0] == 0 and bp[-1] == n and np.all(np.diff(bp) > 0),
          "equal", torch.equal(out, ref), "untouched", int((out[:, 0] == -1).sum()))
