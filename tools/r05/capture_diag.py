"""which operand layout makes hipStreamEndCapture crash on a graph with heavy rows?  each case in its own process
  python tools/r05/capture_diag.py            (driver)     python tools/r05/capture_diag.py <case>   (one case)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CASES = ["padded-hubs", "unpadded-hubs", "unpadded-nohubs", "sf-hubs"]

if len(sys.argv) > 1:
    import numpy as np
    import torch
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    from test_gpu_scripts import _planetoid_like
    case = sys.argv[1]
    dev = torch.device("cuda:0")
    if "nohubs" in case:
        n, src, dst, X = W.citation_graph("cora", seed=0)
    else:
        n, src, dst, X = _planetoid_like()
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Xd = torch.from_numpy(X).to(dev)
    if case.startswith("padded"):
        Xd = ops.pad_rows(Xd)
    if case.startswith("sf"):
        Xd = G.SparseFeatures.from_dense(ops.pad_rows(Xd))
    torch.manual_seed(0)
    model = G.GAE(X.shape[1], [32, 16]).to(dev)
    model.decoder.dropout = 0.0
    opt = Adam(model.parameters(), lr=1e-2)
    g.ndata['h'] = Xd
    l0 = model.reconstruction_loss(g); opt.zero_grad(); ops.backward(l0); opt.step()
    torch.cuda.synchronize()
    print(case, "eager ok", float(l0), flush=True)
    step = CapturedTrainStep(model, opt, g, Xd, warmup=0)
    print(case, "captured", flush=True)
    print(case, "replays", float(step()), float(step()), flush=True)
else:
    for c in CASES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=600)
        print(f"== {c}: rc {r.returncode}\n{r.stdout[-600:]}\n{r.stderr[-800:] if r.returncode else ''}", flush=True)
