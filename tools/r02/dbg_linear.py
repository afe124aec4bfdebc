import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gae_dgl_amd import ops, _lib
dev = torch.device("cuda:0")
for n, fin, fout in ((262161, 32, 16), (262161, 32, 32), (40000, 32, 16), (262161, 16, 16)):
    rng = np.random.default_rng(n + fin + fout)
    M = rng.standard_normal((n, fin)).astype(np.float32)
    W = (rng.standard_normal((fout, fin)) / np.sqrt(fin)).astype(np.float32)
    b = rng.standard_normal(fout).astype(np.float32)
    dY = rng.standard_normal((n, fout)).astype(np.float32)
    Mt = torch.tensor(M, dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    Yref = torch.relu(Mt @ Wt.t() + bt)
    Yref.backward(torch.tensor(dY, dtype=torch.float64))
    for knob in (0, 1, 2):
        _lib.call("gae_tuning_set", b"gemm_rows", knob)
        Mg = torch.tensor(M, device=dev).requires_grad_(True)
        Wd = torch.tensor(W, device=dev).requires_grad_(True); bd = torch.tensor(b, device=dev).requires_grad_(True)
        Y = ops.linear(Mg, Wd, bd, 1)
        Y.backward(torch.tensor(dY, device=dev))
        e = (Mg.grad.double().cpu() - Mt.grad).abs()
        bad = (e.max(dim=1).values > 1e-4).nonzero().flatten()
        print(n, fin, fout, "knob", knob, "Y err", float((Y.double().cpu() - Yref).abs().max()), "dM err", float(e.max()),
              "bad rows", bad.numel(), bad[:5].tolist(), bad[-5:].tolist(),
              "dW err", float((Wd.grad.double().cpu() - Wt.grad).abs().max()))
    _lib.call("gae_tuning_set", b"gemm_rows", 1)
