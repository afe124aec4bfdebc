#!/usr/bin/env python3
"""max error of the Linear kernels relative to the output scale, bf16 x 3 vs fp32-MFMA paths, against fp64"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gae_dgl_amd import _lib, ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for (n, fin, fout, dist) in ((19717, 500, 32, "normal"), (19717, 500, 32, "sparse01"), (2708, 1433, 32, "sparse01"),
                             (3327, 3703, 32, "normal"), (94752, 39, 32, "onehot"), (19717, 32, 16, "normal")):
    if dist == "normal":
        M = rng.standard_normal((n, fin)).astype(np.float32)
    elif dist == "sparse01":
        M = (rng.random((n, fin)) < 0.02).astype(np.float32) * rng.random((n, fin)).astype(np.float32)
    else:
        M = np.zeros((n, fin), np.float32); M[np.arange(n), rng.integers(0, fin, n)] = 1.0
    W = (rng.standard_normal((fout, fin)) / np.sqrt(fin)).astype(np.float32)
    dY = rng.standard_normal((n, fout)).astype(np.float32)
    Yref = torch.tensor(M, dtype=torch.float64) @ torch.tensor(W, dtype=torch.float64).t()
    dWref = torch.tensor(dY, dtype=torch.float64).t() @ torch.tensor(M, dtype=torch.float64)
    Md = ops.pad_rows(torch.from_numpy(M).to(dev)); Wd = torch.from_numpy(W).to(dev); dYd = torch.from_numpy(dY).to(dev)
    line = f"{n} x {fin} -> {fout} ({dist}):"
    for mode in (1, 0):
        _lib.call("gae_tuning_set", b"linear_bf16", mode); _lib.call("gae_tuning_set", b"atb_bf16", mode)
        Y = ops.linear_fwd_raw(Md, Wd, None, 0)
        dW, _, _ = ops.linear_bwd_raw(dYd, None, 0, Md, Wd, True, True, False)
        ey = float((Y.double().cpu() - Yref).abs().max() / Yref.abs().max())
        ew = float((dW.double().cpu() - dWref).abs().max() / dWref.abs().max())
        line += f"   {'bf16x3' if mode else 'fp32  '} fwd {ey:.1e} dW {ew:.1e}"
    print(line)
_lib.call("gae_tuning_set", b"linear_bf16", 0); _lib.call("gae_tuning_set", b"atb_bf16", 1)
