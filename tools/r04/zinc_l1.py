"""ZINC batch-4096 layer-1 aggregation (F = 39, ld = 40), kernel only: the launch as bench.py's ZincWorkload issues it
and variants of its operands (alignment, plan source, block-diagonal hint) -- the round-3 regression hunt"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from gae_dgl_amd import ops
from gae_dgl_amd.dataset import DeviceGraphDataset

dev = torch.device("cuda:0")
B = 4096
ds = DeviceGraphDataset.synthetic_zinc(32768, seed=0, device=dev)
perm = np.random.default_rng(0).permutation(32768)
bg = ds.batch(perm[:B])
ip, ix = bg.csr()
H = bg.ndata['h']
n = bg.number_of_nodes(); E = bg.number_of_edges()
alg = 4 * (n + 1) + 4 * E + 2 * 4 * 39 * n
plan = bg.spmm_plan(False)
print("n", n, "E", E, "H", tuple(H.shape), H.stride(), H.dtype, "ptr%512", H.data_ptr() % 512, "ell_width", ds.ell_width)
for name, t in plan.tensors.items() if hasattr(plan, "tensors") and isinstance(plan.tensors, dict) else []:
    print("  plan tensor", name, tuple(t.shape), t.data_ptr() % 512)

def run(label, fn, iters=200):
    for rep in range(3):
        t = bench.time_launches(fn, iters=iters, warmup=50)
    print(f"{label:60s} {t*1e6:7.2f} us  frac {alg/t/1e9/8000:.3f}", flush=True)

out = ops.pad_rows(torch.empty(H.shape, device=dev))
run("bench: batch operands, table plan, blockdiag", lambda: ops.spmm_raw(ip, ix, H, n, out=out, out_padded=True, blockdiag=bg.block_diag, plan=plan))
run("no blockdiag hint", lambda: ops.spmm_raw(ip, ix, H, n, out=out, out_padded=True, plan=plan))
H2 = ops.pad_rows(torch.rand(n, 39, device=dev)); assert H2.stride(0) == 40
run("fresh random H (fp32, ld 40)", lambda: ops.spmm_raw(ip, ix, H2, n, out=out, out_padded=True, blockdiag=bg.block_diag, plan=plan))
ipc, ixc = ip.clone(), ix.clone()
pb = ops.spmm_plan(ipc, indices=ixc, ell=True, ell_width=ops.ell_width_for_degrees(ipc[1:] - ipc[:-1]))
run("plan from ops.spmm_plan (extras' way)", lambda: ops.spmm_raw(ipc, ixc, H2, n, out=out, out_padded=True, plan=pb))
run("out_padded=False", lambda: ops.spmm_raw(ipc, ixc, H2, n, out=out, plan=pb))
for w in (4, 8, 12, 16):
    try:
        pw = ops.spmm_plan(ipc, indices=ixc, ell=True, ell_width=w)
        run(f"table width {w}", lambda: ops.spmm_raw(ipc, ixc, H2, n, out=out, out_padded=True, plan=pw))
    except Exception as e:
        print("width", w, "failed:", e)
from gae_dgl_amd import _lib
for knob, vals in (("spmm_ell_rpg", (1, 2)),):
    for v in vals:
        try:
            _lib.call("gae_tuning_set", knob.encode(), v)
            run(f"knob {knob}={v}", lambda: ops.spmm_raw(ipc, ixc, H2, n, out=out, out_padded=True, plan=pb))
        except Exception as e:
            print("knob", knob, v, "failed:", e)
    try:
        _lib.call("gae_tuning_set", knob.encode(), 0)
    except Exception:
        pass
