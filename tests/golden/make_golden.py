#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the reference's
OWN model file.

Runs ONLY in the build container (needs the read-only upstream tree at
/root/reference).  It imports ``/root/reference/gae_dgl/gae.py`` unmodified
through importlib; the third-party ``dgl`` package the file needs is absent
from the image, so a small in-memory stand-in for the six DGL symbols
``gae.py`` touches is registered in ``sys.modules`` first.  The stand-in
encodes DGL's documented semantics at the reference's call sites:

* ``g.update_all(copy_src('h','m'), sum('m','h'))``  -> in-edge sum
  (gae.py:18-19,28)
* ``g.apply_nodes(func)`` -> ``func(nodes)`` with ``nodes.data`` = ndata, the
  returned dict written back (gae.py:29,13-16)
* ``dgl.batch`` / ``adjacency_matrix`` / ``in_degrees`` as used by
  train_inductive.py:34,44 and train_transductive.py:55,59.

Only DATA is written (inputs and expected outputs, .npz); no reference source
or bytecode is copied.  Usage:  python tests/golden/make_golden.py

Reproducibility: main() pins torch to ONE thread.  ATen's threaded fp32 GEMMs
split their reductions by thread count, so with 8 threads the wide cases'
layer-1 gradients (f_in 300 / 2003) move in their last bits from run to run
and the three Adam steps amplify that to 1e-5 .. 7e-5 of the weights wherever
a gradient entry is near zero (Adam's first steps are lr * sign-like).  On one
thread two runs in this container write bit-identical files (checked for every
array of every fixture); across machines or torch builds only the tolerance
the tests state holds.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/gae_dgl/gae.py"
OUT = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- DGL stand-in
class _NodeBatch:
    def __init__(self, data):
        self.data = data


class StubGraph:
    def __init__(self, n=0, src=(), dst=()):
        self.n = int(n)
        self.src = torch.as_tensor(np.asarray(src, dtype=np.int64))
        self.dst = torch.as_tensor(np.asarray(dst, dtype=np.int64))
        self.ndata = {}

    def number_of_nodes(self):
        return self.n

    def update_all(self, msg, red):
        assert msg == ("copy_src", "h", "m") and red == ("sum", "m", "h")
        h = self.ndata["h"]
        out = torch.zeros(self.n, h.shape[1], dtype=h.dtype)
        out = out.index_add(0, self.dst, h[self.src])
        self.ndata["h"] = out

    def apply_nodes(self, func):
        self.ndata.update(func(_NodeBatch(self.ndata)))

    def in_degrees(self):
        return torch.bincount(self.dst, minlength=self.n)

    def adjacency_matrix(self):
        idx = torch.stack([self.dst, self.src])
        return torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), (self.n, self.n))


def stub_batch(graphs):
    off, srcs, dsts, hs = 0, [], [], []
    for g in graphs:
        srcs.append(g.src + off)
        dsts.append(g.dst + off)
        hs.append(g.ndata["h"])
        off += g.n
    bg = StubGraph(off, torch.cat(srcs).numpy(), torch.cat(dsts).numpy())
    bg.ndata["h"] = torch.cat(hs, 0)
    return bg


def _install_stub():
    dgl = types.ModuleType("dgl")
    fn = types.ModuleType("dgl.function")
    fn.copy_src = lambda src, out: ("copy_src", src, out)
    fn.sum = lambda msg, out: ("sum", msg, out)
    nn_mod = types.ModuleType("dgl.nn")
    nnpt = types.ModuleType("dgl.nn.pytorch")
    nnpt.GraphConv = object
    dgl.function, dgl.nn, nn_mod.pytorch = fn, nn_mod, nnpt
    dgl.batch = stub_batch
    sys.modules.update({"dgl": dgl, "dgl.function": fn, "dgl.nn": nn_mod, "dgl.nn.pytorch": nnpt})


def load_reference():
    _install_stub()
    spec = importlib.util.spec_from_file_location("ref_gae", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------- cases
def mol_like(rng, n):
    """random spanning tree (max degree 4) + ~1.7 ring bonds, both directions"""
    deg = np.zeros(n, int)
    bonds = []
    for v in range(1, n):
        cand = [u for u in range(v) if deg[u] < 4]
        u = int(rng.choice(cand))
        bonds.append((u, v)); deg[u] += 1; deg[v] += 1
    for _ in range(2):
        u, v = rng.integers(0, n, 2)
        if u != v and deg[u] < 4 and deg[v] < 4 and (min(u, v), max(u, v)) not in bonds:
            bonds.append((int(min(u, v)), int(max(u, v)))); deg[u] += 1; deg[v] += 1
    src, dst = [], []
    for a, b in bonds:
        src += [a, b]; dst += [b, a]
    X = np.zeros((n, 39), np.float32)
    for i, (lo, w) in enumerate([(0, 23), (23, 6), (29, 5), (34, 4)]):
        X[np.arange(n), lo + rng.integers(0, w, n)] = 1
    X[:, 38] = rng.random(n) < 0.3
    return n, np.array(src, np.int64), np.array(dst, np.int64), X


def build_cases():
    rng = np.random.default_rng(0)
    cases = {}
    # (1) tiny hand graph: node 5 has zero in-degree, edge 0->1 duplicated,
    # self-loop at 2, directed edge 3->4 with no reverse.
    src = np.array([0, 0, 1, 2, 2, 3, 5, 4, 1], np.int64)
    dst = np.array([1, 1, 0, 2, 3, 4, 0, 2, 3], np.int64)
    cases["tiny"] = dict(n=6, src=src, dst=dst,
                         X=rng.standard_normal((6, 5)).astype(np.float32), hidden=[4, 3])
    # (2) random symmetric graph, ZINC widths
    n = 200
    a = rng.integers(0, n, 450); b = rng.integers(0, n, 450)
    keep = a != b
    a, b = a[keep], b[keep]
    cases["sym200"] = dict(n=n, src=np.concatenate([a, b]), dst=np.concatenate([b, a]),
                           X=(rng.random((n, 39)) < 0.12).astype(np.float32), hidden=[32, 16])
    # (4) depth rule: 1 and 3 layers on a directed random graph
    n = 50
    cases["deep3"] = dict(n=n, src=rng.integers(0, n, 180), dst=rng.integers(0, n, 180),
                          X=rng.standard_normal((n, 7)).astype(np.float32), hidden=[12, 9, 5])
    cases["single"] = dict(n=n, src=rng.integers(0, n, 180), dst=rng.integers(0, n, 180),
                           X=rng.standard_normal((n, 7)).astype(np.float32), hidden=[6])
    # (3) batch of 8 molecule-like graphs
    mols = [mol_like(rng, int(k)) for k in rng.integers(6, 30, 8)]
    return cases, mols


def build_wide_cases():
    """round 4: inputs WIDE enough (f_in >= 193, f_out <= 32) for the layer order the build now runs by default on such
    layers, act(A (H W^T) + b) -- pinned here by the reference's own act((A H) W^T + b) (gae.py:26-31).  Drawn from a
    generator of their own AFTER everything above, so the earlier fixtures regenerate bit-identically."""
    rng = np.random.default_rng(4)
    cases = {}
    # symmetric graph, bag-of-words-like rows (row-normalised, ~3 % non-zeros), citation widths [32, 16]
    n, f = 384, 300
    a = rng.integers(0, n, 900); b = rng.integers(0, n, 900)
    keep = a != b
    a, b = a[keep], b[keep]
    X = np.where(rng.random((n, f)) < 0.03, rng.random((n, f)) + 0.05, 0.0).astype(np.float32)
    X /= np.maximum(X.sum(1, keepdims=True), 1e-6)
    cases["wide300"] = dict(n=n, src=np.concatenate([a, b]), dst=np.concatenate([b, a]), X=X.astype(np.float32),
                            hidden=[32, 16])
    # directed multigraph (duplicate edges, self-loops, zero-in-degree nodes), signed dense-ish rows, f_in ~ 2000
    n, f = 256, 2003
    src = rng.integers(0, n, 1400); dst = rng.integers(0, n - 16, 1400)      # the last 16 nodes receive nothing
    src[:40] = src[40:80]; dst[:40] = dst[40:80]                              # 40 duplicated edges
    src[80:90] = dst[80:90]                                                   # 10 self-loops
    X = np.where(rng.random((n, f)) < 0.02, rng.standard_normal((n, f)), 0.0).astype(np.float32)
    cases["wide2k"] = dict(n=n, src=src.astype(np.int64), dst=dst.astype(np.int64), X=X, hidden=[32, 16])
    return cases


def run_case(ref, name, n, src, dst, X, hidden, seed):
    torch.manual_seed(seed)
    model = ref.GAE(X.shape[1], hidden)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def fresh():
        g = StubGraph(n, src, dst)
        g.ndata["h"] = torch.from_numpy(X).clone()
        return g

    out = dict(n=np.int64(n), src=np.asarray(src, np.int64), dst=np.asarray(dst, np.int64), X=X,
               hidden=np.asarray(hidden, np.int64))
    for k, v in sd0.items():
        out["sd/" + k] = v.numpy()
    if name == "mol8":
        # a checkpoint FILE exactly as the reference writes it (train_inductive.py:55-57:
        # torch.save(self.model.state_dict(), save_dir/ep{epoch:02}.pkl)) -- the tensors of "sd/*" above
        torch.save(model.state_dict(), os.path.join(OUT, "mol8_ep00.pkl"))
    # encode (gae.py:57-61) -- also records the ndata side effect (A9)
    g = fresh()
    Z = model.encode(g)
    out["Z"] = Z.detach().numpy()
    out["encode_leaves_h"] = np.bool_("h" in g.ndata)
    # forward at dropout 0
    model.decoder.dropout = 0.0
    g = fresh()
    logits0 = model(g)
    out["logits_p0"] = logits0.detach().numpy()
    out["forward_ndata_h"] = g.ndata["h"].detach().numpy()
    # label, pos_weight, loss, grads at dropout 0 (train_inductive.py:44-51)
    adj = fresh().adjacency_matrix().to_dense()
    pw = (adj.shape[0] * adj.shape[0] - adj.sum()) / adj.sum()
    loss = F.binary_cross_entropy_with_logits(logits0, adj, pos_weight=pw)
    model.zero_grad()
    loss.backward()
    out["adj"] = adj.numpy(); out["pos_weight"] = pw.numpy(); out["loss_p0"] = loss.detach().numpy()
    for k, p in model.named_parameters():
        out["grad_p0/" + k] = p.grad.detach().numpy().copy()
    # the hyper-parameter search's criterion at dropout 0 (optuna_gae.py:16,21: nn.MSELoss()(model.forward(g), adj))
    lm = torch.nn.MSELoss()(model(fresh()), adj)
    model.zero_grad(); lm.backward()
    out["mse_p0"] = lm.detach().numpy()
    for k, p in model.named_parameters():
        out["grad_mse_p0/" + k] = p.grad.detach().numpy().copy()
    # forward with the reference's always-on dropout p=0.1 (gae.py:47,64,70):
    # replay the RNG to capture the mask the reference drew.
    model.decoder.dropout = 0.1
    model.eval()  # gae.py:70 ignores eval(): dropout must stay on (A8)
    torch.manual_seed(seed + 1000)
    mask = F.dropout(torch.ones_like(Z), 0.1)
    torch.manual_seed(seed + 1000)
    logits1 = model(fresh())
    out["mask"] = mask.numpy(); out["logits_p01"] = logits1.detach().numpy()
    loss1 = F.binary_cross_entropy_with_logits(logits1, adj, pos_weight=pw)
    model.zero_grad(); loss1.backward()
    out["loss_p01"] = loss1.detach().numpy()
    for k, p in model.named_parameters():
        out["grad_p01/" + k] = p.grad.detach().numpy().copy()
    # norm vector of train_transductive.py:55-58
    degs = fresh().in_degrees().float()
    norm = torch.pow(degs, -0.5); norm[torch.isinf(norm)] = 0
    out["in_degrees"] = fresh().in_degrees().numpy(); out["norm"] = norm.unsqueeze(1).numpy()
    # 3 Adam steps at dropout 0 (train_inductive.py:40,50-52; lr of train_transductive.py:43)
    model.load_state_dict(sd0); model.decoder.dropout = 0.0; model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    losses = []
    for _ in range(3):
        lg = model(fresh())
        ls = F.binary_cross_entropy_with_logits(lg, adj, pos_weight=pw)
        opt.zero_grad(); ls.backward(); opt.step()
        losses.append(ls.item())
    out["adam3_losses"] = np.asarray(losses, np.float64)
    for k, v in model.state_dict().items():
        out["sd_after3/" + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in list(out.items())[:6]})


def main():
    torch.set_num_threads(1)          # see "Reproducibility" above
    ref = load_reference()
    assert ref.gcn_msg == ("copy_src", "h", "m") and ref.gcn_reduce == ("sum", "m", "h")
    cases, mols = build_cases()
    for i, (name, c) in enumerate(cases.items()):
        run_case(ref, name, c["n"], c["src"], c["dst"], c["X"], c["hidden"], seed=10 + i)
    # batched molecules: per-graph inputs + the stub's dgl.batch result
    graphs = []
    for (n, s, d, X) in mols:
        g = StubGraph(n, s, d); g.ndata["h"] = torch.from_numpy(X); graphs.append(g)
    bg = stub_batch(graphs)
    per = {}
    for i, (n, s, d, X) in enumerate(mols):
        per[f"g{i}/n"] = np.int64(n); per[f"g{i}/src"] = s; per[f"g{i}/dst"] = d; per[f"g{i}/X"] = X
    np.savez_compressed(os.path.join(OUT, "mol8_parts.npz"), n_graphs=np.int64(len(mols)), **per)
    run_case(ref, "mol8", bg.n, bg.src.numpy(), bg.dst.numpy(), bg.ndata["h"].numpy(), [32, 16], seed=99)
    for i, (name, c) in enumerate(build_wide_cases().items()):
        run_case(ref, name, c["n"], c["src"], c["dst"], c["X"], c["hidden"], seed=200 + i)


if __name__ == "__main__":
    main()
