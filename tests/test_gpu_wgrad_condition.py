"""Weight-gradient products on ILL-CONDITIONED operands, as tests/test_gpu_r04.py does for the loss.

The reference evaluates dW = dY^T M in fp32 (train_inductive.py:50-52 through ATen).  The library's weight-gradient
kernels run on the matrix cores: exact fp32 MFMAs (gae_xw_wgrad, gae_spx_wgrad) or split bf16 pieces (knob atb_bf16:
gae_linear_bwd's atb_bf16_kernel, gae_gcn2_bwd_dense).  Benign data hides what split operands cost; here

  * `cancel`: both operands carry a +-30 column over 0.3-sigma noise with signs arranged so that the 900-sized terms
    cancel exactly -- the gradient's scale is the noise's, every term is 10^3 x larger;
  * `long`:   10^6 rows of positive data (the sum grows to 2.5e5: accumulation order matters);
  * `cora`:   the gradients of a Cora-shaped model after 200 Adam steps, through the whole backward pass;

each against fp64, at 1e-5 of the gradient's largest entry, with atb_bf16 at its default AND at 0 (exact fp32
products): the default may not be worse than 1e-5 anywhere the exact form is not."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-300))


@pytest.fixture
def knob():
    from gae_dgl_amd import _lib
    touched = {}

    def set_(name, v):
        import ctypes
        if name not in touched:
            old = ctypes.c_int64(0)
            _lib.call("gae_tuning_get", name.encode(), ctypes.byref(old))
            touched[name] = int(old.value)
        _lib.call("gae_tuning_set", name.encode(), int(v))
    yield set_
    for k, v in touched.items():
        _lib.call("gae_tuning_set", k.encode(), v)


def operands(kind, n, f_in, f_out, seed):
    """(dY [n, f_out], M [n, f_in]) fp32 on the host"""
    g = torch.Generator().manual_seed(seed)
    if kind == "cancel":
        dY = 0.3 * torch.randn(n, f_out, generator=g)
        M = 0.3 * torch.randn(n, f_in, generator=g)
        # four equal groups of rows carry the sign pairs (+,+) (+,-) (-,+) (-,-): sum_i s_i s'_i = 0 exactly, so the
        # 900-sized terms of dW[0][0] cancel; the second +-30 column pair cancels against a constant column
        grp = torch.arange(n) % 4
        s = torch.where(grp < 2, 1.0, -1.0); s2 = torch.where(grp % 2 == 0, 1.0, -1.0)
        perm = torch.randperm(n, generator=g)
        s, s2 = s[perm], s2[perm]
        dY[:, 0] += 30.0 * s
        M[:, 0] += 30.0 * s2
        M[:, min(1, f_in - 1)] += 30.0
        return dY, M
    if kind == "long":
        return torch.rand(n, f_out, generator=g), torch.rand(n, f_in, generator=g)
    raise ValueError(kind)


CASES = [("cancel", 20000), ("cancel", 200000), ("long", 1000000)]


def bound(kind, n):
    """what exact fp32 products with the kernels' blocked fp32 accumulation reach on these operands (measured 0.3 .. 1.8e-5
    at 200000 cancelling rows: the condition number grows like sqrt(n))"""
    return TOL * max(1.0, (n / 20000) ** 0.5) if kind == "cancel" else TOL


def check(errs, kind, n, what):
    """errs: {atb knob: error against fp64}.  Exact fp32 products (0) stay inside the bound; the default (1, three
    bf16 pieces) is fp32-grade: inside the bound as well and at most twice the exact form's error (or below 2e-6)"""
    assert errs[0] < bound(kind, n), (what, kind, n, errs)
    assert errs[1] < bound(kind, n) and errs[1] <= max(2.0 * errs[0], 2e-6), (what, kind, n, errs)


@pytest.mark.parametrize("kind,n", CASES)
@pytest.mark.parametrize("f_in,f_out", [(64, 32), (500, 32), (39, 32)])
def test_linear_bwd_wgrad_on_ill_conditioned_operands(kind, n, f_in, f_out, knob):
    """gae_linear_bwd (dense.hip: atb_bf16_kernel / atb_partial_kernel): dW = dY^T M, db = colsum(dY)"""
    from gae_dgl_amd import ops
    if kind == "long" and f_in == 500:
        n = 200000
    dY, M = operands(kind, n, f_in, f_out, seed=n + f_in)
    ref = dY.double().t() @ M.double()
    errs = {}
    for atb in (1, 0, 2):
        knob("atb_bf16", atb)
        dW, db, _ = ops.linear_bwd_raw(dY.to(DEV), None, 0, ops.pad_rows(M.to(DEV)), None, need_dM=False, f_out=f_out)
        errs[atb] = rel(dW, ref)
        assert rel(db, dY.double().sum(0)) < bound(kind, n)
    check(errs, kind, n, f"linear_bwd {f_in}")
    if kind == "cancel":
        assert errs[2] > 3 * errs[1], errs            # (the two-piece form is what this test exists for)


@pytest.mark.parametrize("kind,n", CASES)
def test_gcn2_bwd_dense_on_ill_conditioned_operands(kind, n, knob):
    """gae_gcn2_bwd_dense (tall.hip): dW2 = G^T H1, dW1 = ((G W2) (.) relu')^T M1, with H1 recomputed from M1"""
    from gae_dgl_amd import ops
    G, M1 = operands(kind, n, 32, 16, seed=n + 5)
    g = torch.Generator().manual_seed(n)
    dZ = torch.randn(n, 16, generator=g)
    W1 = torch.randn(32, 32, generator=g) / 32 ** 0.5; b1 = torch.randn(32, generator=g)
    W2 = torch.randn(16, 32, generator=g) / 32 ** 0.5
    d = lambda t: t.to(DEV)
    # (a handful of pre-activations lie within fp32 rounding of zero and may gate differently than in fp64: one term of
    #  10^4 .. 10^6 per sum, far below the bounds)
    H1 = torch.relu(M1.double() @ W1.double().t() + b1.double())
    dY1 = (G.double() @ W2.double()) * (H1 > 0)
    want = {"dW2": G.double().t() @ H1, "dW1": dY1.t() @ M1.double()}
    errs = {"dW1": {}, "dW2": {}}
    for atb in (1, 0):
        knob("atb_bf16", atb)
        dW1, db1, dW2, db2 = ops.gcn2_bwd_dense_raw(d(G), d(dZ), None, 1, d(M1), d(W2), W1=d(W1), b1=d(b1))
        errs["dW1"][atb] = rel(dW1, want["dW1"]); errs["dW2"][atb] = rel(dW2, want["dW2"])
        assert rel(db1, dY1.sum(0)) < bound(kind, n) and rel(db2, dZ.double().sum(0)) < TOL
    check(errs["dW2"], kind, n, "gcn2 dW2")
    check(errs["dW1"], kind, n, "gcn2 dW1")


@pytest.mark.parametrize("kind,n", [("cancel", 20000), ("cancel", 200000), ("long", 200000)])
@pytest.mark.parametrize("f_in", [500, 1433])
def test_xw_wgrad_and_spx_wgrad_on_ill_conditioned_operands(kind, n, f_in):
    """gae_xw_wgrad (exact fp32 MFMAs, one pass over X) and gae_spx_wgrad (the same product from the non-zeros of X);
    gae_xw_fwd (knob xw_p3: three bf16 pieces, six pairs -- the default -- against the exact fp32 MFMAs)"""
    import gae_dgl_amd as Gm
    from gae_dgl_amd import _lib, ops
    G, X = operands(kind, n, f_in, 32, seed=n + f_in + 1)
    gen = torch.Generator().manual_seed(3)
    X = X * (torch.rand(n, f_in, generator=gen) < 0.05)               # bag-of-words sparsity; the +-30 column thinned alike
    ref = G.double().t() @ X.double()
    Xd = ops.pad_rows(X.to(DEV))
    dW, db = ops.xw_wgrad_raw(Xd, G.to(DEV), None, G.to(DEV), None, 32)
    assert rel(dW, ref) < bound(kind, n), (kind, n, f_in, rel(dW, ref))
    assert rel(db, G.double().sum(0)) < bound(kind, n)
    sf = Gm.SparseFeatures.from_dense(Xd)
    dWs, dbs = ops.spx_wgrad_raw(sf, G.to(DEV), G.to(DEV), None, 32)
    assert rel(dWs, ref) < bound(kind, n), (kind, n, f_in, "spx", rel(dWs, ref))
    assert rel(dbs, G.double().sum(0)) < bound(kind, n)
    # forward on the same X: rows of W with a +-30 pair that cancels against X's constant column
    Wt = 0.3 * torch.randn(32, f_in, generator=gen)
    Wt[:, 0] += 30.0; Wt[:, 1] -= 30.0
    reff = X.double() @ Wt.double().t()
    errs = {}
    for p3 in (1, 0):
        _lib.call("gae_tuning_set", b"xw_p3", p3)
        try:
            errs[p3] = rel(ops.xw_fwd_raw(Xd, Wt.to(DEV), None, 0), reff)
        finally:
            _lib.call("gae_tuning_set", b"xw_p3", 1)
    assert errs[0] < TOL and errs[1] <= max(2.0 * errs[0], 2e-6), errs


@pytest.mark.timeout(600)
@pytest.mark.parametrize("layer1", ["transform-first", "reference"])
def test_cora_model_after_200_steps_gradients_match_fp64(layer1, knob):
    """the whole backward pass at the weights 200 Adam steps leave (rows of Z have grown, the loss gradient is peaked on
    the edges): every parameter gradient against the fp64 oracle at 1e-5 of its scale, default knobs and exact fp32"""
    import gae_dgl_amd as G
    from gae_dgl_amd import gae as gae_mod, ops, optim, workloads as W
    from oracle import gae_oracle as O
    n, src, dst, X = W.citation_graph("cora", seed=0)
    torch.manual_seed(0)
    model = G.GAE(X.shape[1], [32, 16]).to(DEV)
    model.decoder.dropout = 0.0
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    Xd = ops.pad_rows(torch.from_numpy(X).to(DEV))
    opt = optim.Adam(model.parameters(), lr=1e-2)
    old = gae_mod.TRANSFORM_FIRST_AUTO
    gae_mod.TRANSFORM_FIRST_AUTO = layer1 == "transform-first"
    try:
        for _ in range(200):
            g.ndata['h'] = Xd
            loss = model.reconstruction_loss(g)
            opt.zero_grad(); ops.backward(loss); opt.step()
        Ws = [l.apply_mod.linear.weight.detach().cpu().numpy().copy() for l in model.layers]
        bs = [l.apply_mod.linear.bias.detach().cpu().numpy().copy() for l in model.layers]
        ref_loss, _, _, dW, db = O.gae_loss_and_grads(src, dst, n, X.astype(np.float64), Ws, bs)
        errs = {}
        for atb in (1, 0):
            knob("atb_bf16", atb)
            knob("xw_p3", atb)
            model.zero_grad()
            g.ndata['h'] = Xd
            loss = model.reconstruction_loss(g)
            loss.backward()
            assert abs(float(loss) - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
            for k, (l, w, b) in enumerate(zip(model.layers, dW, db)):
                errs[(atb, k, "W")] = rel(l.apply_mod.linear.weight.grad, w)
                errs[(atb, k, "b")] = rel(l.apply_mod.linear.bias.grad, b)
        assert max(errs.values()) < TOL, errs
    finally:
        gae_mod.TRANSFORM_FIRST_AUTO = old
