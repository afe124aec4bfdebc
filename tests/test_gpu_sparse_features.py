"""Opt-in compressed input features (gae_dgl_amd.SparseFeatures, csrc/spfeat.hip): exact compression of a dense matrix,
X W^T and G^T X from the non-zeros against fp64, and the whole model on sparse input against the dense path and the
oracle's reference-order step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def rel(a, b):
    a = a.detach().double().cpu(); b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def bag_of_words(rng, n, K, nnz_per_row):
    X = np.zeros((n, K), np.float32)
    cols = rng.integers(0, K, (n, nnz_per_row))
    np.put_along_axis(X, cols, rng.random((n, nnz_per_row)).astype(np.float32) + 0.05, axis=1)
    X[n // 2] = 0.0                                     # an empty row
    X[:, K // 3] = 0.0                                  # an empty feature
    return X


@pytest.mark.parametrize("n,K,per_row", [(700, 300, 12), (19717, 500, 50), (2708, 1433, 18), (50, 64, 64), (3, 5, 1)])
def test_compression_is_exact_and_ordered(n, K, per_row):
    import gae_dgl_amd as G
    rng = np.random.default_rng(n + K)
    X = bag_of_words(rng, n, K, min(per_row, K))
    Xd = torch.from_numpy(X).to(DEV)
    sf = G.SparseFeatures.from_dense(Xd)
    assert sf.shape == (n, K) and sf.nnz == int((X != 0).sum())
    assert torch.equal(sf.to_dense(), Xd)
    rp = sf.rowptr.cpu().numpy(); col = sf.col.cpu().numpy()
    assert np.array_equal(np.diff(rp), (X != 0).sum(1))
    for i in (0, n // 2, n - 1):
        assert np.all(np.diff(col[rp[i]:rp[i + 1]]) > 0)                    # columns ascending inside a row
    trp = sf.t_rowptr.cpu().numpy()
    assert np.array_equal(np.diff(trp), (X != 0).sum(0))
    # segments: every feature at least one, entries of a segment inside its feature's row
    feat, e0, slot = sf.seg_feat.cpu().numpy(), sf.seg_e0.cpu().numpy(), sf.seg_slot.cpu().numpy()
    assert np.array_equal(np.unique(feat), np.arange(K)) and np.all(e0 >= trp[feat]) and np.all(e0 <= trp[feat + 1])
    assert sf.max_segments == slot.max() + 1


@pytest.mark.parametrize("n,K,J,per_row", [(19717, 500, 32, 50), (2708, 1433, 32, 18), (3327, 3703, 32, 47), (900, 300, 16, 9),
                                           (400, 260, 7, 30), (1000, 520, 32, 500)])
def test_spx_products_match_fp64(n, K, J, per_row, tuning):
    """gae_spx_fwd / gae_spx_wgrad against fp64 on the dense matrix: W^T in LDS (K <= 1024) and from global memory,
    features of one and of many segments, masks; the deferred (partial-list) form equals the reduced one"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(3 * n + K)
    X = bag_of_words(rng, n, K, per_row)
    Xd = torch.from_numpy(X).to(DEV)
    sf = G.SparseFeatures.from_dense(Xd)
    g = torch.Generator(device="cpu").manual_seed(n)
    W = torch.randn(J, K, generator=g) / K ** 0.5
    Gm = torch.randn(n, J, generator=g); D = torch.randn(n, J, generator=g); M = torch.randn(n, J, generator=g)
    P = ops.spx_fwd_raw(sf, W.to(DEV))
    assert rel(P, torch.from_numpy(X).double() @ W.double().t()) < TOL
    dW, db = ops.spx_wgrad_raw(sf, Gm.to(DEV), D.to(DEV), M.to(DEV), J)
    assert rel(dW, Gm.double().t() @ torch.from_numpy(X).double()) < TOL
    assert rel(db, (D.double() * (M > 0)).sum(0)) < TOL
    assert bool((dW[:, K // 3] == 0).all())                                # the empty feature
    dW2, none = ops.spx_wgrad_raw(sf, Gm.to(DEV), None, None, J)
    assert none is None and torch.equal(dW2, dW)


@pytest.mark.parametrize("hidden,K", [([32, 16], 500), ([24], 300), ([32, 20, 8], 1433)])
@pytest.mark.parametrize("norm", ["none", "both"])
def test_model_on_sparse_features_matches_dense_and_oracle(hidden, K, norm):
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from oracle import gae_oracle as O
    rng = np.random.default_rng(len(hidden) + K)
    n = 800
    a = rng.integers(0, n, 1800); b = rng.integers(0, n, 1800)
    src = np.concatenate([a, b]).astype(np.int64); dst = np.concatenate([b, a]).astype(np.int64)
    X = bag_of_words(rng, n, K, 20)
    torch.manual_seed(2)
    model = G.GAE(K, hidden, norm=norm).to(DEV)
    model.decoder.dropout = 0.0
    gr = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    sf = G.SparseFeatures.from_dense(torch.from_numpy(X).to(DEV))
    gr.ndata['h'] = sf
    calls = []
    orig = ops.spx_fwd_raw
    ops.spx_fwd_raw = lambda *aa, **kk: (calls.append(1), orig(*aa, **kk))[1]
    try:
        loss = model.reconstruction_loss(gr)
    finally:
        ops.spx_fwd_raw = orig
    assert len(calls) == 1, "the sparse-input layer did not run"
    loss.backward()
    Ws = [l.apply_mod.linear.weight.detach().cpu().double().numpy() for l in model.layers]
    bs = [l.apply_mod.linear.bias.detach().cpu().double().numpy() for l in model.layers]
    nv = None
    if norm == "both":
        deg = np.bincount(dst, minlength=n).astype(np.float64)
        nv = np.where(deg > 0, np.maximum(deg, 1) ** -0.5, 0.0)
    ref_loss, Z, _, dW, db = O.gae_loss_and_grads(src, dst, n, X.astype(np.float64), Ws, bs, norm=nv)
    assert rel(gr.ndata['h'], Z) < TOL and abs(float(loss) - float(ref_loss)) < TOL * abs(float(ref_loss))
    for l, w, b_ in zip(model.layers, dW, db):
        assert rel(l.apply_mod.linear.weight.grad, w) < 5 * TOL and rel(l.apply_mod.linear.bias.grad, b_) < 5 * TOL
    # the dense route on the same model: same embedding to fp32 rounding; encode() leaves no 'h' (gae.py:57-61)
    gr.ndata['h'] = torch.from_numpy(X).to(DEV)
    zd = model.encode(gr)
    assert 'h' not in gr.ndata
    gr.ndata['h'] = sf
    zs = model.encode(gr)
    assert 'h' not in gr.ndata and rel(zs, zd) < 2e-6


def test_captured_step_on_sparse_features_equals_eager():
    """the captured transductive step with sparse input and deferred gradient reductions == the eager steps, bit for bit"""
    import copy
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    n, src, dst, X = W.citation_graph("cora", seed=0)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    sf = G.SparseFeatures.from_dense(torch.from_numpy(X).to(DEV))
    torch.manual_seed(0)
    m_e = G.GAE(X.shape[1], [32, 16]).to(DEV)
    m_e.decoder.seed = 3
    m_c = copy.deepcopy(m_e)
    o_e, o_c = Adam(m_e.parameters(), lr=1e-2), Adam(m_c.parameters(), lr=1e-2)
    losses_e = []
    for _ in range(6):
        g.ndata['h'] = sf
        loss = m_e.reconstruction_loss(g)
        o_e.zero_grad(set_to_none=True); ops.backward(loss, list(m_e.parameters())); o_e.step()
        losses_e.append(float(loss))
    step = CapturedTrainStep(m_c, o_c, g, sf, warmup=2)
    losses_c = [float(step()) for _ in range(4)]
    assert losses_c == losses_e[2:]
    for a, b in zip(m_e.parameters(), m_c.parameters()):
        assert torch.equal(a, b)


def test_maybe_from_dense_compresses_only_where_it_pays():
    """round 4: SparseFeatures.maybe_from_dense -- wide and very sparse features (Citeseer, Cora shapes) come back
    compressed, Pubmed's 500 columns at 10 % and narrow or dense matrices come back as they were"""
    import gae_dgl_amd as G
    from gae_dgl_amd import workloads as W
    dev = "cuda:0"
    for name, want in (("citeseer", True), ("cora", True), ("pubmed", False)):
        _, _, _, X = W.citation_graph(name, seed=0)
        Xd = torch.from_numpy(X).to(dev)
        out = G.SparseFeatures.maybe_from_dense(Xd)
        assert isinstance(out, G.SparseFeatures) == want, name
        if want:
            assert torch.equal(out.to_dense(), Xd)
        else:
            assert out is Xd
    narrow = torch.zeros(100, 64, device=dev)
    assert G.SparseFeatures.maybe_from_dense(narrow) is narrow
    dense = torch.rand(300, 400, device=dev)
    assert G.SparseFeatures.maybe_from_dense(dense) is dense
