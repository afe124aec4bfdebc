"""The headline configuration end to end (round 6): the DEFAULT Pubmed training step bench.py times -- BASELINE
configs[1], 19 717 nodes x 500 features, hidden 32 / 16 -- against the oracle at full size.

The step (train_transductive.py:41-68 on gae.py:26-31,49-55) as the library runs it by default: gae_xw_fwd (P = X W1^T)
-> gae_spmm_csr_epilogue (relu(A P + b1)) -> the fused narrow layer 2 with the loss's prepare step in its epilogue ->
the symmetric fused decoder + BCE kernel (256-row panels, balanced schedule) -> ReLU-gated gather of the backward aggregation ->
gae_xw_wgrad -> Adam.  The oracle evaluates the SAME step in fp64 in the reference's order ((A X) W^T), the N x N part
1024 rows at a time (oracle.gae_loss_and_grads_windowed, pinned against the reference-generated vectors by
tests/test_oracle_golden.py): Z on all rows, the loss, every parameter gradient, then three Adam steps of the captured
HIP-graph step against the oracle's trajectory, and -- where the host has the memory for 19 717 x 19 717 fp32
temporaries -- against oracle.CpuReferenceStep, the dense restatement bench.py times as cpu_baseline."""
import functools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5          # north_star: fp32 features within 1e-5


def O():
    from oracle import gae_oracle
    return gae_oracle


def rel(a, b):
    """max |a - b| / max |b|"""
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@functools.lru_cache(maxsize=None)
def pubmed(degrees):
    from gae_dgl_amd import workloads as W
    return W.citation_graph("pubmed", seed=0, degrees=degrees)


def injected_mask(n, d, seed=5):
    """an inverted-dropout multiplier as F.dropout(z, 0.1) draws it (gae.py:70): 0 or 1 / 0.9"""
    rng = np.random.default_rng(seed)
    return ((rng.random((n, d)) >= 0.1) / 0.9).astype(np.float32)


def build(degrees, seed=0):
    """model, graph and features exactly as bench.py's CitationWorkload('pubmed') sets them up"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    n, src, dst, X = pubmed(degrees)
    torch.manual_seed(seed)
    model = G.GAE(X.shape[1], [32, 16]).to(DEV)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    Xd = ops.pad_rows(torch.from_numpy(X).to(DEV))
    feats = G.SparseFeatures.maybe_from_dense(Xd, 32, graph=g)
    assert isinstance(feats, torch.Tensor), "Pubmed's features stay dense under --features auto (the headline path)"
    return model, g, feats, (n, src, dst, X)


def params_of(model):
    Ws = [l.apply_mod.linear.weight.detach().double().cpu().numpy().copy() for l in model.layers]
    bs = [l.apply_mod.linear.bias.detach().double().cpu().numpy().copy() for l in model.layers]
    return Ws, bs


def calls_since(before):
    from gae_dgl_amd import _lib
    return {k: v - before.get(k, 0) for k, v in _lib.CALLS.items() if v != before.get(k, 0)}


def assert_default_kernels(delta):
    """the launches of the default wide-layer step: one-pass X W^T, epilogue SpMM forward and (ReLU-gated) backward,
    one-pass dW1, the symmetric loss kernel (256-row panels, balanced schedule)"""
    from gae_dgl_amd import _lib
    assert delta.get("gae_xw_fwd", 0) == 1, delta
    assert delta.get("gae_spmm_csr_epilogue", 0) == 2, delta          # relu(A P + b1); G = gate(A^T dM2)
    assert delta.get("gae_xw_wgrad", 0) + delta.get("gae_x_xw_wgrad_partials", 0) == 1, delta
    assert delta.get("gae_x_gcn_layer_fused_prep", 0) == 1, delta     # layer 2 + the loss's prepare step
    assert delta.get("gae_x_decoder_bce_prepared", 0) == 1, delta
    assert "gae_spmm_csr" not in delta and "gae_linear_fwd" not in delta, delta   # no F = 500 aggregation, no separate Linear
    assert _lib.tuning_get("bce_last_kind") == 3, "the loss did not run on the symmetric kernel (256-row panels)"


@pytest.mark.parametrize("degrees", ["uniform", "planetoid"])
@pytest.mark.parametrize("dropout", ["p0", "mask"])
def test_default_pubmed_step_matches_oracle(degrees, dropout):
    """Z (all 19 717 rows), loss, dW1 / db1 / dW2 / db2 of one default step, dropout off and with an injected mask"""
    from gae_dgl_amd import _lib, ops
    model, g, feats, (n, src, dst, X) = build(degrees)
    if degrees == "planetoid":
        assert np.bincount(dst, minlength=n).max() > 128        # the hubs of the real graph (171 there)
    mask = None
    if dropout == "p0":
        model.decoder.dropout = 0.0
    else:
        mask = injected_mask(n, 16)
        model.decoder.mask = torch.from_numpy(mask).to(DEV)
    Ws, bs = params_of(model)
    before = dict(_lib.CALLS)
    g.ndata['h'] = feats
    loss = model.reconstruction_loss(g)
    ops.backward(loss)
    torch.cuda.synchronize()
    assert_default_kernels(calls_since(before))
    Z = g.ndata['h'].detach()                        # GAE.forward leaves the embedding on the graph (gae.py:53)
    ref_loss, Zref, _, dW, db = O().gae_loss_and_grads_windowed(src, dst, n, X, Ws, bs, mask)
    assert Z.shape == (n, 16)
    assert rel(Z, Zref) < TOL
    assert abs(float(loss.detach()) - float(ref_loss)) < TOL * abs(float(ref_loss)), (float(loss.detach()), float(ref_loss))
    for k, l in enumerate(model.layers):
        assert rel(l.apply_mod.linear.weight.grad, dW[k]) < 2 * TOL, (k, "W")
        assert rel(l.apply_mod.linear.bias.grad, db[k]) < 2 * TOL, (k, "b")


def adam_fp64(params, grads, state, lr=1e-2, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam's update rule (train_transductive.py:43 defaults) on numpy fp64 arrays"""
    state["t"] = t = state.get("t", 0) + 1
    out = []
    for k, (p, g_) in enumerate(zip(params, grads)):
        g_ = np.asarray(g_, dtype=np.float64)
        m = state[("m", k)] = b1 * state.get(("m", k), 0.0) + (1 - b1) * g_
        v = state[("v", k)] = b2 * state.get(("v", k), 0.0) + (1 - b2) * g_ * g_
        out.append(p - lr * (m / (1 - b1 ** t)) / (np.sqrt(v / (1 - b2 ** t)) + eps))
    return out


def test_captured_pubmed_steps_follow_the_oracle_trajectory():
    """three replays of the captured default step (what bench.py's timed region runs), dropout off: loss of every step
    against the fp64 oracle stepping the same Adam rule from the same weights; the weights after the third step."""
    from gae_dgl_amd import _lib
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    model, g, feats, (n, src, dst, X) = build("uniform")
    model.decoder.dropout = 0.0
    Ws, bs = params_of(model)
    opt = Adam(model.parameters(), lr=1e-2)
    before = dict(_lib.CALLS)
    step = CapturedTrainStep(model, opt, g, feats, warmup=0)
    delta = calls_since(before)
    assert_default_kernels(delta)
    assert delta.get("gae_x_adam_step_tail", 0) + delta.get("gae_adam_step", 0) == 1, delta
    before = dict(_lib.CALLS)
    losses = []
    for _ in range(3):
        losses.append(float(step()))
    torch.cuda.synchronize()
    assert not calls_since(before), "a replay of the captured step launches nothing from the host"
    ref, state = [], {}
    for _ in range(3):
        l, _, _, dW, db = O().gae_loss_and_grads_windowed(src, dst, n, X, Ws, bs)
        ref.append(float(l))
        new = adam_fp64(Ws + bs, [w.numpy() for w in dW] + [b.numpy() for b in db], state)
        Ws, bs = new[:2], new[2:]
    # step 1: the same weights; later steps inherit the fp32 rounding of the update (lr g / (|g| + eps) amplifies the
    # rounding of gradient entries near zero), hence the wider band there
    assert abs(losses[0] - ref[0]) < TOL * abs(ref[0]), (losses, ref)
    for a, b in zip(losses[1:], ref[1:]):
        assert abs(a - b) < 1e-4 * abs(b), (losses, ref)
    assert ref[2] < ref[0] and losses[2] < losses[0]
    W_now, b_now = params_of(model)
    for k in range(2):
        assert rel(W_now[k], Ws[k]) < 1e-3, k          # |update| = lr per step and entry: three steps move 3e-2
        assert rel(b_now[k], bs[k]) < 1e-3, k


def test_captured_pubmed_steps_follow_cpu_reference_step():
    """the same three replays against oracle.CpuReferenceStep -- the dense N x N restatement of
    train_inductive.py:43-53 that bench.py times as cpu_baseline -- started from the same weights, dropout off"""
    psutil = pytest.importorskip("psutil")
    if psutil.virtual_memory().available < 48e9:
        pytest.skip("CpuReferenceStep holds ~10 dense 19717 x 19717 fp32 temporaries: needs ~48 GB of free host memory")
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    model, g, feats, (n, src, dst, X) = build("uniform")
    model.decoder.dropout = 0.0
    ref = O().CpuReferenceStep(src, dst, n, X, X.shape[1], [32, 16], lr=1e-2, seed=0, dropout=0.0)
    with torch.no_grad():
        for lin, layer in zip(ref.layers, model.layers):
            lin.weight.copy_(layer.apply_mod.linear.weight.detach().cpu())
            lin.bias.copy_(layer.apply_mod.linear.bias.detach().cpu())
    step = CapturedTrainStep(model, Adam(model.parameters(), lr=1e-2), g, feats, warmup=0)
    got = [float(step()) for _ in range(3)]
    want = [ref.step() for _ in range(3)]
    assert abs(got[0] - want[0]) < TOL * abs(want[0]), (got, want)
    for a, b in zip(got[1:], want[1:]):
        assert abs(a - b) < 1e-4 * abs(b), (got, want)
