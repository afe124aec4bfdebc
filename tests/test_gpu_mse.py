"""The hyper-parameter search's criterion (gae_dgl/optuna_gae.py:16,21): ``nn.MSELoss()(model.forward(g),
g.adjacency_matrix().to_dense())``.

ops.decoder_mse evaluates it without the N x N logits / label (oracle/gae_oracle.py: mse_closed_form) from launches the
library already has: gae_spmm_csr on A and A^T, gae_linear_bwd for the Gram matrix Zt^T Zt, gae_linear_fwd for Zt G.
Here: against the values and parameter gradients the reference's own model produced (tests/golden), against the fp64
oracle on directed multigraphs (repeated edges count twice in the label, self-loops, hubs) with a dropout mask, and
against the reference-shaped chain on the device (N x N logits through gae_decoder_dense + torch's MSELoss)."""
import numpy as np
import pytest
import torch

from conftest import golden_params  # noqa: F401  (the `golden` fixture comes from conftest)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def O():
    from oracle import gae_oracle
    return gae_oracle


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(np.asarray(b)).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1.0))


def build_model(g):
    import gae_dgl_amd as G
    model = G.GAE(g["X"].shape[1], [int(h) for h in g["hidden"]])
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")})
    return model.to(DEV)


def fresh_graph(g):
    import gae_dgl_amd as G
    gr = G.DGLGraph()
    gr.add_nodes(int(g["n"]))
    gr.add_edges(g["src"], g["dst"])
    gr.to(DEV)
    gr.ndata['h'] = torch.from_numpy(g["X"]).to(DEV)
    return gr


def test_mse_loss_and_parameter_gradients_match_the_reference_model(golden):
    g = golden
    model = build_model(g)
    model.decoder.dropout = 0.0
    gr = fresh_graph(g)
    loss = model.reconstruction_loss(gr, criterion="mse")
    assert rel(loss, g["mse_p0"]) < TOL
    assert rel(gr.ndata['h'], g["Z"]) < TOL                   # side effect on ndata['h'] as forward() (gae.py:53)
    loss.backward()
    for k, p in model.named_parameters():
        assert rel(p.grad, g["grad_mse_p0/" + k]) < 5 * TOL, k
    # the reference-shaped chain on the device: N x N logits, dense label, torch's MSELoss
    model.zero_grad()
    gr = fresh_graph(g)
    dense = torch.nn.MSELoss()(model(gr), gr.adjacency_matrix().to_dense())
    assert rel(dense, g["mse_p0"]) < TOL
    dense.backward()
    for k, p in model.named_parameters():
        assert rel(p.grad, g["grad_mse_p0/" + k]) < 5 * TOL, k
    with pytest.raises(ValueError):
        model.reconstruction_loss(fresh_graph(g), criterion="l1")


@pytest.mark.parametrize("n,d,e,hub", [(1, 1, 1, False), (17, 3, 40, False), (130, 16, 900, True), (700, 16, 3000, True),
                                       (257, 32, 2000, False), (300, 48, 1000, True), (129, 64, 77, False),
                                       (2000, 100, 9000, True), (5000, 16, 30000, True)])
def test_decoder_mse_vs_oracle_random(n, d, e, hub):
    """directed multigraphs: repeated edges (label 2, 3), self-loops, one hub row and one hub column, a dropout mask;
    any embedding width (no d <= 64 limit: nothing is N x N here)"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(n + d)
    src = rng.integers(0, n, e); dst = rng.integers(0, n, e)
    if hub:
        dst[: e // 3] = 1; src[e // 3: e // 2] = 2
    if e > 8:
        src[:4] = src[4:8]; dst[:4] = dst[4:8]                  # repeated edges
        src[8] = src[4]; dst[8] = dst[4]                        # ... one of them three times
    Z = (rng.standard_normal((n, d)) * 0.7).astype(np.float32)
    mask = ((rng.random((n, d)) >= 0.1) / 0.9).astype(np.float32)
    adj = O().dense_adjacency(src, dst, n, dtype=torch.float64)
    for m in (mask, None):
        Zt = torch.tensor(Z, dtype=torch.float64, requires_grad=True)
        ref = O().mse_mean(O().decoder_logits(Zt, None if m is None else torch.tensor(m, dtype=torch.float64)), adj)
        ref.backward()
        gr = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
        assert abs(float(gr.adjacency_sq_sum()) - float((adj ** 2).sum())) < 1e-9
        Zd = torch.from_numpy(Z).to(DEV).requires_grad_(True)
        loss = ops.decoder_mse(Zd, None if m is None else torch.from_numpy(m).to(DEV), gr)
        assert abs(float(loss) - float(ref)) < 2e-5 * max(abs(float(ref)), 1e-30)
        (3.0 * loss).backward()
        assert rel(Zd.grad, 3.0 * Zt.grad) < 5 * TOL
        with torch.no_grad():
            again = ops.decoder_mse(torch.from_numpy(Z).to(DEV), None if m is None else torch.from_numpy(m).to(DEV), gr)
        assert float(again) == float(loss)


def test_mse_training_reduces_the_loss_and_draws_a_fresh_mask():
    """a few Adam steps with the criterion on a Cora-profile graph with hubs; always-on dropout (gae.py:70): two
    evaluations differ, an injected mask reproduces"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    n, src, dst, X = W.citation_graph("cora", seed=0, degrees="planetoid")
    torch.manual_seed(0)
    model = G.GAE(X.shape[1], [32, 16]).to(DEV)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    Xd = ops.pad_rows(torch.from_numpy(X).to(DEV))
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    losses = []
    for _ in range(30):
        g.ndata['h'] = Xd
        loss = model.reconstruction_loss(g, criterion="mse")
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    with torch.no_grad():
        g.ndata['h'] = Xd; a = float(model.reconstruction_loss(g, criterion="mse"))
        g.ndata['h'] = Xd; b = float(model.reconstruction_loss(g, criterion="mse"))
        assert a != b
        model.decoder.mask = model.decoder.last_mask.clone()
        g.ndata['h'] = Xd; c = float(model.reconstruction_loss(g, criterion="mse"))
        g.ndata['h'] = Xd; d = float(model.reconstruction_loss(g, criterion="mse"))
        assert c == d
    # against the oracle's step at these weights (fp64, injected mask)
    Ws = [l.apply_mod.linear.weight.detach().cpu().numpy() for l in model.layers]
    bs = [l.apply_mod.linear.bias.detach().cpu().numpy() for l in model.layers]
    ref = O().gae_loss_and_grads(src, dst, n, X.astype(np.float64), Ws, bs, mask=model.decoder.mask.double().cpu(),
                                 criterion="mse")[0]
    assert abs(c - float(ref)) < 2e-5 * abs(float(ref))
