"""Round-4 regression tests for the ADVICE findings of round 3 (GPU side)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sym_graph(rng, n, e):
    import gae_dgl_amd as G
    a = rng.integers(0, n, e); b = rng.integers(0, n, e)
    g = G.DGLGraph((np.concatenate([a, b]), np.concatenate([b, a])), num_nodes=n).to(DEV)
    return g, np.concatenate([a, b]).astype(np.int64), np.concatenate([b, a]).astype(np.int64)


@pytest.mark.parametrize("hidden,F_in", [((64, 5), 20), ((64, 16), 40), ((32, 6), 24), ((48, 8), 33)])
def test_vgae_heads_with_shapes_the_fused_backward_cannot_run(hidden, F_in):
    """the packed-heads launch was taken on forward shapes alone; its backward needs the heads'
    input width <= 32 and 2 d a multiple of 4.  Such models must train (through the two separate layers) and agree
    with the oracle's VGAE restatement."""
    from gae_dgl_amd.vgae import VGAE
    from oracle import gae_oracle as O
    rng = np.random.default_rng(hidden[0] + hidden[1])
    n = 300
    g, src, dst = sym_graph(rng, n, 700)
    X = rng.standard_normal((n, F_in)).astype(np.float32)
    torch.manual_seed(1)
    model = VGAE(F_in, hidden).to(DEV)
    eps = torch.from_numpy(rng.standard_normal((n, hidden[1])).astype(np.float32)).to(DEV)
    model.eps = eps
    g.ndata['h'] = torch.from_numpy(X).to(DEV)
    loss = model.loss(g)
    loss.backward()                                          # raised GaeHipError before the guard
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    assert all(torch.isfinite(v).all() for v in grads.values())
    # oracle (Kipf & Welling restatement; unpinned by construction) on the same eps
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    ip, ix = O.csr_from_coo(src, dst, n)
    mu, ls, z = O.vgae_forward(ip, ix, torch.from_numpy(X), P["shared.apply_mod.linear.weight"],
                               P["shared.apply_mod.linear.bias"], P["mu_head.apply_mod.linear.weight"],
                               P["mu_head.apply_mod.linear.bias"], P["logstd_head.apply_mod.linear.weight"],
                               P["logstd_head.apply_mod.linear.bias"], eps.cpu())
    adj = O.dense_adjacency(src, dst, n)
    ref = O.bce_with_logits_mean(z @ z.t(), adj, O.pos_weight_of(adj)) + O.vgae_kl(mu, ls)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
    for k in grads:
        r = P[k].grad.double()
        assert float((grads[k].double() - r).abs().max() / r.abs().max().clamp(min=1e-12)) < 2e-4, k


def test_adam_checkpoint_loaded_after_capture_matches_eager():
    """load_state_dict() on an optimiser whose step is already captured must land in the tensors
    the graph holds (moments AND step counters): replays after the load == eager steps from the same checkpoint"""
    import gae_dgl_amd as G
    from gae_dgl_amd import capture, optim
    rng = np.random.default_rng(3)
    n, F_in = 500, 24
    g, _, _ = sym_graph(rng, n, 1500)
    X = torch.from_numpy(rng.standard_normal((n, F_in)).astype(np.float32)).to(DEV)

    def make(seed=5):
        torch.manual_seed(seed)
        m = G.GAE(F_in, [32, 16]).to(DEV)
        m.decoder.dropout = 0.0
        return m

    # checkpoint after 4 eager steps
    m0 = make(); o0 = optim.Adam(m0.parameters(), lr=1e-2)
    for _ in range(4):
        g.ndata['h'] = X
        loss = m0.reconstruction_loss(g); o0.zero_grad(); loss.backward(); o0.step()
    ck_model = {k: v.clone() for k, v in m0.state_dict().items()}
    ck_opt = copy.deepcopy(o0.state_dict())          # (state_dict() hands out the live moment tensors)
    # eager continuation: 3 more steps
    want = []
    for _ in range(3):
        g.ndata['h'] = X
        loss = m0.reconstruction_loss(g); o0.zero_grad(); loss.backward(); o0.step()
        want.append(float(loss.detach()))
    # a captured step that has already run on another trajectory, then loads the checkpoint
    m1 = make(seed=9); o1 = optim.Adam(m1.parameters(), lr=1e-2)
    step = capture.CapturedTrainStep(m1, o1, g, X, warmup=2)
    step(); step()
    with torch.no_grad():
        for k, v in m1.state_dict().items():
            v.copy_(ck_model[k])
    o1.load_state_dict(ck_opt)
    got = [float(step()) for _ in range(3)]
    np.testing.assert_allclose(got, want, rtol=2e-5)
    for k, v in m1.state_dict().items():
        ref = m0.state_dict()[k]
        assert float((v - ref).abs().max() / ref.abs().max().clamp(min=1e-6)) < 1e-4, k
    assert o1.steps_taken() == 7


def _loss_and_grad(Z, g):
    from gae_dgl_amd import ops
    Zd = torch.tensor(Z, device=DEV).requires_grad_(True)
    loss = ops.decoder_bce(Zd, None, g)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), Zd.grad.detach().cpu()


@pytest.mark.parametrize("bal", [2, 0])
def test_loss_fp16_pieces_match_fp64_and_fall_back_beyond_their_range(bal, tuning):
    """round 4: where the symmetric dense kernel runs, the products of the fused loss use two fp16 pieces per operand
    (22 mantissa bits).  (a) against the fp64 oracle on embeddings whose large components cancel; (b) embeddings
    outside fp16's range make the SAME call fall back to the three-piece bf16 kernel -- bit-identical to that kernel
    selected by knob; (c) the guard re-arms: the next in-range call is bit-identical to the first."""
    from gae_dgl_amd import _lib
    from oracle import gae_oracle as O
    rng = np.random.default_rng(11)
    n, d = 1500, 16
    g, src, dst = sym_graph(rng, n, 4000)
    Z = (rng.standard_normal((n, d)) * 0.3).astype(np.float32)
    sign = rng.choice([-1.0, 1.0], size=n).astype(np.float32)
    Z[:, 0] = 30.0 * sign + Z[:, 0]; Z[:, 1] = 30.0 + Z[:, 1]
    adj = O.dense_adjacency(src, dst, n, dtype=torch.float64)
    Zt = torch.tensor(Z, dtype=torch.float64, requires_grad=True)
    ref = O.bce_with_logits_mean(O.decoder_logits(Zt, None), adj, O.pos_weight_of(adj))
    ref.backward()
    gscale = float(Zt.grad.abs().max())
    Zbig = Z.copy()
    Zbig[7, 3] = 1.0e5; Zbig[n - 1, 15] = -3.0e5; Zbig[700, 0] = 7.0e4      # > 65504: no fp16 value
    _lib.call("gae_tuning_set", b"bce_sym", 2)            # the symmetric kernel from 512 rows on (default: 5120)
    tuning("bce_sym_bal", bal)                            # balanced schedule (the fallback walks its tile ranges with 64 blocks) / 2-D grid
    try:
        l1, g1 = _loss_and_grad(Z, g)
        assert abs(l1 - float(ref)) <= 1e-6 * abs(float(ref))
        err = float((g1.double() - Zt.grad).abs().max()) / gscale
        assert err <= 1e-5, err                            # measured 2e-6 (two bf16 pieces: 7e-5)
        lb, gb = _loss_and_grad(Zbig, g)                   # out of range -> the bf16 launch recomputes everything
        l2, g2 = _loss_and_grad(Z, g)                      # flag cleared by the edge kernel
        assert l2 == l1 and torch.equal(g1, g2)
        _lib.call("gae_tuning_set", b"bce_s_bf16", 2)
        lb3, gb3 = _loss_and_grad(Zbig, g)
        assert np.isfinite(lb) and lb == lb3 and torch.equal(gb, gb3)
        l3, g3 = _loss_and_grad(Z, g)
        assert not torch.equal(g1, g3)                     # (the in-range calls above did run on the fp16 pieces)
    finally:
        _lib.call("gae_tuning_set", b"bce_s_bf16", 3)
        _lib.call("gae_tuning_set", b"bce_sym", 1)


def test_two_models_interleaved_in_one_step_context():
    """the deferred partial sums / loss reductions live in the step's context object (ops.StepContext),
    not in address-keyed module tables.  Two models whose forward, backward and optimiser launches INTERLEAVE inside one
    context end exactly where each ends when it trains alone without any deferral; a context that is left with
    unconsumed partial sums raises, and nothing survives the context."""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, optim
    rng = np.random.default_rng(21)
    ga, _, _ = sym_graph(rng, 3000, 9000)
    gb, _, _ = sym_graph(rng, 1700, 6000)
    Xa = torch.from_numpy(rng.standard_normal((3000, 300)).astype(np.float32)).to(DEV)     # transform-first layer 1
    Xb = torch.from_numpy(rng.standard_normal((1700, 40)).astype(np.float32)).to(DEV)      # fused narrow layers

    def make():
        torch.manual_seed(2)
        a = G.GAE(300, [32, 16]).to(DEV); b = G.GAE(40, [32, 16]).to(DEV)
        a.decoder.dropout = b.decoder.dropout = 0.0
        return a, b, optim.Adam(a.parameters(), lr=1e-2), optim.Adam(b.parameters(), lr=1e-2)

    a0, b0, oa0, ob0 = make()                                   # each alone, nothing deferred
    for _ in range(2):
        for m, o, g, X in ((a0, oa0, ga, Xa), (b0, ob0, gb, Xb)):
            g.ndata['h'] = X
            loss = m.reconstruction_loss(g); o.zero_grad(); ops.backward(loss, list(m.parameters())); o.step()
    a1, b1, oa1, ob1 = make()
    losses = []
    for _ in range(2):
        with ops.StepContext(defer_grads=True, defer_loss=True) as step:
            ga.ndata['h'] = Xa; la = a1.reconstruction_loss(ga)
            gb.ndata['h'] = Xb; lb = b1.reconstruction_loss(gb)
            ops.backward(la, list(a1.parameters()))
            ops.backward(lb, list(b1.parameters()))
            assert len(step.partials) >= 4                      # both models' weight gradients wait in THIS context
            ob1.step()                                          # the other order than the backward passes
            oa1.step()
            assert not step.partials
        assert not ops.current_step().partials and not ops.current_step().tails
        losses.append((float(la), float(lb)))
    for m0, m1 in ((a0, a1), (b0, b1)):
        for p0, p1 in zip(m0.parameters(), m1.parameters()):
            assert torch.equal(p0, p1)
    assert all(np.isfinite(v) for pair in losses for v in pair)
    with pytest.raises(ops.GaeHipError):
        with ops.StepContext(defer_grads=True):
            gb.ndata['h'] = Xb
            ops.backward(b1.reconstruction_loss(gb), list(b1.parameters()))     # no optimiser step inside
    assert not ops.current_step().partials
