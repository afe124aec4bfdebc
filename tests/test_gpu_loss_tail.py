"""The deferred final reduction of the fused loss (gae_x_decoder_bce_defer_finalize / gae_x_decoder_bce_finalize /
gae_x_adam_step_tail): the same bits as the loss's own last launch, whoever runs it."""
import copy
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, seed=0, deg=4):
    import gae_dgl_amd as G
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, deg * n); dst = rng.integers(0, n, deg * n)
    g = G.DGLGraph((np.concatenate([src, dst]), np.concatenate([dst, src])), num_nodes=n).to(DEV)
    g.csr(); g.csc()
    return g


@pytest.mark.parametrize("n,sym", [(700, 1), (2500, 2), (9000, 1)])
def test_standalone_finalize_gives_the_same_loss(n, sym, tuning):
    """armed call + gae_x_decoder_bce_finalize == plain call (loss bits, gradient bits, draw counter); the armed call
    itself leaves the scalar and the counter alone"""
    from gae_dgl_amd import _lib, ops
    tuning("bce_sym", sym)
    g = _graph(n)
    torch.manual_seed(1)
    Z = torch.randn(n, 16, device=DEV) * 0.4
    pw = (n * n - g.number_of_edges()) / g.number_of_edges()
    draws_a = torch.zeros(1, dtype=torch.int64, device=DEV)
    draws_b = torch.zeros(1, dtype=torch.int64, device=DEV)
    mask_a, mask_b = torch.empty_like(Z), torch.empty_like(Z)
    loss_a, dz_a = ops.decoder_bce_raw(Z, mask_a, g.csr(), g.csc(), pw, True, dropout=(0.1, 5, 0, draws_a))
    with ops.deferred_loss_finalize():
        loss_b, dz_b = ops.decoder_bce_raw(Z, mask_b, g.csr(), g.csc(), pw, True, dropout=(0.1, 5, 0, draws_b),
                                           defer_ok=True)
        loss_b.fill_(-7.0)               # stream-ordered behind the armed call: nothing overwrites it ...
        torch.cuda.synchronize()
        assert float(loss_b) == -7.0 and int(draws_b) == 0
        tail, keep = ops.pending_loss_tail()
        assert ops.pending_loss_tail() is None
        _lib.call("gae_x_decoder_bce_finalize", ctypes.byref(tail), ops._stream())
    torch.cuda.synchronize()
    assert int(draws_a) == int(draws_b) == 1
    assert torch.equal(loss_a, loss_b) and torch.equal(dz_a, dz_b) and torch.equal(mask_a, mask_b)


def test_block_exit_flushes_a_reduction_nobody_took():
    from gae_dgl_amd import ops
    n = 1200
    g = _graph(n, seed=3)
    Z = torch.randn(n, 16, device=DEV) * 0.3
    pw = (n * n - g.number_of_edges()) / g.number_of_edges()
    ref, _ = ops.decoder_bce_raw(Z, None, g.csr(), g.csc(), pw, True)
    with ops.deferred_loss_finalize():
        l1, _ = ops.decoder_bce_raw(Z, None, g.csr(), g.csc(), pw, True, defer_ok=True)
        l2, _ = ops.decoder_bce_raw(Z * 0.5, None, g.csr(), g.csc(), pw, True, defer_ok=True)   # flushes l1 first
        l3, _ = ops.decoder_bce_raw(Z, None, g.csr(), g.csc(), pw, True)                        # not deferrable
    torch.cuda.synchronize()
    assert torch.equal(l1, ref) and torch.equal(l3, ref) and float(l2) != float(ref)
    ref2, _ = ops.decoder_bce_raw(Z * 0.5, None, g.csr(), g.csc(), pw, True)
    assert torch.equal(l2, ref2)


@pytest.mark.parametrize("n", [900, 8500])
def test_adam_tail_block_equals_the_separate_launch(n):
    """training steps whose loss is finished by the optimiser launch: losses and weights bit-identical to steps with
    the reduction launch, eager and captured"""
    import gae_dgl_amd as G
    from gae_dgl_amd import capture, ops
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    g = _graph(n, seed=5)
    X = torch.randn(n, 48, device=DEV)
    torch.manual_seed(2)
    m0 = G.GAE(48, [32, 16]).to(DEV)
    m0.decoder.seed = 9
    runs = {}
    for defer in (False, True):
        capture.DEFER_LOSS_FINALIZE = defer
        try:
            m = copy.deepcopy(m0)
            opt = Adam(m.parameters(), lr=1e-2)
            step = CapturedTrainStep(m, opt, g, X, warmup=2)
            losses = []
            for _ in range(4):
                losses.append(float(step()))
            runs[defer] = (losses, [p.detach().clone() for p in m.parameters()], opt.steps_taken())
        finally:
            capture.DEFER_LOSS_FINALIZE = True
    assert runs[False][0] == runs[True][0]
    assert runs[False][2] == runs[True][2] == 6
    for a, b in zip(runs[False][1], runs[True][1]):
        assert torch.equal(a, b)
    assert not ops.current_step().tails and not ops.current_step().partials


def test_a_loss_fn_of_the_callers_keeps_its_reduction_launch():
    """CapturedTrainStep defers only the default reconstruction loss: a caller's loss_fn may read the scalar"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    n = 600
    g = _graph(n, seed=8)
    X = torch.randn(n, 24, device=DEV)
    torch.manual_seed(4)
    m0 = G.GAE(24, [32, 16]).to(DEV)
    m0.decoder.seed = 3
    out = []
    for fn in (None, lambda m, gg: m.reconstruction_loss(gg) * 1.0):
        m = copy.deepcopy(m0)
        step = CapturedTrainStep(m, Adam(m.parameters(), lr=1e-2), g, X, loss_fn=fn, warmup=1)
        out.append([float(step()) for _ in range(3)])
    assert out[0] == out[1]


@pytest.mark.parametrize("n,dims,dropout", [(900, [32, 16], 0.1), (900, [32, 16], 0.0), (8700, [32, 16], 0.1),
                                             (1500, [32, 8], 0.1), (1300, [16], 0.1), (2100, [48, 24, 12], 0.2)])
def test_prepare_step_in_the_last_layers_epilogue(n, dims, dropout):
    """reconstruction_loss with the loss's prepare step folded into the last encoder launch (gae_x_gcn_layer_fused_prep
    + gae_x_decoder_bce_prepared) against the three-launch form: same mask bits, same loss and gradients up to the
    order of the fp64 column sums"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    g = _graph(n, seed=n)
    torch.manual_seed(n)
    X = torch.randn(n, 40, device=DEV)
    m = G.GAE(40, dims).to(DEV)
    m.decoder.seed = 21
    m.decoder.dropout = dropout
    res = []
    for fuse in (True, False):
        ops.FUSE_LOSS_PREPARE = fuse
        try:
            m.decoder._draws = None
            m.zero_grad(set_to_none=True)
            g.ndata['h'] = X
            before = ops.STATS["prepared_losses"]
            loss = m.reconstruction_loss(g)
            assert ops.STATS["prepared_losses"] - before == (1 if fuse else 0)
            ops.backward(loss, list(m.parameters()))
            torch.cuda.synchronize()
            res.append((float(loss), [p.grad.clone() for p in m.parameters()],
                        None if m.decoder.last_mask is None else m.decoder.last_mask.clone(), g.ndata['h'].clone(),
                        int(m.decoder._draws) if m.decoder._draws is not None else 0))
        finally:
            ops.FUSE_LOSS_PREPARE = True
    (la, ga, ma, za, da), (lb, gb, mb, zb, db) = res
    assert abs(la - lb) <= 1e-6 * abs(lb) and da == db == (1 if dropout else 0)
    assert torch.equal(za, zb) and ((ma is None and mb is None) or torch.equal(ma, mb))
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-6)


def test_prepared_loss_on_fixed_capacity_batches():
    """the padded-batch form (device-side true sizes, captured inductive step): an epoch with the prepare step in the
    last layer's epilogue == an epoch with the prepare launch"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    from gae_dgl_amd.capture import CapturedInductiveStep
    from gae_dgl_amd.dataset import DeviceGraphDataset
    from gae_dgl_amd.optim import Adam
    gp, src, dst, X = W.zinc_like(240, seed=4)
    ds = DeviceGraphDataset(gp, src, dst, X, device=DEV)
    torch.manual_seed(1)
    m0 = G.GAE(ds.n_feat, [32, 16]).to(DEV)
    m0.decoder.seed = 2
    order = np.random.default_rng(0).permutation(ds.ids)
    out = []
    for fuse in (True, False):
        ops.FUSE_LOSS_PREPARE = fuse
        try:
            m = copy.deepcopy(m0)
            runner = CapturedInductiveStep(m, Adam(m.parameters(), lr=1e-2), ds, 48)
            out.append(([float(l) for l in runner.epoch(order)], [p.detach().clone() for p in m.parameters()]))
        finally:
            ops.FUSE_LOSS_PREPARE = True
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-6)
    for a, b in zip(out[0][1], out[1][1]):
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-3)


@pytest.mark.parametrize("case", ["wide_last_layer", "norm_both", "given_mask"])
def test_prepare_epilogue_variants(case):
    """the prepare epilogue on 16-lane row groups (last layer with 33..64 inputs), with the symmetric degree
    normalisation, and with a caller-provided dropout mask instead of a drawn one"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    n = 1700
    g = _graph(n, seed=12)
    torch.manual_seed(3)
    X = torch.randn(n, 40, device=DEV)
    dims = [48, 16] if case == "wide_last_layer" else [32, 16]
    m = G.GAE(40, dims, norm="both" if case == "norm_both" else None).to(DEV)
    m.decoder.seed = 6
    if case == "given_mask":
        m.decoder.mask = ops.dropout_mask((n, 16), 0.1, 99, device=DEV)
    res = []
    for fuse in (True, False):
        ops.FUSE_LOSS_PREPARE = fuse
        try:
            m.decoder._draws = None
            m.zero_grad(set_to_none=True)
            g.ndata['h'] = X
            before = ops.STATS["prepared_losses"]
            loss = m.reconstruction_loss(g)
            assert ops.STATS["prepared_losses"] - before == (1 if fuse else 0)
            ops.backward(loss, list(m.parameters()))
            torch.cuda.synchronize()
            res.append((float(loss.detach()), [p.grad.clone() for p in m.parameters()], m.decoder.last_mask.clone()))
        finally:
            ops.FUSE_LOSS_PREPARE = True
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0]) and torch.equal(res[0][2], res[1][2])
    for a, b in zip(res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-6)


def test_vgae_head_kl_and_loss_in_three_launches():
    """ops.VGAEHeadLossFunction (noise + head + KL partials + loss prepare in one launch, KL added by the loss's final
    reduction) against the launch-by-launch path: same noise, z bit for bit; KL / loss / gradients within fp32 rounding;
    the noise counter advances once per loss"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    from gae_dgl_amd.vgae import VGAE
    n, src, dst, X = W.citation_graph("cora", seed=0)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    Xd = ops.pad_rows(torch.from_numpy(X).to(DEV))
    out = {}
    for fused in (True, False):
        ops.VGAE_FUSED_LOSS = fused
        try:
            torch.manual_seed(0)
            model = VGAE(X.shape[1], [32, 16], seed=5).to(DEV)
            vals = []
            for _ in range(2):                               # two draws: the counter must advance in both forms
                g.ndata['h'] = Xd
                before = ops.STATS["prepared_losses"]
                loss = model.loss(g)
                assert ops.STATS["prepared_losses"] - before == (1 if fused else 0)
                model.zero_grad(set_to_none=True)
                ops.backward(loss, list(model.parameters()))
                torch.cuda.synchronize()
                vals.append((float(loss.detach()), {k: v.detach().clone() for k, v in model.last.items()},
                             [p.grad.clone() for p in model.parameters()]))
            assert int(model._draws) == 2
            out[fused] = vals
        finally:
            ops.VGAE_FUSED_LOSS = True
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a[1]["eps"], b[1]["eps"]) and torch.equal(a[1]["z"], b[1]["z"])
        assert abs(a[0] - b[0]) <= 2e-6 * abs(b[0])
        assert abs(float(a[1]["kl"]) - float(b[1]["kl"])) <= 2e-6 * abs(float(b[1]["kl"]))
        assert abs(float(a[1]["rec"]) - float(b[1]["rec"])) <= 2e-6 * abs(float(b[1]["rec"]))
        assert abs(float(a[1]["rec"]) + float(a[1]["kl"]) - a[0]) <= 1e-6 * abs(a[0])
        for ga, gb in zip(a[2], b[2]):
            assert float((ga - gb).abs().max()) <= 2e-6 * max(float(gb.abs().max()), 1e-6)
    assert not torch.equal(out[True][0][1]["eps"], out[True][1][1]["eps"])


def test_vgae_captured_step_with_the_loss_tail_in_adam():
    """CapturedTrainStep(defer_loss=True): the VGAE loss (rec + KL) is finished by the optimiser launch -- losses and
    weights equal the eager steps bit for bit"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    from gae_dgl_amd.vgae import VGAE
    n, src, dst, X = W.citation_graph("cora", seed=0)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    Xd = ops.pad_rows(torch.from_numpy(X).to(DEV))
    runs = {}
    for mode in ("eager", "captured"):
        torch.manual_seed(0)
        model = VGAE(X.shape[1], [32, 16], seed=5).to(DEV)
        opt = Adam(model.parameters(), lr=1e-2)
        params = list(model.parameters())
        if mode == "eager":
            losses = []
            for _ in range(7):
                g.ndata['h'] = Xd
                loss = model.loss(g)
                opt.zero_grad(set_to_none=True); ops.backward(loss, params); opt.step()
                model.last = {}
                losses.append(float(loss.detach()))
        else:
            step = CapturedTrainStep(model, opt, g, Xd, loss_fn=lambda m, gg: m.loss(gg), warmup=2, defer_loss=True)
            losses = [None, None] + [float(step()) for _ in range(5)]
        runs[mode] = (losses, [p.detach().clone() for p in params])
    assert runs["eager"][0][2:] == runs["captured"][0][2:]
    for a, b in zip(runs["eager"][1], runs["captured"][1]):
        assert torch.equal(a, b)
    assert not ops.current_step().tails and not ops.current_step().partials
