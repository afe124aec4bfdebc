"""gae_x_gcn_layer_fused_wgrad: the backward of the fused GCN layer (identity activation) in one launch -- dH from the
fused kernel on A^T, dW / db as side work of the same blocks -- against fp64 and against the two-launch form."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, seed=0, deg=3, norm=False):
    import gae_dgl_amd as G
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, deg * n); dst = rng.integers(0, n, deg * n)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    g.csr(); g.csc()
    return g, src, dst


@pytest.mark.parametrize("n,f_in,f_out", [(1, 32, 16), (37, 32, 16), (1000, 32, 16), (4099, 20, 12), (2500, 8, 32),
                                           (3000, 32, 32), (70001, 32, 16)])
def test_one_launch_backward_against_fp64(n, f_in, f_out):
    from gae_dgl_amd import ops
    g, src, dst = _graph(n, seed=n)
    torch.manual_seed(n)
    H = torch.randn(n, f_in, device=DEV)
    W = (torch.randn(f_out, f_in, device=DEV) * 0.3).requires_grad_()
    b = torch.randn(f_out, device=DEV).requires_grad_()
    Hr = H.clone().requires_grad_()
    Y = ops.GCNLayerFusedFunction.apply(Hr, W, b, g, False, ops.ACT_IDENTITY)
    dY = torch.randn(n, f_out, device=DEV)
    ops.FUSED_LAYER_WGRAD = True
    try:
        dH, dW, db = torch.autograd.grad(Y, (Hr, W, b), dY, retain_graph=True)
        ops.FUSED_LAYER_WGRAD = False
        dH2, dW2, db2 = torch.autograd.grad(Y, (Hr, W, b), dY)
    finally:
        ops.FUSED_LAYER_WGRAD = True
    # fp64 reference: M = A H (row = destination), Y = M W^T + b
    A = torch.zeros(n, n, dtype=torch.float64, device=DEV) if n <= 5000 else None
    if A is not None:
        A.index_put_((torch.as_tensor(dst, device=DEV), torch.as_tensor(src, device=DEV)),
                     torch.ones(len(src), dtype=torch.float64, device=DEV), accumulate=True)
        M = A @ H.double()
        rW = dY.double().t() @ M
        rH = (A.t() @ dY.double()) @ W.detach().double()
        rb = dY.double().sum(0)
        for got, ref in ((dW, rW), (dH, rH), (db, rb)):
            scale = max(float(ref.abs().max()), 1e-6)
            assert float((got.double() - ref).abs().max()) <= 2e-6 * scale * max(1.0, np.sqrt(n / 1000.0))
    assert torch.equal(dH, dH2)                                    # the same kernel produces dH in both forms
    for got, ref in ((dW, dW2), (db, db2)):
        assert float((got - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1e-6)


def test_deferred_partials_reach_adam_with_the_same_bits():
    """inside deferred_grad_reductions() the weight gradient is a partial-sum list that Adam adds up: the written
    gradients and the updated weights equal those of the eager call (reduction launch) bit for bit"""
    import copy
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from gae_dgl_amd.optim import Adam
    n = 5000
    g, _, _ = _graph(n, seed=4)
    torch.manual_seed(0)
    X = torch.randn(n, 40, device=DEV)
    m0 = G.GAE(40, [32, 16]).to(DEV)
    m0.decoder.seed = 1
    out = []
    for defer in (False, True):
        m = copy.deepcopy(m0)
        opt = Adam(m.parameters(), lr=1e-2)
        g.ndata['h'] = X
        loss = m.reconstruction_loss(g)
        params = list(m.parameters())
        if defer:
            with ops.deferred_grad_reductions():
                ops.backward(loss, params)
                assert ops.current_step().partials
                opt.step()
        else:
            ops.backward(loss, params)
            opt.step()
        torch.cuda.synchronize()
        out.append(([p.grad.clone() for p in params], [p.detach().clone() for p in params]))
    for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert torch.equal(a, b)


def test_padding_rows_of_a_capacity_batch_add_nothing():
    """rows behind the true batch size carry dY = 0: their blocks' partials are zero, the sums unchanged"""
    from gae_dgl_amd import ops
    n, n_pad = 700, 1024
    g, src, dst = _graph(n_pad, seed=9)
    keep = (src < n) & (dst < n)
    import gae_dgl_amd as G
    gp = G.DGLGraph((src[keep], dst[keep]), num_nodes=n_pad).to(DEV)
    gs = G.DGLGraph((src[keep], dst[keep]), num_nodes=n).to(DEV)
    torch.manual_seed(5)
    H = torch.randn(n_pad, 32, device=DEV); H[n:] = 0
    W = torch.randn(16, 32, device=DEV) * 0.2
    dY = torch.randn(n_pad, 16, device=DEV); dY[n:] = 0
    res = []
    for gg, rows in ((gp, n_pad), (gs, n)):
        gg.csr(); gg.csc()
        M, _ = ops.gcn_layer_fused_raw(*gg.csr(), H[:rows].contiguous(), rows, gg.spmm_plan(False), W, None,
                                       ops.ACT_IDENTITY)
        res.append(ops.gcn_layer_fused_wgrad_raw(*gg.csc(), dY[:rows].contiguous(), rows, gg.spmm_plan(True), W, M, None))
    assert torch.equal(res[0][0][:n], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


@pytest.mark.parametrize("variant", [0, 2, 6, 14])
@pytest.mark.parametrize("n,f_in,f_out", [(1234, 32, 16), (900, 24, 32)])
def test_every_form_of_the_side_work_gives_the_same_gradient(n, f_in, f_out, variant, tuning):
    """knob ell_side: scalar LDS loop / matrix cores from LDS tiles / operands straight from global memory"""
    from gae_dgl_amd import ops
    g, src, dst = _graph(n, seed=3)
    torch.manual_seed(7)
    dY = torch.randn(n, f_out, device=DEV)
    W = torch.randn(f_out, f_in, device=DEV) * 0.2
    M = torch.randn(n, f_in, device=DEV)
    tuning("ell_side", variant)
    dH, dW, db = ops.gcn_layer_fused_wgrad_raw(*g.csc(), dY, n, g.spmm_plan(True), W, M, None)
    rW = dY.double().t() @ M.double()
    rb = dY.double().sum(0)
    assert float((dW.double() - rW).abs().max()) <= 2e-6 * float(rW.abs().max())
    assert float((db.double() - rb).abs().max()) <= 2e-6 * float(rb.abs().max())
