"""The captured inductive training step (capture.CapturedInductiveStep): fixed-capacity batches, device-side true
sizes, one HIP-graph launch per batch -- against the eager step and against the oracle."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dataset(n=300, seed=11, directed=False):
    from gae_dgl_amd import workloads as W
    from gae_dgl_amd.dataset import DeviceGraphDataset
    gp, src, dst, X = W.zinc_like(n, seed=seed)
    if directed:
        keep = np.ones(len(src), bool); keep[1::6] = False
        src, dst = src[keep], dst[keep]
    return DeviceGraphDataset(gp, src, dst, X, device=DEV), (gp, src, dst, X)


@pytest.mark.parametrize("sym", [1, 2])
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_padded_loss_matches_unpadded(dropout, sym, tuning):
    """gae_decoder_bce_padded on a capacity-padded batch == gae_decoder_bce on the batch itself (loss 1e-6 relative,
    gradient 1e-5 of its scale; padding rows get an exactly zero gradient) and == the fp64 oracle"""
    from gae_dgl_amd import ops
    from oracle import gae_oracle as O
    tuning("bce_sym", sym)            # 2: the symmetric dense kernel already from 512 rows on (default: 8192)
    ds, _ = _dataset()
    ids = np.arange(40, 104)
    bg = ds.batch(ids)
    n, e = bg.number_of_nodes(), bg.number_of_edges()
    cap_n, cap_e = n + 77, e + 300
    torch.manual_seed(0)
    Z = torch.randn(n, 16, device=DEV) * 0.5
    Zp = torch.zeros(cap_n, 16, device=DEV); Zp[:n] = Z; Zp[n:] = 3.0          # garbage in the padding rows
    node_ptr, edge_ptr, _ = ops.batch_plan(ds.graph_ptr, ds.indptr, None, torch.from_numpy(ids).to(DEV))
    counts = torch.zeros(2, dtype=torch.int64, device=DEV)
    ip, ix, feat, table = ops.batch_gather(ds.graph_ptr, ds.indptr, ds.indices, ds.feat, torch.from_numpy(ids).to(DEV),
                                           node_ptr, edge_ptr, cap_n, cap_e, ell_width=ds.ell_width, n_feat=ds.n_feat,
                                           pad_to_capacity=True, counts=counts)
    assert counts.tolist() == [n, e]
    # the padded structure: the batch's CSR, then empty rows / zero features / empty table rows
    assert torch.equal(ip[:n + 1], bg.csr()[0]) and torch.equal(ix[:e], bg.csr()[1])
    assert bool((ip[n:] == e).all()) and bool((feat[n:] == 0).all()) and torch.equal(feat[:n], bg.ndata['h'])
    assert bool((table.view(cap_n, -1)[n:] == -1).all())
    assert torch.equal(table.view(cap_n, -1)[:n], bg.spmm_plan(False).ell.view(n, -1))
    pw = (n * n - e) / e
    drop = None
    if dropout:
        m0 = torch.empty(n, 16, device=DEV); m1 = torch.empty(cap_n, 16, device=DEV)
        d0 = torch.zeros(1, dtype=torch.int64, device=DEV); d1 = torch.zeros(1, dtype=torch.int64, device=DEV)
        l0, g0 = ops.decoder_bce_raw(Z, m0, bg.csr(), bg.csc(), pw, dropout=(dropout, 5, 0, d0))
        l1, g1 = ops.decoder_bce_raw(Zp, m1, (ip, ix), (ip, ix), 0.0, dropout=(dropout, 5, 0, d1), counts=counts)
        assert torch.equal(m0, m1[:n]) and bool((m1[n:] == 0).all()) and int(d1) == 1
        mask = m0
    else:
        l0, g0 = ops.decoder_bce_raw(Z, None, bg.csr(), bg.csc(), pw)
        l1, g1 = ops.decoder_bce_raw(Zp, None, (ip, ix), (ip, ix), 0.0, counts=counts)
        mask = None
    assert abs(float(l1) - float(l0)) <= 1e-6 * abs(float(l0))
    scale = float(g0.abs().max())
    assert float((g1[:n] - g0).abs().max()) <= 1e-5 * scale
    assert bool((g1[n:] == 0).all())
    Zt = (Z if mask is None else Z * mask).cpu()
    c, t = bg.csr(), bg.csc()
    lo, go = O.bce_row_window(Zt, 0, n, c[0].cpu().numpy(), c[1].cpu().numpy(), t[0].cpu().numpy(),
                              t[1].cpu().numpy(), pw)
    if mask is not None:
        go = go * mask.cpu().double()                                  # dLoss/dZ = dLoss/dZt (.) mask
    assert abs(float(l1) - float(lo)) <= 1e-5 * abs(float(lo))
    assert float((g1[:n].cpu().double() - go).abs().max()) <= 1e-5 * float(go.abs().max())


@pytest.mark.parametrize("directed", [False, True])
def test_captured_epoch_matches_eager_epoch(directed):
    """two epochs of batch-32 training, captured vs eager, same initial weights / order / dropout stream: the
    per-batch losses agree to 1e-5 relative and the final weights to 1e-5 of their scale (the rows a reduction is
    split over differ with the padding, nothing else does)"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from gae_dgl_amd.capture import CapturedInductiveStep
    from gae_dgl_amd.optim import Adam
    ds, _ = _dataset(300, directed=directed)
    B = 32
    torch.manual_seed(3)
    m_e = G.GAE(ds.n_feat, [32, 16]).to(DEV)
    m_e.decoder.seed = 77
    m_c = copy.deepcopy(m_e)
    o_e, o_c = Adam(m_e.parameters(), lr=1e-2), Adam(m_c.parameters(), lr=1e-2)
    rng = np.random.default_rng(0)
    orders = [rng.permutation(ds.ids) for _ in range(2)]
    runner = CapturedInductiveStep(m_c, o_c, ds, B)
    losses_c = []
    for order in orders:
        for k, loss in enumerate(runner.epoch(order)):
            losses_c.append(float(loss))
            if k == 0:
                lo = order[:B]
                assert runner.batch_sizes() == (int(ds.sizes_host[lo].sum()), int(ds.edges_host[lo].sum()))
    assert runner.captures == 1 or runner.cap_nodes >= 0       # a second capture only if epoch 2 needed more room
    losses_e = []
    for order in orders:
        d_order = torch.from_numpy(order).to(DEV)
        for lo in range(0, len(order), B):
            bg = ds._assemble(d_order[lo:lo + B], order[lo:lo + B])
            o_e.zero_grad()
            loss = m_e.reconstruction_loss(bg)
            ops.backward(loss)
            o_e.step()
            losses_e.append(float(loss))
    assert len(losses_c) == len(losses_e) == 2 * ((300 + B - 1) // B)
    np.testing.assert_allclose(losses_c, losses_e, rtol=1e-5)
    for pc, pe in zip(m_c.parameters(), m_e.parameters()):
        assert float((pc - pe).abs().max()) <= 1e-5 * max(float(pe.abs().max()), 1e-3) + 2e-6
    assert o_c.steps_taken() == o_e.steps_taken() == len(losses_e)


def test_captured_step_grows_and_rejects():
    """an epoch whose largest batch exceeds the captured capacity is captured again with larger buffers; datasets
    the fixed-capacity path cannot serve are refused up front"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from gae_dgl_amd.capture import CapturedInductiveStep
    from gae_dgl_amd.optim import Adam
    ds, _ = _dataset(200)
    m = G.GAE(ds.n_feat, [32, 16]).to(DEV)
    opt = Adam(m.parameters(), lr=1e-3)
    runner = CapturedInductiveStep(m, opt, ds, 16, margin=1.0)
    small_first = np.argsort(ds.sizes_host, kind="stable")             # smallest molecules first
    n_full = runner.begin_epoch(small_first[:64])
    cap0 = runner.cap_nodes
    for _ in range(n_full):
        runner.step()
    assert runner.captures == 1
    n_full = runner.begin_epoch(small_first[::-1][:64].copy())         # the largest ones: does not fit
    assert runner.captures == 2 and runner.cap_nodes > cap0
    for _ in range(n_full):
        loss = runner.step()
    assert np.isfinite(float(loss))
    assert opt.steps_taken() == 8
    with pytest.raises(ValueError):
        CapturedInductiveStep(m, opt, ds, 0)
    wide = G.GAE(ds.n_feat, [32, 128]).to(DEV)                         # beyond the fused loss kernel's width
    r2 = CapturedInductiveStep(wide, Adam(wide.parameters()), ds, 16)
    with pytest.raises(ops.GaeHipError):
        r2.begin_epoch(ds.ids[:32])


def test_batch_select_and_plan_next_walk_an_epoch_order():
    """gae_batch_select and the fused gae_batch_plan_next: ids of batch `cursor` of an uploaded order, the cursor
    advances on the device, prefix sums equal gae_batch_plan's on the same ids (bit-exact integer work)"""
    from gae_dgl_amd import ops
    ds, _ = _dataset(200)
    B = 24
    order = np.random.default_rng(4).permutation(ds.ids)
    d_order = torch.from_numpy(order).to(DEV)
    cur_a = torch.zeros(1, dtype=torch.int64, device=DEV); cur_b = torch.zeros(1, dtype=torch.int64, device=DEV)
    ids_a = torch.empty(B, dtype=torch.int64, device=DEV); ids_b = torch.empty(B, dtype=torch.int64, device=DEV)
    ptrs = torch.empty(2, B + 1, dtype=torch.int64, device=DEV)
    for k in range(len(order) // B):
        ops.batch_select(d_order, cur_a, B, ids_a)
        node_ptr, edge_ptr, _ = ops.batch_plan_next(ds.graph_ptr, ds.indptr, None, d_order, cur_b, ids_b, ptrs)
        want = order[k * B:(k + 1) * B]
        assert np.array_equal(ids_a.cpu().numpy(), want) and np.array_equal(ids_b.cpu().numpy(), want)
        assert int(cur_a) == int(cur_b) == k + 1
        ref = ops.batch_plan(ds.graph_ptr, ds.indptr, None, ids_a)
        assert torch.equal(node_ptr, ref[0]) and torch.equal(edge_ptr, ref[1])
        assert np.array_equal(node_ptr.cpu().numpy()[1:], np.cumsum(ds.sizes_host[want]))
