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
    tuning("bce_sym", sym)            # 2: the symmetric dense kernel already from 512 rows on (default: 5120)
    ds, _ = _dataset()
    ids = np.arange(40, 104)
    bg = ds.batch(ids)
    n, e = bg.number_of_nodes(), bg.number_of_edges()
    cap_n, cap_e = n + 77, e + 300
    torch.manual_seed(0)
    Z = torch.randn(n, 16, device=DEV) * 0.5
    Zp = torch.zeros(cap_n, 16, device=DEV); Zp[:n] = Z; Zp[n:] = 3.0          # garbage in the padding rows
    node_ptr, edge_ptr, _ = ops.batch_plan(ds.graph_ptr, ds.indptr, None, torch.from_numpy(ids).to(DEV))
    counts = torch.zeros(3, dtype=torch.int64, device=DEV)
    ip, ix, feat, table = ops.batch_gather(ds.graph_ptr, ds.indptr, ds.indices, ds.feat, torch.from_numpy(ids).to(DEV),
                                           node_ptr, edge_ptr, cap_n, cap_e, ell_width=ds.ell_width, n_feat=ds.n_feat,
                                           pad_to_capacity=True, counts=counts)
    assert counts.tolist() == [n, e, 0]
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
    """gae_batch_select and the fused gae_x_batch_plan_next: ids of batch `cursor` of an uploaded order, the cursor
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


def test_gather_guard_drops_what_does_not_fit():
    """device-side guard of gae_batch_gather (fixed-capacity mode): a batch whose prefix sums exceed the buffers is cut
    to the longest prefix of member graphs that fits -- nothing is written behind the arrays, the kept prefix is a
    correct batch, out_counts[2] counts the graphs left out"""
    from gae_dgl_amd import ops
    ds, _ = _dataset()
    ids = np.arange(10, 74)
    gids = torch.from_numpy(ids).to(DEV)
    node_ptr, edge_ptr, _ = ops.batch_plan(ds.graph_ptr, ds.indptr, None, gids)
    npt, ept = node_ptr.cpu().numpy(), edge_ptr.cpu().numpy()
    keep = 41                                                     # capacities between graph 41's and graph 42's end
    cap_n, cap_e = int(npt[keep]) + 3, int(ept[keep]) + 5
    assert cap_n < npt[keep + 1] or cap_e < ept[keep + 1]
    F, ldo, odt = ops.batch_feature_ld(ds.feat, ds.n_feat)
    W = ds.ell_width
    guard = 4096                                                  # sentinel area behind every buffer
    ip = torch.full((cap_n + 1 + guard,), -7, dtype=torch.int32, device=DEV)
    ix = torch.full((cap_e + guard,), -7, dtype=torch.int32, device=DEV)
    feat = torch.full((cap_n + guard // 8, ldo), -7.0, dtype=odt, device=DEV)
    table = torch.full(((cap_n + guard // 8) * W,), -7, dtype=torch.int32, device=DEV)
    counts = torch.zeros(3, dtype=torch.int64, device=DEV)
    out = (ip[:cap_n + 1], ix[:cap_e], feat[:cap_n], table[:cap_n * W])
    ops.batch_gather(ds.graph_ptr, ds.indptr, ds.indices, ds.feat, gids, node_ptr, edge_ptr, cap_n, cap_e,
                     ell_width=W, n_feat=ds.n_feat, out=out, pad_to_capacity=True, counts=counts)
    torch.cuda.synchronize()
    assert counts.tolist() == [int(npt[keep]), int(ept[keep]), len(ids) - keep]
    assert bool((ip[cap_n + 1:] == -7).all()) and bool((ix[cap_e:] == -7).all())
    assert bool((feat[cap_n:] == -7).all()) and bool((table[cap_n * W:] == -7).all())
    ref = ds.batch(ids[:keep])
    n, e = ref.number_of_nodes(), ref.number_of_edges()
    assert torch.equal(ip[:n + 1], ref.csr()[0]) and torch.equal(ix[:e], ref.csr()[1])
    assert bool((ip[n:cap_n + 1] == e).all()) and torch.equal(feat[:n, :F], ref.ndata['h'])
    assert bool((feat[n:cap_n] == 0).all()) and bool((table[n * W:cap_n * W] == -1).all())
    # a second overflowing call ADDS to the counter; a batch that fits leaves it alone
    ops.batch_gather(ds.graph_ptr, ds.indptr, ds.indices, ds.feat, gids, node_ptr, edge_ptr, cap_n, cap_e,
                     ell_width=W, n_feat=ds.n_feat, out=out, pad_to_capacity=True, counts=counts)
    assert int(counts[2]) == 2 * (len(ids) - keep)


def test_capture_with_a_single_full_batch_and_per_step_lr():
    """B <= len(order) < 2 B (the warm-up steps of the capture used to walk past the only full batch into the ragged
    tail, whose size was never checked against the buffers) and a learning rate that changes on every step (the
    re-capture at the LAST full batch had the same problem): no graph is dropped by the device guard, the losses
    equal the eager loop's, the step counter and the dropout stream do not see the warm-up steps"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from gae_dgl_amd.capture import CapturedInductiveStep
    from gae_dgl_amd.optim import Adam
    ds, _ = _dataset(120)
    B = 48
    big_last = np.argsort(ds.sizes_host, kind="stable")             # the tail holds the LARGEST molecules
    order = big_last[:B + 20].copy()                                # one full batch + a ragged tail of 20 large graphs
    torch.manual_seed(5)
    m_e = G.GAE(ds.n_feat, [32, 16]).to(DEV)
    m_e.decoder.seed = 9
    m_c = copy.deepcopy(m_e)
    o_e, o_c = Adam(m_e.parameters(), lr=1e-2), Adam(m_c.parameters(), lr=1e-2)
    runner = CapturedInductiveStep(m_c, o_c, ds, B, margin=1.0)
    losses_c = [float(l) for l in runner.epoch(order)]
    assert runner.dropped_graphs() == 0 and len(losses_c) == 2
    d_order = torch.from_numpy(order).to(DEV)
    losses_e = []
    for lo in range(0, len(order), B):
        bg = ds._assemble(d_order[lo:lo + B], order[lo:lo + B])
        o_e.zero_grad()
        loss = m_e.reconstruction_loss(bg)
        ops.backward(loss); o_e.step()
        losses_e.append(float(loss))
    np.testing.assert_allclose(losses_c, losses_e, rtol=1e-5)
    assert o_c.steps_taken() == o_e.steps_taken() == 2
    with pytest.raises(ops.GaeHipError):
        runner.step()                                               # no full batch left: refused, not replayed
    # ---- a per-step LR schedule: every step re-captures, also the one at the last full batch
    order2 = big_last[::-1][:3 * B].copy()
    n_full = runner.begin_epoch(order2)
    caps = runner.captures
    for k in range(n_full):
        for g in o_c.param_groups:
            g["lr"] = 1e-2 / (k + 2)
        loss = runner.step()
        lo = order2[k * B:(k + 1) * B]
        assert runner.batch_sizes() == (int(ds.sizes_host[lo].sum()), int(ds.edges_host[lo].sum()))
    assert np.isfinite(float(loss)) and runner.captures == caps + n_full and runner.dropped_graphs() == 0
    assert o_c.steps_taken() == 2 + n_full


def test_adam_resume_survives_reload_and_capture():
    """load_state_dict() then state_dict() keeps the step count (the counter does not exist before the first step);
    a capture right after a load restores counter, moments and dropout draws to the resume point"""
    import gae_dgl_amd as G
    from gae_dgl_amd.capture import CapturedInductiveStep
    from gae_dgl_amd.optim import Adam
    ds, _ = _dataset(100)
    torch.manual_seed(1)
    m = G.GAE(ds.n_feat, [32, 16]).to(DEV)
    opt = Adam(m.parameters(), lr=1e-3)
    r = CapturedInductiveStep(m, opt, ds, 25)
    for _ in r.epoch(ds.ids[:75]):
        pass
    assert opt.steps_taken() == 3
    sd = copy.deepcopy(opt.state_dict())
    w0 = [p.detach().clone() for p in m.parameters()]
    opt2 = Adam(m.parameters(), lr=1e-3)
    opt2.load_state_dict(sd)
    assert opt2.steps_taken() == 3                                   # before any step: the checkpoint's count
    sd2 = opt2.state_dict()
    assert all(float(st["step"]) == 3.0 for st in sd2["state"].values())
    m.decoder._draws = None                                          # resume in a fresh process: no draw counter yet
    r2 = CapturedInductiveStep(m, opt2, ds, 25)
    r2.begin_epoch(ds.ids[:75])                                      # captures: warm-up steps must leave no trace
    assert opt2.steps_taken() == 3 and int(m.decoder._draws) == 0
    for p, q in zip(m.parameters(), w0):
        assert torch.equal(p, q)
    for a, b in zip(opt2.state_dict()["state"].values(), sd["state"].values()):
        assert torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"])
    r2.step()
    assert opt2.steps_taken() == 4


@pytest.mark.parametrize("B", [24, 128, 1000])
def test_one_launch_collate_equals_plan_then_gather(B):
    """gae_x_batch_gather_next (select + plan + gather in one launch) == gae_x_batch_plan_next followed by gae_batch_gather:
    ids, prefix sums, CSR, features, packed table and counts bit for bit over consecutive batches of an epoch order; the
    device cursor advances once per launch; a batch that does not fit is cut to the prefix that does"""
    from gae_dgl_amd import ops
    ds, _ = _dataset(max(3 * B + 7, 300))
    order = np.random.default_rng(B).permutation(ds.ids)
    d_order = torch.from_numpy(order).to(DEV)
    n_full = len(order) // B
    need_n = max(int(ds.sizes_host[order[k * B:(k + 1) * B]].sum()) for k in range(n_full))
    need_e = max(int(ds.edges_host[order[k * B:(k + 1) * B]].sum()) for k in range(n_full))
    cap_n, cap_e = need_n + 40, need_e + 100
    F, ldo, odt = ops.batch_feature_ld(ds.feat, ds.n_feat)
    W = ds.ell_width

    def buffers():
        return (torch.full((cap_n + 1,), -5, dtype=torch.int32, device=DEV), torch.full((cap_e,), -5, dtype=torch.int32, device=DEV),
                torch.full((cap_n, ldo), -5.0, dtype=odt, device=DEV), torch.full((cap_n * W,), -5, dtype=torch.int32, device=DEV))
    cur_a = torch.zeros(1, dtype=torch.int64, device=DEV); cur_b = torch.zeros(1, dtype=torch.int64, device=DEV)
    ids_a = torch.empty(B, dtype=torch.int64, device=DEV); ids_b = torch.empty(B, dtype=torch.int64, device=DEV)
    p_a = torch.empty(2, B + 1, dtype=torch.int64, device=DEV); p_b = torch.empty(2, B + 1, dtype=torch.int64, device=DEV)
    c_a = torch.zeros(4, dtype=torch.int64, device=DEV); c_b = torch.zeros(4, dtype=torch.int64, device=DEV)
    for k in range(n_full):
        out_a, out_b = buffers(), buffers()
        npt, ept, _ = ops.batch_plan_next(ds.graph_ptr, ds.indptr, None, d_order, cur_a, ids_a, p_a)
        ops.batch_gather(ds.graph_ptr, ds.indptr, ds.indices, ds.feat, ids_a, npt, ept, cap_n, cap_e, ell_width=W,
                         n_feat=ds.n_feat, out=out_a, pad_to_capacity=True, counts=c_a)
        ops.batch_gather_next(ds.graph_ptr, ds.indptr, ds.indices, ds.feat, d_order, cur_b, ids_b, p_b, cap_n, cap_e, out_b,
                              c_b, ell_width=W, n_feat=ds.n_feat)
        assert torch.equal(ids_a, ids_b) and torch.equal(p_a, p_b) and int(cur_b) == k + 1 == int(cur_a)
        assert c_a[:3].tolist() == c_b[:3].tolist() and int(c_b[3]) == 0
        e = int(c_a[1])
        assert torch.equal(out_a[0], out_b[0]) and torch.equal(out_a[1][:e], out_b[1][:e])
        assert torch.equal(out_a[2], out_b[2]) and torch.equal(out_a[3], out_b[3])
    # ---- capacity guard: buffers one graph too small
    small_n = int(p_a[0][B - 1]) + 2                                   # fits the first B - 1 graphs of the last batch
    ip = torch.full((small_n + 1 + 512,), -7, dtype=torch.int32, device=DEV)
    ix = torch.full((cap_e + 512,), -7, dtype=torch.int32, device=DEV)
    ft = torch.full((small_n + 64, ldo), -7.0, dtype=odt, device=DEV)
    tb = torch.full(((small_n + 64) * W,), -7, dtype=torch.int32, device=DEV)
    cur = torch.full((1,), n_full - 1, dtype=torch.int64, device=DEV)
    c = torch.zeros(4, dtype=torch.int64, device=DEV)
    ops.batch_gather_next(ds.graph_ptr, ds.indptr, ds.indices, ds.feat, d_order, cur, ids_b, p_b, small_n, cap_e,
                          (ip[:small_n + 1], ix[:cap_e], ft[:small_n], tb[:small_n * W]), c, ell_width=W, n_feat=ds.n_feat)
    torch.cuda.synchronize()
    assert int(c[2]) >= 1 and int(c[0]) <= small_n and int(c[0]) == int(p_a[0][B - int(c[2])])
    assert bool((ip[small_n + 1:] == -7).all()) and bool((ft[small_n:] == -7).all()) and bool((tb[small_n * W:] == -7).all())
