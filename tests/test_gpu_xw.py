"""The one-pass layer-1 kernels (csrc/xw.hip, gae_spmm_csr_epilogue) and the transform-first GCN layer built from
them: P = X W^T with W stationary, dW = G^T X in one pass over X, the sparse half with bias / activation / ReLU gate --
each against an fp64 restatement, and the whole layer against the oracle's reference-order step (gae.py:26-31)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def rel(a, b):
    a = a.detach().double().cpu(); b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def rand_graph(rng, n, e):
    a = rng.integers(0, n, e); b = rng.integers(0, n, e)
    return np.concatenate([a, b]).astype(np.int64), np.concatenate([b, a]).astype(np.int64)


SHAPES = [(19717, 500, 32), (2708, 1433, 32), (3327, 3703, 32), (1000, 200, 16), (777, 300, 7), (5, 260, 32),
          (16, 1025, 3), (4100, 513, 17)]


@pytest.mark.parametrize("n,K,J", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_xw_fwd_matches_fp64(n, K, J, dtype, tuning):
    """gae_xw_fwd: P = act(X W^T + b) for fp32 and bf16-stored X, every slice / split / tail shape; pad columns of X
    hold NaN (the kernel must never let them reach a sum)"""
    from gae_dgl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n + K)
    X = torch.randn(n, K, generator=g)
    W = torch.randn(J, K, generator=g) / K ** 0.5
    b = torch.randn(J, generator=g)
    Xd = X.to(DEV).to(dtype)
    ld = ops.padded_ld(K, dtype)
    buf = torch.full((n, ld), float("nan"), dtype=dtype, device=DEV)
    buf[:, :K] = Xd
    Xp = buf[:, :K]
    assert ops.xw_usable(Xp, J)
    ref_x = Xd.double().cpu()                                      # the values as stored
    tol = TOL if dtype == torch.float32 else 3e-5                  # bf16 rows: W = hi + lo carries 16 mantissa bits
    for act, bias in ((0, None), (1, b.to(DEV))):
        P = ops.xw_fwd_raw(Xp, W.to(DEV), bias, act)
        ref = ref_x @ W.double().t() + (b.double() if bias is not None else 0)
        if act:
            ref = torch.relu(ref)
        assert P.shape == (n, J) and rel(P, ref) < tol
    if n >= 1000:                    # other rows-per-block choices (and with them other splits along K): same value
        P0 = ops.xw_fwd_raw(Xp, W.to(DEV), None, 0)
        tuning("xw_rows", 16)
        P1 = ops.xw_fwd_raw(Xp, W.to(DEV), None, 0)
        assert rel(P1, P0) < 2e-6


@pytest.mark.parametrize("n,K,J", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_xw_wgrad_matches_fp64(n, K, J, dtype, tuning):
    """gae_xw_wgrad: dW = (G (.) [Gmask > 0])^T X and db = colsum(D (.) [Dmask > 0]); masks optional; partition count
    changes the association only"""
    from gae_dgl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3 * n + K)
    X = torch.randn(n, K, generator=g)
    G = torch.randn(n, J, generator=g)
    M = torch.randn(n, J, generator=g)
    D = torch.randn(n, J, generator=g)
    Xd = X.to(DEV).to(dtype)
    ld = ops.padded_ld(K, dtype)
    buf = torch.full((n, ld), float("nan"), dtype=dtype, device=DEV)
    buf[:, :K] = Xd
    Xp = buf[:, :K]
    ref_x = Xd.double().cpu()
    Gd, Md, Dd = G.to(DEV), M.to(DEV), D.to(DEV)
    dW, db = ops.xw_wgrad_raw(Xp, Gd, Md, Dd, Md, J)
    gm = G.double() * (M > 0)
    assert rel(dW, gm.t() @ ref_x) < TOL and rel(db, (D.double() * (M > 0)).sum(0)) < TOL
    dW2, db2 = ops.xw_wgrad_raw(Xp, Gd, None, Dd, None, J)
    assert rel(dW2, G.double().t() @ ref_x) < TOL and rel(db2, D.double().sum(0)) < TOL
    dW3, none = ops.xw_wgrad_raw(Xp, Gd, None, None, None, J)
    assert none is None and torch.equal(dW3, dW2)
    _, db4 = ops.xw_wgrad_raw(Xp, Gd, None, Dd, Md, J, need_dW=False)
    assert torch.equal(db4, db)
    # G staged in LDS once per block (default where the partition's rows fit, J > 16) == G loaded per row group: the same
    # products in the same order, bit for bit; with NaN behind the rows' J columns of a wider G (the staging must gate
    # them as the per-group loads do)
    Gw = torch.full((n, J + 4), float("nan"), device=DEV); Gw[:, :J] = Gd
    tuning("xw_glds", 0)
    dW6, db6 = ops.xw_wgrad_raw(Xp, Gw[:, :J], Md, Dd, Md, J)
    tuning("xw_glds", 1)
    dW7, db7 = ops.xw_wgrad_raw(Xp, Gw[:, :J], Md, Dd, Md, J)
    assert torch.equal(dW6, dW) and torch.equal(db6, db) and torch.equal(dW7, dW) and torch.equal(db7, db)
    tuning("xw_parts", 3)                 # (Pubmed: 6573 rows per partition do not fit the LDS stage: per-group loads)
    dW5, _ = ops.xw_wgrad_raw(Xp, Gd, Md, Dd, Md, J)
    assert rel(dW5, gm.t() @ ref_x) < TOL


@pytest.mark.parametrize("F", [32, 16, 7, 48])
@pytest.mark.parametrize("norm", [False, True])
def test_spmm_epilogue_bias_act_and_gate(F, norm):
    """gae_spmm_csr_epilogue: act(rs A cs H + b) equals the plain product followed by bias / ReLU bit for bit, and the
    gated gather equals the plain product of the pre-gated operand bit for bit on every row that fits the table
    (CSR-order sums in both), to rounding on the longer ones"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(F)
    n = 3000
    src, dst = rand_graph(rng, n, 9000)
    src[:40] = 5                                                   # a row longer than the table (CSR continuation)
    gr = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    H = torch.randn(n, F, device=DEV)
    Y = torch.randn(n, F, device=DEV)
    b = torch.randn(F, device=DEV)
    ip, ix = gr.csr()
    tp, tx = gr.csc()
    nv = gr.norm() if norm else None
    def same(a, b, indptr, plan, long_row=None):
        """bit for bit on the rows that fit the packed table; a row that outgrows it is gathered by the whole wave in
        the table kernels (round 5) -- the same terms in another order than the row-group kernel's, which spmm_raw
        falls back to for narrow F: to rounding there"""
        fits = (indptr[1:] - indptr[:-1]) <= plan.ell_width
        assert int((~fits).sum()) < 60 and (long_row is None or not bool(fits[long_row]))
        assert torch.equal(a[fits], b[fits])
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())

    plain = ops.spmm_raw(ip, ix, ops.pad_rows(H), n, nv, nv, plan=gr.spmm_plan(False))
    out = ops.spmm_epilogue_raw(ip, ix, H, n, gr.spmm_plan(False), b, 1, None, nv, nv)
    same(out, torch.relu(plain + b), ip, gr.spmm_plan(False))
    out = ops.spmm_epilogue_raw(ip, ix, H, n, gr.spmm_plan(False), None, 0, None, nv, nv)
    same(out, plain, ip, gr.spmm_plan(False))
    gated = ops.spmm_epilogue_raw(tp, tx, H, n, gr.spmm_plan(True), None, 0, Y, nv, nv)
    ref = ops.spmm_raw(tp, tx, ops.pad_rows(H * (Y > 0)), n, nv, nv, plan=gr.spmm_plan(True))
    same(gated, ref, tp, gr.spmm_plan(True), long_row=5)


@pytest.mark.parametrize("hidden,F_in", [([32, 16], 300), ([24], 260), ([32, 20, 8], 500)])
@pytest.mark.parametrize("norm", ["none", "both"])
def test_transform_first_layer_matches_reference_order_oracle(hidden, F_in, norm):
    """the default ("auto") evaluation order of a wide layer 1, act(A (X W^T) + b), against the oracle's
    reference-order step act((A X) W^T + b) (gae.py:26-31): embeddings, loss and every parameter gradient within the
    fp32 tolerance -- and bit-identical state-dict keys / side effects on g.ndata"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from oracle import gae_oracle as O
    rng = np.random.default_rng(len(hidden) + F_in)
    n = 700
    src, dst = rand_graph(rng, n, 1600)
    X = (rng.random((n, F_in)) < 0.1).astype(np.float32) * rng.random((n, F_in)).astype(np.float32)
    torch.manual_seed(1)
    model = G.GAE(F_in, hidden, norm=norm).to(DEV)
    assert model.layers[0].transform_auto and not model.layers[0].transform_first
    model.decoder.dropout = 0.0
    gr = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    gr.ndata['h'] = torch.from_numpy(X).to(DEV)
    calls = []
    orig = ops.xw_fwd_raw
    ops.xw_fwd_raw = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        loss = model.reconstruction_loss(gr)
    finally:
        ops.xw_fwd_raw = orig
    assert len(calls) == 1, "the one-pass layer did not run"
    loss.backward()
    Ws = [l.apply_mod.linear.weight.detach().cpu().numpy() for l in model.layers]
    bs = [l.apply_mod.linear.bias.detach().cpu().numpy() for l in model.layers]
    nv = None
    if norm == "both":
        deg = np.bincount(dst, minlength=n).astype(np.float64)
        nv = np.where(deg > 0, deg ** -0.5, 0.0)
    ref_loss, Z, _, dW, db = O.gae_loss_and_grads(src, dst, n, X.astype(np.float64), [w.astype(np.float64) for w in Ws],
                                                  [b.astype(np.float64) for b in bs], norm=nv)
    assert rel(gr.ndata['h'], Z) < TOL and abs(float(loss) - float(ref_loss)) < TOL * abs(float(ref_loss))
    for l, w, b in zip(model.layers, dW, db):
        assert rel(l.apply_mod.linear.weight.grad, w) < 5 * TOL and rel(l.apply_mod.linear.bias.grad, b) < 5 * TOL
    # transform_first=False keeps the reference's order (and the dense kernels of that order agree too)
    torch.manual_seed(1)
    ref_model = G.GAE(F_in, hidden, norm=norm, transform_first=False).to(DEV)
    ref_model.decoder.dropout = 0.0
    gr.ndata['h'] = torch.from_numpy(X).to(DEV)
    z2 = ref_model.encode(gr)
    assert 'h' not in gr.ndata and rel(z2, Z) < TOL


@pytest.mark.parametrize("name", ["cora", "citeseer"])
def test_split_partials_are_added_by_the_gather(name):
    """a long f_in is split over thread blocks by gae_xw_fwd; with keep_splits the partial matrices stay and
    gae_spmm_csr_epilogue adds a gathered row's partials in split order: bit for bit the aggregation of the reduced P"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    n, src, dst, X = W.citation_graph(name, seed=0)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    Xd = ops.pad_rows(torch.from_numpy(X).to(DEV))
    Wt = torch.randn(32, X.shape[1], device=DEV) / X.shape[1] ** 0.5
    b = torch.randn(32, device=DEV)
    parts, splits = ops.xw_fwd_raw(Xd, Wt, None, 0, keep_splits=True)
    assert splits > 1 and parts.shape == (splits, n, 32)
    P = ops.xw_fwd_raw(Xd, Wt, None, 0)
    acc = parts[0].clone()
    for sp in range(1, splits):
        acc += parts[sp]
    assert torch.equal(acc, P)                                  # the reduction launch adds in split order too
    ip, ix = g.csr()
    for nv in (None, g.norm()):
        y_ref = ops.spmm_epilogue_raw(ip, ix, P, n, g.spmm_plan(False), b, 1, None, nv, nv)
        y = ops.spmm_epilogue_raw(ip, ix, parts, n, g.spmm_plan(False), b, 1, None, nv, nv)
        assert torch.equal(y, y_ref)


def test_transform_first_hidden_layer_input_gradient():
    """a narrowing HIDDEN layer (256 -> 16) also takes the one-pass route and hands the right gradient upstream"""
    import gae_dgl_amd as G
    from oracle import gae_oracle as O
    rng = np.random.default_rng(9)
    n, F_in, hidden = 500, 64, [256, 16]
    src, dst = rand_graph(rng, n, 1200)
    X = rng.standard_normal((n, F_in)).astype(np.float32)
    torch.manual_seed(2)
    model = G.GAE(F_in, hidden).to(DEV)
    model.decoder.dropout = 0.0
    gr = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    gr.ndata['h'] = torch.from_numpy(X).to(DEV)
    loss = model.reconstruction_loss(gr)
    loss.backward()
    Ws = [l.apply_mod.linear.weight.detach().cpu().double().numpy() for l in model.layers]
    bs = [l.apply_mod.linear.bias.detach().cpu().double().numpy() for l in model.layers]
    ref_loss, Z, _, dW, db = O.gae_loss_and_grads(src, dst, n, X.astype(np.float64), Ws, bs)
    assert rel(gr.ndata['h'], Z) < TOL
    for l, w, b in zip(model.layers, dW, db):
        assert rel(l.apply_mod.linear.weight.grad, w) < 5 * TOL and rel(l.apply_mod.linear.bias.grad, b) < 5 * TOL


def test_linear_forward_uses_the_stream_family():
    """gae_linear_fwd on a wide operand (the reference-order layer 1: Linear on M = A X) runs gae_xw_fwd's kernel: equal
    bit for bit; gae_linear_bwd and gae_xw_wgrad agree within the fp32 tolerance (different kernels, both kept)"""
    from gae_dgl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    n, K, J = 5000, 500, 32
    M = ops.pad_rows(torch.randn(n, K, generator=g).to(DEV))
    W = (torch.randn(J, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(J, generator=g).to(DEV)
    Y = ops.linear_fwd_raw(M, W, b, 1)
    assert torch.equal(Y, ops.xw_fwd_raw(M, W, b, 1))
    dY = torch.randn(n, J, generator=g).to(DEV)
    dW, db, dM = ops.linear_bwd_raw(dY, Y, 1, M, W, True, True, True)
    dW2, db2 = ops.xw_wgrad_raw(M, dY, Y, dY, Y, J)
    assert rel(dW, dW2) < TOL and rel(db, db2) < TOL
    gm = (dY * (Y > 0)).double()
    assert rel(dW, gm.t() @ M.double()) < TOL and rel(db, gm.sum(0)) < TOL and rel(dM, gm @ W.double()) < TOL


def test_deferred_grad_reductions_match_the_separate_launches():
    """ops.deferred_grad_reductions(): the weight-gradient kernels leave partial sums, gae_adam_step adds them inside the
    optimiser launch -- gradients (written back by that launch) and updated weights agree with the ordinary
    backward + step to fp32 rounding of another summation order; leaving the block with unconsumed partials raises"""
    import copy
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from gae_dgl_amd.optim import Adam
    rng = np.random.default_rng(3)
    n, F_in = 900, 320
    src, dst = rand_graph(rng, n, 2500)
    X = torch.from_numpy(rng.standard_normal((n, F_in)).astype(np.float32)).to(DEV)
    torch.manual_seed(4)
    m_a = G.GAE(F_in, [32, 16]).to(DEV)
    m_a.decoder.dropout = 0.0
    m_b = copy.deepcopy(m_a)
    o_a, o_b = Adam(m_a.parameters(), lr=1e-2), Adam(m_b.parameters(), lr=1e-2)
    gr = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    for step in range(3):
        gr.ndata['h'] = X
        la = m_a.reconstruction_loss(gr)
        o_a.zero_grad(); ops.backward(la); o_a.step()
        gr.ndata['h'] = X
        with ops.deferred_grad_reductions():
            lb = m_b.reconstruction_loss(gr)
            ops.backward(lb, list(m_b.parameters()))
            o_b.step()
        assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la))
        for pa, pb in zip(m_a.parameters(), m_b.parameters()):
            assert torch.equal(pb.grad, pa.grad), step         # Adam wrote the reduced gradient back: same order, same bits
            assert torch.equal(pb, pa), step
    assert o_a.steps_taken() == o_b.steps_taken() == 3
    with pytest.raises(ops.GaeHipError):
        with ops.deferred_grad_reductions():
            gr.ndata['h'] = X
            ops.backward(m_b.reconstruction_loss(gr), list(m_b.parameters()))      # no optimiser step inside


@pytest.mark.parametrize("p3", [1, 0])
def test_xw_fwd_every_knob_setting_launches_a_kernel_or_errors(p3, tuning):
    """the experiment knobs xw_dbg / xw_depth / xw_tc select among INSTANTIATED kernels only -- a combination
    without a kernel must come back as an argument error, never as GAE_OK over uninitialised output.  dbg 0 settings
    must equal the default launch; dbg 1..3 (partial kernels for timing experiments) must launch or raise."""
    from gae_dgl_amd import ops, _lib
    g = torch.Generator(device="cpu").manual_seed(7)
    n, K, J = 4100, 500, 32
    Xp = ops.pad_rows(torch.randn(n, K, generator=g).to(DEV))
    W = (torch.randn(J, K, generator=g) / K ** 0.5).to(DEV)
    tuning("xw_p3", p3)
    P0 = ops.xw_fwd_raw(Xp, W, None, 0)
    launched = refused = 0
    for dbg in (0, 1, 2, 3):
        for depth in (0, 2, 3, 4, 5, 7):
            for tc in (0, 3, 4, 6):
                tuning("xw_dbg", dbg); tuning("xw_depth", depth); tuning("xw_tc", tc)
                try:
                    P = ops.xw_fwd_raw(Xp, W, None, 0)
                except _lib.GaeHipError as e:
                    assert "knob combination" in str(e) or (dbg == 3 and "stamp buffer" in str(e)), str(e)
                    refused += 1
                    continue
                launched += 1
                if dbg == 0:
                    assert rel(P, P0) < 2e-6, (dbg, depth, tc)
    tuning("xw_dbg", 0); tuning("xw_depth", 0); tuning("xw_tc", 0)
    assert launched > 0 and refused > 0
