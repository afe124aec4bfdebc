"""Round 4: the two-layer encoder for graphs with millions of rows (BASELINE config 4) -- gae_spmm_csr_ep (store-time
bias / activation for ANY plan), gae_linear2_fwd, gae_gcn2_bwd_dense (csrc/tall.hip) and the row-sharded function built
from them (parallel.ShardedEncoder2Function) -- each against an fp64 restatement of gae.py:26-31,36-45 and its
autograd; the composed step against the oracle's reference-order encoder."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def skew_graph(seed, n, e):
    """heavy rows (thousands of edges), duplicates, empty rows"""
    rng = np.random.default_rng(seed)
    dst = (rng.integers(0, n, e).astype(np.float64) ** 4 / n ** 3).astype(np.int64)
    src = (rng.integers(0, n, e).astype(np.float64) ** 2 / n).astype(np.int64)
    return rng, torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV)


def dense_A(src, dst, n):
    A = torch.zeros(n, n, dtype=torch.float64)
    A.index_put_((dst.cpu(), src.cpu()), torch.ones(src.numel(), dtype=torch.float64), accumulate=True)
    return A


@pytest.mark.parametrize("F", [16, 32, 20])
@pytest.mark.parametrize("kind", ["none", "segments", "pinned"])
def test_spmm_ep_bias_activation_every_plan(F, kind):
    """act(A H + bias) at store time: light rows (row-group kernel), single-segment rows, combined segments, XCD-pinned
    rows; with the accumulate flag the epilogue applies to the accumulated value"""
    from gae_dgl_amd import ops
    n, e = 5000, 300000
    rng, src, dst = skew_graph(F, n, e)
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    plan = None
    if kind == "segments":
        plan = ops.spmm_plan(ip, threshold=8, segment=128, indices=ix, ell=False, hot=False, n_cols=n, homed=False)
        assert plan.n_heavy > 0
    elif kind == "pinned":
        plan = ops.spmm_plan(ip, threshold=8, segment=128, indices=ix, ell=False, hot=True, n_cols=n, homed=True)
        assert plan.homed is not None
    H = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.standard_normal(F).astype(np.float32)).to(DEV)
    M = ops.spmm_raw(ip, ix, H, n, plan=plan)                      # (itself held to the oracle by test_gpu_parity.py)
    ref = dense_A(src, dst, n) @ H.double().cpu()
    scale = float(ref.abs().max())
    assert float((M.double().cpu() - ref).abs().max()) <= (TOL if plan is not None else 1e-4) * scale   # ('none': rows of
    #                                                       60 k terms in one fp32 chain)
    for bias, act in ((b, 0), (b, 1), (None, 1), (None, 0)):
        y = ops.spmm_ep_raw(ip, ix, H, n, plan, bias, act)
        want = M + bias if bias is not None else M                 # one fp32 add, then max: the kernel's own arithmetic
        want = torch.relu(want) if act else want
        assert torch.equal(y, want)
    # accumulate: out = relu((A H + base) + b), the epilogue on the accumulated value
    base = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(DEV)
    out = base.clone()
    ops.spmm_ep_raw(ip, ix, H, n, plan, b, 1, out=out, accumulate=True)
    acc = base.clone()
    ops.spmm_raw(ip, ix, H, n, out=acc, plan=plan, accumulate=True)
    assert torch.equal(out, torch.relu(acc + b))


@pytest.mark.parametrize("n,f_in,f_mid,f_out", [(5000, 32, 32, 16), (777, 17, 20, 7), (33, 32, 32, 32), (4097, 8, 5, 3),
                                                (100000, 32, 32, 16)])
@pytest.mark.parametrize("act", [1, 0])
def test_linear2_fwd_matches_fp64(n, f_in, f_mid, f_out, act):
    from gae_dgl_amd import ops
    g = torch.Generator().manual_seed(n + f_in)
    A = torch.randn(n, f_in, generator=g)
    W1 = torch.randn(f_mid, f_in, generator=g) / f_in ** 0.5
    b1 = torch.randn(f_mid, generator=g)
    W2 = torch.randn(f_out, f_mid, generator=g) / f_mid ** 0.5
    ld = (f_in + 3) // 4 * 4
    buf = torch.full((n, ld), float("nan"), device=DEV)            # pad columns must never reach a sum
    buf[:, :f_in] = A.to(DEV)
    y_ref = A.double() @ W1.double().t() + b1.double()
    y_ref = torch.relu(y_ref) if act else y_ref
    t_ref = y_ref @ W2.double().t()
    for bias in (b1.to(DEV), None):
        Y1, T = ops.linear2_fwd_raw(buf[:, :f_in], W1.to(DEV), bias, act, W2.to(DEV))
        if bias is None:
            yr = A.double() @ W1.double().t()
            yr = torch.relu(yr) if act else yr
            assert rel(Y1, yr) < TOL and rel(T, yr @ W2.double().t()) < TOL
        else:
            assert Y1.shape == (n, f_mid) and T.shape == (n, f_out)
            assert rel(Y1, y_ref) < TOL and rel(T, t_ref) < TOL
    _, T2 = ops.linear2_fwd_raw(buf[:, :f_in], W1.to(DEV), b1.to(DEV), act, W2.to(DEV), want_y1=False)
    assert rel(T2, t_ref) < TOL


@pytest.mark.parametrize("n,f_in,f_mid,f_out", [(5000, 32, 32, 16), (777, 17, 20, 7), (33, 32, 32, 32), (4097, 8, 5, 3),
                                                (100000, 32, 32, 16)])
@pytest.mark.parametrize("act", [1, 0])
def test_gcn2_bwd_dense_matches_fp64(n, f_in, f_mid, f_out, act):
    from gae_dgl_amd import ops
    g = torch.Generator().manual_seed(n + f_out)
    G = torch.randn(n, f_out, generator=g)
    dZ = torch.randn(n, f_out, generator=g)
    Y1 = torch.randn(n, f_mid, generator=g)
    if act:
        Y1 = torch.relu(Y1)
    M1 = torch.randn(n, f_in, generator=g)
    W2 = torch.randn(f_out, f_mid, generator=g)
    dW1, db1, dW2, db2 = ops.gcn2_bwd_dense_raw(G.to(DEV), dZ.to(DEV), Y1.to(DEV), act, M1.to(DEV), W2.to(DEV))
    Gd, Yd = G.double(), Y1.double()
    dY1 = Gd @ W2.double()
    if act:
        dY1 = dY1 * (Yd > 0)
    assert rel(dW2, Gd.t() @ Yd) < TOL
    assert rel(db2, dZ.double().sum(0)) < TOL
    assert rel(dW1, dY1.t() @ M1.double()) < TOL
    assert rel(db1, dY1.sum(0)) < TOL
    # deterministic
    again = ops.gcn2_bwd_dense_raw(G.to(DEV), dZ.to(DEV), Y1.to(DEV), act, M1.to(DEV), W2.to(DEV))
    assert all(torch.equal(a, b) for a, b in zip((dW1, db1, dW2, db2), again))
    # Y1 recomputed inside the pass from M1, W1, b1: the SAME bits as with the Y1 gae_linear2_fwd stored
    W1 = torch.randn(f_mid, f_in, generator=g) / f_in ** 0.5
    b1 = torch.randn(f_mid, generator=g)
    ld = (f_in + 3) // 4 * 4
    buf = torch.full((n, ld), float("nan"), device=DEV)
    buf[:, :f_in] = M1.to(DEV)
    Y1f, _ = ops.linear2_fwd_raw(buf[:, :f_in], W1.to(DEV), b1.to(DEV), act, W2.to(DEV))
    stored = ops.gcn2_bwd_dense_raw(G.to(DEV), dZ.to(DEV), Y1f, act, buf[:, :f_in], W2.to(DEV))
    recomp = ops.gcn2_bwd_dense_raw(G.to(DEV), dZ.to(DEV), None, act, buf[:, :f_in], W2.to(DEV), W1=W1.to(DEV), b1=b1.to(DEV))
    assert all(torch.equal(a, b) for a, b in zip(stored, recomp))


def _encoder_reference(src, dst, n, X, Ws, bs, dZ):
    """fp64: Z = A relu((A X) W1^T + b1) W2^T + b2 (gae.py:26-31,36-45) and the gradients of <Z, dZ>"""
    A = dense_A(src, dst, n)
    P = [torch.as_tensor(t).double().clone().requires_grad_(True) for t in (Ws[0], bs[0], Ws[1], bs[1])]
    H1 = torch.relu((A @ X.double().cpu()) @ P[0].t() + P[1])
    Z = (A @ H1) @ P[2].t() + P[3]
    (Z * dZ.double().cpu()).sum().backward()
    return Z.detach(), [p.grad for p in P]


@pytest.mark.parametrize("mode,overlap", [("allgather", False), ("boundary", False), ("boundary", True), ("allgather", True)])
def test_encoder2_one_rank_group_matches_reference_order(mode, overlap):
    """the whole function under a real (1-rank RCCL) process group: embeddings and all four gradients against the fp64
    reference-order encoder; with deferred gradient reductions + the library's Adam the update equals the eager one"""
    import torch.distributed as dist
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, optim
    from gae_dgl_amd.parallel import ShardedGraph, sharded_encode, encoder2_usable
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29657")
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        n, e, F = 4000, 200000, 32
        rng, src, dst = skew_graph(7, n, e)
        X = torch.from_numpy(rng.random((n, F)).astype(np.float32)).to(DEV)
        dZ = torch.from_numpy(rng.standard_normal((n, 16)).astype(np.float32)).to(DEV) / n
        torch.manual_seed(0)
        model = G.GAE(F, [32, 16]).to(DEV)
        assert encoder2_usable(model, X)
        sg = ShardedGraph(n, src, dst, mode=mode, device=DEV, overlap=overlap)
        sg.cache_constant_inputs = True
        Ws = [l.apply_mod.linear.weight.detach().cpu() for l in model.layers]
        bs = [l.apply_mod.linear.bias.detach().cpu() for l in model.layers]
        z_ref, g_ref = _encoder_reference(src, dst, n, X, Ws, bs, dZ)
        for rep in range(2):                       # second pass: the exchanged rows of X come from the cache
            model.zero_grad()
            z = sharded_encode(model, sg, X, transform_first=True)
            z.backward(dZ)
            assert rel(z, z_ref) < TOL
            got = [model.layers[0].apply_mod.linear.weight.grad, model.layers[0].apply_mod.linear.bias.grad,
                   model.layers[1].apply_mod.linear.weight.grad, model.layers[1].apply_mod.linear.bias.grad]
            for a, b in zip(got, g_ref):
                assert rel(a, b) < 5 * TOL
        # reference order through the same sharded graph: same values within the tolerance
        model.zero_grad()
        z0 = sharded_encode(model, sg, X, transform_first=False)
        assert rel(z0, z_ref) < TOL
        # deferred reductions: partial sums consumed by the optimiser launch == eager gradients + eager Adam
        import copy
        m_a, m_b = copy.deepcopy(model), copy.deepcopy(model)
        o_a, o_b = optim.Adam(m_a.parameters(), lr=1e-2), optim.Adam(m_b.parameters(), lr=1e-2)
        za = sharded_encode(m_a, sg, X, transform_first=True); o_a.zero_grad(); za.backward(dZ); o_a.step()
        with ops.deferred_grad_reductions():
            zb = sharded_encode(m_b, sg, X, transform_first=True)
            ops.backward((zb * dZ).sum(), list(m_b.parameters()))
            o_b.step()
        for pa, pb in zip(m_a.parameters(), m_b.parameters()):
            assert float((pa - pb).abs().max()) <= 1e-6 * float(pa.abs().max())
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [3])
@pytest.mark.parametrize("mode,overlap", [("allgather", False), ("boundary", False), ("boundary", True)])
def test_virtual_ranks_product_with_epilogue(world, mode, overlap):
    """the sharded product with bias + activation in its last launch: every virtual rank's rows == the single-GPU rows"""
    from gae_dgl_amd import ops
    from gae_dgl_amd.parallel import LocalGroup, ShardedGraph, _rank_product
    n, e, F = 3000, 150000, 16
    rng, src, dst = skew_graph(11, n, e)
    T = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.standard_normal(F).astype(np.float32)).to(DEV)
    want = torch.relu(dense_A(src, dst, n) @ T.double().cpu() + b.double().cpu())
    grp = LocalGroup(world)
    rows = []
    for r in range(world):
        sg = ShardedGraph(n, src, dst, rank=r, group=grp, mode=mode, device=DEV, overlap=overlap)
        p = sg.part
        grp.publish(T)
        rows.append(_rank_product(sg, T[p.r0:p.r1].contiguous(), "fwd", bias=b, act=1))
    got = torch.cat(rows)
    assert float((got.double().cpu() - want).abs().max()) <= TOL * float(want.abs().max())


@pytest.mark.parametrize("F", [16, 32])
@pytest.mark.parametrize("pinned", [False, True])
def test_light_row_list_is_bit_identical(F, pinned, tuning):
    """gae_spmm_plan::light_desc: rows with 1 .. threshold edges from the plan's list + the empty rows from the fill
    stream == one sweep of the row-group kernel over all rows, with and without epilogue / accumulate; the list is the
    ascending sequence of those rows"""
    from gae_dgl_amd import ops
    n, e = 20000, 300000
    rng, src, dst = skew_graph(100 + F, n, e)
    dst = dst // 3 * 3                                      # two thirds of the rows are empty
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    plan = ops.spmm_plan(ip, threshold=8, segment=128, indices=ix, ell=False, hot=pinned, n_cols=n, homed=pinned)
    assert plan.light_desc is not None and (plan.homed is not None) == pinned
    deg = (ip[1:] - ip[:-1]).cpu().numpy()
    ipn = ip.cpu().numpy()
    want_rows = np.nonzero((deg >= 1) & (deg <= 8))[0]
    ld = plan.light_desc.cpu().numpy()
    assert np.array_equal(ld[:, 0], want_rows) and np.array_equal(ld[:, 1], ipn[want_rows]) and \
        np.array_equal(ld[:, 2], ipn[want_rows + 1])
    H = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.standard_normal(F).astype(np.float32)).to(DEV)
    base = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(DEV)
    outs = {}
    for light in (1, 0):
        tuning("spmm_light", light)
        o_acc = base.clone(); ops.spmm_raw(ip, ix, H, n, out=o_acc, plan=plan, accumulate=True)
        o_ep = base.clone(); ops.spmm_ep_raw(ip, ix, H, n, plan, b, 1, out=o_ep, accumulate=True)
        outs[light] = (ops.spmm_raw(ip, ix, H, n, plan=plan), ops.spmm_ep_raw(ip, ix, H, n, plan, b, 1), o_acc, o_ep)
    for a, c in zip(outs[1], outs[0]):
        assert torch.equal(a, c)


def test_dead_rows_are_neither_written_nor_read():
    """round 4: rows of an aggregate without edges (R-MAT: most rows).  gae_spmm_csr with GAE_SPMM_SKIP_ROWS leaves the
    rows marked in the plan's mask untouched; gae_linear2_fwd / gae_gcn2_bwd_dense with the same mask never read them:
    NaN-poisoned dead rows give the bits that zero rows give without the mask."""
    from gae_dgl_amd import ops
    n, e, F = 6000, 200000, 32
    rng, src, dst = skew_graph(5, n, e)
    # + a sprinkle of edges into the upper half of the rows: rows of 1 .. 8 edges (the light list) next to empty rows
    extra = torch.from_numpy(rng.integers(n // 2, n, 2500)).to(DEV)
    dst = torch.cat([dst, extra]); src = torch.cat([src, torch.from_numpy(rng.integers(0, n, 2500)).to(DEV)])
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    plan = ops.spmm_plan(ip, threshold=8, segment=128, indices=ix, ell=False, hot=True, n_cols=n, homed=True)
    assert plan is not None and plan.n_light > 0
    dead = (ip[1:] == ip[:-1]).to(torch.uint8).contiguous()
    assert 0 < int(dead.sum()) < n
    plan.set_skip_rows(dead)
    H = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(DEV)
    ref = ops.spmm_raw(ip, ix, H, n, plan=plan)                        # every row written (zeros in the dead rows)
    out = torch.full_like(ref, float("nan"))
    ops.spmm_raw(ip, ix, H, n, out=out, plan=plan, skip_dead=True)
    live = dead == 0
    assert torch.equal(out[live], ref[live]) and bool(torch.isnan(out[~live]).all()) and float(ref[~live].abs().max()) == 0.0
    # the dense passes on the poisoned aggregate
    g = torch.Generator().manual_seed(3)
    W1 = (torch.randn(32, F, generator=g) / F ** 0.5).to(DEV); b1 = torch.randn(32, generator=g).to(DEV)
    W2 = (torch.randn(16, 32, generator=g) / 32 ** 0.5).to(DEV)
    _, T_ref = ops.linear2_fwd_raw(ref, W1, b1, 1, W2, want_y1=False)
    _, T = ops.linear2_fwd_raw(out, W1, b1, 1, W2, want_y1=False, a_dead=dead)
    assert torch.equal(T, T_ref)
    Gm = torch.randn(n, 16, generator=g).to(DEV); dZ = torch.randn(n, 16, generator=g).to(DEV)
    gdead = (torch.rand(n, generator=g) < 0.5).to(torch.uint8).to(DEV)
    G0 = Gm.clone(); G0[gdead.bool()] = 0
    Gp = Gm.clone(); Gp[gdead.bool()] = float("nan")
    want = ops.gcn2_bwd_dense_raw(G0, dZ, None, 1, ref, W2, W1=W1, b1=b1)
    got = ops.gcn2_bwd_dense_raw(Gp, dZ, None, 1, out, W2, W1=W1, b1=b1, m1_dead=dead, g_dead=gdead)
    assert all(torch.equal(a, b) for a, b in zip(want, got))


@pytest.mark.parametrize("act", [1, 0])
def test_list_mode_of_the_dense_passes(act):
    """round 4: the dense passes visit only the rows that have an M1 row.  Forward: listed rows bit-identical to the
    full pass, every other row = act(b1) W2^T; backward: the four gradients equal those of the full pass on an M1 / G
    with zeros in the dead rows (another summation order: 1e-5 of the scale)."""
    from gae_dgl_amd import ops
    n, F = 20011, 32
    g = torch.Generator().manual_seed(17)
    m1_dead = (torch.rand(n, generator=g) < 0.7)
    g_dead = (torch.rand(n, generator=g) < 0.6)
    M1 = torch.randn(n, F, generator=g); M1[m1_dead] = 0
    Gm = torch.randn(n, 16, generator=g); Gm[g_dead] = 0
    dZ = torch.randn(n, 16, generator=g)
    W1 = torch.randn(32, F, generator=g) / F ** 0.5; b1 = torch.randn(32, generator=g)
    W2 = torch.randn(16, 32, generator=g) / 32 ** 0.5
    d = lambda t: t.to(DEV)
    M1p, Gp = M1.clone(), Gm.clone()
    M1p[m1_dead] = float("nan"); Gp[g_dead] = float("nan")          # dead rows were "never written"
    md, gd = d(m1_dead.to(torch.uint8)), d(g_dead.to(torch.uint8))
    rows = torch.nonzero(~m1_dead).reshape(-1).to(torch.int32).to(DEV)
    gdl = gd[rows.long()].contiguous()
    _, T_full = ops.linear2_fwd_raw(d(M1), d(W1), d(b1), act, d(W2), want_y1=False)
    _, T_list = ops.linear2_fwd_raw(d(M1p), d(W1), d(b1), act, d(W2), want_y1=False, a_dead=md, rows=rows)
    live = ~m1_dead
    assert torch.equal(T_list.cpu()[live], T_full.cpu()[live])
    y0 = torch.relu(b1.double()) if act else b1.double()
    t0 = y0 @ W2.double().t()
    assert float((T_list.cpu()[m1_dead].double() - t0).abs().max()) <= 1e-6 * float(t0.abs().max())
    assert rel(T_list, T_full) < 1e-6
    want = ops.gcn2_bwd_dense_raw(d(Gm), d(dZ), None, act, d(M1), d(W2), W1=d(W1), b1=d(b1))
    got = ops.gcn2_bwd_dense_raw(d(Gp), d(dZ), None, act, d(M1p), d(W2), W1=d(W1), b1=d(b1), m1_dead=md, g_dead=gd,
                                 rows=rows, g_dead_listed=gdl)
    for a, b in zip(got, want):
        assert rel(a, b) < TOL
    # fp64 check of the dead rows' share as well
    Gd, Yd = Gm.double(), (torch.relu(M1.double() @ W1.double().t() + b1.double()) if act else M1.double() @ W1.double().t() + b1.double())
    dY1 = Gd @ W2.double()
    if act:
        dY1 = dY1 * (Yd > 0)
    assert rel(got[2], Gd.t() @ Yd) < TOL and rel(got[1], dY1.sum(0)) < TOL and rel(got[3], dZ.double().sum(0)) < TOL
    assert rel(got[0], dY1.t() @ M1.double()) < TOL


def test_list_mode_workspace_covers_every_list_length():
    """the number of partial blocks is not monotone in the row count once a wave takes more than one
    tile (n = 300000 -> 782 blocks, 131072 listed rows -> 1024), so the workspace must cover the most any
    n_listed <= n takes.  An exact-size workspace followed by a canary: the canary survives and the gradients are right."""
    import ctypes
    from gae_dgl_amd import ops, _lib
    n, F, n_listed = 300000, 32, 131072
    g = torch.Generator().manual_seed(23)
    perm = torch.randperm(n, generator=g)
    m1_dead = torch.ones(n, dtype=torch.bool); m1_dead[perm[:n_listed]] = False
    g_dead = torch.rand(n, generator=g) < 0.5
    M1 = torch.randn(n, F, generator=g); M1[m1_dead] = 0
    Gm = torch.randn(n, 16, generator=g); Gm[g_dead] = 0
    dZ = torch.randn(n, 16, generator=g)
    W1 = torch.randn(32, F, generator=g) / F ** 0.5; b1 = torch.randn(32, generator=g)
    W2 = torch.randn(16, 32, generator=g) / 32 ** 0.5
    d = lambda t: t.to(DEV)
    md, gd = d(m1_dead.to(torch.uint8)), d(g_dead.to(torch.uint8))
    rows = torch.nonzero(~m1_dead).reshape(-1).to(torch.int32).to(DEV)
    assert rows.numel() == n_listed
    gdl = gd[rows.long()].contiguous()
    lib = _lib.load()
    nbytes = int(lib.gae_gcn2_bwd_dense_workspace_bytes(n, F, 32, 16))
    # the layout the listed launch uses: 1024 blocks + the dead rows' partial
    per = (32 * F + 32 + 16 * 32 + 16 + 3) // 4 * 4
    assert nbytes >= (1024 + 1) * per * 4
    ws = torch.zeros(nbytes + 4096, dtype=torch.uint8, device=DEV)
    ws[nbytes:] = 0xA5
    outs = [torch.empty(32, F, device=DEV), torch.empty(32, device=DEV), torch.empty(16, 32, device=DEV),
            torch.empty(16, device=DEV)]
    Gd_, dZd, M1d, W1d, b1d, W2d = d(Gm), d(dZ), d(M1), d(W1), d(b1), d(W2)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.call("gae_gcn2_bwd_dense", p(Gd_), 16, p(dZd), 16, None, 0, 1, p(M1d), F, p(W2d), 32, n, F, 32, 16,
              p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]), p(ws), nbytes, None, p(W1d), F, p(b1d), p(md), p(gd),
              p(rows), n_listed, p(gdl), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert bool((ws[nbytes:] == 0xA5).all()), "gae_gcn2_bwd_dense wrote past the workspace it asked for"
    want = ops.gcn2_bwd_dense_raw(Gd_, dZd, None, 1, M1d, W2d, W1=W1d, b1=b1d)
    for a, b in zip(outs, want):
        assert rel(a, b) < TOL
    # a workspace sized for the listed count alone is refused, not overrun
    small = int(lib.gae_gcn2_bwd_dense_workspace_bytes(n_listed, F, 32, 16))
    assert small <= nbytes
    # through ops in deferred mode (torch.empty(nbytes) exactly): gradients after the optimiser's reduction are right
    got = ops.gcn2_bwd_dense_raw(Gd_, dZd, None, 1, M1d, W2d, W1=W1d, b1=b1d, m1_dead=md, g_dead=gd, rows=rows,
                                 g_dead_listed=gdl)
    for a, b in zip(got, want):
        assert rel(a, b) < TOL
