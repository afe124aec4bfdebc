"""Graphs with the degree profile of the REAL citation graphs (round 5).

workloads.citation_graph's default pairs nodes uniformly (longest row ~ 16); the Planetoid graphs a user of the reference
loads (train_transductive.py:37-38) have hubs: Cora 168, Citeseer 99, Pubmed 171 neighbours.  Their plans stay
table-only (ops.TABLE_MAX_ROW): a row that outgrows the 16 slots of the packed table is gathered by the whole wave
(spmm_ell.hip: ell_long_row), in every form of the table kernels -- plain, scaled, ReLU-gated, split-sum, with the fused
dense epilogue --, and the fused loss's edge kernel walks rows of more than 16 edges with the whole wave as well.  Here:
each of those forms on rows of 17 .. 300 edges against fp64, and the whole training step of the default scripts' layer
orders on Planetoid-profile graphs against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def O():
    from oracle import gae_oracle
    return gae_oracle


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def hub_graph(n, e, rng, lengths=(300, 156, 93, 79, 65, 64, 33, 17), symmetric=False):
    """random directed multigraph plus rows (in-edges) of the given lengths at nodes 3, 13, 23, ... and, mirrored, as
    many long columns (out-edges) at nodes 5, 15, ...: both CSRs have rows beyond the table"""
    src = rng.integers(0, n, e); dst = rng.integers(0, n, e)
    for k, L in enumerate(lengths):
        src = np.concatenate([src, rng.choice(n, L, replace=False)]); dst = np.concatenate([dst, np.full(L, 3 + 10 * k)])
        dst = np.concatenate([dst, rng.choice(n, L, replace=False)]); src = np.concatenate([src, np.full(L, 5 + 10 * k)])
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    return src.astype(np.int64), dst.astype(np.int64)


@pytest.mark.parametrize("n", [700, 5000])
def test_auto_plan_policy(n):
    """graphs that get a packed table keep a table-only plan up to TABLE_MAX_ROW edges in a row; beyond it -- and on
    graphs too large for a table at SKEW_MIN_MAXDEG -- the skew plan takes the rows above its threshold"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(n)
    for longest, table_only in ((ops.TABLE_MAX_ROW, True), (ops.TABLE_MAX_ROW + 1, False)):
        if longest > n:
            continue
        src, dst = hub_graph(n, 4 * n, rng, lengths=())
        dst = dst[dst != 7]; src = src[:dst.size]
        src = np.concatenate([src, rng.choice(n, longest, replace=False)]); dst = np.concatenate([dst, np.full(longest, 7)])
        g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
        plan = g.spmm_plan(False)
        assert plan is not None and plan.ell is not None
        assert (plan.n_heavy == 0) == table_only, (longest, plan.n_heavy)
        if not table_only:
            assert int(plan.ell.view(n, plan.ell_width)[7, 0]) == -3           # skip marker: the segment kernels' row


@pytest.mark.parametrize("F,dtype", [(16, torch.float32), (32, torch.float32), (39, torch.float32), (64, torch.float32),
                                     (100, torch.float32), (500, torch.float32), (32, torch.bfloat16), (500, torch.bfloat16)])
@pytest.mark.parametrize("scaled", [False, True])
@pytest.mark.parametrize("e", [1500, 9000, 30000])
def test_table_kernels_gather_long_rows(F, dtype, scaled, e):
    """M = diag(rs) A diag(cs) H through the auto (table-only) plan, lane groups of 8 .. 64 lanes, tables of 4 / 8 / 16
    slots (e), both directions"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(F)
    n = 3000
    src, dst = hub_graph(n, e, rng)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    H = rng.standard_normal((n, F)).astype(np.float32)
    Hd = ops.pad_rows(torch.from_numpy(H).to(DEV).to(dtype))
    Href = Hd[:, :F].double().cpu().numpy()
    A = O().dense_adjacency(src, dst, n, dtype=torch.float64).numpy()
    sc = rng.random(n).astype(np.float32) + 0.5
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-6
    for transposed in (False, True):
        ip, ix = g.csc() if transposed else g.csr()
        plan = g.spmm_plan(transposed)
        assert plan.n_heavy == 0 and plan.ell is not None and plan.ell_width <= {1500: 4, 9000: 16, 30000: 16}[e]
        scd = torch.from_numpy(sc).to(DEV) if scaled else None
        out = ops.spmm_raw(ip, ix, Hd, n, scd, scd, plan=plan)
        Am = A.T if transposed else A
        want = (sc[:, None] * (Am @ (sc[:, None] * Href))) if scaled else Am @ Href
        assert rel(out[:, :F].float(), want) < tol, (F, dtype, scaled, transposed)


@pytest.mark.parametrize("d,e", [(2, 3000), (8, 3000), (16, 9000), (16, 0), (32, 4000), (64, 2500)])
def test_fused_loss_edge_kernel_walks_long_rows_with_the_whole_wave(d, e):
    """lane groups of 1 .. 16 lanes per node (d = 2 .. 64), in- and out-lists of 17 .. 300 edges, ragged tails of the
    64 / 32 / 16-edge trips; a graph that has ONLY long rows (e = 0)"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(d + e)
    n = 1500
    src, dst = hub_graph(n, e, rng)
    Z = (rng.standard_normal((n, d)) * 0.6).astype(np.float32)
    mask = ((rng.random((n, d)) >= 0.1) / 0.9).astype(np.float32)
    adj = O().dense_adjacency(src, dst, n, dtype=torch.float64)
    pw = O().pos_weight_of(adj)
    Zt = torch.tensor(Z, dtype=torch.float64, requires_grad=True)
    ref = O().bce_with_logits_mean(O().decoder_logits(Zt, torch.tensor(mask, dtype=torch.float64)), adj, pw)
    ref.backward()
    gr = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    Zd = torch.from_numpy(Z).to(DEV).requires_grad_(True)
    loss = ops.decoder_bce(Zd, torch.from_numpy(mask).to(DEV), gr)
    assert abs(float(loss) - float(ref)) < 2e-5 * abs(float(ref))
    loss.backward()
    assert rel(Zd.grad, Zt.grad) < 1e-4
    with torch.no_grad():
        assert abs(float(ops.decoder_bce(torch.from_numpy(Z).to(DEV), torch.from_numpy(mask).to(DEV), gr)) - float(ref)) \
            < 2e-5 * abs(float(ref))


@pytest.mark.parametrize("name", ["cora", "citeseer"])
@pytest.mark.parametrize("layer1", ["sparse", "transform-first", "reference"])
def test_training_step_on_planetoid_degree_graphs_matches_oracle(name, layer1):
    """loss and every parameter gradient of the two-layer model on a graph with the real graph's hubs, through each
    layer-1 form the scripts choose from (the non-zeros of X; X W^T first; the reference's order): the table kernels'
    plain, ReLU-gated, split-sum and fused-epilogue forms all meet the long rows here"""
    import gae_dgl_amd as G
    from gae_dgl_amd import gae as gae_mod, ops, workloads as W
    n, src, dst, X = W.citation_graph(name, seed=0, degrees="planetoid")
    assert np.bincount(dst, minlength=n).max() > 64
    torch.manual_seed(0)
    model = G.GAE(X.shape[1], [32, 16]).to(DEV)
    model.decoder.dropout = 0.0
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    assert g.spmm_plan(False).n_heavy == 0 and g.spmm_plan(True).n_heavy == 0
    Xd = ops.pad_rows(torch.from_numpy(X).to(DEV))
    feats = Xd
    if layer1 == "sparse":
        feats = G.SparseFeatures.maybe_from_dense(Xd, 32, graph=g)
        assert isinstance(feats, G.SparseFeatures)
    old = gae_mod.TRANSFORM_FIRST_AUTO
    gae_mod.TRANSFORM_FIRST_AUTO = layer1 != "reference"
    try:
        g.ndata['h'] = feats
        loss = model.reconstruction_loss(g)
        loss.backward()
    finally:
        gae_mod.TRANSFORM_FIRST_AUTO = old
    Ws = [l.apply_mod.linear.weight.detach().cpu().numpy() for l in model.layers]
    bs = [l.apply_mod.linear.bias.detach().cpu().numpy() for l in model.layers]
    ref_loss, _, _, dW, db = O().gae_loss_and_grads(src, dst, n, X.astype(np.float64), Ws, bs)
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
    for k, (l, w, b) in enumerate(zip(model.layers, dW, db)):
        assert rel(l.apply_mod.linear.weight.grad, w) < 2e-5, (k, "W")
        assert rel(l.apply_mod.linear.bias.grad, b) < 2e-5, (k, "b")


def test_captured_step_on_a_planetoid_degree_graph_equals_eager_steps():
    """the default script's captured step (HIP graph replay) on the hubbed Cora profile: same losses as eager steps"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    n, src, dst, X = W.citation_graph("cora", seed=0, degrees="planetoid")
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    Xd = ops.pad_rows(torch.from_numpy(X).to(DEV))

    def run(captured):
        torch.manual_seed(0)
        model = G.GAE(X.shape[1], [32, 16]).to(DEV)
        model.decoder.dropout = 0.0
        opt = Adam(model.parameters(), lr=1e-2)
        out = []
        if captured:
            step = CapturedTrainStep(model, opt, g, Xd, warmup=0)
            return [float(step()) for _ in range(5)]
        for _ in range(5):
            g.ndata['h'] = Xd
            loss = model.reconstruction_loss(g); opt.zero_grad(); ops.backward(loss); opt.step()
            out.append(float(loss.detach()))
        return out
    np.testing.assert_allclose(run(True), run(False), rtol=1e-5)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_vgae_step_on_a_planetoid_degree_graph_matches_oracle(dtype, tol):
    """the VGAE step (fused mu / log-sigma heads: gae_x_gcn_layer_fused2, sampled decoder, BCE + KL) on the hubbed
    Citeseer profile, fp32 and bf16-stored features, against the CPU restatement"""
    import gae_dgl_amd as G
    from gae_dgl_amd import workloads as W
    from gae_dgl_amd.vgae import VGAE
    n, src, dst, X = W.citation_graph("citeseer", seed=0, degrees="planetoid")
    torch.manual_seed(0)
    model = VGAE(X.shape[1], [32, 16], seed=11).to(DEV)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    assert g.spmm_plan(False).n_heavy == 0 and int((g.csr()[0][1:] - g.csr()[0][:-1]).max()) > 64
    Xd = torch.from_numpy(X).to(DEV).to(dtype)
    g.ndata['h'] = Xd
    loss = model.loss(g)
    loss.backward()
    last = {k: v.detach().cpu() for k, v in model.last.items()}
    Xo = Xd.float().cpu()
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    ip, ix = O().csr_from_coo(src, dst, n)
    mu, ls, z = O().vgae_forward(ip, ix, Xo, P["shared.apply_mod.linear.weight"], P["shared.apply_mod.linear.bias"],
                                 P["mu_head.apply_mod.linear.weight"], P["mu_head.apply_mod.linear.bias"],
                                 P["logstd_head.apply_mod.linear.weight"], P["logstd_head.apply_mod.linear.bias"],
                                 last["eps"])
    adj = O().dense_adjacency(src, dst, n)
    rec = O().bce_with_logits_mean(z @ z.t(), adj, O().pos_weight_of(adj))
    kl = O().vgae_kl(mu, ls)
    (rec + kl).backward()

    def rel1(a, b):      # (as tests/test_gpu_parity.py: against max(|b|, 1))
        a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
        return float((a - b).abs().max() / b.abs().max().clamp(min=1.0))
    assert rel1(last["mu"], mu) < tol and rel1(last["logstd"], ls) < tol and rel1(last["z"], z) < tol
    assert rel1(loss, rec + kl) < max(tol, 1e-5)
    for k, p in model.named_parameters():
        assert rel1(p.grad, P[k].grad) < 10 * tol, k
