"""BASELINE.json configurations at their FULL size (the two VERDICT r01 listed as bench-only):

* configs[2]  ZINC-250k inductive, batch = 4096 molecules: one training step through the on-device collate, the
  encoder and the fused 9e9-logit loss (symmetric 256-row-panel kernel), checked on sampled 256-row windows against
  the fp64 oracle (train_inductive.py:44-48) and against the full-square kernel;
* configs[3]  RMAT scale 24 / edge factor 16, F = 32: the SpMM with the automatic degree-skew plan against the plain
  CSR-order launch (bit-identical on light rows) and against oracle/spmm_ref.c on sampled heavy rows.

Sizes the CPU oracle cannot finish are covered through row windows / row samples; everything goes through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return torch.device("cuda:0")


def O():
    from oracle import gae_oracle
    return gae_oracle


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def test_zinc_batch4096_training_step(dev):
    """BASELINE configs[2]: N ~ 95 k nodes, 9e9 logits.  Encoder output vs the fp64 oracle (CSR path, all rows);
    fused loss: three 256-row windows of (loss share, dZ) vs the fp64 oracle restricted to those rows, the
    symmetric RI = 4 panel kernel (what the step runs) vs the full-square kernel, window launches of
    gae_decoder_bce_rows vs the oracle's loss shares; parameter gradients finite and reproducible."""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, _lib
    from gae_dgl_amd.dataset import DeviceGraphDataset
    B = 4096
    ds = DeviceGraphDataset.synthetic_zinc(B, seed=0, device=dev)
    bg = ds.batch(np.arange(B))
    n, e = bg.number_of_nodes(), bg.number_of_edges()
    assert 90000 < n < 100000 and 190000 < e < 220000
    torch.manual_seed(0)
    model = G.GAE(39, [32, 16]).to(dev)
    model.decoder.seed = 123                         # dropout 0.1 stays on (gae.py:70), drawn inside the fused launch
    X = bg.ndata['h'].clone()
    loss = model.reconstruction_loss(bg)
    ops.backward(loss)
    Z = bg.ndata['h'].detach()                       # GAE.forward leaves the embedding on the graph (gae.py:53)
    mask = model.decoder.last_mask
    assert Z.shape == (n, 16) and mask.shape == (n, 16)
    grads = [p.grad.clone() for p in model.parameters()]
    assert all(bool(torch.isfinite(g).all()) for g in grads) and float(loss) > 0

    # ---- encoder at full size vs the fp64 oracle (sparse CSR path: cheap on the CPU)
    ip, ix = (t.cpu().numpy() for t in bg.csr())
    tp, tx = (t.cpu().numpy() for t in bg.csc())
    Ws = [l.apply_mod.linear.weight.detach().double().cpu() for l in model.layers]
    bs = [l.apply_mod.linear.bias.detach().double().cpu() for l in model.layers]
    Zref = O().gae_encode(ip, ix, X[:, :39].double().cpu(), Ws, bs)
    assert rel_err(Z, Zref) < TOL

    # ---- fused loss of the step (symmetric kernel, 256-row panels) on (Z, mask): loss + dZ
    pw = (float(n) * n - e) / e                      # train_inductive.py:46
    Zd = Z.clone().requires_grad_(True)
    l_sym = ops.decoder_bce(Zd, mask, bg)
    ops.backward(l_sym)
    assert torch.equal(l_sym.detach(), loss.detach())        # same launch as inside the step: deterministic
    dZ_sym = Zd.grad.clone()
    # full-square kernel on the same operands
    _lib.call("gae_tuning_set", b"bce_sym", 0)
    try:
        Zf = Z.clone().requires_grad_(True)
        l_full = ops.decoder_bce(Zf, mask, bg)
        ops.backward(l_full)
    finally:
        _lib.call("gae_tuning_set", b"bce_sym", 1)
    assert rel_err(l_sym, l_full) < 2e-6
    assert rel_err(dZ_sym, Zf.grad) < TOL

    # ---- 256-row windows against the fp64 oracle (dense only inside the window: 256 x 95 k logits)
    Zt = (Z * mask).double().cpu()
    rng = np.random.default_rng(0)
    starts = [0, int(rng.integers(256, n - 512)) // 256 * 256 + 37, n - 256]
    dscale = float(dZ_sym.abs().max())
    for r0 in starts:
        r1 = r0 + 256
        share, gt = O().bce_row_window(Zt, r0, r1, ip, ix, tp, tx, pw)
        g = gt * mask[r0:r1].double().cpu()          # dLoss/dZ = dLoss/dZt * mask
        err = float((dZ_sym[r0:r1].double().cpu() - g).abs().max()) / dscale
        assert err < 5 * TOL, (r0, err)
        # the row-window entry point (the row-sharded form of the same loss): loss share of exactly these rows
        wip = torch.as_tensor(ip[r0:r1 + 1] - ip[r0], device=dev, dtype=torch.int32)
        wix = torch.as_tensor(ix[ip[r0]:ip[r1]], device=dev, dtype=torch.int32)
        wtp = torch.as_tensor(tp[r0:r1 + 1] - tp[r0], device=dev, dtype=torch.int32)
        wtx = torch.as_tensor(tx[tp[r0]:tp[r1]], device=dev, dtype=torch.int32)
        lw, gw = ops.decoder_bce_raw(Z, mask, (wip, wix), (wtp, wtx), pw, want_grad=True, row_begin=r0, n_local=256)
        assert abs(float(lw) - float(share)) < TOL * max(float(share), 1e-30) * 10, (r0, float(lw), float(share))
        assert float((gw.double().cpu() - g).abs().max()) / dscale < 5 * TOL


@pytest.mark.parametrize("scale", [24])
def test_rmat_spmm_skew_plan_full_size(scale, dev):
    """BASELINE configs[3] on one GPU: 2^24 nodes, 2^28 directed edges (duplicates kept), F = 32.  The launch with the
    automatic skew plan (what bench.py times) equals the plain CSR-order launch bit for bit on every light row;
    sampled rows (the 64 heaviest, 4000 heavy, 4000 light): the plain launch equals oracle/spmm_ref.c bit for bit
    (same CSR order), the plan launch is within 1e-5 of the oracle's double-accumulator sum on EVERY sampled row
    (hubs of 10^5 in-edges included: segment sums are closer to the exact sum than one fp32 chain, which drifts by
    up to 1e-4 there) and within 1e-5 of the fp32 CSR-order oracle on rows of up to 4096 edges; A 1 = in-degree
    exactly; run-to-run bit-stable."""
    from gae_dgl_amd import ops, workloads as W
    from oracle import c_oracle
    n, F = 1 << scale, 32
    src, dst = W.rmat_edges(scale, 16, seed=0, device=dev)
    e = int(src.numel())
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    del src, dst
    torch.cuda.empty_cache()
    assert int(ip[-1]) == e == 16 << scale
    deg, _ = ops.degree_norm(ip)
    plan = ops.spmm_plan(ip)
    assert plan is not None and plan.n_heavy > 0
    gen = torch.Generator(device=dev).manual_seed(5)
    H = torch.rand(n, F, device=dev, generator=gen)
    out_plan = ops.spmm_raw(ip, ix, H, n, plan=plan)
    out_csr = ops.spmm_raw(ip, ix, H, n)
    light = deg <= ops.SKEW_THRESHOLD
    assert torch.equal(out_plan[light], out_csr[light])
    heavy = ~light
    mid = heavy & (deg <= 4096)
    scale_ = out_csr[mid].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
    assert float(((out_plan[mid] - out_csr[mid]).abs() / scale_).max()) < TOL
    again = ops.spmm_raw(ip, ix, H, n, plan=plan)
    assert torch.equal(again, out_plan)                                   # no atomics: run-to-run bit-stable
    ones = ops.spmm_raw(ip, ix, torch.ones(n, F, device=dev), n, plan=plan)
    assert torch.equal(ones[:, 0], deg.float()) and torch.equal(ones[:, F - 1], deg.float())
    del ones, again
    # ---- sampled rows (the heaviest, random heavy, random light) against the C oracle: sub-CSR of those rows, all
    #      columns (H goes to the host once: 2 GiB)
    order = torch.argsort(deg, descending=True)
    g2 = torch.Generator(device=dev).manual_seed(6)
    hv = torch.nonzero(heavy).flatten()
    lt = torch.nonzero(light).flatten()
    rows = torch.cat([order[:64], hv[torch.randint(0, hv.numel(), (4000,), device=dev, generator=g2)],
                      lt[torch.randint(0, lt.numel(), (4000,), device=dev, generator=g2)]]).unique()
    d = deg[rows].long()
    sub_ip = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=dev)
    sub_ip[1:] = torch.cumsum(d, 0)
    starts = ip[rows].long()
    pos = torch.arange(int(sub_ip[-1]), device=dev) - torch.repeat_interleave(sub_ip[:-1], d) + \
        torch.repeat_interleave(starts, d)
    sub_ix = ix[pos]
    ref = c_oracle.spmm_csr(sub_ip.cpu().numpy().astype(np.int32), sub_ix.cpu().numpy(), H.cpu().numpy())
    got_csr = out_csr[rows].cpu().numpy()
    assert np.array_equal(got_csr, ref)                                   # plain launch: CSR order, bit-exact
    ref64 = c_oracle.spmm_csr_acc64(sub_ip.cpu().numpy().astype(np.int32), sub_ix.cpu().numpy(), H.cpu().numpy())
    got = out_plan[rows].cpu().numpy()
    sc = np.maximum(np.abs(ref64).max(axis=1, keepdims=True), 1.0)
    assert float((np.abs(got - ref64) / sc).max()) < TOL
    assert float((np.abs(ref - ref64) / sc).max()) < 2e-4                # what one fp32 chain of 10^5 terms costs
    assert int(d.max()) > 100000                                          # the sample does contain the hubs


def _aggregate64(rows, cols, H64, n, chunk=1 << 24):
    """out[r] = sum over edges (r, c) of H64[c] in fp64 with ATen's index_add_ (independent of the library's kernels)"""
    out = torch.zeros(n, H64.shape[1], dtype=torch.float64, device=H64.device)
    for c0 in range(0, int(rows.numel()), chunk):
        out.index_add_(0, rows[c0:c0 + chunk], H64[cols[c0:c0 + chunk]])
    return out


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("scale", [24])
def test_rmat_rank_step_full_size(scale, dev):
    """BASELINE configs[3], the STEP `bench.py --workload rmat` times, at the size it times it: the
    one-pass encoder of parallel.ShardedEncoder2Function on 2^24 rows / 2^28 edges under a real (1-rank RCCL) process
    group, structure from the edge slice, boundary exchange + overlap + nnz-balanced blocks as bench.py builds it --
    gae_spmm_csr with GAE_SPMM_SKIP_ROWS, list-mode gae_linear2_fwd + gae_linear2_fill_dead, gae_spmm_csr_ep with the bias,
    backward: gae_spmm_csr on A^T with skipped rows, list-mode gae_gcn2_bwd_dense with the dead rows' rank-one terms.
    Against the reference order (gae.py:26-31,36-45) evaluated in fp64 with plain ATen ops:
      * Z on EVERY row to 1e-5 of the largest |Z|, and on 8000 sampled rows (the heaviest, random heavy, random light,
        rows without in-edges) to 1e-5 of the row's own scale (sum of |terms|);
      * dW1, db1, dW2, db2 to 5e-5 of each gradient's scale; with atb_bf16 = 0 (exact fp32 products) the same;
      * run-to-run bit-stable (forward and backward)."""
    import os
    import torch.distributed as dist
    import gae_dgl_amd as G
    from gae_dgl_amd import _lib, ops, workloads as W
    from gae_dgl_amd.parallel import ShardedGraph, sharded_encode, encoder2_usable
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        n, F = 1 << scale, 32
        src, dst = W.rmat_edges(scale, 16, seed=0, device=dev)
        E = int(src.numel())
        sg = ShardedGraph.from_edge_slice(n, src, dst, None, "boundary", dev, "nnz", overlap=True)
        sg.cache_constant_inputs = True
        gen = torch.Generator(device=dev).manual_seed(1234)
        X = torch.rand(n, F, device=dev, generator=gen)
        dZ = torch.randn(n, 16, device=dev, generator=gen) / n
        torch.manual_seed(0)
        model = G.GAE(F, [32, 16]).to(dev)
        assert encoder2_usable(model, X)
        seen = {}
        inner = _lib.call

        def call(name, *a):
            seen[name] = seen.get(name, 0) + 1
            return inner(name, *a)
        _lib.call = call
        try:
            runs = []
            for rep in range(2):
                model.zero_grad()
                z = sharded_encode(model, sg, X, transform_first=True)
                z.backward(dZ)
                runs.append((z.detach(), [p.grad.clone() for p in model.parameters()]))
        finally:
            _lib.call = inner
        for name in ("gae_spmm_csr", "gae_spmm_csr_ep", "gae_linear2_fwd", "gae_linear2_fill_dead", "gae_gcn2_bwd_dense"):
            assert seen.get(name, 0) >= 2, f"{name}: {seen}"
        assert torch.equal(runs[0][0], runs[1][0]) and all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
        Z, grads = runs[0]
        dead_f, dead_b = sg.dead_rows("fwd").bool(), sg.dead_rows("bwd").bool()
        assert 0.5 < float(dead_f.float().mean()) < 0.9            # R-MAT: most rows have no in-edge (list mode matters)
        assert sg.live_rows("fwd")[0].numel() == int((~dead_f).sum())
        del runs, sg
        torch.cuda.empty_cache()
        # ---- fp64, reference order, ATen only
        W1, b1, W2, b2 = (t.detach().double() for t in (model.layers[0].apply_mod.linear.weight, model.layers[0].apply_mod.linear.bias,
                                                        model.layers[1].apply_mod.linear.weight, model.layers[1].apply_mod.linear.bias))
        M1 = _aggregate64(dst, src, X.double(), n)
        H1 = torch.relu(M1 @ W1.t() + b1)
        M2 = _aggregate64(dst, src, H1, n)
        Zr = M2 @ W2.t() + b2
        zs = float(Zr.abs().max())
        assert float((Z.double() - Zr).abs().max()) / zs < TOL
        # per-row scale of the sampled rows: sum of |terms| of the row's last product
        absM2 = _aggregate64(dst, src, H1.abs(), n) @ W2.abs().t() + b2.abs()
        deg = torch.bincount(dst, minlength=n)
        order = torch.argsort(deg, descending=True)
        g2 = torch.Generator(device=dev).manual_seed(6)
        pick = lambda idx, k: idx[torch.randint(0, idx.numel(), (k,), device=dev, generator=g2)]
        rows = torch.cat([order[:64], pick(torch.nonzero(deg > ops.SKEW_THRESHOLD).flatten(), 3000),
                          pick(torch.nonzero((deg > 0) & (deg <= ops.SKEW_THRESHOLD)).flatten(), 3000),
                          pick(torch.nonzero(deg == 0).flatten(), 2000)]).unique()
        row_err = ((Z[rows].double() - Zr[rows]).abs() / absM2[rows].clamp(min=1e-30)).max()
        assert float(row_err) < TOL, float(row_err)
        assert int(deg[rows].max()) > 100000
        del M2, absM2, Zr
        # backward
        G64 = _aggregate64(src, dst, dZ.double(), n)                    # A^T dZ
        dW2r = G64.t() @ H1
        db2r = dZ.double().sum(0)
        dY1 = (G64 @ W2) * (H1 > 0)
        dW1r = dY1.t() @ M1
        db1r = dY1.sum(0)
        want = [dW1r, db1r, dW2r, db2r]
        for a, b in zip(grads, want):
            assert rel_err(a, b) < 5 * TOL, (a.shape, rel_err(a, b))
        del G64, dY1, M1, H1
        torch.cuda.empty_cache()
        # exact fp32 products in the weight-gradient pass: same bound, and close to the default
        sg = ShardedGraph.from_edge_slice(n, src, dst, None, "boundary", dev, "nnz", overlap=True)
        _lib.call("gae_tuning_set", b"atb_bf16", 0)
        try:
            model.zero_grad()
            z = sharded_encode(model, sg, X, transform_first=True)
            z.backward(dZ)
        finally:
            _lib.call("gae_tuning_set", b"atb_bf16", 1)
        assert torch.equal(z.detach(), Z)
        for p, a, b in zip(model.parameters(), grads, want):
            assert rel_err(p.grad, b) < 5 * TOL
            assert rel_err(a, p.grad) < 2 * TOL
    finally:
        sg = None
        if created:
            dist.destroy_process_group()
