"""Row-sharded path on the GPU: P virtual ranks on ONE device (the box has one
GPU) must reproduce the single-GPU kernels bit for bit, and the real
torch.distributed code path is exercised with a 1-rank RCCL group."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def graph(seed=0, n=1500, e=20000, F=32):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e)
    dst = (rng.integers(0, n, e).astype(np.float64) ** 2 / n).astype(np.int64)   # skewed: heavy rows exist
    near = rng.random(e) < 0.5
    src[near] = np.clip(dst[near] + rng.integers(-6, 7, int(near.sum())), 0, n - 1)
    X = rng.standard_normal((n, F)).astype(np.float32)
    return n, torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV), torch.from_numpy(X).to(DEV)


@pytest.mark.parametrize("world", [1, 3, 4])
@pytest.mark.parametrize("mode", ["allgather", "boundary"])
@pytest.mark.parametrize("balance", ["rows", "nnz"])
def test_virtual_ranks_spmm_fwd_bwd_bit_exact(world, mode, balance):
    from gae_dgl_amd import ops
    from gae_dgl_amd.parallel import LocalGroup, ShardedGraph
    n, src, dst, X = graph()
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    tp, tx = ops.csr_from_coo(src, dst, n, n)
    ref = ops.spmm_raw(ip, ix, X, n, plan=ops.spmm_plan(ip, threshold=ops.SKEW_THRESHOLD))
    dM = torch.randn(n, X.shape[1], device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    refb = ops.spmm_raw(tp, tx, dM, n, plan=ops.spmm_plan(tp, threshold=ops.SKEW_THRESHOLD))
    grp = LocalGroup(world)
    outs, grads = [], []
    for r in range(world):
        sg = ShardedGraph(n, src, dst, rank=r, group=grp, mode=mode, device=DEV, balance=balance)
        p = sg.part
        h = X[p.r0:p.r1].clone().requires_grad_(True)
        grp.publish(X)
        m = sg.spmm(h)
        grp.publish(dM)
        m.backward(dM[p.r0:p.r1])
        outs.append(m.detach()); grads.append(h.grad)
        if mode == "boundary" and world > 1:
            assert sg.exchange_bytes(32) < (n - p.n_local) * 32 * 4             # less than an all-gather
    assert torch.equal(torch.cat(outs), ref)
    assert torch.equal(torch.cat(grads), refb)


@pytest.mark.parametrize("world", [1, 3, 4])
@pytest.mark.parametrize("mode", ["allgather", "boundary"])
def test_virtual_ranks_overlap_form(world, mode):
    """overlap=True: own-column product while the exchange is in flight, then M += remote-column product
    (GAE_SPMM_ACCUMULATE).  Deterministic; equal to the single-GPU rows up to fp32 re-association (1e-5 tolerance)"""
    from gae_dgl_amd import ops
    from gae_dgl_amd.parallel import LocalGroup, ShardedGraph
    n, src, dst, X = graph()
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    tp, tx = ops.csr_from_coo(src, dst, n, n)
    ref = ops.spmm_raw(ip, ix, X, n)
    dM = torch.randn(n, X.shape[1], device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    refb = ops.spmm_raw(tp, tx, dM, n)
    grp = LocalGroup(world)
    for trial in range(2):
        outs, grads = [], []
        for r in range(world):
            sg = ShardedGraph(n, src, dst, rank=r, group=grp, mode=mode, device=DEV, balance="nnz", overlap=True)
            p = sg.part
            h = X[p.r0:p.r1].clone().requires_grad_(True)
            grp.publish(X)
            m = sg.spmm(h)
            grp.publish(dM)
            m.backward(dM[p.r0:p.r1])
            outs.append(m.detach()); grads.append(h.grad)
        out, grad = torch.cat(outs), torch.cat(grads)
        assert float((out - ref).abs().max() / ref.abs().max()) < 1e-5        # hub rows: thousands of terms re-associated
        assert float((grad - refb).abs().max() / refb.abs().max()) < 1e-5
        if trial:
            assert torch.equal(out, first[0]) and torch.equal(grad, first[1])       # deterministic
        first = (out, grad)


@pytest.mark.parametrize("mode,overlap", [("boundary", True), ("allgather", False)])
def test_virtual_ranks_rmat_shards_with_pinned_rows(mode, overlap):
    """RMAT s22 (2^26 edges) on 2 virtual ranks: every rank's structure is large enough for the XCD-pinned part of
    the skew plan (>= ops.HOMED_MIN_EDGES edges in rows of > 256), with hot-column tags and segment descriptors --
    the plans bench.py --gpus N builds.  Forward and backward rows == the single-GPU product without a pinned part to
    1e-5 of the scale (another summation order for the long rows), deterministic."""
    from gae_dgl_amd import ops, workloads as W
    from gae_dgl_amd.parallel import LocalGroup, ShardedGraph
    scale, world, F = 22, 2, 32
    n = 1 << scale
    src, dst = W.rmat_edges(scale, 16, seed=0, device=DEV)
    X = torch.rand(n, F, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    dM = torch.randn(n, F, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    ref = ops.spmm_raw(ip, ix, X, n, plan=ops.spmm_plan(ip, threshold=ops.SKEW_THRESHOLD, indices=ix, ell=False,
                                                        n_cols=n, homed=False))
    del ip, ix
    tp, tx = ops.csr_from_coo(src, dst, n, n)
    refb = ops.spmm_raw(tp, tx, dM, n, plan=ops.spmm_plan(tp, threshold=ops.SKEW_THRESHOLD, indices=tx, ell=False,
                                                         n_cols=n, homed=False))
    del tp, tx
    grp = LocalGroup(world)
    pinned = 0
    for trial in range(2):
        outs, grads = [], []
        for r in range(world):
            sg = ShardedGraph(n, src, dst, rank=r, group=grp, mode=mode, device=DEV, balance="nnz", overlap=overlap)
            p = sg.part
            h = X[p.r0:p.r1].clone().requires_grad_(True)
            grp.publish(X)
            m = sg.spmm(h)
            grp.publish(dM)
            m.backward(dM[p.r0:p.r1])
            outs.append(m.detach()); grads.append(h.grad)
            pinned += sum(pl.homed is not None for pl in sg._plan.values())
            del sg
        out, grad = torch.cat(outs), torch.cat(grads)
        assert float((out - ref).abs().max() / ref.abs().max()) < 1e-5
        assert float((grad - refb).abs().max() / refb.abs().max()) < 1e-5
        if trial:
            assert torch.equal(out, first[0]) and torch.equal(grad, first[1])
        first = (out, grad)
    assert pinned > 0           # the case this test is about


@pytest.mark.parametrize("mode,overlap", [("boundary", True), ("boundary", False), ("allgather", True)])
def test_constant_input_exchange_is_cached(mode, overlap):
    """ShardedGraph.cache_constant_inputs: the forward product of an operand that needs no gradient (the input
    features of a transductive graph) exchanges its remote rows once and reuses them while the same tensor comes
    back unchanged; operands that need a gradient, changed tensors (in-place edit: version counter) and the backward
    always exchange.  Results are bit-identical with and without the cache."""
    from gae_dgl_amd.parallel import LocalGroup, ShardedGraph
    n, src, dst, X = graph()
    world = 3
    grp = LocalGroup(world)
    grp.publish(X)
    for r in range(world):
        sg = ShardedGraph(n, src, dst, rank=r, group=grp, mode=mode, device=DEV, balance="nnz", overlap=overlap)
        p = sg.part
        x = X[p.r0:p.r1].clone()
        ref = sg.spmm(x)
        sg.cache_constant_inputs = True
        sg.timers = {}
        name = "exchange_start" if overlap else "exchange"
        outs = [sg.spmm(x) for _ in range(3)]
        assert len(sg.timers.get(name, [])) == 1                      # one exchange for three products
        assert all(torch.equal(o, ref) for o in outs)
        h = x.clone().requires_grad_(True)                            # needs a gradient: never cached
        m = sg.spmm(h)
        assert len(sg.timers[name]) == 2 and torch.equal(m.detach(), ref)
        grp.publish(torch.randn_like(X))
        m.backward(grp.full[p.r0:p.r1])                               # the backward exchanges
        assert len(sg.timers[name]) == 3
        X2 = X.clone(); X2[p.r0:p.r1] *= 2.0
        grp.publish(X2)
        x.mul_(2.0)                                                   # same storage, new version: exchanged again
        o2 = sg.spmm(x)
        assert len(sg.timers[name]) == 4
        sg.cache_constant_inputs = False
        assert torch.equal(o2, sg.spmm(x))
        grp.publish(X)


def test_spmm_accumulate_flag():
    """GAE_SPMM_ACCUMULATE: out += A H for every kernel family that takes it (row-group, v1, segment plan), scaled
    and unscaled, fp32 and bf16 storage"""
    from gae_dgl_amd import ops, _lib
    n, src, dst, X = graph(seed=7, n=900, e=9000, F=40)
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    deg, norm = ops.degree_norm(ip)
    plan = ops.spmm_plan(ip, threshold=8, segment=64)
    for dtype in (torch.float32, torch.bfloat16):
        H = X.to(dtype)
        base = torch.randn(n, 40, device=DEV).to(dtype)
        for sc in (None, norm):
            for pl in (None, plan):
                for variant in (2, 1):
                    if pl is not None and variant == 1:
                        continue
                    _lib.call("gae_tuning_set", b"spmm_variant", variant)
                    try:
                        prod = ops.spmm_raw(ip, ix, H, n, sc, sc, plan=pl)
                        out = base.clone()
                        ops.spmm_raw(ip, ix, H, n, sc, sc, plan=pl, out=out, accumulate=True)
                    finally:
                        _lib.call("gae_tuning_set", b"spmm_variant", 2)
                    want = (base.float() + prod.float())
                    tol = 1e-6 if dtype == torch.float32 else 2e-2
                    assert float((out.float() - want).abs().max() / want.abs().max()) < tol


@pytest.mark.parametrize("world", [1, 3])
def test_virtual_ranks_fused_loss(world):
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from gae_dgl_amd.parallel import LocalGroup, ShardedGraph
    n, src, dst, _ = graph(seed=2, n=700, e=5000)
    gen = torch.Generator(device=DEV).manual_seed(3)
    Z = torch.randn(n, 16, device=DEV, generator=gen) * 0.6
    mask = ops.dropout_mask((n, 16), 0.1, seed=5, device=DEV)
    g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
    Z0 = Z.clone().requires_grad_(True)
    ref = ops.decoder_bce(Z0, mask, g)
    ref.backward()
    grp = LocalGroup(world)
    grp.publish(Z * mask)
    total, grads = 0.0, []
    for r in range(world):
        # the loss labels pairs by global id: it works in both exchange modes of the SpMM
        sg = ShardedGraph(n, src, dst, rank=r, group=grp, mode=("allgather", "boundary")[r % 2], device=DEV)
        p = sg.part
        z = Z[p.r0:p.r1].clone().requires_grad_(True)
        part = ops.sharded_decoder_bce(z, mask[p.r0:p.r1], sg, n_edges_global=int(src.numel()))
        part.backward()
        total += float(part.detach()); grads.append(z.grad)
    assert abs(total - float(ref.detach())) < 1e-6 * abs(float(ref.detach()))
    err = float((torch.cat(grads) - Z0.grad).abs().max() / Z0.grad.abs().max())
    assert err < 1e-6


def test_one_rank_rccl_group_end_to_end():
    """the real collectives (RCCL, world_size 1): sharded training step == plain step"""
    import torch.distributed as dist
    import gae_dgl_amd as G
    from gae_dgl_amd.parallel import ShardedGraph, allreduce_grads, sharded_loss
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        n, src, dst, X = graph(seed=4, n=900, e=7000, F=39)
        torch.manual_seed(0)
        model = G.GAE(39, [32, 16]).to(DEV)
        model.decoder.dropout = 0.0
        for mode in ("allgather", "boundary"):
            sg = ShardedGraph(n, src, dst, mode=mode, device=DEV)
            full = sg.exchange(X, "fwd")
            assert torch.equal(full[:n], X)
        sg = ShardedGraph(n, src, dst, mode="allgather", device=DEV)
        model.zero_grad()
        loss = sharded_loss(model, sg, X)
        loss.backward()
        allreduce_grads(list(model.parameters()))
        g1 = [p.grad.clone() for p in model.parameters()]
        model.zero_grad()
        g = G.DGLGraph((src, dst), num_nodes=n).to(DEV)
        g.ndata['h'] = X
        ref = model.reconstruction_loss(g)
        ref.backward()
        assert abs(float(loss.detach()) - float(ref.detach())) < 1e-6 * abs(float(ref.detach()))
        for a, p in zip(g1, model.parameters()):      # (fp32 rounding: the plain graph's table kernels add a hub row's
            assert float((a - p.grad).abs().max()) <= 5e-6 * float(p.grad.abs().max())    # terms in another order)
        # narrowing layers as A (H W^T): aggregation (and its backward, and the exchange) at the output width --
        # the same loss and gradients up to fp32 rounding
        for mode in ("allgather", "boundary"):
            sg = ShardedGraph(n, src, dst, mode=mode, device=DEV, overlap=(mode == "boundary"))
            model.zero_grad()
            loss_tf = sharded_loss(model, sg, X, transform_first=True)
            loss_tf.backward()
            assert abs(float(loss_tf.detach()) - float(ref.detach())) < 1e-5 * abs(float(ref.detach()))
            for a, p in zip(g1, model.parameters()):
                assert float((a - p.grad).abs().max()) <= 5e-5 * float(a.abs().max())
    finally:
        if created:
            dist.destroy_process_group()


def _rccl_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        import gae_dgl_amd as G
        from gae_dgl_amd import ops
        from gae_dgl_amd.parallel import ShardedGraph, allreduce_grads, sharded_encode
        dev = f"cuda:{rank}"
        n, src, dst, X = graph(seed=4, n=3000, e=40000, F=32)       # (widths <= 32: the one-pass encoder applies)
        src, dst, X = src.to(dev), dst.to(dev), X.to(dev)
        torch.manual_seed(0)
        model = G.GAE(32, [32, 16]).to(dev)
        g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
        g.ndata['h'] = X
        ref = model.encode(g)
        dZ = torch.randn(n, 16, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
        ref.backward(dZ)
        gref = [p.grad.clone() for p in model.parameters()]
        for mode in ("allgather", "boundary"):
            for overlap in (False, True):
                for balance in ("rows", "nnz"):
                    sg = ShardedGraph(n, src, dst, mode=mode, device=dev, balance=balance, overlap=overlap)
                    p = sg.part
                    model.zero_grad()
                    z = sharded_encode(model, sg, X[p.r0:p.r1])
                    err = float((z - ref[p.r0:p.r1]).abs().max() / ref.abs().max())
                    assert err < 1e-5 and (overlap or torch.equal(z, ref[p.r0:p.r1].detach())), (mode, overlap, err)
                    z.backward(dZ[p.r0:p.r1])
                    allreduce_grads(list(model.parameters()))
                    for a, prm in zip(gref, model.parameters()):
                        assert float((a - prm.grad).abs().max()) <= 5e-5 * float(a.abs().max()), (mode, overlap)
                    # the one-pass encoder (ShardedEncoder2Function: list mode of the dense passes, products that skip the
                    # rows without edges, bias in the epilogue of the remote-column half) on the same shards
                    model.zero_grad()
                    z2 = sharded_encode(model, sg, X[p.r0:p.r1], transform_first=True)
                    assert float((z2 - ref[p.r0:p.r1]).abs().max() / ref.abs().max()) < 1e-5, (mode, overlap, "tf")
                    z2.backward(dZ[p.r0:p.r1])
                    allreduce_grads(list(model.parameters()))
                    for a, prm in zip(gref, model.parameters()):
                        assert float((a - prm.grad).abs().max()) <= 5e-5 * float(a.abs().max()), (mode, overlap, "tf")
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI, world size 2)")
def test_two_rank_rccl_sharded_encoder():
    """real RCCL, one process per GPU: both exchange modes, both block rules, with and without overlap reproduce the
    single-GPU encoder (bit for bit without overlap) and its parameter gradients after the gradient all-reduce"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("F", [32, 16, 7, 33])
def test_rows_pack_is_an_exact_gather(F):
    """gae_rows_pack (the pack kernel of the exchange): out[i] = H[idx[i]] bit for bit, zero rows behind, packing
    into a slice of a larger buffer leaves its other rows alone"""
    from gae_dgl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(F)
    H = torch.randn(1000, F, generator=g).to(dev)
    idx = torch.randint(0, 1000, (777,), generator=g).to(dev)
    out = ops.rows_pack(H, idx)
    assert torch.equal(out, H.index_select(0, idx))
    padded = ops.rows_pack(H, idx, n_out_rows=800)
    assert torch.equal(padded[:777], H[idx]) and bool((padded[777:] == 0).all()) and padded.shape == (800, F)
    big = torch.full((1500, F), 7.0, device=dev)
    ops.rows_pack(H, None, out=big[200:1200])
    assert torch.equal(big[200:1200], H) and bool((big[:200] == 7).all()) and bool((big[1200:] == 7).all())
    Hp = ops.pad_rows(torch.randn(50, 39, device=dev))                  # row-padded source (ld 40)
    assert torch.equal(ops.rows_pack(Hp, idx[:20] % 50), Hp[idx[:20] % 50])
    assert ops.rows_pack(H, idx[:0]).shape == (0, F)


def _one_rank_group():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    return created


@pytest.mark.timeout(300)
@pytest.mark.parametrize("transform_first,capture", [(False, False), (True, False), (True, True)])
def test_sharded_training_step_one_rank_rccl_follows_reference_step(transform_first, capture):
    """the whole data-parallel step (parallel.ShardedTrainStep: sharded encoder, row block of the fused loss + scalar
    all-reduce, backward, gradient all-reduce over RCCL, the library's Adam) against oracle.CpuReferenceStep for three
    steps; ``capture``: the step -- collectives included -- replayed from ONE HIP graph.  The structure comes from
    the rank's edge slice (ShardedGraph.from_edge_slice)."""
    import torch.distributed as dist
    import gae_dgl_amd as G
    from gae_dgl_amd import optim
    from gae_dgl_amd.parallel import ShardedGraph, ShardedTrainStep
    from oracle import gae_oracle as O
    created = _one_rank_group()
    try:
        n, src, dst, X = graph(seed=6, n=800, e=6000, F=32)
        keys = torch.unique(src * n + dst)
        src, dst = keys // n, keys % n
        torch.manual_seed(0)
        ref = O.CpuReferenceStep(src.cpu().numpy(), dst.cpu().numpy(), n, X.cpu().numpy(), 32, [32, 16], lr=1e-2, seed=0,
                                 dropout=0.0)
        model = G.GAE(32, [32, 16]).to(DEV)
        model.decoder.dropout = 0.0
        with torch.no_grad():
            for conv, lin in zip(model.layers, ref.layers):
                conv.apply_mod.linear.weight.copy_(lin.weight); conv.apply_mod.linear.bias.copy_(lin.bias)
        sg = ShardedGraph.from_edge_slice(n, src, dst, None, "boundary", DEV, "nnz", overlap=True)
        opt = optim.Adam(model.parameters(), lr=1e-2)
        step = ShardedTrainStep(model, opt, sg, X, transform_first=transform_first, capture=capture, warmup=1)
        k0 = 1 if capture else 0                       # (the capture's warm-up step trained the model once)
        want = [ref.step() for _ in range(3 + k0)][k0:]
        got = []
        for _ in range(3):
            got.append(float(step()))
        np.testing.assert_allclose(got, want, rtol=5e-5)
        for conv, lin in zip(model.layers, ref.layers):
            w = conv.apply_mod.linear.weight.detach().cpu()
            assert float((w - lin.weight).abs().max()) <= 5e-4 * float(lin.weight.abs().max())
    finally:
        # a captured step holds a HIP graph with RCCL's kernels in it: release it BEFORE the communicator goes away
        step = sg = opt = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        if created:
            dist.destroy_process_group()
