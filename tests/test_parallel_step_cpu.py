"""SURVEY 8(e) on the CPU, world_size-2 gloo: (1) the row-block plan assembled from per-rank EDGE SLICES
(RowPartition.from_edge_slice) equals the plan filtered out of the whole edge list; (2) a WHOLE sharded training step --
sharded encoder, row block of the loss + scalar all-reduce, backward through both exchanges, gradient all-reduce, Adam --
follows oracle.CpuReferenceStep (train_inductive.py:43-53 on one process) for three steps.

The HIP kernels need a GPU, so in (2) the entry points of gae_dgl_amd.ops that parallel.py launches are replaced, in the
worker processes of this test only, by the oracle's arithmetic: what runs for real is the product's host logic --
partition, exchanges in both directions, the autograd functions, the collectives, the optimiser step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_graph(seed=0, n=151, e=900, F=9):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e); dst = rng.integers(0, n, e)
    dst[:70] = 5                                            # a hub row
    near = rng.random(e) < 0.5
    src[near] = np.clip(dst[near] + rng.integers(-4, 5, int(near.sum())), 0, n - 1)
    keys = np.unique(src * n + dst)                          # simple graph: the dense label is 0 / 1
    src, dst = keys // n, keys % n
    X = rng.standard_normal((n, F)).astype(np.float32)
    return n, src.astype(np.int64), dst.astype(np.int64), X


def install_oracle_kernels():
    """(test scaffolding) ops.* entry points used by parallel.py -> CPU arithmetic of the oracle"""
    from gae_dgl_amd import ops
    from oracle import gae_oracle as O

    def csr_from_coo(rows, cols, n_rows, n_cols):
        ip, ix = O.csr_from_coo(cols.numpy(), rows.numpy(), int(n_rows), int(n_cols))
        return torch.as_tensor(np.asarray(ip)), torch.as_tensor(np.asarray(ix))

    def spmm_raw(ip, ix, H, n_rows, out=None, plan=None, accumulate=False, **kw):
        r = torch.as_tensor(O.spmm_csr(ip.numpy(), ix.numpy(), H.detach()))
        if out is None:
            return r
        out.copy_(out + r if accumulate else r)
        return out

    def spmm_ep_raw(ip, ix, H, n_rows, plan, bias, act, out=None, accumulate=False):
        r = spmm_raw(ip, ix, H, n_rows, out=out, accumulate=accumulate)
        if bias is not None:
            r += bias.detach()
        if act:
            r.clamp_(min=0)
        return r

    def linear(M, W, b, act=0):
        y = torch.nn.functional.linear(M, W, b)
        return torch.relu(y) if act == ops.ACT_RELU else y

    def decoder_bce_raw(Zfull, mask, csr, csc, pw, want_grad=True, row_begin=0, n_local=None):
        """rows [row_begin, row_begin + n_local) of the N x N weighted BCE (train_inductive.py:44-48), divided by N^2,
        and the gradient of the GLOBAL loss with respect to those rows of Z (row side + column side)"""
        assert mask is None
        n = Zfull.shape[0]
        Z = Zfull.double()
        rows = slice(row_begin, row_begin + n_local)

        def labels(c):
            ip, ix = (t.numpy() for t in c)
            Y = torch.zeros(n_local, n, dtype=torch.float64)
            r = np.repeat(np.arange(n_local), np.diff(ip))
            Y.index_put_((torch.as_tensor(r), torch.as_tensor(ix.astype(np.int64))), torch.ones(len(ix), dtype=torch.float64),
                         accumulate=True)
            return Y
        x = Z[rows] @ Z.t()
        Y = labels(csr)
        sp = torch.nn.functional.softplus(-x)
        loss = (((1 - Y) * x + (1 + (pw - 1) * Y) * sp).sum() / (float(n) * n)).reshape(1).float()
        if not want_grad:
            return loss, None
        dl = lambda y: (1 - y) - (1 + (pw - 1) * y) * torch.sigmoid(-x)
        g = (dl(Y) + dl(labels(csc))) @ Z / (float(n) * n)       # x is symmetric: (j, i) pairs see the labels of A^T
        return loss, g.float()

    ops.csr_from_coo, ops.spmm_raw, ops.spmm_ep_raw, ops.linear, ops.decoder_bce_raw = \
        csr_from_coo, spmm_raw, spmm_ep_raw, linear, decoder_bce_raw
    ops.spmm_plan = lambda *a, **k: None


def _slices(src, dst, rank, world, how):
    if how == "interleaved":
        return src[rank::world], dst[rank::world]
    cut = [0, len(src) // 3, len(src)]                        # uneven contiguous slices
    return src[cut[rank]:cut[rank + 1]], dst[cut[rank]:cut[rank + 1]]


def _partition_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gae_dgl_amd.parallel import RowPartition
        n, src, dst, _ = make_graph(seed=3)
        for mode in ("allgather", "boundary"):
            for balance in ("rows", "nnz"):
                for overlap in (False, True):
                    for how in ("interleaved", "contiguous"):
                        s, d = _slices(src, dst, rank, world, how)
                        a = RowPartition.from_edge_slice(n, torch.from_numpy(s), torch.from_numpy(d), None, mode,
                                                         balance, overlap)
                        b = RowPartition(n, src, dst, rank, world, mode, balance, overlap)
                        assert np.array_equal(a.bounds, b.bounds) and (a.r0, a.r1) == (b.r0, b.r1)
                        assert a.n_cols == b.n_cols and a.uniform == b.uniform
                        for k in ("fwd", "bwd"):
                            ea = torch.stack((getattr(a, k + "_rows"), getattr(a, k + "_cols")))
                            eb = torch.stack((getattr(b, k + "_rows"), getattr(b, k + "_cols")))
                            key = lambda e: torch.sort(e[0] * (4 * n) + e[1]).values     # the edge MULTISET (CSR build sorts)
                            assert torch.equal(key(ea), key(eb)), (mode, balance, overlap, how, k)
                            if mode == "boundary":
                                assert torch.equal(a.need[k], b.need[k]) and a.n_before[k] == b.n_before[k]
                            if overlap:
                                for part in ("own", "remote"):
                                    pa, pb = a.split[k][part], b.split[k][part]
                                    assert torch.equal(key(torch.stack(pa)), key(torch.stack(pb)))
                                assert a.split[k]["n_remote_cols"] == b.split[k]["n_remote_cols"]
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _step_worker(rank, world, port, q, mode, overlap):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        install_oracle_kernels()
        import gae_dgl_amd as G
        from gae_dgl_amd.parallel import ShardedGraph, ShardedTrainStep
        from oracle import gae_oracle as O
        n, src, dst, X = make_graph(seed=1)
        F, hidden = X.shape[1], [32, 16]
        torch.manual_seed(0)                                       # the same initial weights in every process
        ref = O.CpuReferenceStep(src, dst, n, X, F, hidden, lr=1e-2, seed=0, dropout=0.0)
        model = G.GAE(F, hidden)
        model.decoder.dropout = 0.0
        with torch.no_grad():
            for conv, lin in zip(model.layers, ref.layers):
                conv.apply_mod.linear.weight.copy_(lin.weight); conv.apply_mod.linear.bias.copy_(lin.bias)
        s, d = _slices(src, dst, rank, world, "interleaved")
        sg = ShardedGraph.from_edge_slice(n, torch.from_numpy(s), torch.from_numpy(d), None, mode, "cpu", "nnz", overlap)
        assert sg.n_edges_global() == len(src)
        p = sg.part
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)       # the reference's optimiser (train_transductive.py:43)
        step = ShardedTrainStep(model, opt, sg, torch.from_numpy(X[p.r0:p.r1]), transform_first=False)
        got = [float(step()) for _ in range(3)]
        want = [ref.step() for _ in range(3)]
        np.testing.assert_allclose(got, want, rtol=2e-5)
        for conv, lin in zip(model.layers, ref.layers):            # every rank holds the reference's weights
            w = conv.apply_mod.linear.weight
            assert float((w - lin.weight).abs().max()) <= 2e-4 * float(lin.weight.abs().max())
        # ... and the same ones as the other rank, bit for bit (all-reduced gradients, same update)
        flat = torch.cat([q_.detach().reshape(-1) for q_ in model.parameters()])
        other = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert all(torch.equal(o, flat) for o in other)
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run(target, args, port_base):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port_base + (os.getpid() % 200) * 4
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_partition_from_edge_slices_equals_partition_of_the_whole_list_world2():
    _run(_partition_worker, (), 30500)


@pytest.mark.parametrize("mode,overlap", [("allgather", False), ("boundary", True)])
def test_sharded_training_step_follows_the_reference_step_world2(mode, overlap):
    _run(_step_worker, (mode, overlap), 31500 + (1 if overlap else 0))
