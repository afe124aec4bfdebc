"""N > 1 path on the CPU: world_size-2 gloo processes exercise the row
partition plan and both exchange modes; the oracle is the compute step (the
HIP kernels need a GPU), so what is checked is that every rank ends up with
exactly the inputs whose local SpMM gives its rows of the global result --
forward and backward -- and that the replicated-weight gradient all-reduce
reproduces the single-process gradients."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_graph(seed=0, n=101, e=700, F=7):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e); dst = rng.integers(0, n, e)
    dst[:60] = 3                                            # a hub row
    # locality for the boundary mode: half of the edges stay close to the diagonal
    near = rng.random(e) < 0.5
    src[near] = np.clip(dst[near] + rng.integers(-4, 5, int(near.sum())), 0, n - 1)
    X = rng.standard_normal((n, F)).astype(np.float32)
    return n, src.astype(np.int64), dst.astype(np.int64), X


def _worker(rank, world, port, mode, q, balance="rows"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gae_dgl_amd.parallel import ShardedGraph, allreduce_grads
        from oracle import gae_oracle as O
        n, src, dst, X = make_graph()
        sg = ShardedGraph(n, torch.from_numpy(src), torch.from_numpy(dst), mode=mode, device="cpu", balance=balance)
        p = sg.part
        # ---- forward: rows [r0, r1) of A X
        ip, ix = O.csr_from_coo(src, dst, n)
        ref = O.spmm_csr(ip, ix, X)
        h_local = torch.from_numpy(X[p.r0:p.r1])
        full = sg.exchange(h_local, "fwd")
        lip, lix = O.csr_from_coo(p.fwd_cols.numpy(), p.fwd_rows.numpy(), p.n_local, p.n_cols["fwd"])
        got = O.spmm_csr(lip, lix, full)
        assert torch.equal(got, ref[p.r0:p.r1]), "forward rows differ"
        # ---- backward: rows [r0, r1) of A^T dM
        dM = np.random.default_rng(1).standard_normal((n, X.shape[1])).astype(np.float32)
        tp, tx = O.csc_from_coo(src, dst, n)
        refb = O.spmm_csr(tp, tx, dM)
        fullb = sg.exchange(torch.from_numpy(dM[p.r0:p.r1]), "bwd")
        bip, bix = O.csr_from_coo(p.bwd_cols.numpy(), p.bwd_rows.numpy(), p.n_local, p.n_cols["bwd"])
        assert torch.equal(O.spmm_csr(bip, bix, fullb), refb[p.r0:p.r1]), "backward rows differ"
        # ---- overlap form: A_p = [A_own | A_remote]; own columns read the local block, remote columns read the
        #      received rows where the collective put them; own + remote partials = the rows of the global product
        sgo = ShardedGraph(n, torch.from_numpy(src), torch.from_numpy(dst), mode=mode, device="cpu", balance=balance,
                           overlap=True)
        po = sgo.part
        for which, loc, ref_rows in (("fwd", h_local, ref[p.r0:p.r1]),
                                     ("bwd", torch.from_numpy(dM[p.r0:p.r1]), refb[p.r0:p.r1])):
            recv, wait = sgo.exchange_start(loc, which)
            orow, ocol = po.split[which]["own"]; rrow, rcol = po.split[which]["remote"]
            assert int(ocol.max()) < po.n_local and (rcol.numel() == 0 or int(rcol.max()) < recv.shape[0])
            oip, oix = O.csr_from_coo(ocol.numpy(), orow.numpy(), po.n_local, po.n_local)
            part_own = O.spmm_csr(oip, oix, loc)
            wait()
            rip, rix = O.csr_from_coo(rcol.numpy(), rrow.numpy(), po.n_local, max(recv.shape[0], 1))
            total = part_own + (O.spmm_csr(rip, rix, recv) if rcol.numel() else 0)
            assert torch.allclose(total, ref_rows, rtol=1e-5, atol=1e-5), "overlap partials differ"
            assert orow.numel() + rrow.numel() == (po.fwd_rows if which == "fwd" else po.bwd_rows).numel()
        # ---- exchange volume bookkeeping
        if mode == "allgather":
            assert full.shape[0] == p.padded_n and sg.exchange_bytes(7) == (n - p.n_local) * 7 * 4
            if balance == "nnz":
                assert full.shape[0] == n
        else:
            assert full.shape[0] == p.n_local + p.need["fwd"].numel() < n
        # ---- replicated weights: dW = sum over row blocks (all-reduce)
        W = torch.nn.Parameter(torch.ones(3, X.shape[1]))
        y = (got @ W.t()).sum()
        y.backward()
        allreduce_grads([W])
        Wref = torch.ones(3, X.shape[1], requires_grad=True)
        (ref @ Wref.t()).sum().backward()
        assert torch.allclose(W.grad, Wref.grad, rtol=1e-5, atol=1e-4)
        # ---- row-block loss partial sums add up to the global loss (oracle arithmetic)
        Z = torch.from_numpy(np.random.default_rng(2).standard_normal((n, 4)).astype(np.float32))
        adj = O.dense_adjacency(src, dst, n); pw = O.pos_weight_of(adj)
        x = Z[p.r0:p.r1] @ Z.t()
        y_ = adj[p.r0:p.r1]
        part = ((1 - y_) * x + (1 + (pw - 1) * y_) * torch.nn.functional.softplus(-x)).sum() / (n * n)
        tot = part.clone(); dist.all_reduce(tot)
        assert abs(float(tot) - float(O.bce_with_logits_mean(Z @ Z.t(), adj, pw))) < 1e-5
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,balance", [("allgather", "rows"), ("boundary", "rows"), ("allgather", "nnz"),
                                          ("boundary", "nnz")])
def test_row_sharded_exchange_world2(mode, balance):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29610 + ["allgather", "boundary"].index(mode) + 2 * ["rows", "nnz"].index(balance) + (os.getpid() % 200) * 4
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q, balance)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_partition_plan_covers_graph_exactly():
    """virtual ranks (no process group): the blocks tile the rows, every edge lands in exactly one
    forward block and one backward block, boundary remapping is consistent"""
    from gae_dgl_amd.parallel import RowPartition, block_bounds, nnz_balanced_bounds
    n, src, dst, _ = make_graph(seed=4, n=1000, e=9000)
    for world in (2, 5):     # nnz-balanced blocks: contiguous, cover [0, n), edge counts within 2x of the mean
        b = nnz_balanced_bounds(n, src, dst, world)
        assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0)
        w = np.bincount(dst, minlength=n) + np.bincount(src, minlength=n)
        per = np.array([w[b[r]:b[r + 1]].sum() for r in range(world)])
        assert per.max() <= 2.0 * per.mean() + w.max()
        for r in range(world):
            p = RowPartition(n, src, dst, r, world, "boundary", "nnz")
            assert (p.r0, p.r1) == (b[r], b[r + 1])
    for world in (1, 2, 3, 8):
        b = block_bounds(n, world)
        assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0)
        tot_f = tot_b = 0
        for mode in ("allgather", "boundary"):
            for r in range(world):
                p = RowPartition(n, src, dst, r, world, mode)
                tot_f += p.fwd_rows.numel(); tot_b += p.bwd_rows.numel()
                assert p.fwd_rows.numel() == int(((dst >= p.r0) & (dst < p.r1)).sum())
                assert int(p.fwd_cols.max()) < p.n_cols["fwd"] and int(p.bwd_cols.max()) < p.n_cols["bwd"]
                if mode == "boundary":
                    need = p.need["fwd"].numpy()
                    assert np.all((need < p.r0) | (need >= p.r1)) and np.all(np.diff(need) > 0)
        assert tot_f == 2 * len(src) and tot_b == 2 * len(src)


# ------------------------------------------------------------------------------------------------ world 8 (round 6)
def _hand_graph_world8():
    """a 64-node graph built by hand for 8 ranks of 8 rows: block 0 is a hub block (every other block references rows
    0..2), block 3 references nobody outside itself, block 5 has NO edges at all (empty need lists, empty send lists),
    block 7 references one row of every block; plus a ring between neighbouring blocks.  Directed on purpose: the
    forward (in-edge) and backward (out-edge) tables differ."""
    src, dst = [], []
    for b in range(1, 8):
        if b in (3, 5):
            continue
        for k in range(3):
            src.append(k); dst.append(8 * b + k)             # rows of block b read rows 0..2 of block 0
    for i in range(24, 32):
        src.append(24 + (i + 1) % 8); dst.append(i)          # block 3: edges inside the block only
    for b in range(7):
        if b != 5:
            src.append(8 * b + 4); dst.append(63)            # row 63 (block 7) reads one row of each block
    for b in (0, 1, 2, 6):                                   # ring: last row of block b -> first row of the next used block
        nb = {0: 1, 1: 2, 2: 4, 6: 7}[b]
        src.append(8 * b + 7); dst.append(8 * nb)
    return 64, np.asarray(src, np.int64), np.asarray(dst, np.int64)


def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from gae_dgl_amd import transport
        from gae_dgl_amd.parallel import RowPartition, ShardedGraph
        n, src, dst = _hand_graph_world8()
        X = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)
        sg = ShardedGraph(n, torch.from_numpy(src), torch.from_numpy(dst), mode="boundary", device="cpu")
        p = sg.part
        assert (p.r0, p.r1) == (8 * rank, 8 * rank + 8)
        for which, key_rows, key_cols in (("fwd", dst, src), ("bwd", src, dst)):
            a = sg._a2a[which]
            # ---- the tables against the hand-built graph: what this rank needs = remote columns of its rows
            mine = (key_rows >= p.r0) & (key_rows < p.r1)
            remote = np.unique(key_cols[mine & ((key_cols < p.r0) | (key_cols >= p.r1))])
            assert a["need"].tolist() == remote.tolist()
            assert a["recv_counts"] == [int(((remote >= 8 * r) & (remote < 8 * r + 8)).sum()) for r in range(world)]
            assert a["recv_counts"][rank] == 0 and a["send_counts"][rank] == 0
            # ---- pairwise agreement: what q expects from me is what I send to q (the uneven split tables of
            #      all_to_all_single must be each other's transpose, or RCCL deadlocks / corrupts)
            tables = [None] * world
            dist.all_gather_object(tables, (a["send_counts"], a["recv_counts"]))
            for r in range(world):
                for s in range(world):
                    assert tables[r][0][s] == tables[s][1][r], (which, r, s)
            # ---- the rows arrive where the local CSR indexes them
            full = sg.exchange(X[p.r0:p.r1].contiguous(), which)
            nb = p.n_before[which]
            want = torch.cat([X[remote[remote < p.r0]], X[p.r0:p.r1], X[remote[remote >= p.r1]]])
            assert full.shape[0] == p.n_local + len(remote) and nb == int((remote < p.r0).sum())
            assert torch.equal(full, want), (which, rank)
        if rank == 5:      # the edge-less block exchanges nothing in either direction
            assert sum(sg._a2a["fwd"]["recv_counts"]) == 0 and sum(sg._a2a["bwd"]["send_counts"]) == 0
        if rank == 3:
            # reads nothing remote; row 63 reads its row 28, so the gradient of row 28 needs row 63's in the backward
            assert sum(sg._a2a["fwd"]["recv_counts"]) == 0 and sg._a2a["fwd"]["send_counts"][7] == 1
            assert sg._a2a["bwd"]["recv_counts"] == [0, 0, 0, 0, 0, 0, 0, 1] and sum(sg._a2a["bwd"]["send_counts"]) == 0
        if rank == 0:      # the hub block sends rows 0..2 to five blocks, row 7 to block 1 (ring) and row 4 to block 7
            assert sg._a2a["fwd"]["send_counts"] == [0, 4, 3, 0, 3, 0, 3, 4]
        # ---- every rank holding only a SLICE of the edge list builds the same plan (two all-to-all-v of edge pairs)
        sl = slice(rank, None, world)
        ps = RowPartition.from_edge_slice(n, torch.from_numpy(src[sl]), torch.from_numpy(dst[sl]), None, "boundary")
        for k in ("fwd", "bwd"):
            assert torch.equal(ps.need[k], p.need[k])
            a_rows, a_cols = (ps.fwd_rows, ps.fwd_cols) if k == "fwd" else (ps.bwd_rows, ps.bwd_cols)
            b_rows, b_cols = (p.fwd_rows, p.fwd_cols) if k == "fwd" else (p.bwd_rows, p.bwd_cols)
            assert sorted(zip(a_rows.tolist(), a_cols.tolist())) == sorted(zip(b_rows.tolist(), b_cols.tolist()))
        # ---- uneven all-gather (nnz-balanced blocks): every rank's block lands in its slice
        sgn = ShardedGraph(n, torch.from_numpy(src), torch.from_numpy(dst), mode="allgather", device="cpu", balance="nnz")
        pn = sgn.part
        assert not pn.uniform and int(pn.bounds[-1]) == n
        got = sgn.allgather_rows(X[pn.r0:pn.r1].contiguous())
        assert torch.equal(got, X)
        # ---- capacity agreement of the data-parallel capture (capture.CapturedInductiveStep.begin_epoch): all-reduce
        #      MAX of (nodes, edges) -- every rank must come out with the same pair, the element-wise maximum
        need = torch.tensor([1000 + 13 * ((rank * 5) % 8), 5000 - 7 * rank], dtype=torch.int64)
        transport.all_reduce(need, op=dist.ReduceOp.MAX)
        assert need.tolist() == [1000 + 13 * 7, 5000]
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_boundary_exchange_tables_world8_hand_built():
    """VERDICT r05 #7b: the uneven all_to_all_single split tables, the receive layout, the uneven all-gather and the
    all-reduce-MAX capacity agreement on a hand-built 8-rank partition with an empty block, a self-contained block
    and a hub block (gloo, 8 CPU processes)"""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30900 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]
