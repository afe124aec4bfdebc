"""SURVEY 8(e) with MORE THAN ONE OS PROCESS on the real HIP kernels.

The GPU box has one device and RCCL refuses two ranks on one device, so the ranks of these tests SHARE cuda:0 and the
collectives travel over gloo, staged through host memory (gae_dgl_amd/transport.py).  Everything else is the product
path, nothing is patched: per-rank edge slices -> ShardedGraph.from_edge_slice (degree all-reduce, two all-to-all-v of
edges) -> device CSRs and skew plans (HIP) -> ShardedTrainStep: the sharded encoder through both exchanges, the rank's
row block of the fused loss (HIP) + scalar all-reduce, backward, gradient all-reduce, the library's Adam (HIP).  The
worker records the C-ABI entry points it called and the test asserts the hot ones are among them.

  * 2 and 4 ranks, both exchange modes, overlap on / off, both layer orders: three steps against
    oracle.CpuReferenceStep (train_inductive.py:43-53 on one process) + bit-identical weights on every rank;
  * RMAT s20 (2^20 nodes, 2^24 edges), every rank generating ITS slice of the edge list: the one-pass encoder's Z rows
    and the all-reduced gradients against the 1-rank result on the whole list;
  * `bench.py --gpus 2 --workload rmat --rmat-scale 20 --oversubscribe`: the N > 1 launch the driver runs on 8 GPUs,
    rehearsed end to end; the JSON line's shape is asserted."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def step_graph(seed, n=1200, e=9000, F=32):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e)
    dst = (rng.integers(0, n, e).astype(np.float64) ** 2 / n).astype(np.int64)     # skewed: heavy rows exist
    dst[:300] = 7                                                                   # a hub row (> 256 in-edges)
    near = rng.random(e) < 0.5
    src[near] = np.clip(dst[near] + rng.integers(-6, 7, int(near.sum())), 0, n - 1)
    keys = np.unique(src * n + dst)                                                 # simple graph: labels are 0 / 1
    src, dst = keys // n, keys % n
    X = rng.standard_normal((n, F)).astype(np.float32)
    return n, src.astype(np.int64), dst.astype(np.int64), X


def _init(rank, world, port):
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                       # every rank on the one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
    return dist


def _record_calls():
    """count the C-ABI calls of this process (wrapping, not replacing: the library runs)"""
    from gae_dgl_amd import _lib
    seen = {}
    inner = _lib.call

    def call(name, *a):
        seen[name] = seen.get(name, 0) + 1
        return inner(name, *a)
    _lib.call = call
    return seen


def _step_worker(rank, world, port, q, mode, overlap):
    dist = _init(rank, world, port)
    try:
        seen = _record_calls()
        import gae_dgl_amd as G
        from gae_dgl_amd import optim, ops
        from gae_dgl_amd.parallel import ShardedGraph, ShardedTrainStep
        from oracle import gae_oracle as O
        assert dist.get_backend() == "gloo" and dist.get_world_size() == world
        torch.set_num_threads(2)
        n, src, dst, X = step_graph(seed=11 + world)
        hidden = [32, 16]
        out = {}
        for transform_first in (False, True):
            torch.manual_seed(0)
            ref = O.CpuReferenceStep(src, dst, n, X, X.shape[1], hidden, lr=1e-2, seed=0, dropout=0.0)
            model = G.GAE(X.shape[1], hidden).to(DEV)
            model.decoder.dropout = 0.0
            with torch.no_grad():
                for conv, lin in zip(model.layers, ref.layers):
                    conv.apply_mod.linear.weight.copy_(lin.weight); conv.apply_mod.linear.bias.copy_(lin.bias)
            s = torch.from_numpy(src[rank::world]).to(DEV); d = torch.from_numpy(dst[rank::world]).to(DEV)
            sg = ShardedGraph.from_edge_slice(n, s, d, None, mode, DEV, "nnz", overlap)
            assert sg.n_edges_global() == len(src)
            p = sg.part
            assert 0 < p.n_local < n
            opt = optim.Adam(model.parameters(), lr=1e-2)
            step = ShardedTrainStep(model, opt, sg, torch.from_numpy(X[p.r0:p.r1]).to(DEV),
                                    transform_first=transform_first, capture=False)
            got = [float(step()) for _ in range(3)]
            want = [ref.step() for _ in range(3)]
            np.testing.assert_allclose(got, want, rtol=5e-5)
            for conv, lin in zip(model.layers, ref.layers):
                w = conv.apply_mod.linear.weight.detach().cpu()
                assert float((w - lin.weight).abs().max()) <= 5e-4 * float(lin.weight.abs().max())
            flat = torch.cat([t.detach().reshape(-1) for t in model.parameters()]).cpu()
            other = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(other, flat)
            assert all(torch.equal(o, flat) for o in other), "ranks ended a step with different weights"
            out[transform_first] = got
        for name in ("gae_csr_from_coo", "gae_spmm_csr", "gae_decoder_bce_rows", "gae_x_adam_step_tail", "gae_linear2_fwd",
                     "gae_gcn2_bwd_dense"):
            assert seen.get(name, 0) > 0, f"{name} was never called: {sorted(seen)}"
        q.put((rank, "ok", out))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def _rmat_worker(rank, world, port, q, mode, overlap):
    dist = _init(rank, world, port)
    try:
        seen = _record_calls()
        import gae_dgl_amd as G
        from gae_dgl_amd import workloads as W
        from gae_dgl_amd.parallel import LocalGroup, ShardedGraph, allreduce_grads, sharded_encode
        scale, F = 20, 32
        n = 1 << scale
        # ---- the N-rank result: this rank's slice of the list, its row block of X / dZ
        gen = torch.Generator(device=DEV).manual_seed(5)
        X = torch.rand(n, F, device=DEV, generator=gen)
        dZ = torch.randn(n, 16, device=DEV, generator=gen) / n
        s, d = W.rmat_edges(scale, 16, seed=0, device=DEV, part=(rank, world))
        sg = ShardedGraph.from_edge_slice(n, s, d, None, mode, DEV, "nnz", overlap)
        del s, d
        assert sg.n_edges_global() == 16 << scale
        sg.cache_constant_inputs = True
        p = sg.part
        torch.manual_seed(0)
        model = G.GAE(F, [32, 16]).to(DEV)
        runs = []
        xl, dzl = X[p.r0:p.r1].contiguous(), dZ[p.r0:p.r1].contiguous()
        for rep in range(2):                       # second pass: the exchanged rows of X come from the cache
            model.zero_grad()
            z = sharded_encode(model, sg, xl, transform_first=True)
            z.backward(dzl)
            allreduce_grads(list(model.parameters()))
            runs.append((z.detach().clone(), [t.grad.clone() for t in model.parameters()]))
        assert torch.equal(runs[0][0], runs[1][0]) and all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1])), \
            "the sharded encoder is not run-to-run bit-stable"
        z_n, g_n = runs[0]
        del sg
        # ---- the 1-rank result on the whole list (same kernels; every rank computes it for itself)
        s, d = W.rmat_edges(scale, 16, seed=0, device=DEV)
        sg1 = ShardedGraph(n, s, d, rank=0, group=LocalGroup(1), mode="allgather", device=DEV)
        del s, d
        sg1.group.publish(X)
        model.zero_grad()
        # (LocalGroup: the caller publishes the matrix each exchange assembles -- with one rank, the operand itself)
        from gae_dgl_amd import parallel as P
        inner = P.ShardedGraph.exchange

        def exchange(self, h_local, which="fwd"):
            if isinstance(self.group, LocalGroup):
                self.group.publish(h_local)
            return inner(self, h_local, which)
        P.ShardedGraph.exchange = exchange
        try:
            z1 = sharded_encode(model, sg1, X, transform_first=True)
            z1.backward(dZ)
        finally:
            P.ShardedGraph.exchange = inner
        g1 = [t.grad.clone() for t in model.parameters()]
        scale_z = float(z1.abs().max())
        err = float((z_n - z1[p.r0:p.r1].detach()).abs().max()) / scale_z
        assert err < 1e-5, f"Z rows of rank {rank}: {err}"
        for a, b in zip(g_n, g1):
            e = float((a - b).abs().max()) / float(b.abs().max())
            assert e < 5e-5, f"all-reduced gradient vs 1 rank: {e}"
        for name in ("gae_spmm_csr", "gae_spmm_csr_ep", "gae_linear2_fwd", "gae_gcn2_bwd_dense"):
            assert seen.get(name, 0) > 0, f"{name} was never called: {sorted(seen)}"
        q.put((rank, "ok", {"rows": p.n_local, "z_err": err}))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def _run(target, world, args, port_base, timeout=600):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port_base + (os.getpid() % 300) * 3
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=timeout) for _ in procs]
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()                     # the exact processes this test started
    assert all(r[1] == "ok" for r in res), [r[:2] for r in res]
    return sorted(res)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("mode,overlap", [("allgather", False), ("allgather", True), ("boundary", False), ("boundary", True)])
def test_sharded_training_step_in_separate_processes_on_the_hip_kernels(world, mode, overlap):
    res = _run(_step_worker, world, (mode, overlap), 33100 + 17 * world + (8 if overlap else 0) + (4 if mode == "boundary" else 0))
    # every rank reported the same losses (the scalar all-reduce)
    for r in res[1:]:
        assert r[2] == res[0][2]


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world,mode,overlap", [(2, "allgather", True), (4, "boundary", True), (4, "allgather", False)])
def test_rmat_s20_encoder_shards_in_separate_processes_match_one_rank(world, mode, overlap):
    res = _run(_rmat_worker, world, (mode, overlap), 34300 + 13 * world + (5 if mode == "boundary" else 0), timeout=900)
    assert sum(r[2]["rows"] for r in res) == 1 << 20


@pytest.mark.timeout(1500)
def test_bench_two_ranks_oversubscribed_prints_a_valid_line():
    """what the driver launches for N > 1, rehearsed with two ranks on the one GPU (gloo, staged): one JSON line with
    n_gpus 2, the exchange profile and the same workload timed on one rank"""
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "rmat",
                        "--rmat-scale", "20", "--oversubscribe", "--steps", "3", "--warmup", "2"],
                       capture_output=True, text=True, timeout=1400, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 2
    assert line["unit"] == "edges/s" and line["value"] > 0 and line["ms_per_step"] > 0
    assert line["scaling"] == "strong" and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["config"]["workload"].startswith("rmat-s20") and line["config"]["parallelism"] == "row-shard x2"
    assert "oversubscribed" in line["config"]["transport"]
    assert {"exchange_s_per_step", "spmm_s_per_step", "calls_per_step"} <= set(line["comm"])
    one = line["same_workload_1gpu"]
    assert one["value"] > 0 and one["ms_per_step"] > 0 and "speedup" in one
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1


# ------------------------------------------------------------------------------------------ data-parallel replicas
def _dp_worker(rank, world, port, q, save_dir):
    dist = _init(rank, world, port)
    try:
        seen = _record_calls()
        import gae_dgl_amd as G
        from gae_dgl_amd import train_inductive as TI
        from gae_dgl_amd.dataset import DeviceGraphDataset
        import argparse
        B, steps = 64, 5
        ds = DeviceGraphDataset.synthetic_zinc(B * world * steps + 37, seed=3, device=DEV)

        def make(replicas):
            torch.manual_seed(0)
            model = G.GAE(39, [32, 16]).to(DEV)
            model.decoder.dropout = 0.0                       # (one draw stream per process: the comparison below needs none)
            TI.device = torch.device(DEV)
            return model, TI.Trainer(model, argparse.Namespace(lr=1e-3), fused=True, replicas=replicas)

        # ---- the replicas: this rank's share of a seeded epoch order, gradients averaged by one all-reduce per step
        model, tr = make(True)
        loader = ds.loader(B, shuffle=True, seed=5, shard=(rank, world))
        assert len(loader) == steps + 1                      # 5 full batches + the ragged share of the 37
        losses = [float(tr.iteration(bg, train=True, as_tensor=True)) for bg in loader]
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
        other = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert all(torch.equal(o, flat) for o in other), "replicas ended the epoch with different weights"
        # ---- ONE process fed every replica's batch of a step, loss = their mean (train_inductive.py:44-53 once)
        ref, tr1 = make(False)
        iters = [iter(ds.loader(B, shuffle=True, seed=5, shard=(r, world))) for r in range(world)]
        from gae_dgl_amd import ops
        for _ in range(len(loader)):
            loss = sum(ref.reconstruction_loss(next(it)) for it in iters) / world
            tr1.optim.zero_grad(); ops.backward(loss); tr1.optim.step()
        for a, b in zip(model.parameters(), ref.parameters()):
            assert float((a - b).detach().abs().max()) <= 2e-6 * float(b.detach().abs().max()), \
                "replicas != one process fed all batches"
        for name in ("batch_gather", "decoder_bce", "adam_step"):
            assert any(name in k for k in seen), f"*{name}* was never called: {sorted(seen)}"
        # ---- the script itself: `train_inductive --distributed` (eager here: staged collectives cannot be captured)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), GAE_DIST_SHARE_GPUS="1")
        tl, vl = TI.main(["--distributed", "--synthetic", "700", "-b", "64", "--hidden_dims", "32", "16", "-e", "2",
                          "--no_plot", "--val_size", "100", "-s", save_dir, "--capture", "auto"])
        assert np.isfinite(tl).all() and np.isfinite(vl).all() and len(tl) == 2
        flat = torch.cat([v.reshape(-1).float() for v in TI.main.final_state.values()]).cpu()
        other = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert all(torch.equal(o, flat) for o in other)
        q.put((rank, "ok", {"losses": losses, "script": (tl, vl)}))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_replicas_in_separate_processes_equal_one_process(world, tmp_path):
    """the data-parallel half of SURVEY 8(e) on the real kernels: `world` processes (sharing cuda:0, gloo staged) train
    replicas on their shares of an epoch; they end with bit-identical weights, equal to one process fed all the batches
    of a step; `train_inductive --distributed` runs end to end and every rank reports the same epoch means"""
    res = _run(_dp_worker, world, (str(tmp_path),), 35500 + 11 * world)
    for r in res[1:]:
        assert r[2]["script"] == res[0][2]["script"]
    assert os.path.exists(tmp_path / "ep01.pkl")


@pytest.mark.timeout(1500)
def test_bench_zinc_two_replicas_oversubscribed_prints_a_valid_line():
    """`bench.py --workload zinc --gpus 2` (data-parallel replicas, weak scaling), rehearsed on the one GPU"""
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "zinc",
                        "--batch-graphs", "128", "--oversubscribe", "--steps", "5", "--warmup", "2", "--no-extra"],
                       capture_output=True, text=True, timeout=1400, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["unit"] == "edges/s" and line["value"] > 0
    assert line["config"]["parallelism"].startswith("data-parallel x2")
    assert "oversubscribed" in line["config"]["transport"] and "cpu_baseline" not in line
