"""SURVEY 8(e), the data-parallel half (train_inductive.py:84-96 on N GPUs), host logic on the CPU with world_size-2 gloo:
replicas that train on their shares of an epoch order and average their gradients with ONE all-reduce
(parallel.allreduce_grads(average=True)) end every step with bit-identical weights, equal to ONE process whose loss is
the mean of the replicas' batch losses.  (The HIP kernels need a GPU: tests/test_gpu_multiproc.py runs the same
protocol through Trainer on the real kernels.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_shard_order_deals_the_epoch_in_blocks():
    from gae_dgl_amd.dataset import shard_order
    order = np.random.default_rng(0).permutation(1003)
    for world in (1, 2, 3, 8):
        parts = [shard_order(order, r, world) for r in range(world)]
        assert len({len(p) for p in parts}) == 1 and len(parts[0]) == 1003 // world
        allp = np.concatenate(parts)
        assert len(np.unique(allp)) == len(allp)                      # disjoint
        # global batch k (world * B graphs) = the replicas' k-th batches
        B = 7
        for k in range(3):
            glob = set(order[k * B * world:(k + 1) * B * world].tolist())
            assert glob == set(np.concatenate([p[k * B:(k + 1) * B] for p in parts]).tolist())
        assert set(allp.tolist()) == set(order[:1003 // world * world].tolist())


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(39, 32), torch.nn.ReLU(), torch.nn.Linear(32, 16))


def _batches(seed=1, steps=4, world=2):
    g = torch.Generator().manual_seed(seed)
    return [[(torch.randn(50 + 7 * r, 39, generator=g), torch.randn(50 + 7 * r, 16, generator=g)) for r in range(world)]
            for _ in range(steps)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gae_dgl_amd.parallel import allreduce_grads
        torch.set_num_threads(1)
        model = _model()
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        params = list(model.parameters())
        for step in _batches():
            x, y = step[rank]
            loss = ((model(x) - y) ** 2).mean()
            opt.zero_grad(); loss.backward()
            allreduce_grads(params, None, average=True)
            opt.step()
            flat = torch.cat([p.detach().reshape(-1) for p in params])
            other = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(other, flat)
            assert all(torch.equal(o, flat) for o in other), "replicas diverged"
        q.put((rank, "ok", [p.detach().numpy().copy() for p in params]))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_two_replicas_equal_one_process_fed_both_batches():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32700 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)
    # one process: the loss of a step = mean of the replicas' batch losses
    torch.set_num_threads(1)
    model = _model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for step in _batches():
        loss = sum(((model(x) - y) ** 2).mean() for x, y in step) / world
        opt.zero_grad(); loss.backward(); opt.step()
    for a, b in zip(res[0][2], model.parameters()):
        b = b.detach().numpy()
        assert float(np.abs(a - b).max()) <= 1e-6 * float(np.abs(b).max())
