"""Entry points: the drop-in training scripts run end to end on the GPU and
the device batcher reproduces dgl.batch semantics bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_device_dataset_batch_matches_host_batch():
    import gae_dgl_amd as G
    from gae_dgl_amd import workloads as W
    from gae_dgl_amd.dataset import DeviceGraphDataset
    from oracle import gae_oracle as O
    gp, src, dst, X = W.zinc_like(300, seed=3)
    ds = DeviceGraphDataset(gp, src, dst, X, device="cuda:0")
    ids = [7, 3, 250, 3, 0, 299]                    # order matters, repeats allowed
    bg = ds.batch(ids)
    parts = []
    for gi in ids:
        lo, hi = gp[gi], gp[gi + 1]
        m = (dst >= lo) & (dst < hi)
        parts.append((int(hi - lo), src[m] - lo, dst[m] - lo, X[lo:hi]))
    N, s, d, Xb, gptr = O.batch_graphs(parts)
    ip, ix = O.csr_from_coo(s, d, N)
    tp, tx = O.csc_from_coo(s, d, N)
    assert bg.number_of_nodes() == N and bg.number_of_edges() == len(s)
    assert np.array_equal(bg.csr()[0].cpu().numpy(), ip) and np.array_equal(bg.csr()[1].cpu().numpy(), ix)
    assert np.array_equal(bg.csc()[0].cpu().numpy(), tp) and np.array_equal(bg.csc()[1].cpu().numpy(), tx)
    assert np.array_equal(bg.ndata['h'].cpu().numpy(), Xb.numpy())
    assert np.array_equal(bg.adjacency_matrix().to_dense().cpu().numpy(), O.dense_adjacency(s, d, N).numpy())
    # DataLoader + collate path == explicit batch
    from gae_dgl_amd import train_inductive as TI
    TI.device = torch.device("cuda:0")
    bg2 = TI.collate([ds[i] for i in ids])
    assert torch.equal(bg2.csr()[1], bg.csr()[1]) and torch.equal(bg2.ndata['h'], bg.ndata['h'])


def test_train_inductive_runs_and_learns(tmp_path):
    from gae_dgl_amd import train_inductive as TI
    tr, va = TI.main(["--hidden_dims", "32", "16", "--synthetic", "3000", "-b", "256", "-e", "3", "--lr", "1e-2",
                      "--val_size", "300", "--seed", "0", "-s", str(tmp_path), "--no_plot"])
    assert len(tr) == 3 and all(np.isfinite(tr)) and all(np.isfinite(va))
    assert tr[-1] < tr[0]                                   # loss goes down (README loss-curve sanity band)
    sd = torch.load(tmp_path / "ep02.pkl")
    assert list(sd.keys()) == ["layers.0.apply_mod.linear.weight", "layers.0.apply_mod.linear.bias",
                               "layers.1.apply_mod.linear.weight", "layers.1.apply_mod.linear.bias"]


def test_fused_and_dense_trainers_agree(tmp_path):
    """one iteration of Trainer with the fused loss == the reference-shaped dense path"""
    import argparse
    import gae_dgl_amd as G
    from gae_dgl_amd import train_inductive as TI
    from gae_dgl_amd.dataset import DeviceGraphDataset
    TI.device = torch.device("cuda:0")
    ds = DeviceGraphDataset.synthetic_zinc(64, seed=1, device="cuda:0")
    losses = []
    for fused in (True, False):
        torch.manual_seed(0)
        model = G.GAE(39, [32, 16]).to("cuda:0")
        model.decoder.dropout = 0.0
        tr = TI.Trainer(model, argparse.Namespace(lr=1e-3), fused=fused)
        bg = ds.batch(list(range(64)))
        l0 = tr.iteration(bg)
        l1 = tr.iteration(ds.batch(list(range(64))), train=False)
        losses.append((l0, l1))
    assert abs(losses[0][0] - losses[1][0]) < 1e-5 * abs(losses[1][0])
    assert abs(losses[0][1] - losses[1][1]) < 2e-5 * abs(losses[1][1])


def test_train_transductive_runs(tmp_path):
    from gae_dgl_amd import train_transductive as TT
    losses = TT.main(["--dataset", "cora", "-e", "30", "-s", str(tmp_path), "--seed", "0", "--log_every", "100"])
    assert len(losses) == 30 and np.isfinite(losses).all() and losses[-1] < losses[0]
    losses_n = TT.main(["--dataset", "cora", "-e", "5", "-s", str(tmp_path), "--seed", "0", "--norm", "both",
                        "--log_every", "100"])
    assert np.isfinite(losses_n).all()


def test_hipgraph_captured_step_matches_eager():
    """replaying the captured step trains exactly like the eager step (dropout off), and the
    decoder mask still changes between replays (device-side draw counter) when dropout is on"""
    import gae_dgl_amd as G
    from gae_dgl_amd import workloads as W
    from gae_dgl_amd.capture import CapturedTrainStep
    dev = torch.device("cuda:0")
    n, src, dst, X = W.citation_graph("cora", seed=0)
    Xd = torch.from_numpy(X).to(dev)
    losses = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(0)
        model = G.GAE(X.shape[1], [32, 16]).to(dev)
        model.decoder.dropout = 0.0
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True)
        g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
        out = []
        if mode == "eager":
            for _ in range(8):
                g.ndata['h'] = Xd
                loss = model.reconstruction_loss(g)
                opt.zero_grad(); loss.backward(); opt.step()
                out.append(float(loss.detach()))
        else:
            step = CapturedTrainStep(model, opt, g, Xd, warmup=3)     # 3 eager warm-up steps are real steps
            out = [None] * 3 + [float(step().clone()) for _ in range(5)]
        losses[mode] = out
    np.testing.assert_allclose(losses["graph"][3:], losses["eager"][3:], rtol=2e-5)
    # dropout on: consecutive replays see different masks
    torch.manual_seed(0)
    model = G.GAE(X.shape[1], [32, 16]).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.0, capturable=True)   # lr 0: only the mask changes
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    step = CapturedTrainStep(model, opt, g, Xd, warmup=1)
    a, b = float(step().clone()), float(step().clone())
    assert a != b


def test_transductive_eval_learns_communities(tmp_path):
    """on a graph with planted communities the trained GAE ranks held-out edges above non-edges"""
    import os
    from gae_dgl_amd import train_transductive as TT
    rng = np.random.default_rng(0)
    n, k = 600, 6
    comm = rng.integers(0, k, n)
    a = rng.integers(0, n, 20000); b = rng.integers(0, n, 20000)
    keep = (comm[a] == comm[b]) & (a != b)
    keep |= (rng.random(a.size) < 0.02) & (a != b)
    a, b = a[keep], b[keep]
    feats = np.eye(k, dtype=np.float32)[comm] + 0.1 * rng.standard_normal((n, k)).astype(np.float32)
    os.makedirs(tmp_path / "data", exist_ok=True)
    np.savez(tmp_path / "data" / "cora.npz", src=np.concatenate([a, b]), dst=np.concatenate([b, a]), features=feats, n=n)
    TT.main(["--dataset", "cora", "--data_root", str(tmp_path / "data"), "-e", "150", "-s", str(tmp_path), "--seed", "0",
             "--eval", "--log_every", "1000"])
    assert TT.main.last_eval["auc"] > 0.8 and TT.main.last_eval["ap"] > 0.75
