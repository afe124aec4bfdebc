"""Drop-in counterpart of gae_dgl/train_inductive.py: same flags
(train_inductive.py:18-27), ``collate``, ``Trainer`` (iteration / save) and
epoch loop, running on the HIP kernels of libgae_hip.so.

  python -m gae_dgl_amd.train_inductive --hidden_dims 32 16 -d data/zinc.npz
  python -m gae_dgl_amd.train_inductive --hidden_dims 32 16 --synthetic 20000 -b 4096

Differences, all deliberate: the reference's crashes are not reproduced
(``save_dir`` NameError at :71, ``plt.save()`` at :67); the dataset is a
device-resident block-diagonal CSR instead of a dill pickle of DGLGraphs
(DGL is not a dependency); the loss is evaluated by the fused decoder+BCE
kernel unless ``--loss dense`` asks for the reference-shaped N x N path."""
import argparse
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

import gae_dgl_amd as dgl
from gae_dgl_amd import ops, optim
from gae_dgl_amd.dataset import DeviceGraphDataset, MolDataset  # noqa: F401
from gae_dgl_amd.gae import GAE


# flag, short flag, type, default, help -- the reference's command line (train_inductive.py:18-27), same names,
# short forms and defaults; "hidden_dims" is a list of widths
_REFERENCE_FLAGS = (
    ("n_epochs", "e", int, 10, "passes over the training molecules"),
    ("data_file", "d", str, "data/graphs.pkl", "dataset (flat .npz of DeviceGraphDataset.save)"),
    ("save_dir", "s", str, "../result", "where checkpoints ep{NN}.pkl and the loss curve go"),
    ("in_dim", "i", int, 39, "atom feature width"),
    ("batch_size", "b", int, 128, "molecules per batch"),
    ("lr", None, float, 1e-3, "Adam step size"),
    ("gpu_id", None, int, 0, "which GPU"),
)


def build_parser():
    ap = argparse.ArgumentParser(description="Pre-train GAE")
    for name, short, kind, default, text in _REFERENCE_FLAGS:
        names = ["--" + name] + (["-" + short] if short else [])
        ap.add_argument(*names, type=kind, default=default, help=text)
    ap.add_argument("--hidden_dims", type=int, nargs="+", metavar="N", help="encoder widths, e.g. 32 16")
    # extensions
    ap.add_argument("--synthetic", type=int, default=0, metavar="G",
                    help="generate G ZINC-shaped molecules instead of reading --data_file")
    ap.add_argument("--val_size", type=int, default=10000, help="validation graphs (train_inductive.py:79)")
    ap.add_argument("--loss", choices=["fused", "dense"], default="fused")
    ap.add_argument("--criterion", choices=["bce", "mse"], default="bce",
                    help="bce = train_inductive.py:44-48 (weighted BCE with logits); mse = the hyper-parameter search's "
                         "nn.MSELoss() on the same logits and label (optuna_gae.py:16,21), eager steps only")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--no_plot", action="store_true")
    ap.add_argument("--capture", choices=["auto", "on", "off"], default="auto",
                    help="run the training batches as ONE captured HIP graph each (collate + forward + loss + "
                         "backward + Adam on a fixed-capacity batch, capture.CapturedInductiveStep): 7x faster steps "
                         "at batch 128, where the eager step is host-bound.  auto = on with the fused loss and the "
                         "device-resident iterator")
    ap.add_argument("--distributed", action="store_true",
                    help="data-parallel replicas, one process per GPU (launch with python -m torch.distributed.run "
                         "--nproc-per-node N -m gae_dgl_amd.train_inductive --distributed ...; RANK / LOCAL_RANK / "
                         "WORLD_SIZE / MASTER_* from the environment, backend nccl = RCCL over xGMI, or the backend "
                         "named in GAE_DIST_BACKEND): every replica draws the same epoch orders (--seed, default 0) and "
                         "trains on its share (dataset.shard_order); the parameter gradients are averaged by one small "
                         "all-reduce per step (1 808 floats for 39 -> 32 -> 16), inside the captured step -- the update "
                         "of one process whose loss is the mean of the replicas' batch losses.  Rank 0 saves and plots")
    ap.add_argument("--dataloader", action="store_true",
                    help="batch through torch's DataLoader + collate exactly like the reference (one small pinned "
                         "copy of the graph ids per batch) instead of the device-resident epoch iterator")
    return ap


args = None
device = torch.device("cpu")


def collate(samples):
    """DataLoader collate_fn of the reference (train_inductive.py:31-35): member graphs -> one block-diagonal
    graph on the device (gae_batch_gather for the device-resident dataset)"""
    target = torch.device(device)
    return dgl.batch([g.to(target) or g for g in samples])


class Trainer:
    """train_inductive.py:37-57: owns the optimiser; iteration() = one step (or one evaluation), save() = the
    reference's checkpoint files"""

    def __init__(self, model, args, fused=True, replicas=False, group=None):
        """``replicas``: data-parallel training -- iteration() averages the parameter gradients over the ranks of
        ``group`` before the optimiser step"""
        self.model, self.fused = model, fused
        self.criterion = getattr(args, "criterion", "bce")
        self.replicas, self.group = bool(replicas), group
        self.optim = optim.Adam(model.parameters(), lr=args.lr)     # torch.optim.Adam's rule, one HIP launch
        self._params = list(model.parameters())
        if not replicas or _rank() == 0:
            print("Total Parameters:", sum(p.nelement() for p in model.parameters()))

    def loss(self, g):
        if self.fused:
            return self.model.reconstruction_loss(g, criterion=self.criterion)
        # the reference-shaped path (train_inductive.py:44-48): dense label, pos_weight against the imbalance, N x N logits
        label = g.adjacency_matrix().to_dense().to(device)
        if self.criterion == "mse":
            return torch.nn.MSELoss()(self.model(g), label)                                  # optuna_gae.py:16,21
        n_pairs, n_pos = label.numel(), label.sum()
        return ops.bce_with_logits(self.model(g), label, pos_weight=(n_pairs - n_pos) / n_pos)   # gae_bce_logits

    def iteration(self, g, train=True, as_tensor=False):
        with torch.set_grad_enabled(train):
            loss = self.loss(g)
        if train:
            self.optim.zero_grad()
            ops.backward(loss)            # loss.backward() with a cached unit gradient
            if self.replicas:
                from gae_dgl_amd.parallel import allreduce_grads
                allreduce_grads(self._params, self.group, average=True)
            self.optim.step()
        return loss.detach() if as_tensor else loss.item()

    def save(self, epoch, save_dir):
        torch.save(self.model.state_dict(), os.path.join(save_dir, f"ep{epoch:02}.pkl"))


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_initialized() else 0


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_initialized() else 1


def plot(train_losses, val_losses, save_dir=None):
    try:
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot
    except Exception:  # matplotlib is optional here
        return
    fig, ax = pyplot.subplots()
    for series, name in ((train_losses, "train"), (val_losses, "val")):
        ax.plot(series, label=name)
    ax.set(xlabel="epoch", ylabel="loss")
    ax.grid(True); ax.legend()
    if save_dir:
        fig.savefig(os.path.join(save_dir, "zinc250k.png"))
    pyplot.close(fig)


def load_dataset(args):
    if args.synthetic:
        return DeviceGraphDataset.synthetic_zinc(args.synthetic, seed=args.seed or 0, device=device)
    if not os.path.exists(args.data_file):
        raise FileNotFoundError(f"{args.data_file} not found (use --synthetic G for ZINC-shaped synthetic data)")
    if args.data_file.endswith('.npz'):
        return DeviceGraphDataset.load(args.data_file, device=device)
    raise ValueError("the reference's dill pickle of DGLGraphs needs the `dgl` package; convert it to the flat "
                     ".npz format (DeviceGraphDataset.save: graph_ptr, src, dst, feat)")


def _run_epoch(trainer, loader, train, captured=None):
    """mean loss over the loader's batches; the running sum stays on the device (no host sync per iteration)"""
    total = torch.zeros((), device=device)
    if captured is not None:                     # the same batches (same shuffle), one graph launch each
        n = 0
        for loss in captured.epoch(loader.next_order()):
            total += loss
            n += 1
        return float(total) / max(n, 1)
    for bg in loader:
        for install in (bg.set_e_initializer, bg.set_n_initializer):      # train_inductive.py:93-94
            install(dgl.init.zero_initializer)
        total += trainer.iteration(bg, train=train, as_tensor=True)
    return float(total) / max(len(loader), 1)


def _mean_over_replicas(value):
    """the epoch's mean loss over all replicas (one scalar all-reduce per epoch)"""
    if _world() == 1:
        return value
    import torch.distributed as dist
    from gae_dgl_amd import transport
    t = torch.tensor([value], dtype=torch.float64, device=device)
    transport.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t) / _world()


def main(argv=None):
    global args, device
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("gae_dgl_amd runs on AMD GPUs only (no CPU fallback)")
    shard = None
    if args.distributed:
        import torch.distributed as dist
        created = not dist.is_initialized()
        if created:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(os.environ.get("GAE_DIST_BACKEND", "nccl"), rank=int(os.environ.get("RANK", "0")),
                                    world_size=int(os.environ.get("WORLD_SIZE", "1")))
        # one process per GPU; GAE_DIST_SHARE_GPUS=1: more ranks than devices share them (rehearsal over gloo)
        local = int(os.environ.get("LOCAL_RANK", str(dist.get_rank())))
        if os.environ.get("GAE_DIST_SHARE_GPUS") == "1":
            local %= torch.cuda.device_count()
        args.gpu_id = local
        shard = (dist.get_rank(), dist.get_world_size())
        if args.seed is None:
            args.seed = 0                       # the replicas must agree on the split, the initial weights and the orders
    device = torch.device(f"cuda:{args.gpu_id}")
    torch.cuda.set_device(device)
    if args.seed is not None:
        torch.manual_seed(args.seed); np.random.seed(args.seed)
    if _rank() == 0:
        os.makedirs(args.save_dir, exist_ok=True)

    model = GAE(args.in_dim, args.hidden_dims).to(device)
    if shard:
        # The replicas share ONE seed for the split, the initial weights and the epoch orders -- but each draws its own
        # dropout masks (its own Philox stream): replicas with identical masks on different batches would correlate the
        # dropout noise across the global batch, unlike one process drawing a mask per batch (gae.py:70)
        model.decoder.seed = int(args.seed) + 1000003 * shard[0]
    say = print if _rank() == 0 else (lambda *a, **k: None)
    say("Loading data")
    graphs = load_dataset(args)
    say(f"Loaded {len(graphs)} molecules" + (f" ({_world()} replicas)" if shard else ""))
    order = np.random.permutation(len(graphs))           # train_test_split(graphs, test_size=10000), :79
    n_val = args.val_size if len(graphs) > args.val_size else min(args.val_size, max(1, len(graphs) // 10))
    loaders = {}
    for split, ids, shuffle in (("train", order[n_val:], True), ("val", order[:n_val], False)):
        part = graphs.subset(graphs.ids[ids])
        if args.dataloader:      # train_inductive.py:84-85 verbatim
            loaders[split] = DataLoader(part, batch_size=args.batch_size, shuffle=shuffle, collate_fn=collate)
        else:                    # same batches, assembled from an epoch order that already lives on the device
            loaders[split] = part.loader(args.batch_size, shuffle=shuffle, seed=args.seed, shard=shard)
    if shard and args.dataloader:
        raise ValueError("--distributed uses the device-resident iterator (its epoch orders are seeded and sharded)")
    trainer = Trainer(model, args, fused=(args.loss == "fused"), replicas=shard is not None)
    captured = None
    can_capture = (args.loss == "fused" and args.criterion == "bce" and not args.dataloader
                   and len(loaders["train"].dataset) >= args.batch_size
                   and loaders["train"].dataset.ell_width and loaders["train"].dataset.no_heavy_rows
                   and args.hidden_dims[-1] <= ops.FUSED_MAX_D)
    why = ""
    if shard:
        from gae_dgl_amd import transport
        if transport.backend(None) != "nccl":                                # staged collectives cannot be captured
            can_capture, why = False, "; --distributed: the RCCL backend (collectives staged through the host cannot be captured)"
        if loaders["train"]._n_graphs() < args.batch_size:
            can_capture, why = False, "; --distributed: at least one full batch in every replica's share of the epoch"
        dropped = len(loaders["val"].dataset) % shard[1]
        if dropped:
            say(f"note: {dropped} of the {len(loaders['val'].dataset)} validation molecules fall outside the replicas' "
                f"equal shares and are not evaluated")
    if args.capture == "on" and not can_capture:
        raise ValueError("--capture on needs the fused loss, the device-resident iterator, a low-degree dataset with "
                         "at least one full batch and an embedding width <= %d%s" % (ops.FUSED_MAX_D, why))
    if args.capture != "off" and can_capture:
        from gae_dgl_amd.capture import CapturedInductiveStep
        captured = CapturedInductiveStep(model, trainer.optim, loaders["train"].dataset, args.batch_size,
                                         replicas=shard is not None)
    history = {"train": [], "val": []}
    say("Training Start")
    for epoch in range(args.n_epochs):
        model.train()
        history["train"].append(_mean_over_replicas(_run_epoch(trainer, loaders["train"], train=True, captured=captured)))
        if _rank() == 0:
            trainer.save(epoch, args.save_dir)
        model.eval()         # no effect on the decoder's dropout, exactly like the reference (gae.py:70)
        history["val"].append(_mean_over_replicas(_run_epoch(trainer, loaders["val"], train=False)))
        say(f"Epoch: {epoch:02d} | Train Loss: {history['train'][-1]:.4f} | "
            f"Validation Loss: {history['val'][-1]:.4f}")
    if not args.no_plot and _rank() == 0:
        plot(history["train"], history["val"], args.save_dir)
    main.final_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if args.distributed and created:
        captured = None                    # (a captured step holds RCCL kernels: release it before the communicator)
        import gc
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    return history["train"], history["val"]


if __name__ == '__main__':
    main()
