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
from torch.nn.functional import binary_cross_entropy_with_logits as BCELoss
from torch.utils.data import DataLoader

import gae_dgl_amd as dgl
from gae_dgl_amd import ops, optim
from gae_dgl_amd.dataset import DeviceGraphDataset, MolDataset  # noqa: F401
from gae_dgl_amd.gae import GAE


def build_parser():
    parser = argparse.ArgumentParser(description='Pre-train GAE')
    parser.add_argument('--n_epochs', '-e', type=int, default=10, help='number of epochs')
    parser.add_argument('--data_file', '-d', type=str, default='data/graphs.pkl', help='data file')
    parser.add_argument('--save_dir', '-s', type=str, default='../result', help='result directry')
    parser.add_argument('--in_dim', '-i', type=int, default=39, help='input dimension')
    parser.add_argument('--hidden_dims', metavar='N', type=int, nargs='+', help='list of hidden dimensions')
    parser.add_argument('--batch_size', '-b', type=int, default=128, help='batch size')
    parser.add_argument('--lr', type=float, default=1e-3, help='Adam learning rate')
    parser.add_argument('--gpu_id', type=int, default=0, help='GPU ID to use')
    # extensions
    parser.add_argument('--synthetic', type=int, default=0, metavar='G',
                        help='generate G ZINC-shaped molecules instead of reading --data_file')
    parser.add_argument('--val_size', type=int, default=10000, help='validation graphs (train_inductive.py:79)')
    parser.add_argument('--loss', choices=['fused', 'dense'], default='fused')
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--no_plot', action='store_true')
    return parser


args = None
device = torch.device("cpu")


def collate(samples):
    for g in samples:
        g.to(torch.device(device))
    bg = dgl.batch(samples)
    return bg


class Trainer:
    def __init__(self, model, args, fused=True):
        self.model = model
        self.optim = optim.Adam(self.model.parameters(), lr=args.lr)    # torch.optim.Adam's rule, one HIP launch
        self.fused = fused
        print('Total Parameters:', sum([p.nelement() for p in self.model.parameters()]))

    def loss(self, g):
        if self.fused:
            return self.model.reconstruction_loss(g)
        adj = g.adjacency_matrix().to_dense().to(device)
        # alleviate imbalance
        pos_weight = ((adj.shape[0] * adj.shape[0] - adj.sum()) / adj.sum())
        adj_logits = self.model.forward(g)
        return BCELoss(adj_logits, adj, pos_weight=pos_weight)

    def iteration(self, g, train=True, as_tensor=False):
        if train:
            loss = self.loss(g)
            self.optim.zero_grad()
            ops.backward(loss)            # loss.backward() with a cached unit gradient
            self.optim.step()
        else:
            with torch.no_grad():
                loss = self.loss(g)
        return loss.detach() if as_tensor else loss.item()

    def save(self, epoch, save_dir):
        output_path = os.path.join(save_dir, 'ep{:02}.pkl'.format(epoch))
        torch.save(self.model.state_dict(), output_path)


def plot(train_losses, val_losses, save_dir=None):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:  # matplotlib is optional here
        return
    plt.plot(train_losses, label='train')
    plt.plot(val_losses, label='val')
    plt.legend()
    plt.xlabel('epoch')
    plt.ylabel('loss')
    plt.grid()
    if save_dir:
        plt.savefig(os.path.join(save_dir, 'zinc250k.png'))


def load_dataset(args):
    if args.synthetic:
        return DeviceGraphDataset.synthetic_zinc(args.synthetic, seed=args.seed or 0, device=device)
    if not os.path.exists(args.data_file):
        raise FileNotFoundError(f"{args.data_file} not found (use --synthetic G for ZINC-shaped synthetic data)")
    if args.data_file.endswith('.npz'):
        return DeviceGraphDataset.load(args.data_file, device=device)
    raise ValueError("the reference's dill pickle of DGLGraphs needs the `dgl` package; convert it to the flat "
                     ".npz format (DeviceGraphDataset.save: graph_ptr, src, dst, feat)")


def main(argv=None):
    global args, device
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("gae_dgl_amd runs on AMD GPUs only (no CPU fallback)")
    device = torch.device("cuda:{}".format(args.gpu_id))
    torch.cuda.set_device(device)
    if args.seed is not None:
        torch.manual_seed(args.seed); np.random.seed(args.seed)
    if not os.path.exists(args.save_dir):
        os.makedirs(args.save_dir)

    model = GAE(args.in_dim, args.hidden_dims)
    model.to(device)
    print('Loading data')
    graphs = load_dataset(args)
    print('Loaded {} molecules'.format(len(graphs)))
    perm = np.random.permutation(len(graphs))            # train_test_split(graphs, test_size=10000), :79
    n_val = min(args.val_size, max(1, len(graphs) // 10)) if len(graphs) <= args.val_size else args.val_size
    train_dataset = graphs.subset(graphs.ids[perm[n_val:]])
    val_dataset = graphs.subset(graphs.ids[perm[:n_val]])

    train_loader = DataLoader(train_dataset, batch_size=args.batch_size, shuffle=True, collate_fn=collate)
    val_loader = DataLoader(val_dataset, batch_size=args.batch_size, shuffle=False, collate_fn=collate)
    trainer = Trainer(model, args, fused=(args.loss == 'fused'))
    train_losses, val_losses = [], []
    print('Training Start')
    for epoch in range(args.n_epochs):
        train_loss = torch.zeros((), device=device)
        model.train()
        for bg in train_loader:
            bg.set_e_initializer(dgl.init.zero_initializer)
            bg.set_n_initializer(dgl.init.zero_initializer)
            train_loss += trainer.iteration(bg, as_tensor=True)   # no per-iteration host sync
        train_loss = float(train_loss) / len(train_loader)
        train_losses.append(train_loss)
        trainer.save(epoch, args.save_dir)

        val_loss = torch.zeros((), device=device)
        model.eval()
        for bg in val_loader:
            bg.set_e_initializer(dgl.init.zero_initializer)
            bg.set_n_initializer(dgl.init.zero_initializer)
            val_loss += trainer.iteration(bg, train=False, as_tensor=True)
        val_loss = float(val_loss) / len(val_loader)
        val_losses.append(val_loss)
        print('Epoch: {:02d} | Train Loss: {:.4f} | Validation Loss: {:.4f}'.format(epoch, train_loss, val_loss))
    if not args.no_plot:
        plot(train_losses, val_losses, args.save_dir)
    return train_losses, val_losses


if __name__ == '__main__':
    main()
