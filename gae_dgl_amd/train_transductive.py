"""Counterpart of gae_dgl/train_transductive.py (which cannot run as written,
README.md:19): full-graph GAE training on a citation graph, reproducing the
file's INTENT -- ``GAE(in_feats, [32, 16])``, Adam lr 1e-2, 500 full-graph
epochs, ``pos_weight`` from the dense label (train_transductive.py:41,43,49,
59-60) -- on the HIP kernels.

  python -m gae_dgl_amd.train_transductive --dataset cora [--norm both]

``--norm both`` applies the ``deg^-1/2`` normalisation the reference computes
at :55-58 but never feeds to the model (north-star D^-1/2 A D^-1/2); the
default ``none`` is the reference's actual arithmetic."""
import argparse
import os

import torch

from gae_dgl_amd import DGLGraph
from gae_dgl_amd.data import load_data, register_data_args
from gae_dgl_amd.gae import GAE


def build_parser():
    parser = argparse.ArgumentParser(description='Pre-train GAE')
    register_data_args(parser)
    parser.add_argument('--n_epochs', '-e', type=int, default=500, help='number of epochs (reference hard-codes 500)')
    parser.add_argument('--save_dir', '-s', type=str, default='../result', help='result directry')
    parser.add_argument('--in_dim', '-i', type=int, default=39, help='input dimension (ignored: taken from data)')
    parser.add_argument('--hidden_dims', metavar='N', type=int, nargs='+', default=[32, 16],
                        help='list of hidden dimensions')
    parser.add_argument('--batch_size', '-b', type=int, default=128, help='unused (full graph)')
    parser.add_argument('--lr', type=float, default=1e-2, help='Adam learning rate')
    parser.add_argument('--gpu_id', type=int, default=0, help='GPU ID to use')
    parser.add_argument('--norm', choices=['none', 'both'], default='none')
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--log_every', type=int, default=50)
    parser.add_argument('--eval', action='store_true',
                        help="hold out 5 %% / 10 %% of the edges (the reference's '# TODO: train test split', :35) and "
                             "report link-prediction ROC-AUC / AP on them after training")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("gae_dgl_amd runs on AMD GPUs only (no CPU fallback)")
    device = torch.device("cuda:{}".format(args.gpu_id))
    torch.cuda.set_device(device)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    if not os.path.exists(args.save_dir):
        os.makedirs(args.save_dir)

    data = load_data(args)
    from gae_dgl_amd import ops
    from gae_dgl_amd.optim import Adam
    features = ops.pad_rows(torch.FloatTensor(data.features).to(device))   # rows of whole 16 / 128-byte units
    in_feats = features.shape[1]

    model = GAE(in_feats, args.hidden_dims, norm=args.norm).to(device)
    model.train()
    optim = Adam(model.parameters(), lr=args.lr)      # torch.optim.Adam's rule, one HIP launch

    split = None
    if args.eval:
        from gae_dgl_amd import metrics
        src, dst = (data.graph.src, data.graph.dst) if hasattr(data.graph, "src") else \
            tuple(map(list, zip(*data.graph.edges())))
        train, val, test = metrics.split_edges(src, dst, data.graph.number_of_nodes(), seed=args.seed or 0)
        split = (val, test)
        g = DGLGraph(train, num_nodes=data.graph.number_of_nodes()).to(device)
    else:
        g = DGLGraph(data.graph).to(device)
    # normalization (train_transductive.py:55-58) -- parameter independent, so once, not per epoch
    g.ndata['norm'] = g.norm().unsqueeze(1)

    losses = []
    print('Training Start')
    for epoch in range(args.n_epochs):
        g.ndata['h'] = features
        loss = model.reconstruction_loss(g)
        optim.zero_grad()
        ops.backward(loss)                # loss.backward() with a cached unit gradient
        optim.step()
        losses.append(loss.detach())
        if epoch % args.log_every == 0 or epoch == args.n_epochs - 1:
            print('Epoch: {:02d} | Loss: {:.5f}'.format(epoch, float(loss.detach())))
    torch.save(model.state_dict(), os.path.join(args.save_dir, 'transductive_{}.pkl'.format(args.dataset)))
    if split is not None:
        g.ndata['h'] = features
        with torch.no_grad():
            Z = model.encode(g)
        for name, sp in zip(("val", "test"), split):
            m = metrics.evaluate(Z, sp)
            print('{} ROC-AUC: {:.4f} | AP: {:.4f}'.format(name, m["auc"], m["ap"]))
        main.last_eval = metrics.evaluate(Z, split[1])
    return [float(l) for l in losses]


if __name__ == '__main__':
    main()
