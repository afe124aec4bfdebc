"""Stand-in for ``dgl.data.register_data_args`` / ``load_data``
(gae_dgl/train_transductive.py:6,19,37-38,45): citation datasets.

Real Planetoid files are used when present under ``--data_root``
(``<root>/<name>.npz`` with ``src``, ``dst``, ``features`` [, ``n``]); otherwise
a seeded synthetic graph with the dataset's N / E / F is generated
(workloads.citation_graph) -- the build/bench machines have no network."""
import os

import numpy as np


class EdgeListGraph:
    """what ``data.graph`` is in the reference (a networkx graph there): here a
    minimal edge container that ``DGLGraph(...)`` accepts."""

    def __init__(self, n, src, dst):
        self.n, self.src, self.dst = int(n), np.asarray(src, np.int64), np.asarray(dst, np.int64)

    def number_of_nodes(self):
        return self.n

    def is_directed(self):
        return True

    def edges(self):
        return list(zip(self.src.tolist(), self.dst.tolist()))


class CitationData:
    def __init__(self, name, features, graph, synthetic):
        self.name, self.features, self.graph, self.synthetic = name, features, graph, synthetic


def register_data_args(parser):
    parser.add_argument("--dataset", type=str, required=False, default="cora",
                        help="cora | citeseer | pubmed")
    parser.add_argument("--data_root", type=str, default="data", help="directory with <dataset>.npz (optional)")


def load_data(args):
    name = args.dataset.lower()
    path = os.path.join(getattr(args, "data_root", "data"), name + ".npz")
    if os.path.exists(path):
        z = np.load(path)
        feats = z["features"].astype(np.float32)
        n = int(z["n"]) if "n" in z.files else feats.shape[0]
        return CitationData(name, feats, EdgeListGraph(n, z["src"], z["dst"]), False)
    from . import workloads
    if name not in workloads.CITATION:
        raise ValueError(f"unknown dataset {name!r}")
    n, src, dst, X = workloads.citation_graph(name, seed=0)
    print(f"[gae_dgl_amd] {path} not found: using a seeded synthetic graph with {name}'s N/E/F")
    return CitationData(name, X, EdgeListGraph(n, src, dst), True)
