"""Stand-in for ``dgl.data.register_data_args`` / ``load_data``
(gae_dgl/train_transductive.py:6,19,37-38,45): citation datasets.

Real data is used when present under ``--data_root``: either ``<root>/<name>.npz``
(``src``, ``dst``, ``features`` [, ``n``]) or the Planetoid files themselves
(``<root>/ind.<name>.{x,tx,allx,graph,test.index}``, also looked for in
``<root>/<name>/``: the format DGL's citation loader and Kipf's gcn read --
load_planetoid); otherwise a seeded synthetic graph with the dataset's N / E / F
is generated (workloads.citation_graph) -- the build/bench machines have no network."""
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


def planetoid_dir(root, name):
    """the directory under ``root`` that holds ``ind.<name>.graph`` (root itself or root/<name>), or None"""
    for d in (root, os.path.join(root, name)):
        if os.path.exists(os.path.join(d, f"ind.{name}.graph")):
            return d
    return None


def load_planetoid(root, name):
    """(n, src, dst, features) from the Planetoid files of Yang et al. 2016 as Kipf's gcn/utils.py and DGL 0.4's
    ``citation_graph.py`` read them [format from memory: there are no such files on the build machine; pinned here
    by round-trip tests on files written in the same layout]:

    * ``ind.<name>.allx`` / ``.tx``: scipy CSR feature rows of the training (labelled + unlabelled) / test nodes,
      pickled (python-2 pickles: ``encoding="latin1"``); ``ind.<name>.test.index``: one test-node id per line, in the
      order of the rows of ``tx``; features = vstack(allx, tx) with the test rows permuted back to their ids.  Test
      ids that are missing from the index (Citeseer's isolated nodes) become all-zero rows.
    * ``ind.<name>.graph``: dict {node: [neighbours]}.  As ``nx.DiGraph(nx.from_dict_of_lists(graph))`` builds it:
      an UNDIRECTED simple graph (repeated and mirrored entries collapse), then both directions of every pair; a
      self-loop stays one edge (Pubmed: 88 651 = 2 x 44 324 + 3).
    * features are row-normalised to sum 1, rows without entries stay zero (``_preprocess_features``), and returned
      dense fp32 as ``torch.FloatTensor(data.features)`` wants them (train_transductive.py:38)."""
    import pickle
    import scipy.sparse as sp

    def read(ext):
        with open(os.path.join(root, f"ind.{name}.{ext}"), "rb") as f:
            return pickle.load(f, encoding="latin1")
    allx, tx, graph = read("allx"), read("tx"), read("graph")
    with open(os.path.join(root, f"ind.{name}.test.index")) as f:
        reorder = np.asarray([int(line.strip()) for line in f if line.strip()], dtype=np.int64)
    allx, tx = sp.csr_matrix(allx), sp.csr_matrix(tx)
    if reorder.size != tx.shape[0]:
        raise ValueError(f"ind.{name}.test.index lists {reorder.size} nodes, ind.{name}.tx has {tx.shape[0]} rows")
    lo, hi = (int(reorder.min()), int(reorder.max())) if reorder.size else (allx.shape[0], allx.shape[0] - 1)
    if lo < allx.shape[0]:
        raise ValueError(f"ind.{name}.test.index: test ids must follow the {allx.shape[0]} rows of allx")
    # rows lo .. hi in id order; ids of that range that are not listed keep zero rows
    ext = sp.lil_matrix((hi - lo + 1, allx.shape[1]), dtype=np.float32)
    if reorder.size:
        ext[reorder - lo, :] = tx
    gap = sp.csr_matrix((lo - allx.shape[0], allx.shape[1]), dtype=np.float32)
    feats = sp.vstack([allx.astype(np.float32), gap, ext.tocsr()]).tocsr()
    n = feats.shape[0]
    rowsum = np.asarray(feats.sum(1)).reshape(-1)
    inv = np.where(rowsum != 0, 1.0 / np.where(rowsum != 0, rowsum, 1.0), 0.0).astype(np.float32)
    feats = sp.diags(inv).dot(feats)
    a, b = [], []
    for u, nbrs in graph.items():
        for v in nbrs:
            a.append(int(u)); b.append(int(v))
    a, b = np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64)
    n = max(n, int(max(a.max(initial=-1), b.max(initial=-1))) + 1)
    if n > feats.shape[0]:
        feats = sp.vstack([feats, sp.csr_matrix((n - feats.shape[0], feats.shape[1]), dtype=np.float32)]).tocsr()
    lo_, hi_ = np.minimum(a, b), np.maximum(a, b)
    pairs = np.unique(lo_ * n + hi_)
    u, v = pairs // n, pairs % n
    loops = u == v
    src = np.concatenate([u[~loops], v[~loops], u[loops]])
    dst = np.concatenate([v[~loops], u[~loops], v[loops]])
    return n, src, dst, np.asarray(feats.todense(), dtype=np.float32)


def load_data(args):
    name = args.dataset.lower()
    root = getattr(args, "data_root", "data")
    path = os.path.join(root, name + ".npz")
    if os.path.exists(path):
        z = np.load(path)
        feats = z["features"].astype(np.float32)
        n = int(z["n"]) if "n" in z.files else feats.shape[0]
        return CitationData(name, feats, EdgeListGraph(n, z["src"], z["dst"]), False)
    pdir = planetoid_dir(root, name)
    if pdir is not None:
        n, src, dst, feats = load_planetoid(pdir, name)
        print(f"[gae_dgl_amd] {name}: Planetoid files under {pdir}: {n} nodes, {src.size} edges, {feats.shape[1]} features")
        return CitationData(name, feats, EdgeListGraph(n, src, dst), False)
    from . import workloads
    if name not in workloads.CITATION:
        raise ValueError(f"unknown dataset {name!r}")
    n, src, dst, X = workloads.citation_graph(name, seed=0)
    print(f"[gae_dgl_amd] {path} not found: using a seeded synthetic graph with {name}'s N/E/F")
    return CitationData(name, X, EdgeListGraph(n, src, dst), True)
