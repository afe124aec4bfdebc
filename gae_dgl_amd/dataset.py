"""Datasets for the inductive entry point.

``MolDataset`` mirrors gae_dgl/dataset.py:3-12 (a list wrapper for DataLoader).

``DeviceGraphDataset`` is the MI355X-native replacement of the pickled list of
DGLGraphs (gae_dgl/prepare_data.py:102-103, gae_dgl/train_inductive.py:76-85):
the whole molecule set lives on the GPU as ONE block-diagonal CSR (plus the CSR
of A^T for the backward SpMM) and one feature matrix; ``dgl.batch`` of any
subset of graphs is a single HIP gather kernel (gae_batch_gather), no per-graph
Python objects and no host->device copies inside the epoch loop.

On-disk format (``save`` / ``load``): a .npz with ``graph_ptr`` [G+1] int64,
``src``/``dst`` [E] int64 (global node ids) and ``feat`` [N, F] -- the output
contract of the reference featuriser (per-graph ``ndata['h']`` fp32 [n_atoms,
39], both bond directions, no self loops) flattened."""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import ops
from .graph import Graph


class MolDataset(Dataset):
    def __init__(self, graphs):
        self.graphs = graphs
        print('Dataset includes {:d} graphs'.format(len(graphs)))

    def __len__(self):
        return len(self.graphs)

    def __getitem__(self, item):
        return self.graphs[item]


class GraphView:
    """one member graph of a DeviceGraphDataset (what __getitem__ returns);
    ``batch()`` of such views is gathered on the device"""
    __slots__ = ("_ds", "_gid")

    def __init__(self, ds, gid):
        self._ds, self._gid = ds, int(gid)

    def number_of_nodes(self):
        return int(self._ds.sizes_host[self._gid])

    def to(self, device):  # collate() calls g.to(device) (train_inductive.py:33); data is already resident
        return self

    def materialize(self):
        return self._ds.batch([self._gid])


class DeviceGraphDataset(Dataset):
    def __init__(self, graph_ptr, src, dst, feat, device="cuda", ids=None):
        dev = torch.device(device)
        self.device = dev
        gp = np.asarray(graph_ptr, dtype=np.int64)
        self.graph_ptr_host = gp
        self.sizes_host = np.diff(gp)
        N = int(gp[-1])
        self.n_nodes = N
        self.graph_ptr = torch.from_numpy(gp).to(dev)
        s = torch.as_tensor(np.asarray(src, dtype=np.int64)).to(dev)
        d = torch.as_tensor(np.asarray(dst, dtype=np.int64)).to(dev)
        self.indptr, self.indices = ops.csr_from_coo(d, s, N, N)        # rows = destination
        self.t_indptr, self.t_indices = ops.csr_from_coo(s, d, N, N)    # CSR of A^T
        ip = self.indptr.cpu().numpy().astype(np.int64)
        self.edges_host = ip[gp[1:]] - ip[gp[:-1]]                       # edges per graph (in-edges)
        tp = self.t_indptr.cpu().numpy().astype(np.int64)
        self.t_edges_host = tp[gp[1:]] - tp[gp[:-1]]
        self.feat = ops.pad_rows(torch.as_tensor(feat).to(dev))          # F = 39 stored with ld = 40
        maxdeg = max(int((self.indptr[1:] - self.indptr[:-1]).max()) if N else 0,
                     int((self.t_indptr[1:] - self.t_indptr[:-1]).max()) if N else 0)
        self.no_heavy_rows = maxdeg <= ops.SKEW_MIN_MAXDEG      # same rule as ops.spmm_plan(auto): no plan, no sync
        self.ids = np.arange(len(gp) - 1, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)

    # -------------------------------------------------------------- Dataset protocol
    def __len__(self):
        return len(self.ids)

    def __getitem__(self, item):
        return GraphView(self, self.ids[item])

    def subset(self, ids):
        """a view over a subset of graphs sharing the same device arrays (train/val split)"""
        sub = object.__new__(DeviceGraphDataset)
        sub.__dict__.update(self.__dict__)
        sub.ids = np.asarray(ids, dtype=np.int64)
        return sub

    # -------------------------------------------------------------- dgl.batch on the device
    def batch(self, graph_ids):
        gids = np.asarray(graph_ids, dtype=np.int64)
        B = len(gids)
        node_ptr = np.zeros(B + 1, dtype=np.int64); np.cumsum(self.sizes_host[gids], out=node_ptr[1:])
        edge_ptr = np.zeros(B + 1, dtype=np.int64); np.cumsum(self.edges_host[gids], out=edge_ptr[1:])
        t_edge_ptr = np.zeros(B + 1, dtype=np.int64); np.cumsum(self.t_edges_host[gids], out=t_edge_ptr[1:])
        dev = self.device
        plan = torch.from_numpy(np.concatenate([gids, node_ptr, edge_ptr, t_edge_ptr])).to(dev)
        d_gids, d_np = plan[:B], plan[B:2 * B + 1]
        d_ep, d_tep = plan[2 * B + 1:3 * B + 2], plan[3 * B + 2:]
        nb, eb = int(node_ptr[-1]), int(edge_ptr[-1])
        ip, ix, feat = ops.batch_gather(self.graph_ptr, self.indptr, self.indices, self.feat, d_gids, d_np, d_ep,
                                        nb, eb)
        tp, tx, _ = ops.batch_gather(self.graph_ptr, self.t_indptr, self.t_indices, self.feat[:, :0], d_gids, d_np,
                                     d_tep, nb, int(t_edge_ptr[-1]))
        g = Graph(device=dev)
        g._n = nb
        g._src = g._dst = None          # structure lives in the CSR; edge list derived on demand
        g.set_csr(ip, ix, tp, tx)
        g.ndata['h'] = feat
        g.batch_num_nodes = self.sizes_host[gids].tolist()
        g.no_heavy_rows = self.no_heavy_rows
        g.block_diag = ops.BlockDiag(node_ptr, dev)     # whole molecules per thread block: LDS-staged SpMM
        return g

    # -------------------------------------------------------------- flat on-disk format
    @staticmethod
    def save(path, graph_ptr, src, dst, feat):
        np.savez_compressed(path, graph_ptr=np.asarray(graph_ptr, np.int64), src=np.asarray(src, np.int64),
                            dst=np.asarray(dst, np.int64), feat=np.asarray(feat))

    @classmethod
    def load(cls, path, device="cuda"):
        z = np.load(path)
        return cls(z["graph_ptr"], z["src"], z["dst"], z["feat"], device=device)

    @classmethod
    def synthetic_zinc(cls, n_graphs=249455, seed=0, device="cuda"):
        from . import workloads
        gp, s, d, X = workloads.zinc_like(n_graphs, seed)
        return cls(gp, s, d, X, device=device)
