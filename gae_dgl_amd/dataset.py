"""Datasets for the inductive entry point.

``MolDataset`` keeps the interface of gae_dgl/dataset.py:3-12 (a list of graphs for the DataLoader).

``DeviceGraphDataset`` is the MI355X-native replacement of the pickled list of
DGLGraphs (gae_dgl/prepare_data.py:102-103, gae_dgl/train_inductive.py:76-85):
the whole molecule set lives on the GPU as ONE block-diagonal CSR (plus the CSR
of A^T for the backward SpMM when the set is not symmetric) and one feature
matrix -- one BYTE per 0/1 atom feature (prepare_data.py:31-36) --; ``dgl.batch``
of any subset of graphs is two HIP launches (gae_batch_plan: the prefix sums of
the batch on the device; gae_batch_gather: structure, fp32 features and the
packed neighbour table of the batch in one pass), no per-graph Python objects.

Host <-> device traffic of an epoch: ``epoch()`` uploads the epoch's
permutation once and slices it on the device -- no copy per batch.  ``batch()``
(what the reference's ``collate`` reaches through ``dgl.batch``, with graph ids
that arrive as a Python list) stages the ids in pinned memory: one small
asynchronous copy per batch, nothing else.

On-disk format (``save`` / ``load``): a .npz with ``graph_ptr`` [G+1] int64,
``src``/``dst`` [E] int64 (global node ids) and ``feat`` [N, F] -- the output
contract of the reference featuriser (per-graph ``ndata['h']`` fp32 [n_atoms,
39], both bond directions, no self loops) flattened; ``load`` checks that
contract (``featuriser_contract_errors``)."""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import ops
from ._lib import GaeHipError
from .graph import Graph

# gae_dgl/prepare_data.py:14-16,31-36: one-hot blocks of an atom's feature row (element, degree, formal charge,
# chirality) -- onek_encoding_unk always sets exactly one entry per block -- followed by the aromaticity flag
ATOM_BLOCKS = ((0, 23), (23, 29), (29, 34), (34, 38))
ATOM_FDIM = 39


def featuriser_contract_errors(graph_ptr, src, dst, feat, max_report=5):
    """Check flat arrays against the output contract of gae_dgl/prepare_data.py (host arrays; returns a list of
    messages, empty = conforming):
      * ``feat`` is [N, 39] with values 0 / 1; every one-hot block (23 elements, 6 degrees, 5 charges, 4 chiral tags:
        prepare_data.py:14-16,31-35) holds exactly one 1; column 38 is the aromaticity flag (prepare_data.py:36);
      * ``graph_ptr`` is a non-decreasing int64 [G+1] starting at 0 and ending at N;
      * every edge stays inside one member graph, there are no self loops (RDKit bonds join two different atoms),
        and the edge list holds both directions of every bond as adjacent entries (prepare_data.py:61-64)."""
    errs = []
    gp = np.asarray(graph_ptr, dtype=np.int64)
    src = np.asarray(src, dtype=np.int64); dst = np.asarray(dst, dtype=np.int64)
    feat = np.asarray(feat)
    if gp.ndim != 1 or len(gp) < 1 or gp[0] != 0 or (np.diff(gp) < 0).any():
        return ["graph_ptr must be a non-decreasing 1-D array starting at 0"]
    N = int(gp[-1])
    if feat.ndim != 2 or feat.shape[0] != N:
        errs.append(f"feat has shape {feat.shape}, expected [{N}, {ATOM_FDIM}]")
        return errs
    if feat.shape[1] != ATOM_FDIM:
        errs.append(f"feature width {feat.shape[1]} != {ATOM_FDIM} (23 + 6 + 5 + 4 + 1, prepare_data.py:16)")
        return errs
    if not np.isin(feat, (0, 1)).all():
        errs.append("features must be 0 / 1 (prepare_data.py:26-36 builds them from comparisons)")
    for lo, hi in ATOM_BLOCKS:
        bad = np.nonzero(feat[:, lo:hi].sum(1) != 1)[0]
        if bad.size:
            errs.append(f"one-hot block [{lo}, {hi}) does not hold exactly one 1 in {bad.size} rows "
                        f"(first: {bad[:max_report].tolist()})")
    if len(src) != len(dst):
        errs.append("src / dst length mismatch")
        return errs
    if len(src):
        if src.min() < 0 or dst.min() < 0 or src.max() >= N or dst.max() >= N:
            errs.append("edge endpoint outside [0, N)")
            return errs
        gs = np.searchsorted(gp, src, side="right") - 1
        gd = np.searchsorted(gp, dst, side="right") - 1
        cross = np.nonzero(gs != gd)[0]
        if cross.size:
            errs.append(f"{cross.size} edges join different member graphs (first: {cross[:max_report].tolist()})")
        loops = np.nonzero(src == dst)[0]
        if loops.size:
            errs.append(f"{loops.size} self loops (first: {loops[:max_report].tolist()})")
        if len(src) % 2 or not (np.array_equal(src[0::2], dst[1::2]) and np.array_equal(dst[0::2], src[1::2])):
            errs.append("edges are not (a, b), (b, a) pairs in adjacent positions (prepare_data.py:61-64)")
    return errs


class SequenceDataset(Dataset):
    """any indexable collection as a map-style torch Dataset"""

    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, index):
        return self.items[index]


class MolDataset(SequenceDataset):
    """the reference's dataset class (gae_dgl/dataset.py:3-12): the list of molecule graphs handed to the DataLoader of
    train_inductive.py:84-85, under the attribute name ``graphs``; reports its size like the reference does"""

    def __init__(self, graphs):
        super().__init__(graphs)
        print(f"Dataset includes {len(graphs):d} graphs")

    @property
    def graphs(self):
        return self.items


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
    def __init__(self, graph_ptr, src, dst, feat, device="cuda", ids=None, feat_storage="auto"):
        """``feat_storage``: "uint8" keeps 0/1 features in a byte each (expanded to fp32 by the batch gather),
        "float32" as given, "auto" = uint8 when every value is 0 or 1."""
        dev = torch.device(device)
        self.device = dev
        gp = np.asarray(graph_ptr, dtype=np.int64)
        if gp.ndim != 1 or len(gp) < 1 or gp[0] != 0 or (np.diff(gp) < 0).any():
            raise GaeHipError("DeviceGraphDataset: graph_ptr must be non-decreasing and start at 0")
        self.graph_ptr_host = gp
        self.sizes_host = np.diff(gp)
        N = int(gp[-1])
        self.n_nodes = N
        self.graph_ptr = torch.from_numpy(gp).to(dev)
        s = torch.as_tensor(np.asarray(src, dtype=np.int64)).to(dev)
        d = torch.as_tensor(np.asarray(dst, dtype=np.int64)).to(dev)
        if s.numel() != d.numel():
            raise GaeHipError("DeviceGraphDataset: src / dst length mismatch")
        if s.numel():
            if int(torch.minimum(s.min(), d.min())) < 0 or int(torch.maximum(s.max(), d.max())) >= N:
                raise GaeHipError("DeviceGraphDataset: edge endpoint outside [0, N)")
            # every edge must stay inside one member graph: a cross-graph edge would give column ids outside the
            # batch and the kernels would read out of bounds
            gid = torch.repeat_interleave(torch.arange(len(gp) - 1, device=dev),
                                          torch.as_tensor(self.sizes_host, device=dev), output_size=N)
            if not bool((gid[s] == gid[d]).all()):
                raise GaeHipError("DeviceGraphDataset: an edge joins two different member graphs")
            del gid
        self.indptr, self.indices = ops.csr_from_coo(d, s, N, N)        # rows = destination
        t_indptr, t_indices = ops.csr_from_coo(s, d, N, N)              # CSR of A^T
        # molecule sets hold both directions of every bond: A^T = A, one structure serves forward and backward
        self.symmetric = bool(torch.equal(self.indptr, t_indptr)) and bool(torch.equal(self.indices, t_indices))
        self.t_indptr, self.t_indices = (self.indptr, self.indices) if self.symmetric else (t_indptr, t_indices)
        ip = self.indptr.cpu().numpy().astype(np.int64)
        self.edges_host = ip[gp[1:]] - ip[gp[:-1]]                       # edges per graph (in-edges)
        tp = ip if self.symmetric else self.t_indptr.cpu().numpy().astype(np.int64)
        self.t_edges_host = tp[gp[1:]] - tp[gp[:-1]]
        feat = torch.as_tensor(np.asarray(feat) if not isinstance(feat, torch.Tensor) else feat)
        if feat.dim() != 2 or feat.shape[0] != N:
            raise GaeHipError(f"DeviceGraphDataset: feat has shape {tuple(feat.shape)}, expected [{N}, F]")
        self.n_feat = int(feat.shape[1])
        binary = feat_storage == "uint8" or (feat_storage == "auto" and bool(((feat == 0) | (feat == 1)).all()))
        if feat_storage == "uint8" and not bool(((feat == 0) | (feat == 1)).all()):
            raise GaeHipError("DeviceGraphDataset: uint8 feature storage needs 0 / 1 features")
        if binary:
            ld8 = (self.n_feat + 15) // 16 * 16                          # 16-byte rows (F = 39 -> 48 bytes)
            buf = torch.zeros(N, ld8, dtype=torch.uint8, device=dev)
            buf[:, :self.n_feat] = feat.to(dev).to(torch.uint8)
            self.feat = buf[:, :self.n_feat]
        else:
            self.feat = ops.pad_rows(feat.to(dev).float())               # F = 39 stored with ld = 40
        deg = (self.indptr[1:] - self.indptr[:-1])
        tdeg = deg if self.symmetric else (self.t_indptr[1:] - self.t_indptr[:-1])
        maxdeg = max(int(deg.max()) if N else 0, int(tdeg.max()) if N else 0)
        self.no_heavy_rows = maxdeg <= ops.SKEW_MIN_MAXDEG      # same rule as ops.spmm_plan(auto): no plan, no sync
        # packed neighbour table of every batch, written by the gather itself: the narrowest width that holds all
        # but 1 % of the atoms (bonded atoms have at most a handful of neighbours; the few longer rows continue from
        # the CSR arrays); 0 = none (a set with hub nodes)
        self.ell_width = ops.ell_width_for_degrees(torch.maximum(deg, tdeg)) if 0 < maxdeg <= 64 else 0
        self.max_nodes = int(self.sizes_host.max()) if len(self.sizes_host) else 0
        self.max_edges = int(max(self.edges_host.max(), self.t_edges_host.max())) if len(self.sizes_host) else 0
        self.ids = np.arange(len(gp) - 1, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
        self._pinned = None

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
        sub._pinned = None
        return sub

    # -------------------------------------------------------------- dgl.batch on the device
    def _assemble(self, d_gids, gids_host):
        """block-diagonal batch of the graphs d_gids (device int64) / gids_host (the same ids on the host: only their
        sizes are summed here -- no device read-back, no host -> device copy)"""
        B = len(gids_host)
        nb = int(self.sizes_host[gids_host].sum())
        eb = int(self.edges_host[gids_host].sum())
        dev = self.device
        node_ptr, edge_ptr, t_edge_ptr = ops.batch_plan(self.graph_ptr, self.indptr,
                                                        None if self.symmetric else self.t_indptr, d_gids)
        ip, ix, feat, table = ops.batch_gather(self.graph_ptr, self.indptr, self.indices, self.feat, d_gids, node_ptr,
                                               edge_ptr, nb, eb, ell_width=self.ell_width, n_feat=self.n_feat)
        if self.symmetric:
            tp, tx, t_table = ip, ix, table
        else:
            teb = int(self.t_edges_host[gids_host].sum())
            tp, tx, _, t_table = ops.batch_gather(self.graph_ptr, self.t_indptr, self.t_indices, None, d_gids,
                                                  node_ptr, t_edge_ptr, nb, teb, ell_width=self.ell_width)
        g = Graph(device=dev)
        g._n = nb
        g._src = g._dst = None          # structure lives in the CSR; edge list derived on demand
        g.set_csr(ip, ix, tp, tx)
        g.ndata['h'] = feat
        g.batch_num_nodes = self.sizes_host[gids_host].tolist()
        g._cache["graph_ptr"] = node_ptr                        # already on the device (graph-level readout)
        g.no_heavy_rows = self.no_heavy_rows
        if table is not None:
            g._cache["plan"] = ops.table_plan(table, self.ell_width)
            g._cache["plan_t"] = ops.table_plan(t_table, self.ell_width)
        offs = np.zeros(B + 1, dtype=np.int64); np.cumsum(self.sizes_host[gids_host], out=offs[1:])
        g.block_diag = ops.BlockDiag(offs, dev)         # whole molecules per thread block: LDS-staged SpMM
        return g

    def batch(self, graph_ids):
        """dgl.batch of the listed graphs.  The ids go to the device through a pinned staging buffer (one small
        asynchronous copy); everything else of the plan is computed there."""
        gids = np.asarray(graph_ids, dtype=np.int64)
        B = len(gids)
        # ring of pinned staging buffers: the host may run several batches ahead of the GPU, so a buffer is reused
        # only after the copy that read it has completed (its event)
        if self._pinned is None or self._pinned[0][0].numel() < B:
            self._pinned = [[torch.empty(max(B, 256), dtype=torch.int64).pin_memory(), None] for _ in range(4)]
            self._pin_next = 0
        slot = self._pinned[self._pin_next]
        self._pin_next = (self._pin_next + 1) % len(self._pinned)
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0][:B] = torch.from_numpy(gids)
        with torch.cuda.device(self.device):
            d_gids = slot[0][:B].to(self.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
        return self._assemble(d_gids, gids)

    def epoch(self, batch_size, shuffle=True, rng=None, drop_last=False, shard=None):
        """iterate over the batches of one epoch with NO host <-> device copy per batch: the epoch's order is
        uploaded once, every batch slices it on the device (what DataLoader(shuffle=True, collate_fn=collate) of
        train_inductive.py:84-85 does per batch on the host).  ``shard`` = (rank, world): this replica's share of the
        epoch (shard_order)"""
        order = self.ids.copy()
        if shuffle:
            (rng or np.random.default_rng()).shuffle(order)
        if shard is not None:
            order = shard_order(order, *shard)
        d_order = torch.from_numpy(order).to(self.device)
        n = len(order)
        stop = n - n % batch_size if drop_last else n
        for lo in range(0, stop, batch_size):
            hi = min(lo + batch_size, n)
            yield self._assemble(d_order[lo:hi], order[lo:hi])

    def loader(self, batch_size, shuffle=False, seed=None, shard=None):
        """DataLoader(dataset, batch_size, shuffle, collate_fn=collate) of train_inductive.py:84-85 for a resident
        dataset: an iterable with len() whose batches come from epoch().  ``shard`` = (rank, world): data-parallel
        replicas -- every replica draws the SAME epoch order (same seed) and keeps its share of it"""
        return DeviceLoader(self, batch_size, shuffle, seed, shard)

    # -------------------------------------------------------------- flat on-disk format
    @staticmethod
    def save(path, graph_ptr, src, dst, feat):
        np.savez_compressed(path, graph_ptr=np.asarray(graph_ptr, np.int64), src=np.asarray(src, np.int64),
                            dst=np.asarray(dst, np.int64), feat=np.asarray(feat))

    @classmethod
    def load(cls, path, device="cuda", validate=True):
        """``validate``: refuse files that do not follow the reference featuriser's output contract
        (featuriser_contract_errors: 39-wide one-hot layout, both bond directions, no self loops)"""
        z = np.load(path)
        missing = [k for k in ("graph_ptr", "src", "dst", "feat") if k not in z.files]
        if missing:
            raise GaeHipError(f"{path}: missing arrays {missing}")
        if validate:
            errs = featuriser_contract_errors(z["graph_ptr"], z["src"], z["dst"], z["feat"])
            if errs:
                raise GaeHipError(f"{path} violates the featuriser contract (gae_dgl/prepare_data.py): "
                                  + "; ".join(errs))
        return cls(z["graph_ptr"], z["src"], z["dst"], z["feat"], device=device)

    @classmethod
    def synthetic_zinc(cls, n_graphs=249455, seed=0, device="cuda", **kw):
        from . import workloads
        gp, s, d, X = workloads.zinc_like(n_graphs, seed)
        return cls(gp, s, d, X, device=device, **kw)


def shard_order(order, rank, world):
    """replica ``rank``'s share of an epoch order for data-parallel training (train_inductive.py:84-96 on ``world``
    GPUs): global batch k of B graphs is made of the replicas' k-th batches, so the order is dealt in blocks --
    replica r takes the graphs at positions r, r + world, ...; the ragged remainder (len(order) % world graphs) is
    dropped so that every replica runs the same number of steps (the collectives of a step must pair up)."""
    n = len(order) // world * world
    return np.ascontiguousarray(order[:n][rank::world])


class DeviceLoader:
    """what the reference builds with DataLoader(..., collate_fn=collate) (train_inductive.py:84-85), for a
    DeviceGraphDataset: iterating yields the block-diagonal batches of one epoch (fresh shuffle per epoch, the last
    short batch kept like DataLoader's default); no worker processes, no per-batch host -> device copy"""

    def __init__(self, dataset, batch_size, shuffle=False, seed=None, shard=None):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), bool(shuffle)
        self.rng = np.random.default_rng(seed)
        self.shard = shard
        if shard is not None and shuffle and seed is None:
            raise ValueError("data-parallel replicas must draw the same epoch orders: give the loader a seed")

    def _n_graphs(self):
        n = len(self.dataset)
        return n if self.shard is None else n // self.shard[1]

    def __len__(self):
        return (self._n_graphs() + self.batch_size - 1) // self.batch_size

    def next_order(self):
        """the graph ids of the next epoch, in batch order (consumes the same random numbers as __iter__)"""
        order = self.dataset.ids.copy()
        if self.shuffle:
            self.rng.shuffle(order)
        return order if self.shard is None else shard_order(order, *self.shard)

    def __iter__(self):
        return self.dataset.epoch(self.batch_size, shuffle=self.shuffle, rng=self.rng, shard=self.shard)
