"""Graph container exposing the subset of the DGL 0.4 API the reference uses
(SURVEY.md section 8(b)), backed by a device CSR and the HIP kernels of
libgae_hip.so.

Reference call sites mirrored here:
  DGLGraph() / add_nodes / add_edges          gae_dgl/prepare_data.py:48,53,65
  DGLGraph(networkx graph)                    gae_dgl/train_transductive.py:45
  g.ndata[...] get/set/pop                    gae_dgl/gae.py:27,30,50,53,58
  g.update_all(copy_src, sum)                 gae_dgl/gae.py:28       -> HIP SpMM
  g.apply_nodes(func)                         gae_dgl/gae.py:29
  g.in_degrees()                              gae_dgl/train_transductive.py:55
  g.adjacency_matrix().to_dense()             gae_dgl/train_inductive.py:44
  dgl.batch(samples), g.to(device)            gae_dgl/train_inductive.py:31-35
  set_n_initializer / set_e_initializer       gae_dgl/train_inductive.py:93-94
"""
import numpy as np
import torch

from . import function as fn
from ._lib import GaeHipError


# update_all on a graph WITHOUT ANY edge.  DGL 0.4 -- the API era of the reference: set_n_initializer /
# zero_initializer (train_inductive.py:93-94) exist only there -- short-cuts such a call in its scheduler
# (schedule_update_all: "all the nodes are zero degree; downgrade to apply nodes") and leaves the reduce output field
# untouched: gae.py:27-29 then feeds the layer INPUT to the Linear.  "dgl04" reproduces that (drop-in with the
# reference's DGL); "zeros" gives the mathematical aggregate A H = 0 (what DGL >= 0.5 and the raw SpMM kernel give).
# Graphs with at least one edge are unaffected: their zero-in-degree rows aggregate to 0 either way.
ZERO_EDGE_UPDATE_ALL = "dgl04"


class NodeBatch:
    """what DGL hands to an apply_nodes UDF: ``nodes.data`` is the ndata frame"""

    def __init__(self, data):
        self.data = data


def _as_index(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.int64).reshape(-1)
    return torch.as_tensor(np.asarray(x, dtype=np.int64).reshape(-1), device=device)


class Graph:
    """Directed multigraph with node features; edges are (src -> dst)."""

    def __init__(self, graph_data=None, num_nodes=None, device=None):
        self._device = torch.device(device) if device is not None else torch.device("cpu")
        self._n = 0
        self._src = torch.zeros(0, dtype=torch.int64, device=self._device)
        self._dst = torch.zeros(0, dtype=torch.int64, device=self._device)
        self.ndata = {}
        self.edata = {}
        self.norm_mode = "none"      # "none" = reference behaviour, "both" = D^-1/2 A D^-1/2
        self.batch_num_nodes = None  # node counts per member graph when built by batch()
        self.no_heavy_rows = False   # set by builders that know max degree <= ops.SKEW_THRESHOLD (skips the plan)
        self.block_diag = None       # ops.BlockDiag of a batched graph: enables the LDS-staged block-diagonal SpMM
        self._cache = {}
        if graph_data is not None:
            self._init_from(graph_data, num_nodes)
        elif num_nodes:
            self._n = int(num_nodes)

    # ------------------------------------------------------------ construction
    def _init_from(self, data, num_nodes):
        if isinstance(data, (tuple, list)) and len(data) == 2:
            src, dst = data
        elif hasattr(data, "src") and hasattr(data, "dst") and hasattr(data, "number_of_nodes"):
            src, dst = data.src, data.dst
            num_nodes = data.number_of_nodes() if num_nodes is None else num_nodes
        elif hasattr(data, "edges") and hasattr(data, "number_of_nodes"):  # networkx
            g = data if data.is_directed() else data.to_directed()
            num_nodes = g.number_of_nodes() if num_nodes is None else num_nodes
            e = np.asarray(list(g.edges()), dtype=np.int64).reshape(-1, 2)
            src, dst = e[:, 0], e[:, 1]
        else:
            raise TypeError(f"cannot build a Graph from {type(data)}")
        src = _as_index(src, self._device); dst = _as_index(dst, self._device)
        if num_nodes is None:
            num_nodes = int(max(src.max().item(), dst.max().item())) + 1 if src.numel() else 0
        self._n = int(num_nodes)
        self._src, self._dst = src, dst

    def add_nodes(self, num):
        self._n += int(num)
        self._cache.clear()

    def add_edges(self, u, v):
        self._edge_list()
        u = _as_index(u, self._device); v = _as_index(v, self._device)
        if u.numel() != v.numel():
            raise ValueError("add_edges: src/dst length mismatch")
        if u.numel() and (int(u.max()) >= self._n or int(v.max()) >= self._n or int(u.min()) < 0 or int(v.min()) < 0):
            raise ValueError("add_edges: node id out of range (add_nodes first)")
        self._src = torch.cat([self._src, u]); self._dst = torch.cat([self._dst, v])
        self._cache.clear()

    # ------------------------------------------------------------ queries
    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        if self._src is None:
            return int(self._cache["csr"][1].numel())
        return int(self._src.numel())

    def __len__(self):
        return self._n

    @property
    def device(self):
        return self._device

    def _edge_list(self):
        """(src, dst) int64; expanded from the CSR when the graph was built by the
        device batcher (rows = destination, CSR order)"""
        if self._src is None:
            indptr, indices = self._cache["csr"]
            counts = (indptr[1:] - indptr[:-1]).to(torch.int64)
            self._dst = torch.repeat_interleave(torch.arange(self._n, device=self._device), counts)
            self._src = indices.to(torch.int64)
        return self._src, self._dst

    def edges(self):
        return self._edge_list()

    def to(self, device):
        """in place like DGL 0.4 (the reference discards the result, train_inductive.py:33)"""
        device = torch.device(device)
        if device != self._device:
            self._edge_list()
            self._device = device
            self._src = self._src.to(device); self._dst = self._dst.to(device)
            self.ndata = {k: v.to(device) for k, v in self.ndata.items()}
            self.edata = {k: v.to(device) for k, v in self.edata.items()}
            self._cache.clear()
        return self

    def set_n_initializer(self, initializer, field=None):
        return None  # zero rows are what the SpMM writes for message-less nodes

    def set_e_initializer(self, initializer, field=None):
        return None

    # ------------------------------------------------------------ device structure
    def _require_gpu(self, what):
        if self._device.type != "cuda":
            raise GaeHipError(f"Graph.{what} runs HIP kernels and needs the graph on an AMD GPU "
                              f"(graph is on {self._device}); gae_dgl_amd has no CPU fallback")

    def _follow(self, tensor):
        """move the structure next to the features (the reference's collate leaves
        that to DGL, train_inductive.py:33)"""
        if isinstance(tensor, torch.Tensor) and tensor.is_cuda and self._device != tensor.device:
            self.to(tensor.device)

    def csr(self):
        """(indptr, indices) of A: rows = destination, cols = source"""
        if "csr" not in self._cache:
            from . import ops
            self._require_gpu("csr")
            self._edge_list()
            self._cache["csr"] = ops.csr_from_coo(self._dst, self._src, self._n, self._n)
        return self._cache["csr"]

    def csc(self):
        """CSR of A^T (rows = source): the structure of the backward SpMM"""
        if "csc" not in self._cache:
            from . import ops
            self._require_gpu("csc")
            self._edge_list()
            self._cache["csc"] = ops.csr_from_coo(self._src, self._dst, self._n, self._n)
        return self._cache["csc"]

    def spmm_plan(self, transposed=False):
        """plan of the CSR (or of the CSR of A^T): skew plan and / or packed neighbour table (ops.spmm_plan);
        None when the graph needs neither"""
        key = "plan_t" if transposed else "plan"
        if key not in self._cache:
            from . import ops
            indptr, indices = self.csc() if transposed else self.csr()
            self._cache[key] = None if self.no_heavy_rows else ops.spmm_plan(indptr, indices=indices)
        return self._cache[key]

    def scattered(self, row_bytes=2048):
        """True when the gathers of rows of ``row_bytes`` bytes have poor locality (ops.gather_scattered);
        block-diagonal batches never"""
        if "gather_distance" not in self._cache:
            from . import ops
            local = self.no_heavy_rows or self.block_diag is not None
            self._cache["gather_distance"] = 0 if local else ops.gather_distance(*self.csr())
        from . import ops
        return ops.gather_scattered(None, None, row_bytes, distance=self._cache["gather_distance"])

    def graph_ptr(self):
        """int64 [G + 1] node offsets of the member graphs (a graph that was not built by batch() is its own only
        member) on the graph's device"""
        if "graph_ptr" not in self._cache:
            counts = self.batch_num_nodes if self.batch_num_nodes is not None else [self._n]
            gp = np.zeros(len(counts) + 1, dtype=np.int64)
            np.cumsum(np.asarray(counts, dtype=np.int64), out=gp[1:])
            self._cache["graph_ptr"] = torch.from_numpy(gp).to(self._device)
        return self._cache["graph_ptr"]

    def set_csr(self, indptr, indices, t_indptr=None, t_indices=None):
        """adopt an already-built device CSR (used by the device dataset batcher)"""
        self._cache["csr"] = (indptr, indices)
        if t_indptr is not None:
            self._cache["csc"] = (t_indptr, t_indices)

    def in_degrees(self):
        from . import ops
        self._require_gpu("in_degrees")
        if "deg" not in self._cache:
            self._cache["deg"], self._cache["norm"] = ops.degree_norm(self.csr()[0])
        return self._cache["deg"].to(torch.int64)

    def norm(self):
        """in_degree^-1/2 with inf -> 0  (train_transductive.py:55-58), fp32 [N]"""
        self.in_degrees()
        return self._cache["norm"]

    def adjacency_matrix(self, transpose=False):
        """sparse COO [N, N]; rows = destination, cols = source (DGL 0.4
        ``transpose=False``); ``.to_dense()`` adds duplicate edges."""
        self._edge_list()
        r, c = (self._src, self._dst) if transpose else (self._dst, self._src)
        vals = torch.ones(r.numel(), dtype=torch.float32, device=self._device)
        return torch.sparse_coo_tensor(torch.stack([r, c]), vals, (self._n, self._n))

    def adjacency_sq_sum(self):
        """sum_ij a_ij^2 of the label ``adjacency_matrix().to_dense()`` (a_ij = multiplicity of the edge j -> i; E on a
        graph without repeated edges) as a float64 scalar on the device: the constant term of ops.decoder_mse.  Counted
        once per graph from the CSR (its rows hold their column ids sorted: repeated edges are neighbours)."""
        if "adj_sq" not in self._cache:
            indptr, indices = self.csr()
            e = int(indices.numel())
            if e == 0:
                self._cache["adj_sq"] = torch.zeros((), dtype=torch.float64, device=indices.device)
            else:
                rows = torch.repeat_interleave(torch.arange(self._n, device=indices.device),
                                               (indptr[1:] - indptr[:-1]).to(torch.int64), output_size=e)
                keys = rows * self._n + indices.to(torch.int64)
                _, cnt = torch.unique_consecutive(keys, return_counts=True)
                self._cache["adj_sq"] = (cnt.to(torch.float64) ** 2).sum()
        return self._cache["adj_sq"]

    def dense_adjacency(self):
        """the same label through the HIP kernel (duplicates add)"""
        from . import ops
        indptr, indices = self.csr()
        return ops.csr_to_dense(indptr, indices, self._n, self._n)

    # ------------------------------------------------------------ message passing
    def update_all(self, message_func, reduce_func, norm=None):
        if not (isinstance(message_func, fn.CopySrc) and isinstance(reduce_func, fn.SumReduce)
                and message_func.out == reduce_func.msg):
            raise NotImplementedError("only update_all(copy_src(src, m), sum(m, out)) is implemented "
                                      "(the pair gae.py:18-19 uses)")
        from . import ops
        h = self.ndata[message_func.src]
        self._follow(h)
        self._require_gpu("update_all")
        mode = self.norm_mode if norm is None else norm
        if mode not in ("none", "both"):
            raise ValueError(f"norm must be 'none' or 'both', got {mode!r}")
        if self.number_of_edges() == 0 and ZERO_EDGE_UPDATE_ALL == "dgl04":
            self.ndata.setdefault(reduce_func.out, h)     # DGL 0.4 leaves the frame as it is (see the note above)
            return
        self.ndata[reduce_func.out] = ops.spmm(self, h, use_norm=(mode == "both"))

    def apply_nodes(self, func):
        self.ndata.update(func(NodeBatch(self.ndata)))


DGLGraph = Graph


def readout_nodes(g, feat='h'):
    """[mean | sum | max] of ``g.ndata[feat]`` over the nodes of every member graph of a batched graph -> [G, 3 d]:
    the molecule feature of the reference's ESOL experiment (README.md:54; DGL: mean_nodes / sum_nodes / max_nodes).
    ``feat`` may also be the [N, d] tensor itself (e.g. ``model.encode(g)``)."""
    from . import ops
    z = g.ndata[feat] if isinstance(feat, str) else feat
    g._follow(z)
    return ops.segment_readout(z, g.graph_ptr())


def batch(graphs):
    """dgl.batch (train_inductive.py:34): block-diagonal union, node ids offset
    by the exclusive prefix sum of node counts, ndata concatenated on dim 0.

    Samples drawn from a :class:`gae_dgl_amd.dataset.DeviceGraphDataset` are
    gathered by one HIP kernel from the device-resident dataset CSR."""
    graphs = list(graphs)
    if graphs and all(getattr(g, "_ds", None) is not None for g in graphs) \
            and len({id(g._ds) for g in graphs}) == 1:
        return graphs[0]._ds.batch([g._gid for g in graphs])
    if not graphs:
        return Graph()
    dev = graphs[0].device
    counts = np.asarray([g.number_of_nodes() for g in graphs], dtype=np.int64)
    offs = np.zeros(len(graphs) + 1, dtype=np.int64)
    np.cumsum(counts, out=offs[1:])
    bg = Graph(device=dev)
    bg._n = int(offs[-1])
    bg._src = torch.cat([g.to(dev)._src + int(o) for g, o in zip(graphs, offs[:-1])])
    bg._dst = torch.cat([g._dst + int(o) for g, o in zip(graphs, offs[:-1])])
    keys = set(graphs[0].ndata)
    for k in keys:
        if all(k in g.ndata for g in graphs):
            bg.ndata[k] = torch.cat([g.ndata[k] for g in graphs], dim=0)
    bg.batch_num_nodes = counts.tolist()
    bg.norm_mode = graphs[0].norm_mode
    if dev.type == "cuda":
        from . import ops
        bg.block_diag = ops.BlockDiag(offs, dev)      # member graphs are closed under adjacency
    return bg
