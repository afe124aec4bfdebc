"""Compressed input features (opt-in: ``SparseFeatures.from_dense(X)`` always, ``SparseFeatures.maybe_from_dense(X)`` when it pays).

The reference hands its citation features to the model as a dense FloatTensor (train_transductive.py:37-38);
they are bag-of-words rows with 1-10 % non-zeros.  ``SparseFeatures.from_dense(X)`` keeps the non-zeros of X (rows:
nodes) and of X^T (rows: features) in compressed form; set as ``g.ndata['h']`` it makes layer 1 run from those --
same values (the skipped terms are exact zeros), a fifth to a hundredth of the bytes:

    g.ndata['h'] = gae_dgl_amd.SparseFeatures.from_dense(X)      # once; X is constant across epochs
    loss = model.reconstruction_loss(g)

Built by the library's HIP kernels (gae_dense_to_csr_count / _fill) plus prefix sums; columns ascending inside a row,
so sums meet their terms in the order the dense kernels would."""
import torch

from . import _lib, ops

SEGMENT = 64       # non-zeros of one feature per lane group of gae_spx_wgrad (csrc/spfeat.hip: kSeg)


def _compress(X):
    """(rowptr int32 [n + 1], col int32 [nnz], val fp32 [nnz]) of a dense fp32 device matrix"""
    X = ops._gpu(X, "X")
    if X.dtype != torch.float32 or X.dim() != 2:
        raise ops.GaeHipError("SparseFeatures: a 2-D fp32 device tensor is expected")
    if X.stride(1) != 1:
        X = X.contiguous()
    n, K = X.shape
    ld = X.stride(0) if n > 1 else max(K, 1)
    dev = X.device
    with ops._on_device(dev):
        cnt = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.call("gae_dense_to_csr_count", ops._ptr(X), ld, n, K, ops._ptr(cnt), ops._stream())
        rowptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        torch.cumsum(cnt, 0, out=rowptr[1:])
        nnz = int(rowptr[-1]) if n else 0
        col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
        _lib.call("gae_dense_to_csr_fill", ops._ptr(X), ld, n, K, ops._ptr(rowptr), ops._ptr(col), ops._ptr(val),
                  ops._stream())
    return rowptr, col[:nnz], val[:nnz]


class SparseFeatures:
    """compressed rows of X [n, K] and of X^T, plus the segment list gae_spx_wgrad walks"""

    def __init__(self, shape, rowptr, col, val, t_rowptr, t_row, t_val):
        self.shape = tuple(shape)
        self.rowptr, self.col, self.val = rowptr, col, val
        self.t_rowptr, self.t_row, self.t_val = t_rowptr, t_row, t_val
        K = self.shape[1]
        nnz_k = (t_rowptr[1:] - t_rowptr[:-1]).to(torch.int64)
        segs = ((nnz_k + SEGMENT - 1) // SEGMENT).clamp(min=1)           # an empty feature keeps one empty segment
        first = torch.cumsum(segs, 0) - segs
        feat = torch.repeat_interleave(torch.arange(K, device=segs.device), segs)
        slot = torch.arange(feat.numel(), device=segs.device) - first[feat]
        self.seg_feat = feat.to(torch.int32).contiguous()
        self.seg_slot = slot.to(torch.int32).contiguous()
        self.seg_e0 = (t_rowptr[:-1].to(torch.int64)[feat] + SEGMENT * slot).to(torch.int32).contiguous()
        self.max_segments = int(segs.max()) if K else 1

    # cost model of layer 1 per direction (measured on MI355X, tools/r04/spx_bench.py): the dense one-pass kernels move
    # 4 bytes per ENTRY of X; the kernels on the non-zeros move 8 bytes per non-zero plus one 128-byte line of the gather
    # operand (W^T / G, from L2) -- and pay about two dependent round trips more.  They win where X is wide and very
    # sparse (Citeseer, 3703 columns at 1.3 %: 17.5 vs 34.7 us for the pair), break even on Cora (1433 columns at 1.3 %:
    # 15.0 vs 16.8 us) and lose on Pubmed (500 columns at 9.5 %: 27.7 vs 26.4 us).
    BYTES_PER_NONZERO = 136
    WORTH_IT = 0.5
    MAX_SEGMENTS = 256      # densest column: at most 16384 non-zeros (workspace 32 K floats per slot)

    @classmethod
    def maybe_from_dense(cls, X, f_out=32, graph=None):
        """``X`` itself, or its compressed form when layer 1 runs faster from the non-zeros (one count pass over X; the
        compression only if it pays).  For features that stay CONSTANT across the steps that use the result.
        ``graph``: the graph the features will be set on -- the kernels on the non-zeros need plans that carry only a
        packed neighbour table (ops.sparse_input_usable: at most ops.ELL_MAX_ROWS rows, none longer than
        ops.TABLE_MAX_ROW -- the real Planetoid graphs with their hubs of 100-170 neighbours qualify); on any other graph the layer would densify the features
        again in every step, so X stays dense."""
        X = ops._gpu(X, "X")
        if X.dtype != torch.float32 or X.dim() != 2 or f_out > 32 or X.shape[1] < 193 or X.shape[0] == 0:
            return X
        if graph is not None and not ops.sparse_input_usable(graph, X.shape[0], X.shape[1], f_out):
            return X
        n, K = X.shape
        Xc = X if X.stride(1) == 1 else X.contiguous()
        cnt = torch.empty(n, dtype=torch.int32, device=X.device)
        with ops._on_device(X.device):
            _lib.call("gae_dense_to_csr_count", ops._ptr(Xc), Xc.stride(0) if n > 1 else max(K, 1), n, K, ops._ptr(cnt),
                      ops._stream())
        nnz = int(cnt.sum())
        if nnz * cls.BYTES_PER_NONZERO >= cls.WORTH_IT * 4.0 * n * K:
            return X
        sf = cls.from_dense(X)
        # gae_spx_wgrad's workspace and zero-fill scale with the DENSEST column (max_segments slots of 32 K floats): one
        # near-dense column (a bias feature, a stop word) on many rows would cost more than the dense pass it replaces
        if sf.max_segments > cls.MAX_SEGMENTS:
            return X
        return sf

    @classmethod
    def from_dense(cls, X):
        X = ops._gpu(X, "X")
        rowptr, col, val = _compress(X)
        t_rowptr, t_row, t_val = _compress(X.t().contiguous())
        return cls(X.shape, rowptr, col, val, t_rowptr, t_row, t_val)

    # ---- the little of the tensor interface the host mirror touches
    @property
    def device(self):
        return self.val.device

    @property
    def is_cuda(self):
        return True

    @property
    def dtype(self):
        return torch.float32

    @property
    def requires_grad(self):
        return False

    @property
    def nnz(self):
        return int(self.val.numel())

    def dim(self):
        return 2

    def to(self, device):
        dev = torch.device(device)
        if dev == self.device:
            return self
        moved = [t.to(dev) for t in (self.rowptr, self.col, self.val, self.t_rowptr, self.t_row, self.t_val)]
        return SparseFeatures(self.shape, *moved)

    def to_dense(self, cache=False):
        """the dense [n, K] matrix.  ``cache``: keep it on the object (the features are constant) -- what a layer does
        that cannot run from the non-zeros, so that it densifies once, outside any stream capture, not per step"""
        if getattr(self, "_dense", None) is not None:
            return self._dense
        n, K = self.shape
        out = ops.pad_rows(torch.zeros(n, K, dtype=torch.float32, device=self.device))     # rows as the loaders pad them
        if self.nnz:
            rows = torch.repeat_interleave(torch.arange(n, device=self.device),
                                           (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64), output_size=self.nnz)
            out[rows, self.col.to(torch.int64)] = self.val
        if cache:
            self._dense = out
        return out

    def __repr__(self):
        n, K = self.shape
        return f"SparseFeatures({n} x {K}, nnz = {self.nnz}, {100.0 * self.nnz / max(n * K, 1):.2f} %)"
