"""Layer 1 on wide input features: the transform-first GCN layer from one-pass kernels (gae_xw_fwd / gae_spmm_csr_epilogue
/ gae_xw_wgrad), and the same layer from the non-zeros of sparse input features (gae_spx_*).

Part of the package gae_dgl_amd.ops (one module until round 6).  Functions look each other up in the PACKAGE
namespace (`_ops.<name>`) when they run: setting a flag or replacing a function on `gae_dgl_amd.ops` reaches every caller."""
import ctypes

import torch

import gae_dgl_amd.ops as _ops
from .. import _lib
from .._lib import ACT_IDENTITY, ACT_RELU, GaeHipError
from ._base import _dtype_code, _f32, _gpu, _on_device, _ptr, _rowmajor, _stream, _workspace

__all__ = [
    'xw_usable', 'xw_fwd_raw', 'xw_wgrad_raw', 'spmm_epilogue_raw', '_table_only', 'gcn_transform_first_usable',
    'GCNTransformFirstFunction', 'gcn_layer_transform_first', 'spx_fwd_raw', 'spx_wgrad_raw',
    'GCNSparseInputFunction', 'sparse_input_usable', 'gcn_layer_sparse_input',
]


# ------------------------------------------------------------------ transform-first GCN layer (wide in, narrow out)
def xw_usable(X, n_out):
    """can gae_xw_fwd / gae_xw_wgrad take this operand?  (fp32 or bf16 rows of whole 16-byte vectors, f_in >= 193,
    f_out <= 32, X below 3.5 GiB)"""
    if not isinstance(X, torch.Tensor) or not X.is_cuda or X.dim() != 2 or X.dtype not in (torch.float32, torch.bfloat16):
        return False
    if X.shape[0] == 0 or X.stride(1) != 1:
        return False
    return bool(_lib.load().gae_xw_usable(_ptr(X), X.stride(0), _dtype_code(X), X.shape[0], X.shape[1], int(n_out)))


def xw_fwd_raw(X, W, b, act, keep_splits=False):
    """P = act(X W^T + b) with X read once and W stationary in registers (gae_xw_fwd); X fp32 or bf16 storage.
    ``keep_splits`` (b None, identity): when the library splits a long f_in over thread blocks, return the partial
    products [splits, n, f_out] instead of their sum (a consumer adds them: spmm_epilogue_raw) -- returns (P, 1) when
    there is no split"""
    W = _f32(_gpu(W, "W"), "xw_fwd: W")
    if W.stride(1) != 1:
        W = W.contiguous()
    _f32(b, "xw_fwd: b")
    n, f_in = X.shape
    f_out = W.shape[0]
    code = _dtype_code(X)
    lib = _lib.load()
    _ops.STATS["xw_fwd"] += 1
    splits = int(lib.gae_xw_fwd_splits(n, f_in, f_out, code)) if keep_splits else 1
    keep = keep_splits and splits > 1
    with _on_device(X.device):
        nbytes = lib.gae_xw_fwd_workspace_bytes(n, f_in, f_out, code)
        if nbytes < 0:
            _lib.check(int(nbytes), "gae_xw_fwd_workspace_bytes")
        if keep:       # the partials ARE the result: a buffer of their own, not the shared scratch
            parts = torch.empty(splits, n, f_out, dtype=torch.float32, device=X.device)
            ws, ws_n, P = parts, parts.numel() * 4, None
        else:
            ws = _workspace(nbytes, X.device) if nbytes > 0 else None
            ws_n = ws.numel() if ws is not None else 0
            P = torch.empty(n, f_out, dtype=torch.float32, device=X.device)

        def launch():
            _lib.call("gae_xw_fwd", _ptr(X), X.stride(0), code, n, f_in, _ptr(W), W.stride(0), _ptr(b), f_out, int(act),
                      _ptr(P), max(f_out, 1), _ptr(ws), ws_n, 1 if keep else 0, _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("xw_fwd", n, f_in, f_out, str(X.dtype)), launch)
        else:
            launch()
    if keep_splits:
        return (parts, splits) if keep else (P, 1)
    return P


def xw_wgrad_raw(X, G, Gmask, D, Dmask, f_out, need_dW=True, need_db=True):
    """(dW [f_out, f_in] = (G (.) [Gmask > 0])^T X, db [f_out] = colsum(D (.) [Dmask > 0])) in one pass over X
    (gae_xw_wgrad); masks may be None"""
    n, f_in = X.shape
    code = _dtype_code(X)
    dev = X.device
    _ops.STATS["xw_wgrad"] += 1
    G, ldg = _rowmajor(_f32(G, "xw_wgrad: G"), "G")
    ldgm = ldd = lddm = 0
    if Gmask is not None:
        Gmask, ldgm = _rowmajor(_f32(Gmask, "xw_wgrad: Gmask"), "Gmask")
    if D is not None:
        D, ldd = _rowmajor(_f32(D, "xw_wgrad: D"), "D")
    if Dmask is not None:
        Dmask, lddm = _rowmajor(_f32(Dmask, "xw_wgrad: Dmask"), "Dmask")
    dW = torch.empty(f_out, f_in, dtype=torch.float32, device=dev) if need_dW else None
    db = torch.empty(f_out, dtype=torch.float32, device=dev) if need_db and D is not None else None
    if _ops.current_step().defer_grads and (dW is not None or db is not None):
        with _on_device(dev):
            ws = torch.empty(_lib.load().gae_xw_wgrad_workspace_bytes(n, f_in, code), dtype=torch.uint8, device=dev)
            lay = (ctypes.c_int64 * 8)()
            _lib.call("gae_x_xw_wgrad_partials", _ptr(X), X.stride(0), code, n, f_in, _ptr(G), ldg, _ptr(Gmask), ldgm,
                      _ptr(D), ldd, _ptr(Dmask), lddm, int(f_out), int(dW is not None), int(db is not None), _ptr(ws),
                      ws.numel(), lay, _stream())
        if dW is not None:
            _ops.current_step().add_partials(dW, (ws, ws.data_ptr(), lay[0], lay[1], f_in, lay[2]))
        if db is not None:
            _ops.current_step().add_partials(db, (ws, ws.data_ptr() + 4 * lay[3], lay[4], lay[5], f_out, f_out))
        return dW, db
    with _on_device(dev):
        ws = _workspace(_lib.load().gae_xw_wgrad_workspace_bytes(n, f_in, code), dev)

        def launch():
            _lib.call("gae_xw_wgrad", _ptr(X), X.stride(0), code, n, f_in, _ptr(G), ldg, _ptr(Gmask), ldgm, _ptr(D), ldd,
                      _ptr(Dmask), lddm, int(f_out), _ptr(dW), max(f_in, 1), _ptr(db), _ptr(ws), ws.numel(), _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("xw_wgrad", n, f_in, f_out, str(X.dtype)), launch)
        else:
            launch()
    return dW, db


def spmm_epilogue_raw(indptr, indices, H, n_rows, plan, bias=None, act=ACT_IDENTITY, Hmask=None, row_scale=None,
                      col_scale=None):
    """Y = act(diag(rs) A diag(cs) (H (.) [Hmask > 0]) + bias) in one launch of the packed-table kernel
    (gae_spmm_csr_epilogue); H fp32 [n_cols, F <= 64] with rows of whole 16-byte vectors -- or a contiguous stack
    [splits, n_cols, F] of partial matrices (xw_fwd_raw(keep_splits=True)): a gathered row is then the sum of its
    partial rows in split order"""
    n_splits, split_stride = 1, 0
    if H.dim() == 3:
        if Hmask is not None or not H.is_contiguous() or H.shape[2] % 4 or H.data_ptr() % 16:
            H = H.sum(0)                                    # (not reached by the library's own callers)
        else:
            n_splits, split_stride = int(H.shape[0]), int(H.shape[1] * H.shape[2])
            H = H[0]
    H, ldh = _rowmajor(_f32(_gpu(H, "H"), "spmm_epilogue: H"), "H")
    if ldh % 4 or H.data_ptr() % 16:
        H = _ops.pad_rows(H); ldh = H.stride(0)
    n_cols, F = H.shape
    if Hmask is not None:
        Hmask, ldk = _rowmajor(_f32(_gpu(Hmask, "Hmask"), "spmm_epilogue: Hmask"), "Hmask")
        if ldk != ldh or Hmask.data_ptr() % 16:
            buf = torch.empty(n_cols, ldh, dtype=torch.float32, device=H.device)[:, :F]
            buf.copy_(Hmask)
            Hmask = buf
    _f32(bias, "spmm_epilogue: bias")
    ldy = (F + 3) // 4 * 4
    Y = torch.empty(n_rows, ldy, dtype=torch.float32, device=H.device)[:, :F]
    with _on_device(H.device):
        def launch():
            _lib.call("gae_spmm_csr_epilogue", _ptr(indptr), _ptr(indices), n_rows, n_cols, _ptr(H), ldh, _ptr(Hmask),
                      _ptr(Y), ldy, F, _ptr(row_scale), _ptr(col_scale), ctypes.byref(plan.c), _ptr(bias), int(act),
                      n_splits, split_stride, _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch)
        else:
            launch()
    return Y


def _table_only(plan):
    return plan is not None and plan.ell is not None and plan.n_heavy == 0 and plan.homed is None


def gcn_transform_first_usable(graph, H, n_out):
    """can GCNTransformFirstFunction run this layer?  A layer that narrows wide features (gae_xw_usable), on a graph
    whose plans carry a packed neighbour table and no heavy rows (every citation / molecule graph)"""
    if not _ops.xw_usable(H, n_out) or graph.number_of_edges() == 0:
        return False
    n = graph.number_of_nodes()
    if n != H.shape[0] or n * ((n_out + 3) // 4 * 4) * 4 + (1 << 16) >= (1 << 32):
        return False
    return _table_only(graph.spmm_plan(False)) and _table_only(graph.spmm_plan(True))


class GCNTransformFirstFunction(torch.autograd.Function):
    """GCN.forward (gae.py:26-31) of a layer that narrows its features, evaluated as Y = act(A (H W^T) + b) -- the
    value of the reference's act((A H) W^T + b) up to fp32 rounding -- in three launches forward (gae_xw_fwd,
    gae_spmm_csr_epilogue) and backward (gae_spmm_csr_epilogue on A^T with the ReLU gate in the gather,
    gae_xw_wgrad): H is read once per direction and nothing of the input width is written."""

    @staticmethod
    def forward(ctx, H, W, b, graph, use_norm, act):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        # (a long f_in is split over thread blocks: the aggregation adds the partial rows itself, no reduction launch)
        # (gae_spmm_csr_epilogue addresses the stack of partials through one raw buffer: splits * n * f_out * 4 < 2^27 bytes;
        # larger operands let gae_xw_fwd reduce the splits itself)
        f_out = W.shape[0]
        splits = int(_lib.load().gae_xw_fwd_splits(H.shape[0], H.shape[1], f_out, _dtype_code(H))) if f_out % 4 == 0 else 1
        keep = f_out % 4 == 0 and splits > 1 and splits * H.shape[0] * f_out * 4 < (1 << 27)
        P = _ops.xw_fwd_raw(H, W, None, ACT_IDENTITY, keep_splits=keep)
        if keep:
            P = P[0]
        Y = _ops.spmm_epilogue_raw(indptr, indices, P, n, graph.spmm_plan(False), b, act, None, norm, norm)
        ctx.act, ctx.has_bias = act, b is not None
        ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True))
        ctx.save_for_backward(H, W, Y if act == ACT_RELU else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        H, W, Y = ctx.saved_tensors
        (t_indptr, t_indices), n, norm, plan_t = ctx.bwd
        need_dH, need_dW = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        dW = db = dH = None
        dYc, _ = _rowmajor(_f32(dY, "dY"), "dY")
        G = None
        if need_dH or need_dW:
            G = _ops.spmm_epilogue_raw(t_indptr, t_indices, dYc, n, plan_t, None, ACT_IDENTITY, Y, norm, norm)   # A^T dYm
        if need_dW or need_db:
            dW, db = _ops.xw_wgrad_raw(H, G if need_dW else dYc, None, dYc if need_db else None, Y, W.shape[0],
                                  need_dW=need_dW, need_db=need_db)
        if need_dH:
            if H.dtype != torch.float32:
                raise GaeHipError("transform-first layer: a bf16-stored input cannot receive a gradient")
            _, _, dH = _ops.linear_bwd_raw(G, None, ACT_IDENTITY, H, W, False, False, True)      # dH = G W
        return dH, dW, db, None, None, None


def gcn_layer_transform_first(graph, H, W, b, act, use_norm=False):
    """the layer as GCNTransformFirstFunction, or None when the shapes / the graph do not allow it"""
    if not isinstance(H, torch.Tensor) or not H.is_cuda:
        return None
    if H.dtype == torch.float32 and (H.stride(1) != 1 or H.stride(0) % 4 or H.data_ptr() % 16):
        H = _ops.pad_rows(H)
    if not _ops.gcn_transform_first_usable(graph, H, W.shape[0]):
        return None
    return _ops.GCNTransformFirstFunction.apply(H, W, b, graph, use_norm, act)


# ------------------------------------------------------------------ layer 1 on sparse input features (opt-in)
def spx_fwd_raw(sf, W):
    """P = X W^T from the compressed rows of X (gae_spx_fwd); sf: sparse.SparseFeatures"""
    W = _f32(_gpu(W, "W"), "spx_fwd: W")
    if W.stride(1) != 1:
        W = W.contiguous()
    n, K = sf.shape
    J = W.shape[0]
    ldp = (J + 3) // 4 * 4
    P = torch.empty(n, ldp, dtype=torch.float32, device=W.device)[:, :J]
    with _on_device(W.device):
        ws = _workspace(K * 32 * 4 + 256, W.device)

        def launch():
            _lib.call("gae_spx_fwd", _ptr(sf.rowptr), _ptr(sf.col), _ptr(sf.val), n, K, _ptr(W), W.stride(0), J, _ptr(P),
                      ldp, _ptr(ws), ws.numel(), _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("spx_fwd", n, K, J, sf.nnz), launch)
        else:
            launch()
    return P


def spx_wgrad_raw(sf, G, D, Dmask, f_out, need_dW=True, need_db=True):
    """(dW = G^T X from the compressed rows of X^T, db = colsum(D (.) [Dmask > 0])) -- gae_spx_wgrad; inside
    deferred_grad_reductions() the sums stay partial lists for the optimiser launch"""
    n, K = sf.shape
    dev = G.device
    G, ldg = _rowmajor(_f32(G, "spx_wgrad: G"), "G")
    ldd = lddm = 0
    if D is not None:
        D, ldd = _rowmajor(_f32(D, "spx_wgrad: D"), "D")
    if Dmask is not None:
        Dmask, lddm = _rowmajor(_f32(Dmask, "spx_wgrad: Dmask"), "Dmask")
    dW = torch.empty(f_out, K, dtype=torch.float32, device=dev) if need_dW else None
    db = torch.empty(f_out, dtype=torch.float32, device=dev) if need_db and D is not None else None
    lay = (ctypes.c_int64 * 5)()
    _lib.call("gae_spx_wgrad_layout", n, K, sf.max_segments, lay)
    defer = _ops.current_step().defer_grads and (dW is not None or db is not None)
    with _on_device(dev):
        ws = torch.empty(lay[4], dtype=torch.uint8, device=dev) if defer else _workspace(lay[4], dev)

        def launch():
            _lib.call("gae_spx_wgrad", _ptr(sf.t_rowptr), _ptr(sf.t_row), _ptr(sf.t_val), _ptr(sf.seg_feat),
                      _ptr(sf.seg_e0), _ptr(sf.seg_slot), sf.seg_feat.numel(), sf.max_segments, n, K, _ptr(G), ldg,
                      _ptr(D), ldd, _ptr(Dmask), lddm, int(f_out), _ptr(dW), max(K, 1), _ptr(db), 0 if defer else 1,
                      _ptr(ws), ws.numel(), _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("spx_wgrad", n, K, f_out, sf.nnz), launch)
        else:
            launch()
    if defer:
        if dW is not None:
            _ops.current_step().add_partials(dW, (ws, ws.data_ptr(), lay[0], lay[1], f_out * K, f_out * K))
        if db is not None:
            _ops.current_step().add_partials(db, (ws, ws.data_ptr() + 4 * lay[2], lay[3], 32, f_out, f_out))
    return dW, db


class GCNSparseInputFunction(torch.autograd.Function):
    """GCNTransformFirstFunction on compressed input features: P = X W^T and dW = G^T X come from the non-zeros of X
    (gae_spx_fwd / gae_spx_wgrad); the sparse halves (gae_spmm_csr_epilogue) are the same launches"""

    @staticmethod
    def forward(ctx, W, b, sf, graph, use_norm, act):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        P = _ops.spx_fwd_raw(sf, W)
        Y = _ops.spmm_epilogue_raw(indptr, indices, P, n, graph.spmm_plan(False), b, act, None, norm, norm)
        ctx.act, ctx.has_bias, ctx.sf, ctx.f_out = act, b is not None, sf, W.shape[0]
        ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True))
        ctx.save_for_backward(Y if act == ACT_RELU else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        (Y,) = ctx.saved_tensors
        (t_indptr, t_indices), n, norm, plan_t = ctx.bwd
        need_dW = ctx.needs_input_grad[0]
        need_db = ctx.has_bias and ctx.needs_input_grad[1]
        dW = db = None
        dYc, _ = _rowmajor(_f32(dY, "dY"), "dY")
        if need_dW or need_db:
            G = _ops.spmm_epilogue_raw(t_indptr, t_indices, dYc, n, plan_t, None, ACT_IDENTITY, Y, norm, norm) \
                if need_dW else dYc
            dW, db = _ops.spx_wgrad_raw(ctx.sf, G, dYc if need_db else None, Y, ctx.f_out, need_dW=need_dW, need_db=need_db)
        return dW, db, None, None, None, None


def sparse_input_usable(graph, n, K, f_out):
    """can GCNSparseInputFunction run a K -> f_out layer on ``graph`` with compressed [n, K] features?  (the check
    sparse.SparseFeatures.maybe_from_dense makes BEFORE compressing)"""
    if graph.number_of_nodes() != n or f_out > 32 or graph.number_of_edges() == 0:
        return False
    if n * ((f_out + 3) // 4 * 4) * 4 + (1 << 16) >= (1 << 32):
        return False
    return _table_only(graph.spmm_plan(False)) and _table_only(graph.spmm_plan(True))


def gcn_layer_sparse_input(graph, sf, W, b, act, use_norm=False):
    """the layer on sparse.SparseFeatures input, or None when graph / widths do not allow it (the caller then
    densifies, once: SparseFeatures.to_dense(cache=True))"""
    if W.shape[1] != sf.shape[1] or not _ops.sparse_input_usable(graph, sf.shape[0], sf.shape[1], W.shape[0]):
        return None
    return _ops.GCNSparseInputFunction.apply(W, b, sf, graph, use_norm, act)
