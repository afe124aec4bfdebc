"""Fused narrow GCN layers: aggregate + Linear + activation in one launch, its backward with the weight gradient as side
work, the loss's prepare step in a producer's epilogue, two heads on one aggregate.

Part of the package gae_dgl_amd.ops (one module until round 6).  Functions look each other up in the PACKAGE
namespace (`_ops.<name>`) when they run: setting a flag or replacing a function on `gae_dgl_amd.ops` reaches every caller."""
import ctypes
import os

import torch

import gae_dgl_amd.ops as _ops
from .. import _lib
from .._lib import ACT_IDENTITY, ACT_RELU, GaeHipError
from ._base import _f32, _gpu, _on_device, _ptr, _rowmajor, _stream, _workspace
from .aggregate import _scattered

__all__ = [
    'gcn_layer_fused_usable', 'gcn_layer_fused_raw', 'FUSE_LOSS_PREPARE', 'STATS', 'loss_prepare_request',
    'gcn_layer_fused_prep_raw', 'gcn_layer_fused_wgrad_raw', 'FUSED_LAYER_WGRAD', 'GCNLayerFusedFunction',
    '_split_pending', 'GCNTwoHeadFunction', 'gcn_two_heads', 'gcn_layer',
]


def gcn_layer_fused_usable(H, n_out, plan):
    """can gae_gcn_layer_fused run this layer?  fp32 rows of <= 64 features made of whole 16-byte vectors, <= 32
    outputs, a plan that carries a packed neighbour table and neither heavy nor XCD-pinned rows (their table rows
    are skip markers: the fused kernel would leave them unwritten)"""
    return (plan is not None and plan.ell is not None and plan.n_heavy == 0 and plan.homed is None
            and H.dtype == torch.float32
            and H.dim() == 2 and 1 <= H.shape[1] <= _ops.FUSED_LAYER_MAX_IN and 1 <= n_out <= _ops.FUSED_LAYER_MAX_OUT
            and H.shape[0] > 0 and H.stride(1) == 1 and H.stride(0) % 4 == 0 and H.data_ptr() % 16 == 0
            and H.shape[0] * H.stride(0) * 4 + (1 << 16) < (1 << 32))


def gcn_layer_fused_raw(indptr, indices, H, n_rows, plan, W, bias, act, row_scale=None, col_scale=None,
                        w_transposed=False, want_m=True):
    """(M or None, Y): the aggregation of spmm_raw and Y = act(M W^T + b) in one launch (gae_gcn_layer_fused).
    ``w_transposed``: use W^T, i.e. Y = M W for W [F, J] stored as nn.Linear keeps it ([out = F][in = J]) -- the
    backward form dH = (A^T dY) W."""
    H, ldh = _rowmajor(_f32(_gpu(H, "H"), "gcn_layer_fused: H"), "H")
    W = _f32(_gpu(W, "W"), "gcn_layer_fused: W")
    if W.stride(1) != 1:
        W = W.contiguous()
    n_cols, F = H.shape
    if w_transposed:
        J, so, sk = W.shape[1], 1, W.stride(0)
        if W.shape[0] != F:
            raise GaeHipError("gcn_layer_fused: weight shape does not match the features")
    else:
        J, so, sk = W.shape[0], W.stride(0), 1
        if W.shape[1] != F:
            raise GaeHipError("gcn_layer_fused: weight shape does not match the features")
    _f32(bias, "gcn_layer_fused: bias")
    M = torch.empty(n_rows, _ops.padded_ld(F, torch.float32), dtype=torch.float32, device=H.device)[:, :F] if want_m else None
    Y = torch.empty(n_rows, J, dtype=torch.float32, device=H.device)
    with _on_device(H.device):
        def launch():
            _lib.call("gae_gcn_layer_fused", _ptr(indptr), _ptr(indices), n_rows, n_cols, _ptr(H), ldh, _ptr(M),
                      M.stride(0) if M is not None else 0, F, _ptr(row_scale), _ptr(col_scale), ctypes.byref(plan.c),
                      _ptr(W), so, sk, _ptr(bias), J, int(act), _ptr(Y), max(J, 1), _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch)
        else:
            launch()
    return M, Y


# ---- the loss's prepare step in the epilogue of the layer that produces Z --------------------------------------
# ``with loss_prepare_request(graph, d, mask, dropout) as req:`` around the LAST encoder layer: if that layer runs as
# the fused launch (GCNLayerFusedFunction, identity activation, <= 16 outputs), the launch also writes Zt / hi / lo /
# column sums (and draws the dropout mask) into a loss workspace, and ``req.token`` describes it for
# ``decoder_bce(..., prepared=req.token)`` -- the loss then starts at its dense kernel (one kernel node fewer per step).
FUSE_LOSS_PREPARE = os.environ.get("GAE_FUSE_LOSS_PREPARE", "1") != "0"


STATS = {"prepared_losses": 0,      # losses that started at the dense kernel (tests read this)
         "xw_fwd": 0, "xw_wgrad": 0}  # launches of the one-pass layer-1 kernels (transform-first order)


class loss_prepare_request:
    def __init__(self, graph, d, mask, dropout):
        """``mask``: a given [n, d] multiplier (or None); ``dropout`` = (p, seed, offset, draw_counter) to draw one"""
        self.graph, self.d, self.mask, self.dropout, self.token = graph, int(d), mask, dropout, None

    def __enter__(self):
        self.step = _ops.current_step()             # the request belongs to the step it was made in
        self.prev = self.step.prep_req
        self.step.prep_req = self if _ops.FUSE_LOSS_PREPARE and self.d <= 16 else None
        return self

    def __exit__(self, *exc):
        self.step.prep_req = self.prev


def gcn_layer_fused_prep_raw(indptr, indices, H, n, plan, W, bias, row_scale, req, want_m=True):
    """(M or None, Z, token): gae_x_gcn_layer_fused_prep -- the fused layer (identity activation) with the prepare step
    of the loss in its epilogue; ``token`` goes to decoder_bce_raw(prepared=...)"""
    H, ldh = _rowmajor(_f32(_gpu(H, "H"), "gcn_layer_fused_prep: H"), "H")
    W = _f32(_gpu(W, "W"), "gcn_layer_fused_prep: W")
    if W.stride(1) != 1:
        W = W.contiguous()
    _f32(bias, "gcn_layer_fused_prep: bias")
    F, J = H.shape[1], W.shape[0]
    dev = H.device
    M = torch.empty(n, _ops.padded_ld(F, torch.float32), dtype=torch.float32, device=dev)[:, :F] if want_m else None
    Z = torch.empty(n, J, dtype=torch.float32, device=dev)
    p_drop, seed, offset, draws = req.dropout if req.dropout is not None else (0.0, 0, 0, None)
    mask = req.mask
    if p_drop:
        mask = torch.empty(n, J, dtype=torch.float32, device=dev)
    elif mask is not None:
        mask = _f32(_gpu(mask, "mask"), "gcn_layer_fused_prep: mask").contiguous()
    counts = getattr(req.graph, "batch_counts", None)
    with _on_device(dev):
        nbytes = _lib.load().gae_decoder_bce_workspace_bytes(n, n, J)
        if nbytes < 0:
            _lib.check(int(nbytes), "gae_decoder_bce_workspace_bytes")
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)       # lives until the loss has run: not the scratch cache
        lay = _lib.BcePrep()
        _lib.call("gae_x_decoder_bce_prep_layout", n, J, _ptr(ws), ws.numel(), ctypes.byref(lay))
        blocks = ctypes.c_int64(0)

        def launch():
            _lib.call("gae_x_gcn_layer_fused_prep", _ptr(indptr), _ptr(indices), n, _ptr(H), ldh, _ptr(M),
                      M.stride(0) if M is not None else 0, F, _ptr(row_scale), _ptr(row_scale), ctypes.byref(plan.c),
                      _ptr(W), W.stride(0), 1, _ptr(bias), J, _ptr(Z), J, ctypes.byref(lay), _ptr(mask), J, float(p_drop),
                      int(seed) & (2 ** 64 - 1), int(offset), _ptr(draws), _ptr(counts), ctypes.byref(blocks), _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("spmm", n, n, F, str(H.dtype)), launch)
        else:
            launch()
    token = {"ws": ws, "blocks": int(blocks.value), "mask": mask, "z_ptr": Z.data_ptr(), "n": n, "d": J,
             "dropout": req.dropout, "counts": counts}
    return M, Z, token


def gcn_layer_fused_wgrad_raw(t_indptr, t_indices, dY, n, plan_t, W, M, norm, want_dW=True, want_db=True):
    """(dH, dW, db): the identity-activation backward of the fused layer in one launch (gae_x_gcn_layer_fused_wgrad).
    Inside ``deferred_grad_reductions()`` dW / db are left as per-block partial sums for optim.Adam.step()."""
    dY, lddy = _rowmajor(_f32(_gpu(dY, "dY"), "gcn_layer_fused_wgrad: dY"), "dY")
    W = W if W.stride(1) == 1 else W.contiguous()
    f_out, f_in = W.shape
    if dY.shape[1] != f_out or M.shape[1] != f_in or M.stride(1) != 1:
        raise GaeHipError("gcn_layer_fused_wgrad: operand shapes do not match the weight")
    dev = dY.device
    dH = torch.empty(n, f_in, dtype=torch.float32, device=dev)
    dW = torch.empty(f_out, f_in, dtype=torch.float32, device=dev) if want_dW else None
    db = torch.empty(f_out, dtype=torch.float32, device=dev) if want_db else None
    defer = _ops.current_step().defer_grads and (want_dW or want_db)
    with _on_device(dev):
        nbytes = _lib.load().gae_x_gcn_layer_fused_wgrad_workspace_bytes(n, f_out, f_in)
        if nbytes < 0:
            _lib.check(int(nbytes), "gae_x_gcn_layer_fused_wgrad_workspace_bytes")
        # deferred partials outlive the call: they must not sit in the per-stream scratch cache
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev) if defer else _workspace(nbytes, dev)
        lay = (ctypes.c_int64 * 3)()

        def launch():
            _lib.call("gae_x_gcn_layer_fused_wgrad", _ptr(t_indptr), _ptr(t_indices), n, _ptr(dY), lddy, f_out, _ptr(norm),
                      _ptr(norm), ctypes.byref(plan_t.c), _ptr(W), W.stride(0), f_in, _ptr(dH), f_in, _ptr(M),
                      M.stride(0), None if defer else _ptr(dW), None if defer else _ptr(db), _ptr(ws), ws.numel(), lay,
                      _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("spmm", n, n, f_out, str(dY.dtype)), launch)
        else:
            launch()
    if defer:
        if dW is not None:
            _ops.current_step().add_partials(dW, (ws, ws.data_ptr(), lay[0], lay[1], f_out * f_in, f_out * f_in))
        if db is not None:
            _ops.current_step().add_partials(db, (ws, ws.data_ptr() + 4 * lay[2], lay[0], lay[1], f_out, f_out))
    return dH, dW, db


FUSED_LAYER_WGRAD = os.environ.get("GAE_FUSED_LAYER_WGRAD", "1") != "0"      # False: the fused layer's backward keeps its separate weight-gradient launch (experiments)


class GCNLayerFusedFunction(torch.autograd.Function):
    """GCN.forward (gae.py:26-31) as one launch: Y = act((A H) W^T + b).  Backward: dW = dYm^T M and db from the
    stored aggregate (gae_linear_bwd), and dH = A^T (dYm W) -- for an identity activation as ONE launch of the
    same kernel on the CSR of A^T, dH = (A^T dY) W (the aggregation then runs at the output width)."""

    @staticmethod
    def forward(ctx, H, W, b, graph, use_norm, act):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        need_w = ctx.needs_input_grad[1] or (b is not None and ctx.needs_input_grad[2])
        req = _ops.current_step().prep_req
        if (req is not None and req.token is None and req.graph is graph and act == ACT_IDENTITY
                and W.shape[0] == req.d and H.shape[0] == n and n > 0):
            # the last encoder layer of a training step: the loss's prepare step rides in this launch's epilogue
            M, Y, req.token = _ops.gcn_layer_fused_prep_raw(indptr, indices, H, n, graph.spmm_plan(False), W, b, norm, req,
                                                       want_m=need_w)
        else:
            M, Y = _ops.gcn_layer_fused_raw(indptr, indices, H, n, graph.spmm_plan(False), W, b, act, norm, norm,
                                       want_m=need_w)
        ctx.act, ctx.has_bias = act, b is not None
        if ctx.needs_input_grad[0]:
            ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True), graph.block_diag, _scattered(graph, H))
        ctx.save_for_backward(M, W, Y if act == ACT_RELU else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        M, W, Y = ctx.saved_tensors
        need_dH, need_dW = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        dW = db = dH = None
        fused_bwd = need_dH and ctx.act == ACT_IDENTITY and _ops.gcn_layer_fused_usable(dY.contiguous(), W.shape[1],
                                                                                 ctx.bwd[3])
        if fused_bwd and _ops.FUSED_LAYER_WGRAD and need_dW and M is not None and W.shape[0] <= 32 and W.shape[1] <= 32:
            # dH, dW and db from ONE launch: the blocks of the backward gather also add up dY^T M over their own rows
            (t_indptr, t_indices), n, norm, plan_t, _, _ = ctx.bwd
            dH, dW, db = _ops.gcn_layer_fused_wgrad_raw(t_indptr, t_indices, dY.contiguous(), n, plan_t, W, M, norm,
                                                   True, need_db)
            return dH, dW, db, None, None, None
        if need_dW or need_db or (need_dH and not fused_bwd):
            dW, db, dM = _ops.linear_bwd_raw(dY, Y, ctx.act, M if M is not None else dY.new_zeros(dY.shape[0], W.shape[1]),
                                        W, need_dW, need_db, need_dH and not fused_bwd)
        if need_dH:
            (t_indptr, t_indices), n, norm, plan_t, blockdiag, sc = ctx.bwd
            if fused_bwd:
                _, dH = _ops.gcn_layer_fused_raw(t_indptr, t_indices, dY.contiguous(), n, plan_t, W, None, ACT_IDENTITY,
                                            norm, norm, w_transposed=True, want_m=False)
            else:
                dH = _ops.spmm_raw(t_indptr, t_indices, dM, n, norm, norm, plan=plan_t, blockdiag=blockdiag, scattered=sc)
        return dH, dW, db, None, None, None


def _split_pending(t, rows):
    """a deferred gradient ``t`` [R, ...] handed out as the two row blocks t[:rows], t[rows:]: register the partial
    lists of the halves (same workspace, second one offset)"""
    step = _ops.current_step()
    ent = step.take_partials(t)
    if ent is None:
        return
    ws, ptr, n_part, stride, _, _ = ent
    per_row = t[0].numel() if t.dim() > 1 else 1
    n0, n1 = rows * per_row, (t.shape[0] - rows) * per_row
    _ops.current_step().add_partials(t[:rows], (ws, ptr, n_part, stride, max(n0, 1), max(n0, 1)))
    _ops.current_step().add_partials(t[rows:], (ws, ptr + 4 * n0, n_part, stride, max(n1, 1), max(n1, 1)))


class GCNTwoHeadFunction(torch.autograd.Function):
    """two identity-activation GCN layers on the same input (VGAE's mu and log sigma heads) as ONE fused launch
    (gae_x_gcn_layer_fused2): ML = [(A H) W1^T + b1 | (A H) W2^T + b2].  Backward: one dW launch for both heads
    (dML^T M, split by rows), one fused launch dH = (A^T dML) [W1; W2] -- instead of two of each plus an add."""

    @staticmethod
    def forward(ctx, H, W1, b1, W2, b2, graph, use_norm):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        plan = graph.spmm_plan(False)
        Hc, ldh = _rowmajor(_f32(_gpu(H, "H"), "two heads: H"), "H")
        W1 = W1 if W1.stride(1) == 1 else W1.contiguous()
        W2 = W2 if W2.stride(1) == 1 and W2.stride(0) == W1.stride(0) else W2.contiguous()
        F, d1, d2 = Hc.shape[1], W1.shape[0], W2.shape[0]
        need_w = any(ctx.needs_input_grad[1:5])
        M = torch.empty(n, _ops.padded_ld(F, torch.float32), dtype=torch.float32, device=Hc.device)[:, :F] if need_w else None
        Y = torch.empty(n, d1 + d2, dtype=torch.float32, device=Hc.device)
        with _on_device(Hc.device):
            _lib.call("gae_x_gcn_layer_fused2", _ptr(indptr), _ptr(indices), n, Hc.shape[0], _ptr(Hc), ldh, _ptr(M),
                      M.stride(0) if M is not None else 0, F, _ptr(norm), _ptr(norm), ctypes.byref(plan.c), _ptr(W1),
                      _ptr(W2), d1, 0, W1.stride(0), 1, _ptr(b1), _ptr(b2), d1 + d2, ACT_IDENTITY, _ptr(Y), d1 + d2,
                      _stream())
        ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True), d1, d2, b1 is not None)
        ctx.save_for_backward(M, W1, W2)
        return Y

    @staticmethod
    def backward(ctx, dY):
        M, W1, W2 = ctx.saved_tensors
        (t_indptr, t_indices), n, norm, plan_t, d1, d2, has_bias = ctx.bwd
        need_dH = ctx.needs_input_grad[0]
        need_dW = ctx.needs_input_grad[1] or ctx.needs_input_grad[3]
        need_db = has_bias and (ctx.needs_input_grad[2] or ctx.needs_input_grad[4])
        dW1 = db1 = dW2 = db2 = dH = None
        dYc = dY.contiguous()
        if (_ops.FUSED_LAYER_WGRAD and need_dH and need_dW and M is not None and d1 + d2 <= 32 and W1.shape[1] <= 32
                and M.stride(1) == 1):
            # dH, dW and db of both heads from ONE launch (gae_x_gcn_layer_fused2_wgrad: side work of the gather's blocks)
            f_out, f_in = d1 + d2, W1.shape[1]
            dev = dYc.device
            dH = torch.empty(n, f_in, dtype=torch.float32, device=dev)
            dW = torch.empty(f_out, f_in, dtype=torch.float32, device=dev)
            db = torch.empty(f_out, dtype=torch.float32, device=dev) if need_db else None
            defer = _ops.current_step().defer_grads
            with _on_device(dev):
                nbytes = _lib.load().gae_x_gcn_layer_fused_wgrad_workspace_bytes(n, f_out, f_in)
                if nbytes < 0:
                    _lib.check(int(nbytes), "gae_x_gcn_layer_fused_wgrad_workspace_bytes")
                ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev) if defer else _workspace(nbytes, dev)
                lay = (ctypes.c_int64 * 3)()
                _lib.call("gae_x_gcn_layer_fused2_wgrad", _ptr(t_indptr), _ptr(t_indices), n, _ptr(dYc), f_out, f_out,
                          _ptr(norm), _ptr(norm), ctypes.byref(plan_t.c), _ptr(W1), _ptr(W2), d1, W1.stride(0), f_in,
                          _ptr(dH), f_in, _ptr(M), M.stride(0), None if defer else _ptr(dW), None if defer else _ptr(db),
                          _ptr(ws), ws.numel(), lay, _stream())
            if defer:
                _ops.current_step().add_partials(dW, (ws, ws.data_ptr(), lay[0], lay[1], f_out * f_in, f_out * f_in))
                if db is not None:
                    _ops.current_step().add_partials(db, (ws, ws.data_ptr() + 4 * lay[2], lay[0], lay[1], f_out, f_out))
            _split_pending(dW, d1)
            dW1, dW2 = dW[:d1], dW[d1:]
            if db is not None:
                _split_pending(db, d1)
                db1, db2 = db[:d1], db[d1:]
            return dH, dW1, db1, dW2, db2, None, None
        if need_dW or need_db:
            dW, db, _ = _ops.linear_bwd_raw(dYc, None, ACT_IDENTITY, M, None, need_dW, need_db, False, f_out=d1 + d2)
            if dW is not None:
                _split_pending(dW, d1)
                dW1, dW2 = dW[:d1], dW[d1:]
            if db is not None:
                _split_pending(db, d1)
                db1, db2 = db[:d1], db[d1:]
        if need_dH:
            F = W1.shape[1]
            dH = torch.empty(n, F, dtype=torch.float32, device=dYc.device)
            with _on_device(dYc.device):
                # dH = (A^T dY) [W1; W2]: the stacked matrix addressed transposed (element (o, k) at row k, column o)
                _lib.call("gae_x_gcn_layer_fused2", _ptr(t_indptr), _ptr(t_indices), n, n, _ptr(dYc), d1 + d2, None, 0,
                          d1 + d2, _ptr(norm), _ptr(norm), ctypes.byref(plan_t.c), _ptr(W1), _ptr(W2), d1, 1, 1,
                          W1.stride(0), None, None, F, ACT_IDENTITY, _ptr(dH), F, _stream())
        return dH, dW1, db1, dW2, db2, None, None


def gcn_two_heads(graph, H, lin1, lin2, use_norm=False):
    """[head1 | head2] of two identity-activation GCN layers (nn.Linear modules lin1, lin2) on ``H`` in one launch, or
    None when the shapes / the graph do not allow the fused layer"""
    if not isinstance(H, torch.Tensor) or not H.is_cuda or H.dtype != torch.float32 or graph.number_of_edges() == 0:
        return None
    Hc, _ = _rowmajor(H, "H")
    if Hc.stride(0) % 4 or Hc.data_ptr() % 16:
        Hc = _ops.pad_rows(Hc)
    d1, d2 = lin1.weight.shape[0], lin2.weight.shape[0]
    plan, plan_t = graph.spmm_plan(False), graph.spmm_plan(True)
    if lin1.weight.shape[1] != lin2.weight.shape[1] or (lin1.bias is None) != (lin2.bias is None):
        return None
    if not _ops.gcn_layer_fused_usable(Hc, d1 + d2, plan) or not _ops._table_only(plan_t) or d1 + d2 > _ops.FUSED_LAYER_MAX_IN:
        return None
    # the backward is gae_x_gcn_layer_fused2(_wgrad) on the CSR of A^T with dML [n, d1 + d2] as the gathered operand and
    # the heads' INPUT width as its output: rows of whole 16-byte vectors and <= 32 outputs, or the two separate
    # layers (which have their own fallbacks) must run instead
    if lin1.weight.shape[1] > _ops.FUSED_LAYER_MAX_OUT or (d1 + d2) % 4 != 0:
        return None
    bd = graph.block_diag
    if bd is not None and bd.usable(Hc, Hc.shape[1], Hc.stride(0), Hc.stride(0)):
        return None
    return _ops.GCNTwoHeadFunction.apply(Hc, lin1.weight, lin1.bias, lin2.weight, lin2.bias, graph, use_norm)


def gcn_layer(graph, H, W, b, act, use_norm=False):
    """one GCN layer on ``graph``: fused launch when the shapes allow it (gcn_layer_fused_usable), None otherwise
    (the caller then runs update_all + apply_nodes as two launches)"""
    if not isinstance(H, torch.Tensor) or not H.is_cuda or H.dtype != torch.float32:
        return None
    if graph.number_of_edges() == 0:
        return None
    Hc, _ = _rowmajor(H, "H")
    if Hc.stride(0) % 4 or Hc.data_ptr() % 16:
        Hc = _ops.pad_rows(Hc)
    bd = graph.block_diag
    if bd is not None and bd.usable(Hc, Hc.shape[1], Hc.stride(0), Hc.stride(0)):
        return None                  # whole-set molecule launches: the LDS-staged block-diagonal kernel is faster
    plan = graph.spmm_plan(False)
    if not _ops.gcn_layer_fused_usable(Hc, W.shape[0], plan):
        return None
    return _ops.GCNLayerFusedFunction.apply(Hc, W, b, graph, use_norm, act)
