"""Dense node-apply operators: Linear forward / backward (K3-K5), the one-pass dense halves of the two-layer encoder on
very tall operands, and their autograd Function.

Part of the package gae_dgl_amd.ops (one module until round 6).  Functions look each other up in the PACKAGE
namespace (`_ops.<name>`) when they run: setting a flag or replacing a function on `gae_dgl_amd.ops` reaches every caller."""
import ctypes

import torch

import gae_dgl_amd.ops as _ops
from .. import _lib
from .._lib import ACT_IDENTITY, ACT_RELU, GaeHipError
from ._base import _f32, _gpu, _on_device, _ptr, _rowmajor, _stream, _workspace

__all__ = [
    'linear2_usable', '_dead_mask', 'linear2_fwd_raw', 'gcn2_bwd_dense_raw', 'linear_fwd_raw', 'linear_bwd_raw',
    'LinearFunction', 'linear',
]


def linear2_usable(A, f_mid, f_out):
    """can gae_linear2_fwd / gae_gcn2_bwd_dense take this operand?  fp32 rows of whole 16-byte vectors, widths <= 32"""
    return (isinstance(A, torch.Tensor) and A.is_cuda and A.dtype == torch.float32 and A.dim() == 2 and A.shape[0] > 0
            and 1 <= A.shape[1] <= 32 and 1 <= f_mid <= 32 and 1 <= f_out <= 32)


def _dead_mask(t, n, what):
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 1 and t.numel() == n
            and t.is_contiguous()):
        raise GaeHipError(f"{what}: a dead-row mask is a contiguous uint8 device tensor with one entry per row")
    return t


def linear2_fwd_raw(A, W1, b1, act1, W2, want_y1=True, a_dead=None, rows=None, fill=None):
    """(Y1, T): Y1 = act1(A W1^T + b1), T = Y1 W2^T in ONE pass over A (gae_linear2_fwd); Y1 None when not wanted.
    Both outputs have rows of whole 16-byte vectors.  ``a_dead`` (uint8 [n]): rows of A that are zero and were never
    written (spmm_raw(skip_dead=True)) -- not read.  ``rows`` (int32, ascending; with ``a_dead`` marking all the others):
    list mode -- the two products run on the listed rows only, the other rows of T get their common value
    act1(b1) W2^T (gae_linear2_fill_dead); Y1 is not available then.  ``fill`` (uint8 [n], default ``a_dead``): the
    rows that receive that value -- a caller who knows that some dead rows of T are never read (T as the gather operand
    of the next aggregation: nodes without out-edges) leaves them unwritten."""
    A, lda = _rowmajor(_f32(_gpu(A, "A"), "linear2: A"), "A")
    if lda % 4 or A.data_ptr() % 16:
        A = _ops.pad_rows(A); lda = A.stride(0)
    W1 = _f32(_gpu(W1, "W1"), "linear2: W1"); W2 = _f32(_gpu(W2, "W2"), "linear2: W2")
    W1 = W1 if W1.stride(1) == 1 else W1.contiguous()
    W2 = W2 if W2.stride(1) == 1 else W2.contiguous()
    _f32(b1, "linear2: b1")
    n, f_in = A.shape
    f_mid, f_out = W1.shape[0], W2.shape[0]
    if W1.shape[1] != f_in or W2.shape[1] != f_mid:
        raise GaeHipError("linear2: weight shapes do not chain")
    dev = A.device
    ld1, ld2 = (f_mid + 3) // 4 * 4, (f_out + 3) // 4 * 4
    Y1 = torch.empty(n, ld1, dtype=torch.float32, device=dev)[:, :f_mid] if want_y1 else None
    T = torch.empty(n, ld2, dtype=torch.float32, device=dev)[:, :f_out]
    a_dead = _dead_mask(a_dead, n, "linear2")
    if rows is not None:
        if a_dead is None or want_y1 or rows.dtype != torch.int32 or not rows.is_cuda or not rows.is_contiguous():
            raise GaeHipError("linear2: list mode takes an int32 device list, the dead-row mask of all other rows, and no Y1")
    with _on_device(dev):
        def launch():
            if rows is not None:
                _lib.call("gae_linear2_fwd", _ptr(A), lda, n, f_in, _ptr(W1), W1.stride(0), _ptr(b1), f_mid, int(act1),
                          _ptr(W2), W2.stride(0), f_out, None, ld1, _ptr(T), ld2, None, _ptr(rows), int(rows.numel()),
                          _stream())
                _lib.call("gae_linear2_fill_dead", _ptr(b1), f_mid, int(act1), _ptr(W2), W2.stride(0), f_out,
                          _ptr(a_dead if fill is None else _dead_mask(fill, n, "linear2")), n, _ptr(T), ld2, _stream())
                return
            _lib.call("gae_linear2_fwd", _ptr(A), lda, n, f_in, _ptr(W1), W1.stride(0), _ptr(b1), f_mid, int(act1),
                      _ptr(W2), W2.stride(0), f_out, _ptr(Y1), ld1, _ptr(T), ld2, _ptr(a_dead), None, 0, _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("linear2", n, f_in, f_mid, f_out), launch)
        else:
            launch()
    return Y1, T


def gcn2_bwd_dense_raw(G, dZ, Y1, act1, M1, W2, W1=None, b1=None, m1_dead=None, g_dead=None, rows=None,
                       g_dead_listed=None):
    """(dW1, db1, dW2, db2) of a two-layer encoder from G = A^T dZ in ONE pass over G, dZ, Y1, M1 (gae_gcn2_bwd_dense):
    dW2 = G^T Y1, db2 = colsum(dZ), dY1 = (G W2) (.) act1'(Y1), dW1 = dY1^T M1, db1 = colsum(dY1).  Inside
    ``deferred_grad_reductions()`` the four gradients stay per-block partial sums for optim.Adam.step().
    ``Y1`` None: the pass recomputes Y1 = act1(M1 W1^T + b1) itself (bit-identical to linear2_fwd_raw's), ``W1`` / ``b1``
    needed; then ``m1_dead`` / ``g_dead`` (uint8 [n]) mark rows of M1 / G that are zero and were never written: not read.
    ``rows`` (int32, ascending = the rows that have an M1 row; ``m1_dead`` marks exactly the others; ``g_dead_listed`` =
    g_dead at the listed rows): list mode -- the pass visits the listed rows, the share of the others (a rank-one term of
    the column sums of their G / dZ rows) is one more partial of the list."""
    G, ldg = _rowmajor(_f32(_gpu(G, "G"), "gcn2_bwd: G"), "G")
    dZ, lddz = _rowmajor(_f32(_gpu(dZ, "dZ"), "gcn2_bwd: dZ"), "dZ")
    if ldg % 4 or G.data_ptr() % 16:
        G = _ops.pad_rows(G); ldg = G.stride(0)
    if lddz % 4 or dZ.data_ptr() % 16:
        dZ = _ops.pad_rows(dZ); lddz = dZ.stride(0)
    M1, ldm1 = _rowmajor(_f32(M1, "gcn2_bwd: M1"), "M1")
    W2 = _f32(W2, "gcn2_bwd: W2")
    W2 = W2 if W2.stride(1) == 1 else W2.contiguous()
    n, f_out = G.shape
    f_mid, f_in = W2.shape[1], M1.shape[1]
    ldy1 = ldw1 = 0
    if Y1 is not None:
        Y1, ldy1 = _rowmajor(_f32(Y1, "gcn2_bwd: Y1"), "Y1")
        if Y1.shape != (n, f_mid):
            raise GaeHipError("gcn2_bwd: operand shapes do not match")
    else:
        if W1 is None:
            raise GaeHipError("gcn2_bwd: recomputing Y1 needs W1")
        W1 = _f32(_gpu(W1, "W1"), "gcn2_bwd: W1")
        W1 = W1 if W1.stride(1) == 1 else W1.contiguous()
        _f32(b1, "gcn2_bwd: b1")
        ldw1 = W1.stride(0)
        if W1.shape != (f_mid, f_in):
            raise GaeHipError("gcn2_bwd: W1 does not match the operands")
        if ldm1 % 4 or M1.data_ptr() % 16:
            M1 = _ops.pad_rows(M1); ldm1 = M1.stride(0)
    if dZ.shape != G.shape or W2.shape[0] != f_out or M1.shape[0] != n:
        raise GaeHipError("gcn2_bwd: operand shapes do not match")
    dev = G.device
    m1_dead, g_dead = _dead_mask(m1_dead, n, "gcn2_bwd"), _dead_mask(g_dead, n, "gcn2_bwd")
    if Y1 is not None and (m1_dead is not None or g_dead is not None or rows is not None):
        raise GaeHipError("gcn2_bwd: dead-row masks / row lists go with the recomputing form (Y1 = None)")
    if rows is not None:
        if m1_dead is None or rows.dtype != torch.int32 or not rows.is_cuda or not rows.is_contiguous():
            raise GaeHipError("gcn2_bwd: list mode takes an int32 device list and the dead-row mask of all other rows")
        g_dead_listed = _dead_mask(g_dead_listed, int(rows.numel()), "gcn2_bwd")
    dW1 = torch.empty(f_mid, f_in, dtype=torch.float32, device=dev)
    db1 = torch.empty(f_mid, dtype=torch.float32, device=dev)
    dW2 = torch.empty(f_out, f_mid, dtype=torch.float32, device=dev)
    db2 = torch.empty(f_out, dtype=torch.float32, device=dev)
    with _on_device(dev):
        nbytes = _lib.load().gae_gcn2_bwd_dense_workspace_bytes(n, f_in, f_mid, f_out)
        if nbytes < 0:
            _lib.check(int(nbytes), "gae_gcn2_bwd_dense_workspace_bytes")
        defer = _ops.current_step().defer_grads
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev) if defer else _workspace(nbytes, dev)
        lay = (ctypes.c_int64 * 5)()

        def launch():
            _lib.call("gae_gcn2_bwd_dense", _ptr(G), ldg, _ptr(dZ), lddz, _ptr(Y1), ldy1, int(act1), _ptr(M1), ldm1,
                      _ptr(W2), W2.stride(0), n, f_in, f_mid, f_out, _ptr(dW1), _ptr(db1), _ptr(dW2), _ptr(db2), _ptr(ws),
                      ws.numel(), lay if defer else None, _ptr(W1) if Y1 is None else None, ldw1,
                      _ptr(b1) if Y1 is None else None, _ptr(m1_dead), _ptr(g_dead), _ptr(rows),
                      0 if rows is None else int(rows.numel()), _ptr(g_dead_listed), _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("gcn2_bwd", n, f_in, f_mid, f_out), launch)
        else:
            launch()
    if defer:
        base = ws.data_ptr()
        for t, off, ne in ((dW1, 0, f_mid * f_in), (db1, lay[2], f_mid), (dW2, lay[3], f_out * f_mid), (db2, lay[4], f_out)):
            _ops.current_step().add_partials(t, (ws, base + 4 * off, lay[0], lay[1], ne, ne))
    return dW1, db1, dW2, db2


def linear_fwd_raw(M, W, b, act):
    M, ldm = _rowmajor(_f32(M, "linear: M"), "M")
    W = _f32(_gpu(W, "W"), "linear: W").contiguous()
    _f32(b, "linear: b")
    n, f_in = M.shape
    f_out = W.shape[0]
    Y = torch.empty(n, f_out, dtype=torch.float32, device=M.device)
    with _on_device(M.device):
        nbytes = _lib.load().gae_linear_fwd_workspace_bytes(n, f_in, f_out)
        ws = _workspace(nbytes, M.device) if nbytes > 0 else None
        _lib.call("gae_linear_fwd", _ptr(M), ldm, n, f_in, _ptr(W), _ptr(b), f_out, act, _ptr(Y), max(f_out, 1),
                  _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    return Y


def linear_bwd_raw(dY, Y, act, M, W, need_dW=True, need_db=True, need_dM=True, f_out=None):
    dY, lddy = _rowmajor(_f32(dY, "linear backward: dY"), "dY")
    M, ldm = _rowmajor(_f32(M, "linear backward: M"), "M")
    if W is not None:
        W = _f32(W, "linear backward: W").contiguous()
    elif need_dM or f_out is None:
        raise GaeHipError("linear backward: dM needs W (and f_out must be given without it)")
    _f32(Y, "linear backward: Y")
    n, f_in = M.shape
    f_out = W.shape[0] if W is not None else int(f_out)
    dev = M.device
    dW = torch.empty(f_out, f_in, dtype=torch.float32, device=dev) if need_dW else None
    db = torch.empty(f_out, dtype=torch.float32, device=dev) if need_db else None
    dM = torch.empty(n, f_in, dtype=torch.float32, device=dev) if need_dM else None
    ldy = 0
    if Y is not None:
        Y, ldy = _rowmajor(Y, "Y")
    with _on_device(dev):
        if _ops.current_step().defer_grads and n > 0 and f_in > 0 and (dW is not None or db is not None):
            # (dW, db) stay partial sums for the optimiser launch; dM, if wanted, comes from the ordinary entry point
            wsp = torch.empty(_lib.load().gae_linear_bwd_workspace_bytes(n, f_in, f_out), dtype=torch.uint8, device=dev)
            lay = (ctypes.c_int64 * 4)()
            _lib.call("gae_x_linear_bwd_partials", _ptr(dY), lddy, _ptr(Y), ldy, act, _ptr(M), ldm, n, f_in, f_out,
                      int(dW is not None), int(db is not None), _ptr(wsp), wsp.numel(), lay, _stream())
            if dW is not None:
                _ops.current_step().add_partials(dW, (wsp, wsp.data_ptr(), lay[0], lay[1], f_out * f_in, f_out * f_in))
            if db is not None:
                _ops.current_step().add_partials(db, (wsp, wsp.data_ptr() + 4 * lay[2], lay[0], lay[1], f_out, f_out))
            if dM is None:
                return dW, db, dM
            ws = _workspace(_lib.load().gae_linear_bwd_workspace_bytes(n, f_in, f_out), dev)
            _lib.call("gae_linear_bwd", _ptr(dY), lddy, _ptr(Y), ldy, act, _ptr(M), ldm, _ptr(W), n, f_in, f_out,
                      None, None, _ptr(dM), max(f_in, 1), _ptr(ws), ws.numel(), _stream())
            return dW, db, dM
        ws = _workspace(_lib.load().gae_linear_bwd_workspace_bytes(n, f_in, f_out), dev)
        _lib.call("gae_linear_bwd", _ptr(dY), lddy, _ptr(Y), ldy, act, _ptr(M), ldm, _ptr(W), n, f_in, f_out,
                  _ptr(dW), _ptr(db), _ptr(dM), max(f_in, 1), _ptr(ws), ws.numel(), _stream())
    return dW, db, dM


class LinearFunction(torch.autograd.Function):
    """NodeApplyModule: act(M W^T + b)  (gae.py:13-16)."""

    @staticmethod
    def forward(ctx, M, W, b, act):
        Y = _ops.linear_fwd_raw(M, W, b, act)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.save_for_backward(M, W, Y if act == ACT_RELU else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        M, W, Y = ctx.saved_tensors
        need_dM, need_dW = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        dW, db, dM = _ops.linear_bwd_raw(dY, Y, ctx.act, M, W, need_dW, need_db, need_dM)
        return dM, dW, db, None


def linear(M, W, b, act=ACT_IDENTITY):
    return _ops.LinearFunction.apply(M, W, b, act)
