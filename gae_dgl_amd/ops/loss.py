"""Decoder and loss: dropout masks / noise, dense decoder, the fused never-materialised decoder + weighted BCE, MSE closed
form, the row-sharded loss, and backward() -- the step's entry into autograd.

Part of the package gae_dgl_amd.ops (one module until round 6).  Functions look each other up in the PACKAGE
namespace (`_ops.<name>`) when they run: setting a flag or replacing a function on `gae_dgl_amd.ops` reaches every caller."""
import ctypes

import torch

import gae_dgl_amd.ops as _ops
from .. import _lib
from .._lib import ACT_IDENTITY, GaeHipError
from ._base import _f32, _gpu, _on_device, _ptr, _rowmajor, _stream, _workspace

__all__ = [
    'dropout_mask', 'normal_noise', 'decoder_dense_raw', 'decoder_dense_bwd_raw', 'decoder_bce_raw',
    'DecoderDenseFunction', 'DecoderBCEFunction', 'bce_logits_raw', 'DecoderDenseBCEFunction', 'BCELogitsFunction',
    'bce_with_logits', 'FUSED_MAX_D', '_UNIT', '_is_unit', 'backward', 'decoder_bce', 'gram_raw',
    'DecoderMSEFunction', 'decoder_mse', 'ShardedDecoderBCEFunction', 'sharded_decoder_bce', 'decoder_dense',
]


def dropout_mask(shape, p, seed, offset=0, device="cuda", draw_counter=None):
    """inverted-dropout multiplier; ``draw_counter`` (int64 device tensor [1]) selects the
    draw on the device so that captured HIP graphs advance the stream between replays"""
    mask = torch.empty(shape, dtype=torch.float32, device=device)
    with _on_device(mask.device):
        _lib.call("gae_dropout_mask", _ptr(mask), mask.numel(), float(p), int(seed) & (2 ** 64 - 1),
                  int(offset) & (2 ** 64 - 1), _ptr(draw_counter), _stream())
    return mask


def normal_noise(shape, seed, offset=0, device="cuda", draw_counter=None):
    """eps ~ N(0, 1) from the library's Philox + Box-Muller generator (VGAE reparameterisation)"""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    with _on_device(out.device):
        _lib.call("gae_normal_noise", _ptr(out), out.numel(), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
                  _ptr(draw_counter), _stream())
    return out


def decoder_dense_raw(Z, mask=None):
    Z, ldz = _rowmajor(_f32(Z, "decoder: Z"), "Z")
    _f32(mask, "decoder: mask")
    if mask is not None:
        mask = _gpu(mask, "mask").contiguous()
        if Z.stride(0) != mask.stride(0) and Z.shape[0] > 1:
            Z = Z.contiguous(); ldz = max(Z.shape[1], 1)
    n, d = Z.shape
    out = torch.empty(n, n, dtype=torch.float32, device=Z.device)
    with _on_device(Z.device):
        _lib.call("gae_decoder_dense", _ptr(Z), _ptr(mask), ldz, n, d, _ptr(out), max(n, 1), _stream())
    return out


def decoder_dense_bwd_raw(G, Z, mask=None):
    G, ldg = _rowmajor(_f32(G, "decoder backward: G"), "G")
    Z = _f32(_gpu(Z, "Z"), "decoder backward: Z").contiguous()
    _f32(mask, "decoder backward: mask")
    if mask is not None:
        mask = mask.contiguous()
    n, d = Z.shape
    dZ = torch.empty(n, d, dtype=torch.float32, device=Z.device)
    with _on_device(Z.device):
        ws = _workspace(_lib.load().gae_decoder_dense_bwd_workspace_bytes(n, d), Z.device)
        _lib.call("gae_decoder_dense_bwd", _ptr(G), ldg, _ptr(Z), _ptr(mask), max(d, 1), n, d, _ptr(dZ), max(d, 1),
                  _ptr(ws), ws.numel(), _stream())
    return dZ


def decoder_bce_raw(Z, mask, csr, csc, pos_weight, want_grad=True, row_begin=0, n_local=None, dropout=None,
                    counts=None, defer_ok=False, prepared=None):
    """fused decoder + weighted BCE (mean): returns (loss[1], dZ or None).
    ``counts`` (int64[2] on the device: true {nodes, edges}) = Z / csr are a fixed-capacity batch
    (gae_decoder_bce_padded): pos_weight and the mean come from the counts, ``pos_weight`` is ignored.
    ``row_begin/n_local`` select a row window (row-sharded form): Z/mask stay the
    full [n, d] arrays, csr/csc are the window's local row blocks.
    ``dropout`` = (p, seed, offset, draw_counter): the mask of this draw is generated inside the launch, written
    to ``mask`` (an [n, d] output buffer then) and the device draw counter is advanced by the library.
    ``defer_ok``: inside ``deferred_loss_finalize()`` the final reduction may be left to the optimiser launch -- the
    returned scalar is then NOT valid before ``optim.Adam.step()`` (or the end of the block) has run.
    ``prepared``: token of gcn_layer_fused_prep_raw -- the producer of Z already ran the prepare step (mask drawn,
    workspace filled): mask / dropout / counts are the token's."""
    Z = _f32(_gpu(Z, "Z"), "decoder_bce: Z").contiguous()
    if prepared is not None:
        if prepared["z_ptr"] != Z.data_ptr() or tuple(Z.shape) != (prepared["n"], prepared["d"]) or row_begin or \
                (n_local is not None and n_local != Z.shape[0]):
            raise GaeHipError("decoder_bce: the prepared workspace belongs to another embedding")
        mask, dropout, counts = prepared["mask"], prepared["dropout"], prepared["counts"]
    if mask is not None:
        mask = _f32(_gpu(mask, "mask"), "decoder_bce: mask").contiguous()
    p_drop, seed, offset, draws = dropout if dropout is not None else (0.0, 0, 0, None)
    if p_drop and (mask is None or mask.shape != Z.shape):
        raise GaeHipError("decoder_bce: in-kernel dropout needs an [n, d] mask output buffer")
    n, d = Z.shape
    n_local = n if n_local is None else int(n_local)
    dev = Z.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dZ = torch.empty(n_local, d, dtype=torch.float32, device=dev) if want_grad else None
    indptr, indices = csr
    t_indptr, t_indices = csc if csc is not None else (None, None)
    with _on_device(dev):
        if prepared is not None:
            ws = prepared["ws"]
        else:
            nbytes = _lib.load().gae_decoder_bce_workspace_bytes(n, n_local, d)
            if nbytes < 0:
                _lib.check(int(nbytes), "gae_decoder_bce_workspace_bytes")
            # a deferred final reduction reads the partial sums in the optimiser launch: they must not sit in the
            # per-stream scratch cache, which any launch in between (dM of gae_linear_bwd, a weight-gradient
            # reduction) may hand out again
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev) if (_ops.current_step().defer_loss and defer_ok and want_grad) \
                else _workspace(nbytes, dev)

        def launch():
            tail = None
            if _ops.current_step().tails:
                _ops.current_step().flush_loss_tails()                 # an earlier loss nobody took: its partial sums may live in the
                                                   # cached workspace this call is about to reuse
            if _ops.current_step().defer_loss and defer_ok and want_grad:
                tail = _lib.BceTail()
                _lib.call("gae_x_decoder_bce_defer_finalize", ctypes.byref(tail))
            try:
                launch_kernels()
            except Exception:
                if tail is not None:
                    _lib.call("gae_x_decoder_bce_defer_finalize", None)
                raise
            if tail is not None:
                _ops.current_step().tails.append((tail, (loss, ws, draws, counts)))

        def launch_kernels():
            if prepared is not None:
                _ops.STATS["prepared_losses"] += 1
                _lib.call("gae_x_decoder_bce_prepared", _ptr(mask), max(d, 1), n, d, _ptr(indptr), _ptr(indices),
                          _ptr(t_indptr), _ptr(t_indices), float(pos_weight), _ptr(counts), float(p_drop), _ptr(draws),
                          prepared["blocks"], _ptr(loss), _ptr(dZ), max(d, 1), _ptr(ws), ws.numel(), _stream())
                return
            if counts is not None:
                if row_begin or n_local != n:
                    raise GaeHipError("decoder_bce: a fixed-capacity batch has no row window")
                _lib.call("gae_decoder_bce_padded", _ptr(Z), _ptr(mask), max(d, 1), n, d, _ptr(indptr),
                          _ptr(indices), _ptr(t_indptr), _ptr(t_indices), _ptr(counts), float(p_drop),
                          int(seed) & (2 ** 64 - 1), int(offset), _ptr(draws), _ptr(loss), _ptr(dZ), max(d, 1),
                          _ptr(ws), ws.numel(), _stream())
                return
            _lib.call("gae_decoder_bce_rows", _ptr(Z), _ptr(mask), max(d, 1), n, d, int(row_begin), n_local,
                      _ptr(indptr), _ptr(indices), _ptr(t_indptr), _ptr(t_indices), float(pos_weight), float(p_drop),
                      int(seed) & (2 ** 64 - 1), int(offset), _ptr(draws), _ptr(loss), _ptr(dZ), max(d, 1), _ptr(ws),
                      ws.numel(), _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("decoder_bce", n, d, want_grad), launch)
        else:
            launch()
    return loss, dZ


class DecoderDenseFunction(torch.autograd.Function):
    """InnerProductDecoder: (Z m)(Z m)^T  (gae.py:70-71)."""

    @staticmethod
    def forward(ctx, Z, mask):
        ctx.save_for_backward(Z, mask)
        return _ops.decoder_dense_raw(Z, mask)

    @staticmethod
    def backward(ctx, G):
        Z, mask = ctx.saved_tensors
        return _ops.decoder_dense_bwd_raw(G, Z, mask), None


class DecoderBCEFunction(torch.autograd.Function):
    """train_inductive.py:44-48 fused: label from the graph's CSR, pos_weight,
    (Z m)(Z m)^T, BCE-with-logits mean.  The gradient w.r.t. Z is produced by
    the same launch sequence as the loss (flash-style) and scaled in backward."""

    @staticmethod
    def forward(ctx, Z, mask, graph, dropout=None, prepared=None):
        n = graph.number_of_nodes()
        nnz = graph.number_of_edges()
        counts = getattr(graph, "batch_counts", None)             # fixed-capacity batch: true sizes on the device
        pw = 0.0 if counts is not None else (float(n) * float(n) - float(nnz)) / float(nnz)  # train_inductive.py:46
        need = ctx.needs_input_grad[0]
        loss, dZ = _ops.decoder_bce_raw(Z, mask, graph.csr(), graph.csc() if need else None, pw, want_grad=need,
                                   dropout=dropout, counts=counts, defer_ok=True, prepared=prepared)
        ctx.save_for_backward(dZ)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dZ,) = ctx.saved_tensors
        if _is_unit(g):
            return dZ, None, None, None, None    # upstream gradient is the cached constant 1 (ops.backward)
        return dZ * g, None, None, None, None


def bce_logits_raw(logits, labels, pos_weight, want_grad=True):
    """F.binary_cross_entropy_with_logits(logits, labels, pos_weight) (mean) on materialised [n, m] fp32 matrices:
    returns (loss[1], dLoss/dLogits or None).  The gradient is written over ``logits``."""
    X, ldx = _rowmajor(_f32(logits, "bce_logits: logits"), "logits")
    Y, ldy = _rowmajor(_f32(labels, "bce_logits: labels"), "labels")
    if X.shape != Y.shape:
        raise GaeHipError("bce_logits: logits / labels shape mismatch")
    n, m = X.shape
    loss = torch.empty(1, dtype=torch.float32, device=X.device)
    with _on_device(X.device):
        ws = _workspace(_lib.load().gae_bce_logits_workspace_bytes(), X.device)
        _lib.call("gae_bce_logits", _ptr(X), ldx, _ptr(Y), ldy, n, m, float(pos_weight), _ptr(loss),
                  _ptr(X) if want_grad else None, ldx, _ptr(ws), ws.numel(), _stream())
    return loss, (X if want_grad else None)


class DecoderDenseBCEFunction(torch.autograd.Function):
    """train_inductive.py:44-48 in the reference's own shape -- dense label, N x N logits, weighted BCE (mean) -- as a
    chain of HIP kernels (gae_decoder_dense, gae_csr_to_dense, gae_bce_logits, gae_decoder_dense_bwd).  Used for
    embedding widths above FUSED_MAX_D, which the never-materialised kernel does not take; O(N^2) memory."""

    @staticmethod
    def forward(ctx, Z, mask, graph):
        n, nnz = graph.number_of_nodes(), graph.number_of_edges()
        pw = (float(n) * float(n) - float(nnz)) / float(nnz)      # train_inductive.py:46
        need = ctx.needs_input_grad[0]
        logits = _ops.decoder_dense_raw(Z, mask)
        loss, G = _ops.bce_logits_raw(logits, graph.dense_adjacency(), pw, want_grad=need)
        ctx.save_for_backward(G, Z, mask)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        G, Z, mask = ctx.saved_tensors
        dZ = _ops.decoder_dense_bwd_raw(G, Z, mask)
        return (dZ if _is_unit(g) else dZ * g), None, None


class BCELogitsFunction(torch.autograd.Function):
    """F.binary_cross_entropy_with_logits(logits, label, pos_weight=pw) (mean) on materialised matrices -- the
    reference's own loss call (train_inductive.py:48) -- by gae_bce_logits: loss and dLoss/dLogits in one pass."""

    @staticmethod
    def forward(ctx, logits, label, pos_weight):
        need = ctx.needs_input_grad[0]
        X, ldx = _rowmajor(_f32(_gpu(logits, "logits"), "bce_logits: logits"), "logits")
        Y, ldy = _rowmajor(_f32(_gpu(label, "label"), "bce_logits: labels"), "labels")
        if X.shape != Y.shape:
            raise GaeHipError("bce_with_logits: logits / labels shape mismatch")
        n, m = X.shape
        loss = torch.empty(1, dtype=torch.float32, device=X.device)
        G = torch.empty(n, m, dtype=torch.float32, device=X.device) if need else None
        with _on_device(X.device):
            ws = _workspace(_lib.load().gae_bce_logits_workspace_bytes(), X.device)
            _lib.call("gae_bce_logits", _ptr(X), ldx, _ptr(Y), ldy, n, m, float(pos_weight), _ptr(loss), _ptr(G),
                      max(m, 1), _ptr(ws), ws.numel(), _stream())
        ctx.save_for_backward(G)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (G,) = ctx.saved_tensors
        return (G if _is_unit(g) else G * g), None, None


def bce_with_logits(logits, label, pos_weight):
    """drop-in for ``BCELoss(logits, label, pos_weight=pw)`` of the reference's Trainer (train_inductive.py:48) on the
    HIP kernel; ``pos_weight`` may be a 0-dim / 1-element tensor like the reference's (read back once per call)"""
    return _ops.BCELogitsFunction.apply(logits, label, float(pos_weight))


FUSED_MAX_D = 64     # widest embedding of the fused decoder + BCE kernels (gae_decoder_bce)


_UNIT = {}


def _is_unit(g):
    one = _UNIT.get(g.device)
    return one is not None and g.dim() == 0 and g.data_ptr() == one.data_ptr()


def backward(loss, params=None):
    """``loss.backward()`` with the upstream gradient 1 handed over as a cached device constant: the fused loss
    recognises it and returns its stored gradient as is (saves the fill and the multiply launch of a plain
    ``loss.backward()``; 9 us of a 190 us Cora step).
    ``params``: write the gradients of exactly these tensors through ``torch.autograd.grad`` (p.grad is REPLACED, not
    accumulated).  No AccumulateGrad node takes part then -- those remember the stream of the iteration that
    created them, which breaks a HIP-graph capture that follows eager steps on another stream."""
    one = _UNIT.get(loss.device)
    if one is None:
        one = _UNIT[loss.device] = torch.ones((), dtype=loss.dtype, device=loss.device)
    if params is None:
        loss.backward(gradient=one)
        return
    params = [p for p in params if p.requires_grad]
    for p, g in zip(params, torch.autograd.grad(loss, params, grad_outputs=one, allow_unused=True)):
        p.grad = g


def decoder_bce(Z, mask, graph, dropout=None, prepared=None):
    """``dropout`` = (p, seed, offset, draw_counter): draw the mask inside the fused launch into ``mask``.
    ``prepared``: token of a producer launch that already ran the prepare step (loss_prepare_request).
    Embeddings wider than FUSED_MAX_D take the dense HIP chain (same value, O(N^2) memory)."""
    if prepared is not None:
        return _ops.DecoderBCEFunction.apply(Z, prepared["mask"], graph, prepared["dropout"], prepared)
    if Z.shape[1] > _ops.FUSED_MAX_D:
        if getattr(graph, "batch_counts", None) is not None:
            raise GaeHipError(f"fixed-capacity batches need an embedding width <= {_ops.FUSED_MAX_D}")
        if dropout is not None and dropout[0]:
            p_drop, seed, offset, draws = dropout
            mask.copy_(_ops.dropout_mask(tuple(Z.shape), p_drop, seed, offset, Z.device, draw_counter=draws))
            if draws is not None:
                draws += 1
        return _ops.DecoderDenseBCEFunction.apply(Z, mask, graph)
    return _ops.DecoderBCEFunction.apply(Z, mask, graph, dropout)


def gram_raw(M):
    """M^T M ([f, f], fp32-grade products) through gae_linear_bwd's weight-gradient kernel (dW = dY^T M with dY = M);
    never deferred to an optimiser launch (it is a value of the forward pass, not a parameter gradient)"""
    M, ldm = _rowmajor(_f32(_gpu(M, "M"), "gram: M"), "M")
    n, f = M.shape
    G = torch.empty(f, f, dtype=torch.float32, device=M.device)
    with _on_device(M.device):
        ws = _workspace(_lib.load().gae_linear_bwd_workspace_bytes(n, f, f), M.device)
        _lib.call("gae_linear_bwd", _ptr(M), ldm, None, 0, ACT_IDENTITY, _ptr(M), ldm, None, n, f, f,
                  _ptr(G), None, None, max(f, 1), _ptr(ws), ws.numel(), _stream())
    return G


class DecoderMSEFunction(torch.autograd.Function):
    """The criterion of the reference's hyper-parameter search, optuna_gae.py:16,21: ``nn.MSELoss()(model.forward(g),
    g.adjacency_matrix().to_dense())`` -- the mean over all N^2 ordered pairs of (s_ij - a_ij)^2, s = Zt Zt^T,
    Zt = Z (.) mask -- WITHOUT the N x N matrices:
        sum_ij (s_ij - a_ij)^2 = ||Zt^T Zt||_F^2 - 2 <Zt, A Zt> + sum_ij a_ij^2
        dL/dZt = (2 / N^2) (2 Zt (Zt^T Zt) - A Zt - A^T Zt)
    (oracle/gae_oracle.py: mse_closed_form): O(N d^2 + E d) work from launches the library already has --
    gae_spmm_csr on A and on A^T, gae_linear_bwd for the d x d Gram matrix (fp32-grade products), gae_linear_fwd for
    Zt G -- and two sums of N d products, taken in fp64."""

    @staticmethod
    def forward(ctx, Z, mask, graph):
        if getattr(graph, "batch_counts", None) is not None:
            raise GaeHipError("decoder_mse: fixed-capacity batches are not supported (the BCE loss's captured step only)")
        n = graph.number_of_nodes()
        Z = _f32(_gpu(Z, "Z"), "decoder_mse: Z")
        if Z.shape[0] != n:
            raise GaeHipError(f"decoder_mse: Z has {Z.shape[0]} rows, the graph {n} nodes")
        d = Z.shape[1]
        Zt = _ops.pad_rows(Z if mask is None else Z * mask)
        ip, ix = graph.csr()
        AZ = _ops.spmm_raw(ip, ix, Zt, n, plan=graph.spmm_plan(False))
        G = _ops.gram_raw(Zt)                                                      # Zt^T Zt
        inv = 1.0 / (float(n) * float(n))
        loss = ((G.double() ** 2).sum() - 2.0 * (Zt.double() * AZ.double()).sum() + graph.adjacency_sq_sum()) * inv
        if ctx.needs_input_grad[0]:
            tp, tx = graph.csc()
            AtZ = _ops.spmm_raw(tp, tx, Zt, n, plan=graph.spmm_plan(True))
            ZG = _ops.linear_fwd_raw(Zt, G, None, ACT_IDENTITY)                   # Zt G (G is symmetric)
            dZ = (2.0 * ZG - AZ - AtZ) * (2.0 * inv)
            if mask is not None:
                dZ = dZ * mask
            ctx.save_for_backward(dZ)
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (dZ,) = ctx.saved_tensors
        return (dZ if _is_unit(g) else dZ * g), None, None


def decoder_mse(Z, mask, graph):
    """mean((Zt Zt^T - A)^2) over all N^2 ordered pairs, Zt = Z (.) mask (optuna_gae.py:16,21), never materialised"""
    return _ops.DecoderMSEFunction.apply(Z, mask, graph)


class ShardedDecoderBCEFunction(torch.autograd.Function):
    """Row block of the fused loss on a row-sharded graph (parallel.ShardedGraph):
    Zt = Z (.) mask is all-gathered (N x d, small), each rank evaluates its rows
    against all columns, the partial means are summed with a scalar all-reduce."""

    @staticmethod
    def forward(ctx, z_local, mask_local, sg, n_edges_global):
        p = sg.part
        zt_local = z_local if mask_local is None else z_local * mask_local
        full = sg.allgather_rows(zt_local)
        n = p.n
        pw = (float(n) * float(n) - float(n_edges_global)) / float(n_edges_global)
        need = ctx.needs_input_grad[0]
        # labels are looked up by GLOBAL column id, whatever the exchange mode of the SpMM
        loss, dzt = _ops.decoder_bce_raw(full[:n], None, sg.csr_global("fwd"), sg.csr_global("bwd") if need else None, pw,
                                    want_grad=need, row_begin=p.r0, n_local=p.n_local)
        sg.allreduce_sum(loss)
        ctx.save_for_backward(dzt, mask_local)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dzt, mask_local = ctx.saved_tensors
        dz = dzt * g
        if mask_local is not None:
            dz = dz * mask_local
        return dz, None, None, None


def sharded_decoder_bce(z_local, mask_local, sg, n_edges_global=None):
    if n_edges_global is None:
        n_edges_global = sg.n_edges_global()          # (cached: no host read-back per step)
    return _ops.ShardedDecoderBCEFunction.apply(z_local, mask_local, sg, n_edges_global)


def decoder_dense(Z, mask=None):
    return _ops.DecoderDenseFunction.apply(Z, mask)
