"""VGAE heads: reparameterisation + KL (separate, packed, and fused with the loss's prepare step).

Part of the package gae_dgl_amd.ops (one module until round 6).  Functions look each other up in the PACKAGE
namespace (`_ops.<name>`) when they run: setting a flag or replacing a function on `gae_dgl_amd.ops` reaches every caller."""
import ctypes
import os

import torch

import gae_dgl_amd.ops as _ops
from .. import _lib
from ._base import _f32, _gpu, _on_device, _ptr, _stream, _vp, _workspace

__all__ = [
    'VGAEHeadFunction', 'VGAEPackedHeadFunction', 'vgae_head_packed', 'VGAE_FUSED_LOSS', 'VGAEHeadLossFunction',
    'vgae_head_loss', 'vgae_head',
]


class VGAEHeadFunction(torch.autograd.Function):
    """z = mu + eps exp(logstd) and the KL term of Kipf & Welling's VGAE, fused (gae_vgae_head_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, mu, logstd, eps):
        mu = _gpu(mu, "mu").contiguous(); logstd = logstd.contiguous(); eps = eps.contiguous()
        n, d = mu.shape
        z = torch.empty_like(mu)
        kl = torch.empty(1, dtype=torch.float32, device=mu.device)
        with _on_device(mu.device):
            ws = _workspace(_lib.load().gae_vgae_head_workspace_bytes(n * d), mu.device)
            _lib.call("gae_vgae_head_fwd", _ptr(mu), _ptr(logstd), d, _ptr(eps), n, d, _ptr(z), _ptr(kl), _ptr(ws),
                      ws.numel(), _stream())
        ctx.save_for_backward(mu, logstd, eps)
        return z, kl.reshape(())

    @staticmethod
    def backward(ctx, dz, dkl):
        mu, logstd, eps = ctx.saved_tensors
        n, d = mu.shape
        dmu = torch.empty_like(mu); dls = torch.empty_like(mu)
        dz = None if dz is None else dz.contiguous()
        gkl = (torch.zeros(1, device=mu.device) if dkl is None else dkl.reshape(1).float().contiguous())
        with _on_device(mu.device):
            _lib.call("gae_vgae_head_bwd", _ptr(dz), _ptr(mu), _ptr(logstd), d, _ptr(eps), _ptr(gkl), n, d, _ptr(dmu),
                      _ptr(dls), _stream())
        return dmu, dls, None


class VGAEPackedHeadFunction(torch.autograd.Function):
    """the same on both heads PACKED in one [n, 2 d] matrix [mu | logstd] (what GCNTwoHeadFunction produces): one
    gradient matrix goes back, no slicing / concatenation kernels in between"""

    @staticmethod
    def forward(ctx, ml, eps):
        ml = _gpu(ml, "ml").contiguous(); eps = eps.contiguous()
        n, d2 = ml.shape
        d = d2 // 2
        z = torch.empty(n, d, dtype=torch.float32, device=ml.device)
        kl = torch.empty(1, dtype=torch.float32, device=ml.device)
        with _on_device(ml.device):
            ws = _workspace(_lib.load().gae_vgae_head_workspace_bytes(n * d), ml.device)
            _lib.call("gae_vgae_head_fwd", _ptr(ml), _vp(ml.data_ptr() + 4 * d), d2, _ptr(eps), n, d, _ptr(z), _ptr(kl),
                      _ptr(ws), ws.numel(), _stream())
        ctx.save_for_backward(ml, eps)
        return z, kl.reshape(())

    @staticmethod
    def backward(ctx, dz, dkl):
        ml, eps = ctx.saved_tensors
        n, d2 = ml.shape
        d = d2 // 2
        dml = torch.empty_like(ml)
        dz = None if dz is None else dz.contiguous()
        gkl = (torch.zeros(1, device=ml.device) if dkl is None else dkl.reshape(1).float().contiguous())
        with _on_device(ml.device):
            _lib.call("gae_vgae_head_bwd", _ptr(dz), _ptr(ml), _vp(ml.data_ptr() + 4 * d), d2, _ptr(eps), _ptr(gkl), n, d,
                      _ptr(dml), _vp(dml.data_ptr() + 4 * d), _stream())
        return dml, None


def vgae_head_packed(ml, eps):
    return _ops.VGAEPackedHeadFunction.apply(ml, eps)


VGAE_FUSED_LOSS = os.environ.get("GAE_VGAE_FUSED_LOSS", "1") != "0"


class VGAEHeadLossFunction(torch.autograd.Function):
    """The VGAE head, the KL term and the fused reconstruction loss on the PACKED heads [mu | logstd] (d = 16) as
    three launches: gae_x_vgae_head_prep (noise of this draw, z, KL partials, the loss's prepare step), then the dense
    and the edge kernel of gae_x_decoder_bce_prepared; the scalar rec + KL comes out of the loss's final reduction,
    which also adds the KL partials (gae_bce_tail::kl_*) -- inside ``deferred_loss_finalize()`` as one block of the
    optimiser launch.  Replaces gae_normal_noise, gae_vgae_head_fwd (2 launches), the prepare and final-reduction
    launches of the loss, the draw-counter increment and the ``rec + kl`` addition: 5 launches instead of 12.
    Returns (loss, z, kl, rec, eps); only ``loss`` carries a gradient (to ``ml``)."""

    @staticmethod
    def forward(ctx, ml, graph, eps, noise):
        ml = _f32(_gpu(ml, "ml"), "vgae loss: ml").contiguous()
        n, d2 = ml.shape
        d = d2 // 2
        dev = ml.device
        draw = eps is None
        seed, offset, draws = noise if noise is not None else (0, 0, None)
        eps_t = torch.empty(n, d, dtype=torch.float32, device=dev) if draw else _f32(_gpu(eps, "eps"), "eps").contiguous()
        z = torch.empty(n, d, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad[0]
        nnz = graph.number_of_edges()
        pw = (float(n) * float(n) - float(nnz)) / float(nnz)
        indptr, indices = graph.csr()
        t_indptr, t_indices = graph.csc() if need else (None, None)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        kl = torch.empty(1, dtype=torch.float32, device=dev)
        rec = torch.empty(1, dtype=torch.float32, device=dev)
        dZ = torch.empty(n, d, dtype=torch.float32, device=dev) if need else None
        with _on_device(dev):
            nbytes = _lib.load().gae_decoder_bce_workspace_bytes(n, n, d)
            if nbytes < 0:
                _lib.check(int(nbytes), "gae_decoder_bce_workspace_bytes")
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            klp = torch.empty((n + 63) // 64, dtype=torch.float64, device=dev)
            lay = _lib.BcePrep()
            _lib.call("gae_x_decoder_bce_prep_layout", n, d, _ptr(ws), ws.numel(), ctypes.byref(lay))
            blocks = ctypes.c_int64(0)
            if _ops.current_step().tails:
                _ops.current_step().flush_loss_tails()
            _lib.call("gae_x_vgae_head_prep", _ptr(ml), _vp(ml.data_ptr() + 4 * d), d2, _ptr(eps_t), 1 if draw else 0,
                      int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _ptr(draws) if draw else None, n, d, _ptr(z),
                      ctypes.byref(lay), _ptr(klp), klp.numel(), ctypes.byref(blocks), _stream())
            tail = _lib.BceTail()
            _lib.call("gae_x_decoder_bce_defer_finalize", ctypes.byref(tail))
            try:
                _ops.STATS["prepared_losses"] += 1
                _lib.call("gae_x_decoder_bce_prepared", None, d, n, d, _ptr(indptr), _ptr(indices), _ptr(t_indptr),
                          _ptr(t_indices), float(pw), None, 0.0, None, int(blocks.value), _ptr(loss), _ptr(dZ), d, _ptr(ws),
                          ws.numel(), _stream())
            except Exception:
                _lib.call("gae_x_decoder_bce_defer_finalize", None)
                raise
            tail.kl_partial = klp.data_ptr(); tail.n_kl = int(blocks.value); tail.kl_scale = -0.5 / (float(n) * float(n))
            tail.kl_out = kl.data_ptr(); tail.rec_out = rec.data_ptr()
            if draw and draws is not None:
                tail.bump_draw = draws.data_ptr()          # the noise counter advances with the loss's last block
            keep = (loss, ws, klp, kl, rec, draws)
            if _ops.current_step().defer_loss and need:
                _ops.current_step().tails.append((tail, keep))
            else:
                _lib.call("gae_x_decoder_bce_finalize", ctypes.byref(tail), _stream())
        ctx.save_for_backward(ml, eps_t, dZ)
        kl0, rec0 = kl.reshape(()), rec.reshape(())
        ctx.mark_non_differentiable(z, kl0, rec0, eps_t)     # (the returned objects themselves: autograd would otherwise
        ctx.set_materialize_grads(False)                      #  fill a zero gradient for each of them in every backward)
        return loss.reshape(()), z, kl0, rec0, eps_t

    @staticmethod
    def backward(ctx, g, *unused):
        ml, eps, dZ = ctx.saved_tensors
        n, d2 = ml.shape
        d = d2 // 2
        dml = torch.empty_like(ml)
        unit = _ops._is_unit(g)
        dz = dZ if unit else dZ * g
        gkl = g.reshape(1).float().contiguous()              # d loss / d kl = the upstream gradient
        with _on_device(ml.device):
            _lib.call("gae_vgae_head_bwd", _ptr(dz), _ptr(ml), _vp(ml.data_ptr() + 4 * d), d2, _ptr(eps), _ptr(gkl), n, d,
                      _ptr(dml), _vp(dml.data_ptr() + 4 * d), _stream())
        return dml, None, None, None


def vgae_head_loss(ml, graph, eps=None, noise=None):
    """(loss, z, kl, rec, eps) -- see VGAEHeadLossFunction; None when the fused form does not apply (d != 16, no
    edges, fixed-capacity batch)"""
    if (not _ops.VGAE_FUSED_LOSS or not isinstance(ml, torch.Tensor) or not ml.is_cuda or ml.dim() != 2 or ml.shape[1] != 32
            or ml.dtype != torch.float32 or graph.number_of_edges() == 0
            or getattr(graph, "batch_counts", None) is not None or ml.shape[0] != graph.number_of_nodes()):
        return None
    return _ops.VGAEHeadLossFunction.apply(ml, graph, eps, noise)


def vgae_head(mu, logstd, eps):
    return _ops.VGAEHeadFunction.apply(mu, logstd, eps)
