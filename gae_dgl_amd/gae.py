"""Host-side mirror of the reference model file gae_dgl/gae.py: same class
names, constructor arguments, attribute names and state-dict keys
(``layers.{i}.apply_mod.linear.{weight,bias}``), same side effects on
``g.ndata['h']`` -- with every arithmetic op executed by the HIP kernels of
libgae_hip.so (SpMM, fp32-MFMA Linear+activation, inner-product decoder).

Extensions over the reference are keyword-only: ``norm="none"|"both"``
(gae.py applies no normalisation although train_transductive.py:55-58
computes one), an injectable dropout mask/seed for reproducible tests, and
the evaluation ORDER of a layer that narrows wide features (layer 1: 500 /
1433 / 3703 -> 32): ``transform_first=None`` (default, "auto") evaluates
such a layer as ``act(A (H W^T) + b)`` -- the value of the reference's
``act((A H) W^T + b)`` up to fp32 rounding (2e-7 of the scale measured; the
parity suite holds it to the same 1e-5 as everything else) -- because the
aggregation then runs at the OUTPUT width and the 40 MB aggregate ``A H`` is
neither written nor re-read (gae_xw_fwd / gae_spmm_csr_epilogue /
gae_xw_wgrad); ``transform_first=False`` keeps the reference's order
everywhere, ``True`` asks for the reorder on every narrowing layer (it takes
effect wherever the one-pass kernels apply, like "auto").  ``cache_aggregate``
(opt-in) keeps ``A H`` of a parameter-independent input across steps."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import function as fn
from . import ops
from .sparse import SparseFeatures
from ._lib import ACT_IDENTITY, ACT_RELU


# GCN layers with <= 64 input and <= 32 output features run as ONE launch (gae_gcn_layer_fused: aggregation + Linear +
# bias + activation, backward of the identity-activation layer likewise) when the graph carries a packed neighbour
# table; False restores the two launches per layer (update_all, apply_nodes) everywhere.  Same values up to fp32
# rounding of the 32 -> 16 product.
FUSE_NARROW_LAYERS = True
# Layers that narrow WIDE features (f_in >= 193, f_out <= 32) built with transform_first=None run as
# act(A (H W^T) + b) through the one-pass kernels; False makes "auto" mean the reference's order (experiments)
TRANSFORM_FIRST_AUTO = True


def identity(x):
    return x


def _act_code(activation):
    """fused-epilogue code of an activation callable, or None (apply it after)"""
    if activation is None or activation is identity:
        return ACT_IDENTITY
    if activation is F.relu or activation is torch.relu:
        return ACT_RELU
    if getattr(activation, "__name__", "") == "<lambda>":
        # the reference spells identity as ``lambda x: x`` (gae.py:43,45,47)
        code = getattr(activation, "__code__", None)
        if code is not None and code.co_argcount == 1 and code.co_code == (lambda x: x).__code__.co_code:
            return ACT_IDENTITY
    return None


class NodeApplyModule(nn.Module):
    """gae.py:7-16 -- Linear (+bias) then activation, as one fused HIP kernel."""

    def __init__(self, in_feats, out_feats, activation):
        super().__init__()
        self.linear = nn.Linear(in_feats, out_feats)  # default init, weight [out, in] (gae.py:10)
        self.activation = activation

    def forward(self, node):
        feats = node.data['h']
        if feats.dtype != torch.float32:  # bf16-stored features: aggregation ran in bf16 storage / fp32 accumulate
            feats = ops.float_rows(feats) if feats.is_cuda and feats.dim() == 2 else feats.float()
        fused = _act_code(self.activation)
        out = ops.linear(feats, self.linear.weight, self.linear.bias, ACT_IDENTITY if fused is None else fused)
        return {'h': out if fused is not None else self.activation(out)}


# the message / reduce pair the reference hands to update_all (gae.py:18-19): copy the source feature, sum it
gcn_msg, gcn_reduce = fn.copy_src(src='h', out='m'), fn.sum(msg='m', out='h')


class GCN(nn.Module):
    """gae.py:21-31 -- aggregate over in-edges (HIP SpMM) -> NodeApplyModule.

    ``transform_first`` (opt-in, layers that narrow the features only): ``act(A (H W^T) + b)``.  Mathematically the
    reference's ``act((A H) W^T + b)``; the rounding differs (measured 2e-7 relative), the aggregation moves
    F_out instead of F_in floats per edge and nothing of width F_in is written.
    ``cache_aggregate`` (opt-in): reuse ``A H`` while the same parameter-free input tensor comes back unchanged."""

    def __init__(self, in_feats, out_feats, activation, norm=None, *, transform_first=None, cache_aggregate=False):
        super().__init__()
        self.norm = norm
        self.apply_mod = NodeApplyModule(in_feats, out_feats, activation)
        # True: reorder whenever the layer narrows; None ("auto"): reorder where the one-pass kernels apply (wide
        # input, <= 32 outputs, graph with a packed neighbour table); False: the reference's order
        self.transform_first = bool(transform_first) and in_feats > out_feats
        self.transform_auto = transform_first is None and in_feats > out_feats and not cache_aggregate
        self.cache_aggregate = bool(cache_aggregate)
        self._agg_key = self._agg = None

    def _one_pass(self, g, feature):
        """the layer through ops.GCNTransformFirstFunction, or None when shapes / graph do not allow it"""
        code = _act_code(self.apply_mod.activation)
        if code is None or not isinstance(feature, torch.Tensor) or not feature.is_cuda:
            return None
        g._follow(feature)
        mode = g.norm_mode if self.norm is None else self.norm
        if mode not in ("none", "both"):
            return None
        lin = self.apply_mod.linear
        return ops.gcn_layer_transform_first(g, feature, lin.weight, lin.bias, code, use_norm=(mode == "both"))

    def _forward_sparse_input(self, g, sf):
        """``feature`` is a sparse.SparseFeatures (compressed input): the layer from its non-zeros, or from its dense
        form when the one-pass route does not apply"""
        code = _act_code(self.apply_mod.activation)
        if sf.device != g.device and sf.device.type == "cuda":
            g.to(sf.device)
        mode = g.norm_mode if self.norm is None else self.norm
        lin = self.apply_mod.linear
        if code is not None and mode in ("none", "both"):
            g.ndata['h'] = sf                    # same traffic on g.ndata['h'] as the dense path (gae.py:27,30)
            out = ops.gcn_layer_sparse_input(g, sf, lin.weight, lin.bias, code, use_norm=(mode == "both"))
            if out is not None:
                g.ndata.pop('h')
                return out
        return self.forward(g, sf.to_dense(cache=True))      # (constant features: densified once, then reused)

    def forward(self, g, feature):
        if isinstance(feature, SparseFeatures):
            return self._forward_sparse_input(g, feature)
        # same traffic on g.ndata['h'] as the reference: set (gae.py:27), reduced in place (:28), transformed in
        # place (:29), removed (:30)
        g.ndata['h'] = feature
        code = _act_code(self.apply_mod.activation)
        if self.transform_first or (self.transform_auto and TRANSFORM_FIRST_AUTO):
            # (where the one-pass kernels do not apply -- narrow inputs, graphs with heavy rows -- the layer keeps
            #  the reference's order)
            out = self._one_pass(g, feature)
            if out is not None:
                g.ndata.pop('h')
                return out
        if FUSE_NARROW_LAYERS and code is not None and not self.cache_aggregate and isinstance(feature, torch.Tensor) \
                and feature.is_cuda:
            g._follow(feature)
            mode = g.norm_mode if self.norm is None else self.norm
            lin = self.apply_mod.linear
            out = ops.gcn_layer(g, feature, lin.weight, lin.bias, code, use_norm=(mode == "both")) \
                if mode in ("none", "both") else None
            if out is not None:
                g.ndata.pop('h')
                return out
        if self.cache_aggregate and not feature.requires_grad:
            key = (id(g), g.number_of_edges(), feature.data_ptr(), feature._version, tuple(feature.shape), self.norm)
            if self._agg_key != key:
                g.update_all(gcn_msg, gcn_reduce, norm=self.norm)
                self._agg_key, self._agg = key, g.ndata['h']
            g.ndata['h'] = self._agg
        else:
            g.update_all(gcn_msg, gcn_reduce, norm=self.norm)
        g.apply_nodes(func=self.apply_mod)
        return g.ndata.pop('h')


class GAE(nn.Module):
    """gae.py:33-61.  ReLU on layers 0..L-2, identity on the last layer; a
    single hidden dim gives one identity layer (gae.py:36-45)."""

    def __init__(self, in_dim, hidden_dims, *, norm=None, transform_first=None, cache_first_aggregate=False):
        super().__init__()
        widths = [in_dim] + list(hidden_dims)
        last = len(widths) - 2
        self.layers = nn.ModuleList(
            GCN(widths[k], widths[k + 1], identity if k == last else F.relu, norm, transform_first=transform_first,
                cache_aggregate=cache_first_aggregate and k == 0) for k in range(last + 1))
        self.decoder = InnerProductDecoder(activation=identity)

    def _embed(self, g, write_back):
        z = g.ndata['h']
        for layer in self.layers:
            z = layer(g, z)
        if write_back:
            g.ndata['h'] = z     # forward() leaves the embedding on the graph (gae.py:53); encode() does not
        return z

    def forward(self, g):
        return self.decoder(self._embed(g, write_back=True))

    def encode(self, g):
        return self._embed(g, write_back=False)

    def reconstruction_loss(self, g, criterion="bce"):
        """The training loss of train_inductive.py:44-48 (dense label from g,
        pos_weight, BCE-with-logits mean over all N^2 ordered pairs) evaluated
        by the fused HIP kernel: numerically the same quantity as
        ``BCELoss(self.forward(g), adj, pos_weight)`` without the N x N logits /
        label matrices.  Side effect on ``g.ndata['h']`` as in forward().
        ``criterion="mse"``: the hyper-parameter search's ``nn.MSELoss()(self.forward(g), adj)``
        (optuna_gae.py:16,21), likewise without the N x N matrices (ops.decoder_mse)."""
        z = g.ndata['h']
        if criterion == "mse":
            for layer in self.layers:
                z = layer(g, z)
            g.ndata['h'] = z
            return self.decoder.loss_mse(z, g)
        if criterion != "bce":
            raise ValueError(f"criterion: 'bce' or 'mse', not {criterion!r}")
        for layer in self.layers[:-1]:
            z = layer(g, z)
        # the last layer may run the loss's prepare step in its epilogue (ops.loss_prepare_request)
        with self.decoder.prepare_request(g, z, self.layers[-1].apply_mod.linear.out_features) as req:
            z = self.layers[-1](g, z)
        g.ndata['h'] = z
        return self.decoder.loss(z, g, prepared=req.token)


class InnerProductDecoder(nn.Module):
    """gae.py:63-72.  Dropout is applied regardless of train()/eval() exactly
    like the reference (``F.dropout(z, self.dropout)`` omits ``training=``).
    The mask comes from the library's Philox counter RNG; set ``self.mask`` to
    inject a precomputed multiplier (0 or 1/(1-p)) instead."""

    def __init__(self, activation=torch.sigmoid, dropout=0.1, seed=None):
        super().__init__()
        self.dropout = dropout
        self.activation = activation
        self.mask = None
        self.seed = seed
        self._draws = None       # device-side draw counter (int64[1]); advanced by a device op per forward
        self.last_mask = None

    def _draw_mask(self, z):
        if self.mask is not None:
            return self.mask
        if not self.dropout:
            return None
        seed = self.seed if self.seed is not None else int(torch.initial_seed())
        if self._draws is None or self._draws.device != z.device:
            self._draws = torch.zeros(1, dtype=torch.int64, device=z.device)
        mask = ops.dropout_mask(tuple(z.shape), self.dropout, seed, 0, z.device, draw_counter=self._draws)
        self._draws += 1         # device op: a captured HIP graph draws a fresh mask every replay
        return mask

    def forward(self, z):
        self.last_mask = self._draw_mask(z)
        return self.activation(ops.decoder_dense(z, self.last_mask))

    def _loss_dropout(self, device):
        """(p, seed, offset, draw counter) of a mask drawn inside a launch, or None (given mask / no dropout)"""
        if self.mask is not None or not self.dropout:
            return None
        seed = self.seed if self.seed is not None else int(torch.initial_seed())
        if self._draws is None or self._draws.device != device:
            self._draws = torch.zeros(1, dtype=torch.int64, device=device)
        return (self.dropout, seed, 0, self._draws)

    def prepare_request(self, g, h, d):
        """the request a producer of Z answers by running this loss's prepare step in its own launch"""
        on_gpu = isinstance(h, torch.Tensor) and h.is_cuda
        return ops.loss_prepare_request(g if on_gpu else None, d, self.mask, self._loss_dropout(h.device) if on_gpu else None)

    def loss_mse(self, z, g):
        """nn.MSELoss()(self.forward(z), adj) (optuna_gae.py:16,21; identity activation) without the N x N matrices; the
        dropout mask is drawn as forward() draws it and kept in ``last_mask``"""
        if not (isinstance(z, torch.Tensor) and z.is_cuda):
            raise ops.GaeHipError("InnerProductDecoder.loss_mse: the HIP path needs device tensors")
        self.last_mask = self._draw_mask(z)
        return ops.decoder_mse(z, self.last_mask, g)

    def loss(self, z, g, prepared=None):
        """fused decoder + weighted BCE (identity activation = logits, gae.py:47).  The dropout mask of this call
        is drawn inside the fused launch (same Philox stream as _draw_mask) and kept in ``last_mask``.
        ``prepared``: the producer of z already ran the prepare step (prepare_request)."""
        if prepared is not None:
            self.last_mask = prepared["mask"]
            return ops.decoder_bce(z, None, g, prepared=prepared)
        drop = self._loss_dropout(z.device)
        if drop is None:
            self.last_mask = self.mask
            return ops.decoder_bce(z, self.mask, g)
        mask = torch.empty(tuple(z.shape), dtype=torch.float32, device=z.device)
        self.last_mask = mask
        return ops.decoder_bce(z, mask, g, dropout=drop)
