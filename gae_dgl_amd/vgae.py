"""Variational graph auto-encoder (BASELINE config 5).  The reference only
cites the paper (README.md:58, Kipf & Welling 2016); this follows the paper
with the reference's building blocks: a shared GCN layer (ReLU), two GCN heads
with identity activation for mu and log(sigma), z = mu + eps * exp(log sigma),
loss = weighted BCE reconstruction (the fused decoder+BCE kernel, identical to
GAE's) + KL = -(0.5 / N) * mean_i sum_j (1 + 2 log sigma - mu^2 - sigma^2).

Feature storage may be bf16 (``g.ndata['h']`` in torch.bfloat16): the
layer-1 aggregation then runs the bf16-storage / fp32-accumulate SpMM."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .gae import GCN, InnerProductDecoder, identity


# mu and log sigma heads as ONE fused launch on the shared aggregate (gae_x_gcn_layer_fused2) and one packed gradient
# path; False runs them as two GCN layers (same values up to the fp32 rounding of the dW summation order)
FUSE_HEADS = True


class VGAE(nn.Module):
    def __init__(self, in_dim, hidden_dims=(32, 16), *, norm=None, seed=None):
        super().__init__()
        h0, d = int(hidden_dims[0]), int(hidden_dims[-1])
        self.shared = GCN(in_dim, h0, F.relu, norm)
        self.mu_head = GCN(h0, d, identity, norm)
        self.logstd_head = GCN(h0, d, identity, norm)
        self.decoder = InnerProductDecoder(activation=identity, dropout=0.0)
        self.seed = seed
        self._draws = None
        self.eps = None          # inject a fixed noise tensor for tests; None = draw from the library RNG
        self.last = {}

    def _heads_packed(self, g, h):
        """[mu | logstd] from ONE fused launch on the shared aggregate (ops.GCNTwoHeadFunction), or None"""
        if not FUSE_HEADS:
            return None
        g._follow(h)
        mode = g.norm_mode if self.mu_head.norm is None else self.mu_head.norm
        if mode not in ("none", "both"):
            return None
        return ops.gcn_two_heads(g, h, self.mu_head.apply_mod.linear, self.logstd_head.apply_mod.linear,
                                 use_norm=(mode == "both"))

    def encode(self, g):
        h = self.shared(g, g.ndata['h'])
        ml = self._heads_packed(g, h)
        if ml is not None:
            d = ml.shape[1] // 2
            return ml[:, :d], ml[:, d:]
        mu = self.mu_head(g, h)
        logstd = self.logstd_head(g, h)
        return mu, logstd

    def _noise(self, like):
        if self.eps is not None:
            return self.eps
        if self._draws is None or self._draws.device != like.device:
            self._draws = torch.zeros(1, dtype=torch.int64, device=like.device)
        seed = self.seed if self.seed is not None else int(torch.initial_seed())
        eps = ops.normal_noise(tuple(like.shape), seed, 0, like.device, draw_counter=self._draws)
        self._draws += 1
        return eps

    def _fused_loss(self, g, ml):
        """head + KL + reconstruction loss as ops.VGAEHeadLossFunction (noise drawn inside its first launch with the
        stream _noise() uses), or None"""
        if self.decoder.dropout or self.decoder.mask is not None or not ml.is_cuda:
            return None
        noise = None
        if self.eps is None:
            if self._draws is None or self._draws.device != ml.device:
                self._draws = torch.zeros(1, dtype=torch.int64, device=ml.device)
            seed = self.seed if self.seed is not None else int(torch.initial_seed())
            noise = (seed, 0, self._draws)
        return ops.vgae_head_loss(ml, g, self.eps, noise)

    def loss(self, g):
        """reconstruction BCE (train_inductive.py:44-48 semantics, fused) + KL"""
        h = self.shared(g, g.ndata['h'])
        ml = self._heads_packed(g, h)
        if ml is not None:                       # both heads in one launch, [mu | logstd] packed end to end
            d = ml.shape[1] // 2
            mu, logstd = ml[:, :d], ml[:, d:]
            fused = self._fused_loss(g, ml)
            if fused is not None:
                loss, z, kl, rec, eps = fused
                g.ndata['h'] = z
                self.decoder.last_mask = None
                self.last = {"mu": mu, "logstd": logstd, "eps": eps, "z": z, "kl": kl, "rec": rec}
                return loss
            eps = self._noise(mu)
            z, kl = ops.vgae_head_packed(ml, eps)
        else:
            mu = self.mu_head(g, h)
            logstd = self.logstd_head(g, h)
            eps = self._noise(mu)
            z, kl = ops.vgae_head(mu, logstd, eps)
        g.ndata['h'] = z
        rec = self.decoder.loss(z, g)
        self.last = {"mu": mu, "logstd": logstd, "eps": eps, "z": z, "kl": kl, "rec": rec}
        return rec + kl

    def forward(self, g):
        """sampled Z Z^T logits (dense parity / inference path)"""
        mu, logstd = self.encode(g)
        z, _ = ops.vgae_head(mu, logstd, self._noise(mu))
        g.ndata['h'] = z
        return self.decoder(z)
