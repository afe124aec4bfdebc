"""CPU oracle for the GCN-encoder hot path of shionhonda/gae-dgl.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gae_dgl_amd/`` imports this module;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may use it, and only as the checker / the timed CPU baseline.

What it restates (citations are relative to the upstream reference tree):

* ``gae_dgl/gae.py:7-72``     NodeApplyModule / GCN / GAE / InnerProductDecoder
* ``gae_dgl/train_inductive.py:31-35,43-53``  dgl.batch collate, dense label,
  pos_weight, BCE-with-logits(mean), backward
* ``gae_dgl/train_transductive.py:55-60``     in-degree^-1/2 norm, pos_weight

Parity pin status
-----------------
* The reference holds NO tests, golden vectors or fixtures for this path.
* ``gae.py`` itself is pinned: ``tests/golden/make_golden.py`` imports the
  reference's own ``gae.py`` unmodified (in the build container only) and the
  resulting vectors are committed under ``tests/golden/*.npz``;
  ``tests/test_oracle_golden.py`` checks every function below against them.
* The sparse arithmetic lives in the third-party package ``dgl`` which is
  neither vendored nor version-pinned by the reference (bare name in
  README.md:10-16; API era = DGL 0.4.x) and is absent from the image.  Its
  published semantics are restated here and anchored on the reference's call
  sites: ``update_all(copy_src('h','m'), sum('m','h'))`` = in-edge sum
  (gae.py:18-19,28), ``dgl.batch`` = block-diagonal union with node ids
  offset by the exclusive prefix sum of node counts (train_inductive.py:34),
  ``adjacency_matrix().to_dense()`` = COO->dense with duplicate edges added
  (train_inductive.py:44).  At that DGL boundary parity is therefore
  "restated, not pinned by reference vectors" -- the golden generator's
  in-memory DGL stand-in encodes the same documented semantics and a dense
  fp64 matrix restatement cross-checks both.

Integer/index results are exact; floating point is fp32 (or fp64 when the
caller passes fp64 tensors) on the CPU through plain torch ops.
"""
from __future__ import annotations

import numpy as np
import torch

# --------------------------------------------------------------------------
# graph structure (integer work, numpy, bit-exact)
# --------------------------------------------------------------------------


def csr_from_coo(src, dst, n_rows: int, n_cols: int | None = None):
    """CSR of the aggregation matrix A (rows = destination, cols = source).

    gae.py:18-19,28 -- ``update_all(copy_src, sum)`` reduces over IN-edges, so
    ``M[v] = sum_{(u->v)} H[u]``; row v of A lists the sources u.  Inside a row
    the columns are sorted ascending (duplicates kept, multigraph-additive).
    Returns int32 ``indptr[n_rows+1]``, int32 ``indices[E]``.
    """
    src = np.asarray(src, dtype=np.int64).ravel()
    dst = np.asarray(dst, dtype=np.int64).ravel()
    if n_cols is None:
        n_cols = n_rows
    assert src.shape == dst.shape
    if src.size:
        assert 0 <= src.min() and src.max() < n_cols, "source id out of range"
        assert 0 <= dst.min() and dst.max() < n_rows, "destination id out of range"
    order = np.lexsort((src, dst))  # primary key dst, secondary src; stable
    counts = np.bincount(dst, minlength=n_rows)
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr.astype(np.int32), src[order].astype(np.int32)


def csc_from_coo(src, dst, n_rows: int, n_cols: int | None = None):
    """CSR of A^T (rows = source, cols = destination) used by the backward
    ``dH = A^T dM`` (autograd of gae.py:28, train_inductive.py:51)."""
    if n_cols is None:
        n_cols = n_rows
    return csr_from_coo(dst, src, n_cols, n_rows)


def in_degrees(dst, n: int):
    """``g.in_degrees()`` (train_transductive.py:55): #edges arriving at v."""
    return np.bincount(np.asarray(dst, dtype=np.int64).ravel(), minlength=n).astype(np.int64)


def norm_from_in_degrees(deg):
    """train_transductive.py:55-58: ``norm = deg^-0.5``; ``inf -> 0``; fp32."""
    deg = torch.as_tensor(np.asarray(deg)).float()
    norm = torch.pow(deg, -0.5)
    norm[torch.isinf(norm)] = 0
    return norm


def batch_graphs(graphs):
    """``dgl.batch(samples)`` (train_inductive.py:31-35).

    ``graphs`` = list of ``(n_nodes, src, dst, X)``.  Node ids of graph i are
    offset by the exclusive prefix sum of node counts in list order, edge
    lists are concatenated in list order, ``ndata['h']`` is concatenated on
    dim 0.  Returns ``(N, src, dst, X, graph_ptr)``.
    """
    n_list = [int(g[0]) for g in graphs]
    graph_ptr = np.zeros(len(graphs) + 1, dtype=np.int64)
    np.cumsum(n_list, out=graph_ptr[1:])
    src = [np.asarray(g[1], dtype=np.int64) + graph_ptr[i] for i, g in enumerate(graphs)]
    dst = [np.asarray(g[2], dtype=np.int64) + graph_ptr[i] for i, g in enumerate(graphs)]
    src = np.concatenate(src) if src else np.zeros(0, np.int64)
    dst = np.concatenate(dst) if dst else np.zeros(0, np.int64)
    X = torch.cat([torch.as_tensor(g[3]) for g in graphs], dim=0)
    return int(graph_ptr[-1]), src, dst, X, graph_ptr


def dense_adjacency(src, dst, n: int, dtype=torch.float32):
    """``g.adjacency_matrix().to_dense()`` (train_inductive.py:44,
    train_transductive.py:59).  Rows = destination, cols = source (DGL 0.4
    ``transpose=False``); duplicate edges ADD (value 2.0)."""
    A = torch.zeros(n, n, dtype=dtype)
    idx = (torch.as_tensor(np.asarray(dst, dtype=np.int64)),
           torch.as_tensor(np.asarray(src, dtype=np.int64)))
    A.index_put_(idx, torch.ones(idx[0].numel(), dtype=dtype), accumulate=True)
    return A


# --------------------------------------------------------------------------
# K1/K2  SpMM (the DGL fused copy_src+sum), row-order summation
# --------------------------------------------------------------------------


def spmm_csr(indptr, indices, H, row_scale=None, col_scale=None):
    """``M = diag(row_scale) A diag(col_scale) H`` with A given as CSR.

    ``row_scale = col_scale = None`` is the reference's behaviour (gae.py:28,
    plain in-edge sum, no normalisation).  Passing ``norm`` for both gives the
    north-star ``D^-1/2 A D^-1/2`` (train_transductive.py:55-57).
    Accumulates in row order (CSR order) in the dtype of ``H``.
    """
    indptr = torch.as_tensor(np.asarray(indptr, dtype=np.int64))
    indices = torch.as_tensor(np.asarray(indices, dtype=np.int64))
    H = torch.as_tensor(H)
    n_rows = indptr.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n_rows), indptr[1:] - indptr[:-1])
    G = H[indices]
    if col_scale is not None:
        G = G * torch.as_tensor(col_scale).to(H.dtype).reshape(-1)[indices].unsqueeze(1)
    M = torch.zeros(n_rows, H.shape[1], dtype=H.dtype)
    M.index_add_(0, rows, G)
    if row_scale is not None:
        M = M * torch.as_tensor(row_scale).to(H.dtype).reshape(-1, 1)
    return M


def spmm_csr_loops(indptr, indices, H, row_scale=None, col_scale=None):
    """Pure-Python triple loop of the same op (tiny cases only); an
    independent cross-check of :func:`spmm_csr`."""
    H = np.asarray(H, dtype=np.float64)
    n_rows = len(indptr) - 1
    M = np.zeros((n_rows, H.shape[1]), dtype=np.float64)
    for v in range(n_rows):
        for e in range(int(indptr[v]), int(indptr[v + 1])):
            u = int(indices[e])
            c = 1.0 if col_scale is None else float(col_scale[u])
            M[v] += c * H[u]
        if row_scale is not None:
            M[v] *= float(row_scale[v])
    return M


# --------------------------------------------------------------------------
# K3-K5 GCN layer, encoder, K6-K7 decoder, K8 loss
# --------------------------------------------------------------------------


def activation_rule(n_layers: int):
    """gae.py:36-45: ReLU on layers 0..L-2, identity on the last; a single
    layer is identity only."""
    return ["relu"] * (n_layers - 1) + ["identity"]


def gcn_layer(indptr, indices, H, W, b, act, norm=None):
    """gae.py:26-31 + 13-16: aggregate -> Linear(+bias) -> activation.
    ``W`` is ``[out, in]`` like ``nn.Linear.weight`` (gae.py:10)."""
    M = spmm_csr(indptr, indices, H, norm, norm)
    Y = M @ torch.as_tensor(W).t() + torch.as_tensor(b)
    return torch.relu(Y) if act == "relu" else Y


def gae_encode(indptr, indices, X, weights, biases, norm=None):
    """gae.py:57-61: the encoder loop; returns Z ``[N, d]``."""
    h = torch.as_tensor(X)
    acts = activation_rule(len(weights))
    for W, b, a in zip(weights, biases, acts):
        h = gcn_layer(indptr, indices, h, W, b, a, norm)
    return h


def decoder_logits(Z, mask=None):
    """gae.py:69-72 with the activation GAE passes (identity, gae.py:47).
    ``mask`` is the already-scaled inverted-dropout multiplier (0 or
    1/(1-p)); ``None`` = dropout disabled (p = 0)."""
    Z = torch.as_tensor(Z)
    Zt = Z if mask is None else Z * torch.as_tensor(mask).to(Z.dtype)
    return Zt @ Zt.t()


def pos_weight_of(adj):
    """train_inductive.py:46: ``(N*N - sum(adj)) / sum(adj)``."""
    n = adj.shape[0]
    s = adj.sum()
    return (n * n - s) / s


def bce_with_logits_mean(logits, adj, pos_weight):
    """train_inductive.py:48 -- F.binary_cross_entropy_with_logits with
    ``pos_weight`` and the default mean reduction, written out:
    ``l = (1-y) x + (1 + (pw-1) y) softplus(-x)``."""
    x = torch.as_tensor(logits)
    y = torch.as_tensor(adj).to(x.dtype)
    pw = torch.as_tensor(pos_weight).to(x.dtype)
    lw = 1 + (pw - 1) * y
    loss = (1 - y) * x + lw * torch.nn.functional.softplus(-x)
    return loss.mean()


def mse_mean(logits, adj):
    """optuna_gae.py:16,21 -- the hyper-parameter search's criterion, ``nn.MSELoss()(model.forward(g),
    g.adjacency_matrix().to_dense())`` (mean over all N^2 ordered pairs) on the logits GAE.forward returns."""
    x = torch.as_tensor(logits)
    y = torch.as_tensor(adj).to(x.dtype)
    return ((x - y) ** 2).mean()


def mse_closed_form(Zt, indptr, indices):
    """The same number without the N x N matrices (what ops.decoder_mse evaluates on the device):
        sum_ij (s_ij - a_ij)^2 = ||Zt^T Zt||_F^2 - 2 <Zt, A Zt> + sum_ij a_ij^2,   s = Zt Zt^T,
    and its gradient  dL/dZt = (2 / N^2) (2 Zt (Zt^T Zt) - A Zt - A^T Zt).  ``indptr`` / ``indices``: CSR of A (rows =
    destination); duplicate edges count (a_ij = multiplicity).  Returns (loss, dZt) in Zt's dtype."""
    Zt = torch.as_tensor(Zt)
    ip = np.asarray(indptr, dtype=np.int64); ix = np.asarray(indices, dtype=np.int64)
    n = Zt.shape[0]
    rows = np.repeat(np.arange(n), np.diff(ip))
    A = torch.zeros(n, n, dtype=Zt.dtype)
    A.index_put_((torch.as_tensor(rows), torch.as_tensor(ix)), torch.ones(ix.size, dtype=Zt.dtype), accumulate=True)
    G = Zt.t() @ Zt
    AZ = A @ Zt
    loss = ((G * G).sum() - 2 * (Zt * AZ).sum() + (A * A).sum()) / (n * n)
    dZt = (2.0 / (n * n)) * (2 * Zt @ G - AZ - A.t() @ Zt)
    return loss, dZt


def bce_row_window(Zt, r0, r1, indptr, indices, t_indptr, t_indices, pos_weight):
    """Rows ``[r0, r1)`` of the loss of train_inductive.py:44-48 for graphs whose
    dense N x N label does not fit (ZINC batch of 4096 molecules: 9e9 logits).
    ``Zt`` is the dropped embedding (gae.py:70), ``(indptr, indices)`` the CSR
    of the label (rows = destination: ``y_ij`` = #edges j->i, duplicates add,
    train_inductive.py:44), ``(t_indptr, t_indices)`` the CSR of its transpose
    (``y_ji``).  Returns ``(sum_{i in window, j} l_ij / N^2, dLoss/dZt[r0:r1])``
    in fp64 with ``l = (1-y) x + (1+(pw-1) y) softplus(-x)`` and
    ``dZt_i = sum_j (G_ij + G_ji) Zt_j``, ``G = dl/dx / N^2`` (x is symmetric,
    the label need not be)."""
    Zt = torch.as_tensor(Zt).double()
    n = Zt.shape[0]
    pw = float(pos_weight)
    x = Zt[r0:r1] @ Zt.t()

    def label(ip, ix):
        y = torch.zeros(r1 - r0, n, dtype=torch.float64)
        ip = np.asarray(ip, dtype=np.int64)
        cols = torch.as_tensor(np.asarray(ix[ip[r0]:ip[r1]], dtype=np.int64))
        rows = torch.repeat_interleave(torch.arange(r1 - r0), torch.as_tensor(np.diff(ip[r0:r1 + 1])))
        y.index_put_((rows, cols), torch.ones(len(cols), dtype=torch.float64), accumulate=True)
        return y

    y_row, y_col = label(indptr, indices), label(t_indptr, t_indices)
    sp = torch.nn.functional.softplus(-x)
    loss = ((1 - y_row) * x + (1 + (pw - 1) * y_row) * sp).sum() / (float(n) * n)
    sig_neg = torch.sigmoid(-x)
    g = ((1 - y_row) - (1 + (pw - 1) * y_row) * sig_neg) + ((1 - y_col) - (1 + (pw - 1) * y_col) * sig_neg)
    return loss, (g @ Zt) / (float(n) * n)


def segment_readout(Z, graph_ptr):
    """README.md:54: "concatenation of mean, sum, and max aggregation of the hidden vector H in R^{N x 16}" per
    molecule -> [G, 3 d].  (The reference ships no code for it; with DGL it is mean_nodes / sum_nodes / max_nodes
    on the batched graph.)  An empty graph gives zeros."""
    Z = np.asarray(Z, dtype=np.float64)
    gp = np.asarray(graph_ptr, dtype=np.int64)
    d = Z.shape[1]
    out = np.zeros((len(gp) - 1, 3 * d))
    for g in range(len(gp) - 1):
        seg = Z[gp[g]:gp[g + 1]]
        if len(seg):
            out[g, :d], out[g, d:2 * d], out[g, 2 * d:] = seg.mean(0), seg.sum(0), seg.max(0)
    return out


def gae_loss_and_grads(src, dst, n, X, weights, biases, mask=None, norm=None, criterion="bce"):
    """One reference training step up to the gradients
    (train_inductive.py:44-51): dense label, pos_weight, GAE.forward with an
    injected dropout mask, BCE-with-logits mean, backward.  Returns
    ``(loss, Z, logits, dW list, db list)``; dtype follows ``X``.
    ``criterion="mse"``: nn.MSELoss() on the same logits and label (optuna_gae.py:16,21)."""
    indptr, indices = csr_from_coo(src, dst, n)
    X = torch.as_tensor(X)
    Ws = [torch.as_tensor(w).to(X.dtype).clone().requires_grad_(True) for w in weights]
    bs = [torch.as_tensor(b).to(X.dtype).clone().requires_grad_(True) for b in biases]
    Z = gae_encode(indptr, indices, X, Ws, bs, norm)
    logits = decoder_logits(Z, mask)
    adj = dense_adjacency(src, dst, n, dtype=X.dtype)
    pw = pos_weight_of(adj)
    loss = bce_with_logits_mean(logits, adj, pw) if criterion == "bce" else mse_mean(logits, adj)
    grads = torch.autograd.grad(loss, Ws + bs)
    L = len(Ws)
    return (loss.detach(), Z.detach(), logits.detach(),
            [g.detach() for g in grads[:L]], [g.detach() for g in grads[L:]])


def gae_loss_and_grads_windowed(src, dst, n, X, weights, biases, mask=None, norm=None, window=1024):
    """:func:`gae_loss_and_grads` (criterion "bce") for graphs whose dense fp64 N x N label / logits / autograd
    temporaries do not fit comfortably (Pubmed: 3.1 GB each): the same step -- train_transductive.py:59-66 on
    gae.py:49-55 -- with the N x N part evaluated ``window`` rows at a time by :func:`bce_row_window` (loss share and
    dLoss/dZt of those rows) and the encoder differentiated by autograd from the assembled dLoss/dZ.  Always fp64.
    Returns ``(loss, Z, dZ, dW list, db list)`` (no logits: they are never whole).  tests/test_oracle_golden.py
    pins it against :func:`gae_loss_and_grads` on the golden graphs."""
    indptr, indices = csr_from_coo(src, dst, n)
    t_indptr, t_indices = csc_from_coo(src, dst, n)
    X = torch.as_tensor(X).double()
    Ws = [torch.as_tensor(w).double().clone().requires_grad_(True) for w in weights]
    bs = [torch.as_tensor(b).double().clone().requires_grad_(True) for b in biases]
    Z = gae_encode(indptr, indices, X, Ws, bs, None if norm is None else torch.as_tensor(norm).double())
    m = None if mask is None else torch.as_tensor(mask).double()
    Zt = (Z if m is None else Z * m).detach()
    e = float(len(indices))
    pw = (float(n) * n - e) / e                       # train_inductive.py:46 on the label's sum (duplicates add)
    loss = torch.zeros((), dtype=torch.float64)
    dZt = torch.empty_like(Zt)
    for r0 in range(0, n, window):
        r1 = min(r0 + window, n)
        share, g = bce_row_window(Zt, r0, r1, indptr, indices, t_indptr, t_indices, pw)
        loss += share
        dZt[r0:r1] = g
    dZ = dZt if m is None else dZt * m
    grads = torch.autograd.grad(Z, Ws + bs, grad_outputs=dZ)
    L = len(Ws)
    return loss, Z.detach(), dZ, [g.detach() for g in grads[:L]], [g.detach() for g in grads[L:]]


def dense_restatement_encode(src, dst, n, X, weights, biases, norm=None):
    """Independent fp64 dense-matrix restatement
    ``Z = A relu((A X) W1^T + b1) W2^T + b2 ...`` used to cross-check the CSR
    path."""
    A = dense_adjacency(src, dst, n, dtype=torch.float64)
    if norm is not None:
        nv = torch.as_tensor(norm).double().reshape(-1)
        A = nv.unsqueeze(1) * A * nv.unsqueeze(0)
    h = torch.as_tensor(X).double()
    acts = activation_rule(len(weights))
    for W, b, a in zip(weights, biases, acts):
        h = (A @ h) @ torch.as_tensor(W).double().t() + torch.as_tensor(b).double()
        if a == "relu":
            h = torch.relu(h)
    return h


# --------------------------------------------------------------------------
# VGAE (BASELINE config 5; not in the reference, Kipf & Welling 2016 --
# README.md:58 only cites the paper).  Parity for it is pinned only by this
# restatement.
# --------------------------------------------------------------------------


def vgae_forward(indptr, indices, X, W1, b1, Wmu, bmu, Wls, bls, eps, norm=None):
    h = gcn_layer(indptr, indices, X, W1, b1, "relu", norm)
    mu = gcn_layer(indptr, indices, h, Wmu, bmu, "identity", norm)
    logstd = gcn_layer(indptr, indices, h, Wls, bls, "identity", norm)
    z = mu + torch.as_tensor(eps).to(mu.dtype) * torch.exp(logstd)
    return mu, logstd, z


def vgae_kl(mu, logstd):
    """``-(0.5/N) * mean_i sum_j (1 + 2 logstd - mu^2 - exp(2 logstd))``."""
    n = mu.shape[0]
    return -(0.5 / n) * torch.mean(torch.sum(1 + 2 * logstd - mu ** 2 - torch.exp(2 * logstd), dim=1))


# --------------------------------------------------------------------------
# full CPU training step used as bench.py's cpu_baseline ("port")
# --------------------------------------------------------------------------


class CpuReferenceStep:
    """The reference step of train_inductive.py:43-53 in plain PyTorch CPU:
    dense adjacency label, pos_weight, GAE.forward (aggregate-then-Linear
    order of gae.py:26-31, SpMM as torch CSR sparse.mm), dense Z Z^T, BCE,
    backward, Adam."""

    def __init__(self, src, dst, n, X, in_dim, hidden_dims, lr=1e-2, seed=0, dropout=0.1):
        g = torch.Generator().manual_seed(seed)
        self.n = n
        self.src = np.asarray(src, dtype=np.int64)
        self.dst = np.asarray(dst, dtype=np.int64)
        indptr, indices = csr_from_coo(self.src, self.dst, n)
        self.A = torch.sparse_csr_tensor(torch.as_tensor(indptr.astype(np.int64)),
                                         torch.as_tensor(indices.astype(np.int64)),
                                         torch.ones(len(indices)), size=(n, n))
        self.X = torch.as_tensor(X).float()
        dims = [in_dim] + list(hidden_dims)
        self.layers = torch.nn.ModuleList(
            [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(hidden_dims))])
        self.acts = activation_rule(len(hidden_dims))
        self.dropout = dropout
        self.optim = torch.optim.Adam(self.layers.parameters(), lr=lr)
        self.n_edges = len(indices)
        self._g = g

    def encode(self):
        h = self.X
        for lin, a in zip(self.layers, self.acts):
            h = lin(torch.sparse.mm(self.A, h))
            if a == "relu":
                h = torch.relu(h)
        return h

    def step(self):
        adj = dense_adjacency(self.src, self.dst, self.n)
        pw = pos_weight_of(adj)
        z = self.encode()
        z = torch.nn.functional.dropout(z, self.dropout)  # always on, gae.py:70
        logits = z @ z.t()
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, adj, pos_weight=pw)
        self.optim.zero_grad()
        loss.backward()
        self.optim.step()
        return float(loss)
