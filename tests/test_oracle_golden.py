"""Pins the CPU oracle against the vectors produced by the reference's own
gae.py (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import gae_oracle as O
from conftest import golden_params, load_golden

TOL = 1e-5  # north_star: fp32 features within 1e-5


def close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    assert float(np.abs(a - b).max()) / scale <= tol if b.size else True


def test_encode_matches_reference(golden):
    g = golden
    Ws, bs = golden_params(g)
    indptr, indices = O.csr_from_coo(g["src"], g["dst"], int(g["n"]))
    Z = O.gae_encode(indptr, indices, g["X"], Ws, bs)
    close(Z, g["Z"])
    # A9: encode leaves no 'h'; forward leaves Z
    assert not bool(g["encode_leaves_h"])
    close(g["forward_ndata_h"], g["Z"])


def test_dense_fp64_restatement(golden):
    g = golden
    Ws, bs = golden_params(g)
    Z = O.dense_restatement_encode(g["src"], g["dst"], int(g["n"]), g["X"], Ws, bs)
    close(Z, g["Z"])


def test_logits_dropout0_and_mask(golden):
    g = golden
    close(O.decoder_logits(g["Z"]), g["logits_p0"])
    close(O.decoder_logits(g["Z"], g["mask"]), g["logits_p01"])
    # A8: mask values are 0 or 1/(1-0.1)
    m = np.unique(g["mask"])
    assert all(np.isclose(v, 0) or np.isclose(v, 1 / 0.9) for v in m)


def test_label_posweight_loss(golden):
    g = golden
    n = int(g["n"])
    adj = O.dense_adjacency(g["src"], g["dst"], n)
    assert np.array_equal(adj.numpy(), g["adj"])  # exact incl. duplicates -> 2.0
    pw = O.pos_weight_of(adj)
    close(pw, g["pos_weight"], 1e-6)
    close(O.bce_with_logits_mean(g["logits_p0"], adj, pw), g["loss_p0"], 1e-6)
    close(O.bce_with_logits_mean(g["logits_p01"], adj, pw), g["loss_p01"], 1e-6)


def test_grads(golden):
    g = golden
    Ws, bs = golden_params(g)
    n = int(g["n"])
    for tag, mask in (("p0", None), ("p01", g["mask"])):
        loss, Z, logits, dW, db = O.gae_loss_and_grads(g["src"], g["dst"], n, g["X"], Ws, bs, mask)
        close(loss, g["loss_" + tag], 1e-6)
        for i in range(len(Ws)):
            close(dW[i], g[f"grad_{tag}/layers.{i}.apply_mod.linear.weight"])
            close(db[i], g[f"grad_{tag}/layers.{i}.apply_mod.linear.bias"])


def test_windowed_grads_match_reference_vectors(golden):
    """the row-window form of the step (what the full-size Pubmed parity test checks the device against): same loss
    and parameter gradients as the reference-generated vectors, with windows that divide the graph unevenly"""
    g = golden
    Ws, bs = golden_params(g)
    n = int(g["n"])
    for tag, mask in (("p0", None), ("p01", g["mask"])):
        for window in (max(1, n // 3 + 1), n + 5):
            loss, Z, dZ, dW, db = O.gae_loss_and_grads_windowed(g["src"], g["dst"], n, g["X"], Ws, bs, mask, window=window)
            close(loss, g["loss_" + tag], 1e-6)
            close(Z, g["Z"])
            for i in range(len(Ws)):
                close(dW[i], g[f"grad_{tag}/layers.{i}.apply_mod.linear.weight"])
                close(db[i], g[f"grad_{tag}/layers.{i}.apply_mod.linear.bias"])


def test_mse_criterion(golden):
    """optuna_gae.py:16,21: nn.MSELoss() on the logits of GAE.forward -- value and parameter gradients the reference
    model produced (dropout 0), the dense restatement and the closed form the device path evaluates"""
    g = golden
    Ws, bs = golden_params(g)
    n = int(g["n"])
    adj = O.dense_adjacency(g["src"], g["dst"], n)
    close(O.mse_mean(g["logits_p0"], adj), g["mse_p0"], 1e-6)
    loss, Z, logits, dW, db = O.gae_loss_and_grads(g["src"], g["dst"], n, g["X"], Ws, bs, criterion="mse")
    close(loss, g["mse_p0"], 1e-6)
    for i in range(len(Ws)):
        close(dW[i], g[f"grad_mse_p0/layers.{i}.apply_mod.linear.weight"])
        close(db[i], g[f"grad_mse_p0/layers.{i}.apply_mod.linear.bias"])
    ip, ix = O.csr_from_coo(g["src"], g["dst"], n)
    Zt = torch.tensor(np.asarray(g["Z"]), dtype=torch.float64, requires_grad=True)
    ref = O.mse_mean(O.decoder_logits(Zt), adj.double())
    ref.backward()
    l2, dZ = O.mse_closed_form(Zt.detach(), ip, ix)
    close(l2, ref.detach(), 1e-12)
    close(dZ, Zt.grad, 1e-12)
    close(l2, g["mse_p0"], 1e-5)


def test_degrees_norm(golden):
    g = golden
    deg = O.in_degrees(g["dst"], int(g["n"]))
    assert np.array_equal(deg, g["in_degrees"])
    assert np.array_equal(O.norm_from_in_degrees(deg).unsqueeze(1).numpy(), g["norm"])


def test_csr_structure_exact(golden):
    g = golden
    n = int(g["n"])
    indptr, indices = O.csr_from_coo(g["src"], g["dst"], n)
    assert indptr.dtype == np.int32 and indices.dtype == np.int32
    assert indptr[0] == 0 and indptr[-1] == len(g["src"])
    assert np.array_equal(np.diff(indptr), g["in_degrees"])
    for v in range(n):
        row = indices[indptr[v]:indptr[v + 1]]
        assert np.array_equal(row, np.sort(g["src"][g["dst"] == v]))
    # CSC = CSR of the transpose
    cptr, cidx = O.csc_from_coo(g["src"], g["dst"], n)
    for u in range(n):
        assert np.array_equal(cidx[cptr[u]:cptr[u + 1]], np.sort(g["dst"][g["src"] == u]))


def test_spmm_loops_vs_vectorised(golden):
    g = golden
    if int(g["n"]) > 64:
        return
    indptr, indices = O.csr_from_coo(g["src"], g["dst"], int(g["n"]))
    norm = g["norm"].ravel()
    a = O.spmm_csr(indptr, indices, torch.from_numpy(g["X"]).double(), norm, norm)
    b = O.spmm_csr_loops(indptr, indices, g["X"], norm, norm)
    close(a, b, 1e-12)


def test_batch_matches_stub_dgl_batch():
    parts = load_golden("mol8_parts")
    whole = load_golden("mol8")
    graphs = [(int(parts[f"g{i}/n"]), parts[f"g{i}/src"], parts[f"g{i}/dst"], parts[f"g{i}/X"])
              for i in range(int(parts["n_graphs"]))]
    N, src, dst, X, gptr = O.batch_graphs(graphs)
    assert N == int(whole["n"])
    assert np.array_equal(src, whole["src"]) and np.array_equal(dst, whole["dst"])
    assert np.array_equal(X.numpy(), whole["X"])
    assert gptr[-1] == N and len(gptr) == len(graphs) + 1


def test_tiny_semantics():
    """A1-A4: in-edge sum, no implicit self loops, duplicates add, zero rows."""
    g = load_golden("tiny")
    indptr, indices = O.csr_from_coo(g["src"], g["dst"], 6)
    M = O.spmm_csr(indptr, indices, g["X"]).numpy()
    X = g["X"]
    np.testing.assert_allclose(M[1], 2 * X[0], rtol=1e-6)          # duplicate edge 0->1 twice
    np.testing.assert_allclose(M[5], 0 * X[0])                     # zero in-degree
    np.testing.assert_allclose(M[2], X[2] + X[4], rtol=1e-6)       # self loop present in data
    np.testing.assert_allclose(M[4], X[3], rtol=1e-6)              # directed 3->4 only
    assert g["adj"][1, 0] == 2.0 and g["adj"][4, 3] == 1.0 and g["adj"][3, 4] == 0.0


def test_adam3_losses(golden):
    """three Adam steps of the CPU restatement reproduce the reference's losses"""
    g = golden
    Ws, bs = golden_params(g)
    n = int(g["n"])
    indptr, indices = O.csr_from_coo(g["src"], g["dst"], n)
    Wt = [torch.tensor(w, requires_grad=True) for w in Ws]
    bt = [torch.tensor(b, requires_grad=True) for b in bs]
    # parameter order of model.parameters(): w0,b0,w1,b1,...
    params = [p for pair in zip(Wt, bt) for p in pair]
    opt = torch.optim.Adam(params, lr=1e-2)
    adj = O.dense_adjacency(g["src"], g["dst"], n)
    pw = O.pos_weight_of(adj)
    losses = []
    for _ in range(3):
        Z = O.gae_encode(indptr, indices, g["X"], Wt, bt)
        loss = O.bce_with_logits_mean(O.decoder_logits(Z), adj, pw)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g["adam3_losses"], rtol=2e-5)


def test_c_oracle_matches_python(golden):
    from oracle import c_oracle as C
    g = golden
    Ws, bs = golden_params(g)
    indptr, indices = O.csr_from_coo(g["src"], g["dst"], int(g["n"]))
    norm = g["norm"].ravel()
    close(C.spmm_csr(indptr, indices, g["X"]), O.spmm_csr(indptr, indices, g["X"]), 1e-6)
    close(C.spmm_csr(indptr, indices, g["X"], norm, norm), O.spmm_csr(indptr, indices, g["X"], norm, norm), 1e-6)
    # full encoder through the C pieces == reference Z
    h = g["X"]
    acts = O.activation_rule(len(Ws))
    for W, b, a in zip(Ws, bs, acts):
        h = C.linear(C.spmm_csr(indptr, indices, h), W, b, a == "relu")
    close(h, g["Z"])


def test_bce_row_window_matches_dense_oracle():
    """the row-window form of the loss (used at sizes where the N x N label does not fit) == the dense oracle on a
    directed multigraph: loss shares add up to the loss, gradient rows equal autograd's"""
    rng = np.random.default_rng(7)
    n, d = 60, 5
    src = rng.integers(0, n, 300); dst = rng.integers(0, n, 300)
    src[:3] = src[3:6]; dst[:3] = dst[3:6]                       # duplicate edges
    Zt = torch.tensor(rng.standard_normal((n, d)) * 0.7, dtype=torch.float64, requires_grad=True)
    adj = O.dense_adjacency(src, dst, n, dtype=torch.float64)
    pw = O.pos_weight_of(adj)
    loss = O.bce_with_logits_mean(O.decoder_logits(Zt), adj, pw)
    loss.backward()
    ip, ix = O.csr_from_coo(src, dst, n)
    tp, tx = O.csc_from_coo(src, dst, n)
    total = 0.0
    for r0, r1 in ((0, 17), (17, 40), (40, 60)):
        share, g = O.bce_row_window(Zt.detach(), r0, r1, ip, ix, tp, tx, float(pw))
        total += float(share)
        close(g, Zt.grad[r0:r1], 1e-10)
    assert abs(total - float(loss)) < 1e-12


def test_segment_readout_on_golden_molecules():
    """README.md:54 readout (mean | sum | max per molecule) on the 8-molecule golden batch: the oracle's loop equals
    an independent torch restatement per member graph"""
    parts = load_golden("mol8_parts")
    whole = load_golden("mol8")
    sizes = [int(parts[f"g{i}/n"]) for i in range(int(parts["n_graphs"]))]
    gp = np.concatenate([[0], np.cumsum(sizes)])
    Z = whole["Z"]
    got = O.segment_readout(Z, gp)
    assert got.shape == (len(sizes), 3 * Z.shape[1])
    for g, zs in enumerate(torch.split(torch.tensor(Z, dtype=torch.float64), sizes)):
        ref = torch.cat([zs.mean(0), zs.sum(0), zs.max(0).values]).numpy()
        np.testing.assert_allclose(got[g], ref, rtol=1e-12, atol=1e-12)
    empty = O.segment_readout(Z, np.array([0, 0, len(Z)]))
    assert (empty[0] == 0).all() and np.allclose(empty[1, :Z.shape[1]], Z.mean(0))
