"""link-prediction utilities: edge split invariants, AUC / AP against scikit-learn"""
import numpy as np
import torch

from gae_dgl_amd import metrics as M


def test_split_edges_invariants():
    rng = np.random.default_rng(0)
    n = 300
    a = rng.integers(0, n, 2000); b = rng.integers(0, n, 2000)
    src, dst = np.concatenate([a, b]), np.concatenate([b, a])
    (ts, td), val, test = M.split_edges(src, dst, n, 0.05, 0.1, seed=1)
    key = lambda s, d: set((np.minimum(s, d) * n + np.maximum(s, d)).tolist())
    all_pairs = key(src[src != dst], dst[src != dst])
    tr, va, te = key(ts, td), key(*val["pos"]), key(*test["pos"])
    assert tr | va | te == all_pairs and not (tr & va) and not (tr & te) and not (va & te)
    assert set(zip(ts.tolist(), td.tolist())) == set(zip(td.tolist(), ts.tolist()))      # symmetric train graph
    for sp in (val, test):
        neg = key(*sp["neg"])
        assert not (neg & all_pairs) and sp["neg"].shape == sp["pos"].shape and np.all(sp["neg"][0] != sp["neg"][1])


def test_auc_ap_match_sklearn():
    from sklearn.metrics import average_precision_score, roc_auc_score
    rng = np.random.default_rng(2)
    for ties in (False, True):
        pos = rng.normal(0.6, 1.0, 500); neg = rng.normal(0.0, 1.0, 700)
        if ties:
            pos, neg = np.round(pos, 1), np.round(neg, 1)
        y = np.concatenate([np.ones_like(pos), np.zeros_like(neg)]); s = np.concatenate([pos, neg])
        assert abs(M.roc_auc(torch.tensor(pos), torch.tensor(neg)) - roc_auc_score(y, s)) < 1e-12
        assert abs(M.average_precision(torch.tensor(pos), torch.tensor(neg)) - average_precision_score(y, s)) < 1e-12
