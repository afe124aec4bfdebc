"""Link-prediction evaluation the reference only alludes to (``# TODO: train test
split`` at gae_dgl/train_transductive.py:35, Kipf & Welling's AUC / AP protocol):
edge split into train / validation / test positives with sampled negatives,
ROC-AUC and average precision of ``sigmoid(z_i . z_j)``.

Evaluation utilities, not part of the HIP hot path: index bookkeeping and
rank statistics through torch ops on whatever device the tensors live on."""
import numpy as np
import torch


def split_edges(src, dst, n, val_frac=0.05, test_frac=0.10, seed=0):
    """Undirected split (both directions of a pair stay together).  Returns
    ``train (src, dst)`` with both directions, and ``val`` / ``test`` dicts with
    ``pos`` and ``neg`` int64 arrays of shape [2, k] (one direction per pair;
    negatives are sampled non-edges, no self pairs)."""
    src = np.asarray(src, dtype=np.int64); dst = np.asarray(dst, dtype=np.int64)
    rng = np.random.default_rng(seed)
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    keep = lo != hi
    pairs = np.unique(np.stack([lo[keep], hi[keep]], 1), axis=0)
    perm = rng.permutation(len(pairs))
    n_val, n_test = int(len(pairs) * val_frac), int(len(pairs) * test_frac)
    val_p, test_p, train_p = pairs[perm[:n_val]], pairs[perm[n_val:n_val + n_test]], pairs[perm[n_val + n_test:]]
    edge_keys = set((pairs[:, 0] * n + pairs[:, 1]).tolist())

    def negatives(k):
        out = []
        while len(out) < k:
            a = rng.integers(0, n, 2 * (k - len(out)) + 8); b = rng.integers(0, n, a.size)
            for x, y in zip(np.minimum(a, b).tolist(), np.maximum(a, b).tolist()):
                if x != y and (x * n + y) not in edge_keys:
                    edge_keys.add(x * n + y)
                    out.append((x, y))
                    if len(out) == k:
                        break
        return np.asarray(out, dtype=np.int64).reshape(-1, 2).T

    train = (np.concatenate([train_p[:, 0], train_p[:, 1]]), np.concatenate([train_p[:, 1], train_p[:, 0]]))
    return train, {"pos": val_p.T.copy(), "neg": negatives(n_val)}, {"pos": test_p.T.copy(), "neg": negatives(n_test)}


def edge_scores(Z, pairs):
    """sigmoid(z_i . z_j) for pairs [2, k] (the decoder of gae.py:69-72 restricted to the listed pairs)"""
    pairs = torch.as_tensor(pairs, device=Z.device)
    return torch.sigmoid((Z[pairs[0]] * Z[pairs[1]]).sum(1))


def roc_auc(pos_scores, neg_scores):
    """area under the ROC curve = P(score_pos > score_neg) + 0.5 P(equal), via average ranks"""
    s = torch.cat([pos_scores, neg_scores]).double()
    n_pos, n_neg = pos_scores.numel(), neg_scores.numel()
    order = torch.argsort(s)
    sv = s[order]
    ranks = torch.arange(1, s.numel() + 1, device=s.device, dtype=torch.float64)
    # average ranks over ties
    uniq, inv, counts = torch.unique_consecutive(sv, return_inverse=True, return_counts=True)
    ends = torch.cumsum(counts, 0).double()
    avg = ends - (counts.double() - 1) / 2
    r = torch.empty_like(ranks)
    r[order] = avg[inv]
    return float((r[:n_pos].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg))


def average_precision(pos_scores, neg_scores):
    """sum_k (R_k - R_{k-1}) P_k over distinct thresholds (sklearn's definition)"""
    s = torch.cat([pos_scores, neg_scores]).double()
    y = torch.cat([torch.ones_like(pos_scores), torch.zeros_like(neg_scores)]).double()
    order = torch.argsort(s, descending=True)
    s, y = s[order], y[order]
    tp = torch.cumsum(y, 0)
    k = torch.arange(1, s.numel() + 1, device=s.device, dtype=torch.float64)
    last = torch.ones_like(y, dtype=torch.bool)
    last[:-1] = s[1:] != s[:-1]                       # evaluate only at the end of each tie group
    prec, rec = (tp / k)[last], (tp / y.sum())[last]
    rec_prev = torch.cat([torch.zeros(1, device=s.device, dtype=torch.float64), rec[:-1]])
    return float(((rec - rec_prev) * prec).sum())


def evaluate(Z, split):
    p, q = edge_scores(Z, split["pos"]), edge_scores(Z, split["neg"])
    return {"auc": roc_auc(p, q), "ap": average_precision(p, q)}
