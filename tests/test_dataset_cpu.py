"""Host-side logic of the inductive data path: the featuriser output contract (gae_dgl/prepare_data.py) and the
reference-written checkpoint fixture.  No GPU needed."""
import os

import numpy as np
import torch

from conftest import GOLDEN, load_golden


def test_featuriser_contract_accepts_zinc_like_and_golden_molecules():
    from gae_dgl_amd import workloads as W
    from gae_dgl_amd.dataset import featuriser_contract_errors
    gp, src, dst, X = W.zinc_like(500, seed=2)
    assert featuriser_contract_errors(gp, src, dst, X) == []
    parts = load_golden("mol8_parts")             # molecules of the golden generator (prepare_data.py layout)
    sizes = [int(parts[f"g{i}/n"]) for i in range(int(parts["n_graphs"]))]
    gp = np.concatenate([[0], np.cumsum(sizes)])
    src = np.concatenate([parts[f"g{i}/src"] + gp[i] for i in range(len(sizes))])
    dst = np.concatenate([parts[f"g{i}/dst"] + gp[i] for i in range(len(sizes))])
    X = np.concatenate([parts[f"g{i}/X"] for i in range(len(sizes))])
    assert featuriser_contract_errors(gp, src, dst, X) == []


def test_featuriser_contract_rejects_violations():
    from gae_dgl_amd import workloads as W
    from gae_dgl_amd.dataset import featuriser_contract_errors as check
    gp, src, dst, X = W.zinc_like(50, seed=1)
    assert any("width" in e for e in check(gp, src, dst, X[:, :38]))
    bad = X.copy(); bad[3, 0:23] = 0                                   # element block without a 1
    assert any("[0, 23)" in e for e in check(gp, src, dst, bad))
    bad = X.copy(); bad[5, 23:29] = 1                                  # degree block with six 1s
    assert any("[23, 29)" in e for e in check(gp, src, dst, bad))
    bad = X.copy(); bad[0, 38] = 0.5
    assert any("0 / 1" in e for e in check(gp, src, dst, bad))
    assert any("self loops" in e for e in check(gp, np.append(src, [4, 4]), np.append(dst, [4, 4]), X))
    cross_s = np.append(src, [0, gp[1]]); cross_d = np.append(dst, [gp[1], 0])
    assert any("different member graphs" in e for e in check(gp, cross_s, cross_d, X))
    assert any("pairs" in e for e in check(gp, src[:-1], dst[:-1], X))   # one direction of a bond missing
    assert any("outside" in e for e in check(gp, np.append(src, [10 ** 9, 0]), np.append(dst, [0, 10 ** 9]), X))
    assert check(np.array([0, 5, 3]), src, dst, X) != []


def test_reference_checkpoint_file_fixture():
    """tests/golden/mol8_ep00.pkl was written by the reference's own GAE through torch.save(model.state_dict())
    (train_inductive.py:55-57, make_golden.py): keys / shapes / values are those of the golden state dict"""
    sd = torch.load(os.path.join(GOLDEN, "mol8_ep00.pkl"))
    g = load_golden("mol8")
    assert list(sd) == ["layers.0.apply_mod.linear.weight", "layers.0.apply_mod.linear.bias",
                        "layers.1.apply_mod.linear.weight", "layers.1.apply_mod.linear.bias"]
    for k, v in sd.items():
        assert v.dtype == torch.float32 and np.array_equal(v.numpy(), g["sd/" + k])


def test_citation_graph_degree_profiles():
    """workloads.citation_graph: the published N / E / F either way; "planetoid" has the real graphs' hubs (longest row
    close to the published maximum, a heavy tail of rows beyond the 16-slot packed table), "uniform" has none; both are
    symmetric, without self-loops, and reproducible from the seed"""
    import numpy as np
    from gae_dgl_amd import workloads as W
    for name, (n, e, f) in W.CITATION.items():
        for degrees in ("uniform", "planetoid"):
            n2, src, dst, X = W.citation_graph(name, seed=0, degrees=degrees)
            assert n2 == n and src.size == e == dst.size and X.shape == (n, f)
            assert (src != dst).all()
            half = e // 2
            assert np.array_equal(src[:half], dst[half:2 * half]) and np.array_equal(dst[:half], src[half:2 * half])
            deg = np.bincount(dst, minlength=n)
            if degrees == "uniform":
                assert deg.max() <= 24
            else:
                dmax = W.PLANETOID_MAX_DEGREE[name]
                assert 0.85 * dmax <= deg.max() <= dmax + 8
                assert 0.005 * n < (deg > 16).sum() < 0.06 * n and (deg == 1).sum() > 0.25 * n
            assert np.array_equal(X, W.citation_graph(name, seed=0)[3])          # the same features for both profiles
            again = W.citation_graph(name, seed=0, degrees=degrees)
            assert np.array_equal(again[1], src) and np.array_equal(again[2], dst) and np.array_equal(again[3], X)
    import pytest
    with pytest.raises(ValueError):
        W.citation_graph("cora", degrees="zipf")


def _write_planetoid(root, name, n_all, test_ids_in_file_order, feats, adjacency_lists):
    """files in the Planetoid layout (python-2 style pickles of scipy CSR blocks + a dict of neighbour lists + the
    test index as text): rows 0 .. n_all-1 of ``feats`` go to allx, the rows of the listed test ids to tx in FILE order"""
    import pickle
    from collections import defaultdict
    import scipy.sparse as sp
    os.makedirs(root, exist_ok=True)
    allx = sp.csr_matrix(feats[:n_all])
    tx = sp.csr_matrix(feats[np.asarray(test_ids_in_file_order)])
    g = defaultdict(list)
    for u, nbrs in adjacency_lists.items():
        g[u] = list(nbrs)
    for ext, obj in (("allx", allx), ("tx", tx), ("x", allx[:10]), ("graph", g)):
        with open(os.path.join(root, f"ind.{name}.{ext}"), "wb") as f:
            pickle.dump(obj, f, protocol=2)
    with open(os.path.join(root, f"ind.{name}.test.index"), "w") as f:
        f.write("\n".join(str(i) for i in test_ids_in_file_order) + "\n")


def test_planetoid_files_load_as_the_reference_loaders_read_them(tmp_path):
    """data.load_planetoid against (i) networkx's own ``DiGraph(from_dict_of_lists(graph))`` -- what DGL 0.4's citation
    loader hands to ``DGLGraph(data.graph)`` (train_transductive.py:45) -- and (ii) the feature assembly of Kipf's
    gcn/utils.py written out with scipy (vstack, extended tx for missing test ids, in-place permutation, row
    normalisation): Citeseer-style isolated test ids, repeated / mirrored neighbour entries, self-loops"""
    import argparse
    import networkx as nx
    import scipy.sparse as sp
    from gae_dgl_amd import data as D
    rng = np.random.default_rng(5)
    n_all, n, f = 60, 83, 37
    missing = {64, 71, 80}                                         # test ids without a row in tx (isolated nodes)
    test_ids = np.asarray([i for i in range(n_all, n) if i not in missing])
    rng.shuffle(test_ids)
    feats = np.where(rng.random((n, f)) < 0.15, rng.integers(1, 4, (n, f)), 0).astype(np.float32)
    feats[5] = 0                                                   # a node without words
    feats[list(missing)] = 0
    adj = {}
    for u in range(n):
        if u in missing or u == 17:
            adj[u] = []
            continue
        adj[u] = [int(v) for v in rng.integers(0, n, rng.integers(1, 6)) if int(v) not in missing]
    adj[3] += [3, 9, 9]                                             # self-loop, repeated entry
    adj[9] += [3]                                                   # mirrored entry
    adj[40] = [int(v) for v in range(41, 60)]                       # a hub
    root = str(tmp_path / "citeseer")
    _write_planetoid(root, "citeseer", n_all, test_ids.tolist(), feats, adj)
    assert D.planetoid_dir(str(tmp_path), "citeseer") == root and D.planetoid_dir(str(tmp_path), "cora") is None
    n2, src, dst, X = D.load_planetoid(root, "citeseer")
    # (i) the graph
    ref = nx.DiGraph(nx.from_dict_of_lists(adj))
    assert n2 == n == ref.number_of_nodes()
    assert sorted(zip(src.tolist(), dst.tolist())) == sorted(ref.edges())
    assert len(src) == ref.number_of_edges() and (3, 3) in set(zip(src.tolist(), dst.tolist()))
    # (ii) the features
    allx, tx = sp.csr_matrix(feats[:n_all]), sp.csr_matrix(feats[test_ids])
    order = np.sort(test_ids)
    full = range(int(test_ids.min()), int(test_ids.max()) + 1)
    tx_ext = sp.lil_matrix((len(full), f))
    tx_ext[order - order.min(), :] = tx
    fm = sp.vstack((allx, tx_ext)).tolil()
    fm[test_ids, :] = fm[order, :]
    fm = np.asarray(fm.todense(), dtype=np.float64)
    rs = fm.sum(1)
    want = np.where(rs[:, None] != 0, fm / np.where(rs != 0, rs, 1.0)[:, None], 0.0)
    assert X.shape == (n, f) and X.dtype == np.float32
    np.testing.assert_allclose(X, want, rtol=1e-6, atol=0)
    assert not X[5].any() and not X[list(missing)].any()
    np.testing.assert_allclose(X.sum(1)[rs != 0], 1.0, rtol=1e-5)
    # through the scripts' entry point (root/<name>/ind.<name>.*), next to the npz route and the synthetic fallback
    got = D.load_data(argparse.Namespace(dataset="citeseer", data_root=str(tmp_path)))
    assert not got.synthetic and got.graph.number_of_nodes() == n and np.array_equal(got.features, X)
    assert np.array_equal(got.graph.src, src) and np.array_equal(got.graph.dst, dst)
    assert D.load_data(argparse.Namespace(dataset="cora", data_root=str(tmp_path))).synthetic
