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
            again = W.citation_graph(name, seed=0, degrees=degrees)
            assert np.array_equal(again[1], src) and np.array_equal(again[2], dst) and np.array_equal(again[3], X)
    import pytest
    with pytest.raises(ValueError):
        W.citation_graph("cora", degrees="zipf")
