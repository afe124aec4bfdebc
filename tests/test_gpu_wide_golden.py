"""The layer order the build runs BY DEFAULT on layers that narrow wide features -- act(A (H W^T) + b) through
gae_xw_fwd / gae_spmm_csr_epilogue / gae_xw_wgrad -- held directly to vectors the reference's own gae.py produced in
its own order act((A H) W^T + b) (gae.py:26-31): tests/golden/wide300.npz (n = 384, f_in = 300, symmetric graph) and
wide2k.npz (n = 256, f_in = 2003, directed multigraph with duplicate edges, self-loops and nodes without in-edges),
made by tests/golden/make_golden.py.  Every test asserts that the one-pass kernels really ran."""
import numpy as np
import pytest
import torch

from conftest import WIDE_CASES, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def rel(a, b):
    a = a.detach().double().cpu(); b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1.0))


def build(g):
    import gae_dgl_amd as G
    model = G.GAE(g["X"].shape[1], [int(h) for h in g["hidden"]])          # default arguments: transform_first "auto"
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")})
    return model.to(DEV)


def graph_of(g):
    import gae_dgl_amd as G
    gr = G.DGLGraph()
    gr.add_nodes(int(g["n"]))
    gr.add_edges(g["src"], g["dst"])
    gr.to(DEV)
    gr.ndata['h'] = torch.from_numpy(g["X"]).to(DEV)
    return gr


@pytest.mark.parametrize("name", WIDE_CASES)
def test_default_path_is_the_one_pass_layer_and_matches_the_reference(name):
    from gae_dgl_amd import ops
    g = load_golden(name)
    model = build(g)
    before = dict(ops.STATS)
    Z = model.encode(graph_of(g))
    assert ops.STATS["xw_fwd"] == before["xw_fwd"] + 1, "layer 1 did not run through gae_xw_fwd"
    assert rel(Z, g["Z"]) < TOL
    model.decoder.dropout = 0.0
    gr = graph_of(g)
    assert rel(model(gr), g["logits_p0"]) < TOL
    assert rel(gr.ndata['h'], g["Z"]) < TOL
    model.decoder.dropout = 0.1
    model.decoder.mask = torch.from_numpy(g["mask"]).to(DEV)
    assert rel(model(graph_of(g)), g["logits_p01"]) < TOL


@pytest.mark.parametrize("name", WIDE_CASES)
@pytest.mark.parametrize("tag", ["p0", "p01"])
def test_default_path_loss_and_gradients(name, tag):
    """fused loss + backward through gae_spmm_csr_epilogue (ReLU gate) and gae_xw_wgrad against the reference's
    autograd gradients"""
    from gae_dgl_amd import ops
    g = load_golden(name)
    model = build(g)
    model.decoder.dropout = 0.0 if tag == "p0" else 0.1
    model.decoder.mask = None if tag == "p0" else torch.from_numpy(g["mask"]).to(DEV)
    before = dict(ops.STATS)
    loss = model.reconstruction_loss(graph_of(g))
    loss.backward()
    assert ops.STATS["xw_fwd"] == before["xw_fwd"] + 1 and ops.STATS["xw_wgrad"] == before["xw_wgrad"] + 1
    assert abs(float(loss) - float(g["loss_" + tag])) < TOL * max(1.0, abs(float(g["loss_" + tag])))
    for k, p in model.named_parameters():
        assert rel(p.grad, g[f"grad_{tag}/{k}"]) < 5 * TOL, k


@pytest.mark.parametrize("name", WIDE_CASES)
@pytest.mark.parametrize("captured", [False, True])
def test_default_path_three_adam_steps(name, captured):
    """the reference's three Adam steps (lr 1e-2, dropout 0): eager steps and the captured HIP graph of the step"""
    from gae_dgl_amd import capture, ops, optim
    g = load_golden(name)
    model = build(g)
    model.decoder.dropout = 0.0
    opt = optim.Adam(model.parameters(), lr=1e-2)
    gr = graph_of(g)
    x = gr.ndata['h']
    before = dict(ops.STATS)
    losses = []
    if captured:
        step = capture.CapturedTrainStep(model, opt, gr, x, warmup=0)
        for _ in range(3):
            losses.append(float(step()))
    else:
        for _ in range(3):
            gr.ndata['h'] = x
            loss = model.reconstruction_loss(gr)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(float(loss.detach()))
    assert ops.STATS["xw_fwd"] > before["xw_fwd"] and ops.STATS["xw_wgrad"] > before["xw_wgrad"]
    np.testing.assert_allclose(losses, g["adam3_losses"], rtol=5e-5)
    for k, v in model.state_dict().items():
        assert rel(v, g["sd_after3/" + k]) < 1e-4, k
