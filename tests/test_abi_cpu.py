"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a
GPU and exports every symbol include/gae_hip.h declares; the ctypes table
mirrors the header; the host mirror keeps the reference's API surface."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gae_hip.h")                      # the boundary a maintainer binds
SEAMS = os.path.join(ROOT, "include", "gae_hip_experimental.h")          # the library's own seams (its host mirror only)


def _symbols_of(path):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gae_[a-z0-9_]+)\s*\(", src)))


def declared_symbols():
    return sorted(set(_symbols_of(HEADER)) | set(_symbols_of(SEAMS)))


def test_the_boundary_header_stays_small():
    """include/gae_hip.h is what a maintainer binds -- at most 45 entry points, none of them a gae_x_*
    seam; the seams live in gae_hip_experimental.h, and no symbol is declared twice"""
    core, seams = _symbols_of(HEADER), _symbols_of(SEAMS)
    assert len(core) <= 45, len(core)
    assert not any(s.startswith("gae_x_") for s in core)
    assert not set(core) & set(seams)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    ask = text[:text.index("### Seams of the library's own host mirror")]      # the part a maintainer follows
    for s in seams:
        if s.startswith("gae_x_"):
            assert s not in ask, f"INTEGRATION.md asks a maintainer to bind the seam {s}"


def test_header_declares_entry_points():
    syms = declared_symbols()
    for must in ["gae_spmm_csr", "gae_csr_from_coo", "gae_linear_fwd", "gae_linear_bwd", "gae_decoder_dense",
                 "gae_last_error", "gae_version"]:
        assert must in syms


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from gae_dgl_amd import _lib
    lib = _lib.load()
    for s in declared_symbols():
        assert hasattr(lib, s), f"libgae_hip.so does not export {s}"
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert lib.gae_version() >= 100
    assert isinstance(lib.gae_last_error(), bytes)


def test_argument_errors_do_not_need_a_gpu():
    from gae_dgl_amd import _lib
    lib = _lib.load()
    # negative sizes / NULL pointers are rejected before any launch
    rc = lib.gae_spmm_csr(None, None, -1, 0, None, 0, None, 0, 0, 0, None, None, None, None, 0, 0, None)
    assert rc == -2 and b"negative" in lib.gae_last_error()
    rc = lib.gae_spmm_csr(None, None, 4, 4, None, 8, None, 8, 8, 7, None, None, None, None, 0, 0, None)
    assert rc == -4
    assert lib.gae_spmm_workspace_bytes(None, 32) == 0
    assert lib.gae_spmm_plan_sizes(None, 8, 0, 256, 512, None, None, 0, None) == -6
    rc = lib.gae_linear_fwd(None, 4, 4, 4, None, None, 4, 9, None, 4, None, 0, None)
    assert rc == -4
    rc = lib.gae_dropout_mask(None, 8, ctypes.c_float(1.5), 0, 0, None, None)
    assert rc == -6
    assert lib.gae_linear_bwd_workspace_bytes(1000, 39, 32) > 0


def test_product_path_has_no_cpu_fallback():
    import gae_dgl_amd as G
    from gae_dgl_amd._lib import GaeHipError
    g = G.DGLGraph()
    g.add_nodes(3)
    g.add_edges([0, 1], [1, 2])
    g.ndata['h'] = torch.ones(3, 4)
    with pytest.raises(GaeHipError):
        g.update_all(G.gcn_msg, G.gcn_reduce)
    with pytest.raises(GaeHipError):
        g.in_degrees()


def test_module_api_and_state_dict_keys():
    import gae_dgl_amd as G
    m = G.GAE(39, [32, 16])
    assert list(m.state_dict().keys()) == [
        "layers.0.apply_mod.linear.weight", "layers.0.apply_mod.linear.bias",
        "layers.1.apply_mod.linear.weight", "layers.1.apply_mod.linear.bias"]
    assert m.layers[0].apply_mod.linear.weight.shape == (32, 39)
    assert sum(p.nelement() for p in m.parameters()) == 1808
    assert m.decoder.dropout == 0.1
    from gae_dgl_amd.gae import _act_code, identity
    import torch.nn.functional as F
    acts = [_act_code(l.apply_mod.activation) for l in G.GAE(5, [4, 3, 2]).layers]
    assert acts == [1, 1, 0]
    assert [_act_code(l.apply_mod.activation) for l in G.GAE(5, [4]).layers] == [0]
    assert _act_code(lambda x: x) == 0 and _act_code(F.relu) == 1 and _act_code(torch.tanh) is None
    assert G.gcn_msg.src == 'h' and G.gcn_msg.out == 'm' and G.gcn_reduce.msg == 'm' and G.gcn_reduce.out == 'h'
    assert G.InnerProductDecoder().activation is torch.sigmoid


def test_graph_host_logic_and_batch():
    import numpy as np
    import gae_dgl_amd as G
    from conftest import load_golden
    parts = load_golden("mol8_parts"); whole = load_golden("mol8")
    gs = []
    for i in range(int(parts["n_graphs"])):
        g = G.DGLGraph()
        g.add_nodes(int(parts[f"g{i}/n"]))
        g.add_edges(parts[f"g{i}/src"], parts[f"g{i}/dst"])
        g.ndata['h'] = torch.from_numpy(parts[f"g{i}/X"])
        gs.append(g)
    bg = G.batch(gs)
    assert bg.number_of_nodes() == int(whole["n"]) and bg.number_of_edges() == len(whole["src"])
    s, d = bg.edges()
    assert np.array_equal(s.numpy(), whole["src"]) and np.array_equal(d.numpy(), whole["dst"])
    assert np.array_equal(bg.ndata['h'].numpy(), whole["X"])
    assert np.array_equal(bg.adjacency_matrix().to_dense().numpy(), whole["adj"])
    with pytest.raises(ValueError):
        gs[0].add_edges([0], [10 ** 6])
    bg.set_n_initializer(G.init.zero_initializer); bg.set_e_initializer(G.init.zero_initializer)


def test_row_layout_helpers_are_host_logic():
    """pad_rows / row_quantum decide the HBM layout (16-byte vectors, whole 128-byte lines from 512 B per row on)"""
    from gae_dgl_amd import ops
    assert ops.row_quantum(39, torch.float32) == 4 and ops.row_quantum(127, torch.float32) == 4
    assert ops.row_quantum(128, torch.float32) == 32 and ops.row_quantum(3703, torch.float32) == 32
    assert ops.row_quantum(255, torch.bfloat16) == 8 and ops.row_quantum(256, torch.bfloat16) == 64
    assert ops.padded_ld(500, torch.float32) == 544 and ops.padded_ld(1433, torch.float32) == 1440    # odd line counts
    assert ops.padded_ld(39, torch.float32) == 40 and ops.padded_ld(256, torch.bfloat16) == 320
    already = torch.zeros(3, 512)[:, :500]
    assert ops.pad_rows(already) is already or already.data_ptr() % 128      # whole-line rows are taken as they are
    for f, ld in ((39, 40), (500, 544), (1433, 1440), (3703, 3744), (32, 32)):
        x = torch.arange(3 * f, dtype=torch.float32).reshape(3, f)
        y = ops.pad_rows(x)
        assert y.shape == (3, f) and y.stride() == (ld, 1) and torch.equal(y, x)
        if y.data_ptr() % 128 == 0:                                  # (host allocations are only 64-byte aligned)
            assert ops.pad_rows(y) is y                              # already laid out: no copy
        assert y.storage_offset() == 0 and float(y._base[:, f:].abs().sum()) == 0.0 if ld != f else True


def test_atb_and_loss_plans_are_consistent_without_a_gpu():
    """workspace queries are pure host functions of the shapes (and grow monotonically enough to be cached)"""
    from gae_dgl_amd import _lib
    lib = _lib.load()
    assert lib.gae_linear_fwd_workspace_bytes(19717, 500, 32) == 0          # enough row tiles: no split-K
    assert lib.gae_linear_fwd_workspace_bytes(2708, 1433, 32) > 0            # Cora: split along f_in
    assert lib.gae_linear_bwd_workspace_bytes(19717, 500, 32) >= 32 * 500 * 4
    small, big = lib.gae_decoder_bce_workspace_bytes(2708, 2708, 16), lib.gae_decoder_bce_workspace_bytes(19717, 19717, 16)
    assert 0 < small < big
    assert lib.gae_decoder_bce_workspace_bytes(19717, 19717, 16) > 19717 * 19717 // 8   # symmetric path: strip buffer (256-row panels: N^2 / 8 bytes)
    assert lib.gae_decoder_bce_workspace_bytes(100, 200, 16) < 0            # n_local > n is an argument error


def test_tuning_knobs_round_trip_and_are_process_wide():
    """gae_tuning_set / gae_tuning_get: one value per process -- a thread other than the one that set a knob sees
    it too (PyTorch launches the backward kernels from its autograd worker thread: a per-thread knob set from Python
    never reached them)"""
    import ctypes
    import threading
    from gae_dgl_amd import _lib
    lib = _lib.load()

    def get(name):
        v = ctypes.c_int64(-1)
        assert lib.gae_tuning_get(name, ctypes.byref(v)) == 0
        return v.value
    default = get(b"spmm_tile_vecs")
    assert lib.gae_tuning_set(b"spmm_tile_vecs", 8) == 0 and get(b"spmm_tile_vecs") == 8
    seen = []
    t = threading.Thread(target=lambda: seen.append(get(b"spmm_tile_vecs")))
    t.start(); t.join()
    assert seen == [8]
    assert lib.gae_tuning_set(b"spmm_tile_vecs", default) == 0
    assert lib.gae_tuning_get(b"no_such_knob", ctypes.byref(ctypes.c_int64())) != 0
    assert lib.gae_tuning_set(b"no_such_knob", 1) != 0 and b"unknown knob" in lib.gae_last_error()
