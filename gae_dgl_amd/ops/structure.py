"""Graph structure on the device: CSR from COO, degrees / norms, row packing, on-device dgl.batch (select / plan /
gather), per-graph readout.

Part of the package gae_dgl_amd.ops (one module until round 6).  Functions look each other up in the PACKAGE
namespace (`_ops.<name>`) when they run: setting a flag or replacing a function on `gae_dgl_amd.ops` reaches every caller."""
import torch

import gae_dgl_amd.ops as _ops
from .. import _lib
from .._lib import F32, GaeHipError
from ._base import _dtype_code, _f32, _gpu, _on_device, _ptr, _rowmajor, _stream, _workspace

__all__ = [
    'csr_from_coo', 'degree_norm', 'rows_pack', 'csr_to_dense', 'batch_plan', 'batch_select', 'batch_plan_next',
    'batch_feature_ld', 'batch_gather', 'batch_gather_next', 'segment_readout',
]


# ------------------------------------------------------------------ structure
def csr_from_coo(row, col, n_rows, n_cols, validate=True):
    """CSR (int32 indptr, int32 indices) with rows ascending, columns ascending
    inside a row, duplicates kept.  ``row``/``col`` are int64 device tensors."""
    _gpu(row, "row"); _gpu(col, "col")
    row = row.to(torch.int64).contiguous(); col = col.to(torch.int64).contiguous()
    E = row.numel()
    if col.numel() != E:
        raise GaeHipError("csr_from_coo: row/col length mismatch")
    dev = row.device
    with _on_device(dev):
        indptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
        indices = torch.empty(E, dtype=torch.int32, device=dev)
        nbytes = _lib.load().gae_csr_from_coo_workspace_bytes(E, n_rows)
        if nbytes < 0:
            _lib.check(int(nbytes), "gae_csr_from_coo_workspace_bytes")
        ws = _workspace(nbytes, dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("gae_csr_from_coo", _ptr(row), _ptr(col), E, n_rows, n_cols, _ptr(indptr), _ptr(indices),
                  _ptr(ws), ws.numel(), _ptr(status), _stream())
        if validate and E and int(status.item()) != 0:
            raise GaeHipError("csr_from_coo: edge endpoint outside [0, n_rows) x [0, n_cols)")
    return indptr, indices


def degree_norm(indptr, want_deg=True, want_norm=True):
    _gpu(indptr, "indptr")
    n = indptr.numel() - 1
    dev = indptr.device
    deg = torch.empty(n, dtype=torch.int32, device=dev) if want_deg else None
    norm = torch.empty(n, dtype=torch.float32, device=dev) if want_norm else None
    with _on_device(dev):
        _lib.call("gae_degree_norm", _ptr(indptr), n, _ptr(deg), _ptr(norm), _stream())
    return deg, norm


def rows_pack(H, idx=None, out=None, n_out_rows=None):
    """out[i] = H[idx[i]] (idx None: H[i]); rows behind the packed ones up to ``n_out_rows`` are zeroed
    (gae_rows_pack).  ``out``: a [>= n_out_rows, F] fp32 row-major destination (a slice of an exchange buffer)."""
    H, ldh = _rowmajor(_f32(_gpu(H, "H"), "rows_pack: H"), "H")
    n_src, F = H.shape
    n_rows = n_src if idx is None else int(idx.numel())
    n_out = n_rows if n_out_rows is None else int(n_out_rows)
    if idx is not None:
        idx = _gpu(idx, "idx").to(torch.int64).contiguous()
    if out is None:
        out = torch.empty(n_out, F, dtype=torch.float32, device=H.device)
    if out.dtype != torch.float32 or out.dim() != 2 or out.shape[1] != F or out.shape[0] < n_out or \
            (F and out.stride(1) != 1):
        raise GaeHipError("rows_pack: `out` must be an fp32 [>= n_out_rows, F] row-major tensor")
    ldo = out.stride(0) if out.shape[0] > 1 else max(F, 1)
    with _on_device(H.device):
        _lib.call("gae_rows_pack", _ptr(H), ldh, n_src, _ptr(idx), n_rows, n_out, F, _ptr(out), max(ldo, F, 1), _stream())
    return out[:n_out]


def csr_to_dense(indptr, indices, n_rows, n_cols):
    _gpu(indptr, "indptr")
    out = torch.empty(n_rows, n_cols, dtype=torch.float32, device=indptr.device)
    with _on_device(indptr.device):
        _lib.call("gae_csr_to_dense", _ptr(indptr), _ptr(indices), n_rows, n_cols, _ptr(out), max(n_cols, 1),
                  _stream())
    return out


def batch_plan(graph_ptr, ds_indptr, ds_t_indptr, graph_ids, out=None):
    """exclusive prefix sums of the selected graphs' node / edge / transposed-edge counts, computed on the device
    (gae_batch_plan): (node_ptr, edge_ptr, t_edge_ptr or None), int64 [B + 1] each.  ``ds_t_indptr`` None = the
    dataset is symmetric.  ``out``: an int64 [3 or 2, B + 1] buffer to write into."""
    gids = _gpu(graph_ids, "graph_ids")
    B = gids.numel()
    dev = gids.device
    rows = 3 if ds_t_indptr is not None else 2
    buf = torch.empty(rows, B + 1, dtype=torch.int64, device=dev) if out is None else out
    if buf.shape != (rows, B + 1) or buf.dtype != torch.int64 or not buf.is_contiguous():
        raise GaeHipError("batch_plan: `out` must be a contiguous int64 [rows, B + 1] buffer")
    with _on_device(dev):
        _lib.call("gae_batch_plan", _ptr(graph_ptr), _ptr(ds_indptr), _ptr(ds_t_indptr), _ptr(gids), B, _ptr(buf[0]),
                  _ptr(buf[1]), _ptr(buf[2]) if ds_t_indptr is not None else None, _stream())
    return buf[0], buf[1], (buf[2] if ds_t_indptr is not None else None)


def batch_select(order, cursor, batch_graphs, out_ids):
    """out_ids[b] = order[cursor * batch_graphs + b]; cursor += 1 -- all on the device (gae_batch_select), so a
    replayed HIP graph walks an epoch order that was uploaded once.  ``cursor`` int64[1], ``out_ids`` int64[B]."""
    order = _gpu(order, "order")
    with _on_device(order.device):
        _lib.call("gae_batch_select", _ptr(order), order.numel(), _ptr(cursor), int(batch_graphs), _ptr(out_ids),
                  _stream())
    return out_ids


def batch_plan_next(graph_ptr, ds_indptr, ds_t_indptr, order, cursor, out_ids, out):
    """batch_select + batch_plan in one launch (gae_x_batch_plan_next): the ids of batch ``cursor`` of ``order`` go to
    ``out_ids`` (int64 [B]), their prefix sums to ``out`` (int64 [3 or 2, B + 1]), the cursor advances"""
    B = out_ids.numel()
    rows = 3 if ds_t_indptr is not None else 2
    if out.shape != (rows, B + 1) or out.dtype != torch.int64 or not out.is_contiguous():
        raise GaeHipError("batch_plan_next: `out` must be a contiguous int64 [rows, B + 1] buffer")
    with _on_device(order.device):
        _lib.call("gae_x_batch_plan_next", _ptr(graph_ptr), _ptr(ds_indptr), _ptr(ds_t_indptr), _ptr(order),
                  order.numel(), _ptr(cursor), B, _ptr(out_ids), _ptr(out[0]), _ptr(out[1]),
                  _ptr(out[2]) if rows == 3 else None, _stream())
    return out[0], out[1], (out[2] if rows == 3 else None)


def batch_feature_ld(ds_feat, n_feat=None):
    """(F, leading dimension, dtype) of the feature matrix gae_batch_gather writes for ``ds_feat``"""
    F = ds_feat.shape[1] if n_feat is None else int(n_feat)
    odt = torch.float32 if ds_feat.dtype == torch.uint8 else ds_feat.dtype
    q = 4 if odt == torch.float32 else 8
    return F, max((F + q - 1) // q * q, 1), odt                  # batch features keep 16-byte rows


def batch_gather(graph_ptr, ds_indptr, ds_indices, ds_feat, graph_ids, node_ptr, edge_ptr, n_nodes, n_edges,
                 ell_width=0, n_feat=None, out=None, pad_to_capacity=False, counts=None):
    """dgl.batch of the graphs ``graph_ids`` of a device-resident dataset (gae_batch_gather): returns
    (indptr, indices, feat or None, packed table or None).  ``ds_feat`` None = structure only; uint8 features come
    back as fp32; ``n_feat`` = number of feature columns when ``ds_feat`` carries pad columns.
    ``out`` = (indptr, indices, feat, table) buffers to write into (static buffers of a captured step);
    ``pad_to_capacity``: n_nodes / n_edges are CAPACITIES, the rows behind the batch become isolated zero-feature
    nodes and the true {nodes, edges} go to ``counts`` (int64[2], device)."""
    dev = ds_indptr.device
    if counts is not None and (counts.dtype != torch.int64 or counts.numel() < 3):
        raise GaeHipError("batch_gather: `counts` must be an int64[3] device tensor {nodes, edges, graphs dropped}")
    if out is not None:
        out_indptr, out_indices, out_feat, table = out
    else:
        out_indptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
        out_indices = torch.empty(n_edges, dtype=torch.int32, device=dev)
        table = torch.empty(n_nodes * ell_width, dtype=torch.int32, device=dev) if ell_width else None
        out_feat = None
    feat = None
    ldf = F = ldo = 0
    code = F32
    if ds_feat is not None:
        feat, ldf = _rowmajor(ds_feat, "ds_feat")
        F, ldo, odt = _ops.batch_feature_ld(feat, n_feat)
        code = _lib.U8 if feat.dtype == torch.uint8 else _dtype_code(feat)
        if out_feat is None:
            out_feat = torch.empty(n_nodes, ldo, dtype=odt, device=dev)   # pad columns are zeroed by the kernel
        elif out_feat.shape != (n_nodes, ldo) or out_feat.dtype != odt or not out_feat.is_contiguous():
            raise GaeHipError("batch_gather: `out` feature buffer has the wrong shape / dtype")
    else:
        out_feat = None
    if out_indptr.numel() != n_nodes + 1 or out_indices.numel() < n_edges or (
            ell_width and (table is None or table.numel() != n_nodes * ell_width)):
        raise GaeHipError("batch_gather: `out` structure buffers have the wrong size")
    with _on_device(dev):
        _lib.call("gae_batch_gather", _ptr(graph_ptr), _ptr(ds_indptr), _ptr(ds_indices), _ptr(feat), max(ldf, F), F,
                  code, _ptr(graph_ids), graph_ids.numel(), _ptr(node_ptr), _ptr(edge_ptr),
                  n_nodes, n_edges, _ptr(out_indptr), _ptr(out_indices), _ptr(out_feat), max(ldo, F),
                  _ptr(table) if ell_width else None, int(ell_width), n_nodes if pad_to_capacity else 0,
                  _ptr(counts), _stream())
    return out_indptr, out_indices, (out_feat[:, :F] if out_feat is not None else None), table


def batch_gather_next(graph_ptr, ds_indptr, ds_indices, ds_feat, order, cursor, out_ids, ptrs, cap_nodes, cap_edges, out,
                      counts, ell_width=0, n_feat=None):
    """select + plan + gather of the next batch of an uploaded epoch order in ONE launch (gae_x_batch_gather_next;
    batches of <= 1024 graphs, symmetric datasets): the ids go to ``out_ids`` (int64 [B]), the prefix sums to ``ptrs``
    (int64 [2, B + 1]), the capacity-padded batch to ``out`` = (indptr, indices, feat, table), the true sizes to
    ``counts`` (int64[4], zero before the first call), the cursor advances"""
    B = out_ids.numel()
    out_indptr, out_indices, out_feat, table = out
    feat, ldf = _rowmajor(ds_feat, "ds_feat")
    F, ldo, odt = _ops.batch_feature_ld(feat, n_feat)
    code = _lib.U8 if feat.dtype == torch.uint8 else _dtype_code(feat)
    if ptrs.shape[0] < 2 or ptrs.shape[1] != B + 1 or ptrs.dtype != torch.int64 or not ptrs.is_contiguous():
        raise GaeHipError("batch_gather_next: `ptrs` must be a contiguous int64 [>= 2, B + 1] buffer")
    if counts.dtype != torch.int64 or counts.numel() < 4:
        raise GaeHipError("batch_gather_next: `counts` must be an int64[4] device tensor")
    if out_feat.shape != (cap_nodes, ldo) or out_feat.dtype != odt or out_indptr.numel() != cap_nodes + 1 or \
            out_indices.numel() < cap_edges or (ell_width and table.numel() != cap_nodes * ell_width):
        raise GaeHipError("batch_gather_next: `out` buffers have the wrong shape / dtype")
    with _on_device(order.device):
        _lib.call("gae_x_batch_gather_next", _ptr(graph_ptr), _ptr(ds_indptr), _ptr(ds_indices), _ptr(feat), max(ldf, F), F,
                  code, _ptr(order), order.numel(), _ptr(cursor), B, _ptr(out_ids), _ptr(ptrs[0]), _ptr(ptrs[1]),
                  int(cap_nodes), int(cap_edges), _ptr(out_indptr), _ptr(out_indices), _ptr(out_feat), max(ldo, F),
                  _ptr(table) if ell_width else None, int(ell_width), _ptr(counts), _stream())
    return ptrs[0], ptrs[1]


def segment_readout(Z, graph_ptr):
    """[mean | sum | max] of the rows of Z [N, d] per member graph (README.md:54 of the reference: the 48-d molecule
    feature); ``graph_ptr`` int64 [G + 1] node offsets on the device.  Returns [G, 3 d] fp32.  Inference-side op:
    no autograd."""
    Z, ldz = _rowmajor(_gpu(Z, "Z").detach(), "Z")
    if Z.dtype != torch.float32:
        raise GaeHipError(f"segment_readout: fp32 embeddings expected, got {Z.dtype}")
    gp = _gpu(graph_ptr, "graph_ptr").to(torch.int64).contiguous()
    n, d = Z.shape
    G_ = gp.numel() - 1
    out = torch.empty(max(G_, 0), 3 * d, dtype=torch.float32, device=Z.device)
    with _on_device(Z.device):
        _lib.call("gae_segment_readout", _ptr(Z), ldz, n, d, _ptr(gp), G_, _ptr(out), max(3 * d, 1), _stream())
    return out
