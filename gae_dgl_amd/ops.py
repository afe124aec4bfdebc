"""Torch-facing wrappers of the C ABI (include/gae_hip.h).

Tensors are only carriers of device pointers here: every op below launches
hand-written HIP kernels from libgae_hip.so on PyTorch's current HIP stream.
There is no CPU / eager fallback -- a CPU tensor raises."""
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import ACT_IDENTITY, ACT_RELU, BF16, F32, GaeHipError

_vp = ctypes.c_void_p


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_handle(device_index=None):
    """raw HIP stream of PyTorch's current stream (the private fast getter costs ~1 us, current_stream() ~10)"""
    if device_index is None:
        device_index = torch.cuda.current_device()
    if _raw_stream is not None:
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _stream():
    return _vp(_stream_handle())


def _ptr(t):
    return None if t is None else _vp(t.data_ptr())


def _gpu(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GaeHipError(f"{name}: expected a tensor on an AMD GPU (gae_dgl_amd has no CPU fallback), "
                          f"got {getattr(t, 'device', type(t))}")
    return t


def row_quantum(f, dtype):
    """leading-dimension multiple (elements) of an [N, f] operand: rows of 512 bytes and more are made of whole
    128-byte lines (XCD feature tiles of gae_spmm_csr then never share a line), narrower rows of whole 16-byte
    vectors"""
    size = 4 if dtype == torch.float32 else 2
    return (128 if f * size >= 512 else 16) // size


def padded_ld(f, dtype):
    """leading dimension (elements) this package allocates for an [N, f] operand: 16-byte rows below 512 bytes; from
    there on whole 128-byte lines and an ODD number of them (F = 500 fp32: 17 lines = 544 floats, not 16): with a
    power-of-two row pitch the column slice an XCD gathers under feature tiling maps to a fraction of the L1 tag
    banks / L2 channels (Pubmed layer-1 SpMM 19.1 -> 18.1 us; the pad line is never read or written)"""
    size = 4 if dtype == torch.float32 else 2
    q = row_quantum(f, dtype)
    ld = (f + q - 1) // q * q
    if f * size >= 512 and (ld * size // 128) % 2 == 0:
        ld += 128 // size
    return ld


def pad_rows(t, multiple=None):
    """view of ``t`` [N, F] inside a buffer whose rows are padded (zero pad) to 16 bytes, or to whole 128-byte lines
    (an odd number of them, see padded_ld) for rows of 512 bytes and more: gives every kernel the aligned vector
    path for odd feature widths (F = 39 -> ld 40, 500 -> 544, 1433 -> 1440, 3703 -> 3744).  A tensor that already has
    16-byte / whole-line rows is returned as it is."""
    n, f = t.shape
    q = multiple or row_quantum(f, t.dtype)
    if t.stride(1) == 1 and t.stride(0) % q == 0 and t.stride(0) >= f and t.data_ptr() % (q * t.element_size()) == 0:
        return t
    ld = (f + q - 1) // q * q if multiple else padded_ld(f, t.dtype)
    buf = torch.zeros(n, ld, dtype=t.dtype, device=t.device)
    buf[:, :f] = t
    return buf[:, :f]


def float_rows(t):
    """fp32 copy of a bf16-stored [N, F] operand in a row-padded buffer (see pad_rows): the Linear and dW kernels
    that read it next then take their aligned 16-byte paths (a plain ``.float()`` of F = 3703 has rows of 14812
    bytes: scalar loads; Citeseer VGAE layer-1 Linear 31 -> 22.5 us, dW 37.9 -> 14.9 us)"""
    n, f = t.shape
    ld = padded_ld(f, torch.float32)
    buf = torch.empty(n, ld, dtype=torch.float32, device=t.device)
    if ld > f:
        buf[:, f:].zero_()
    out = buf[:, :f]
    out.copy_(t)
    return out


def _rowmajor(t, name):
    """(tensor, ld) with unit inner stride; copies only when needed."""
    _gpu(t, name)
    if t.dim() != 2:
        raise GaeHipError(f"{name}: expected a 2-D tensor, got {tuple(t.shape)}")
    if t.shape[1] > 0 and t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return t, max(ld, t.shape[1], 1)


def _f32(t, name):
    """the dense kernels read raw fp32 memory: any other dtype would be reinterpreted, not converted"""
    if t is not None and t.dtype != torch.float32:
        raise GaeHipError(f"{name}: fp32 tensor expected, got {t.dtype} (cast explicitly; the kernels do not convert)")
    return t


def _dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise GaeHipError(f"unsupported dtype {t.dtype} (fp32 and bf16 storage are supported)")


_WS_CACHE = {}

# ------------------------------------------------------------------ the step context
# What ONE training step holds between its backward pass and its optimiser launch.  Two things can be left to the
# optimiser launch of gae_dgl_amd.optim.Adam (round 3: 3 launches less per captured step):
#   defer_grads  the weight-gradient kernels (gae_xw_wgrad, gae_linear_bwd, the fused layers' side work, gae_gcn2_bwd_dense)
#                leave their per-block partial sums in a private workspace and return UNINITIALISED gradient tensors; the
#                optimiser looks every gradient up in the context, adds its partials inside its launch (gae_adam_step's
#                deferred reduction) and writes the sum to the gradient tensor.  Only for steps in which that optimiser's
#                step() follows the backward pass directly: anything reading .grad in between reads garbage.
#   defer_loss   the fused loss (decoder_bce_raw) leaves its last launch -- the reduction of the per-block partial sums
#                to the scalar, which the backward pass does not read -- to the optimiser launch (gae_x_adam_step_tail).  The
#                returned loss tensor is filled only then.  A reduction nobody took is launched when the context closes.
# The context also carries the request of the loss's prepare step (loss_prepare_request).
#
# The state lives in the StepContext OBJECT that the step's owner opens (capture.CapturedTrainStep /
# CapturedInductiveStep per step; ``with ops.StepContext(...)`` / deferred_grad_reductions() in a hand-written loop) -- not
# in module-level tables: two models whose steps interleave each see their own partial sums, and nothing outlives its
# step.
# Inside a context a gradient is found by the address of its storage; the context keeps the tensor alive until the
# optimiser took the entry (or the context closed), so the address cannot be handed out again while the entry exists.
class StepContext:
    def __init__(self, defer_grads=False, defer_loss=False):
        self.defer_grads, self.defer_loss = bool(defer_grads), bool(defer_loss)
        self.partials = {}      # storage address -> (gradient tensor, (keep-alive workspace, partials ptr, n_partials,
        #                                              partial_stride, row_len, row_pitch))
        self.tails = []         # [(BceTail, keep-alive tensors)]
        self.prep_req = None
        self._outer = None
        self._flags = None

    # -- used by the kernels' wrappers
    def add_partials(self, grad, entry):
        self.partials[grad.data_ptr()] = (grad, entry)

    def take_partials(self, grad):
        ent = self.partials.pop(grad.data_ptr(), None) if self.partials else None
        return None if ent is None else ent[1]

    def take_loss_tail(self):
        return self.tails.pop() if self.tails else None

    def flush_loss_tails(self):
        while self.tails:
            tail, keep = self.tails.pop()
            with _on_device(keep[0].device):
                _lib.call("gae_x_decoder_bce_finalize", ctypes.byref(tail), _stream())

    # -- scope
    def __enter__(self):
        stack = _step_stack()
        if stack:
            # a context opened inside another one JOINS it (same step): it shares the outer tables and adds its flags
            # for its own duration
            outer = stack[-1]
            self._outer = outer
            self._flags = (outer.defer_grads, outer.defer_loss)
            outer.defer_grads |= self.defer_grads
            outer.defer_loss |= self.defer_loss
            stack.append(outer)
            return outer
        stack.append(self)
        return self

    def __exit__(self, *exc):
        stack = _step_stack()
        top = stack.pop()
        if self._outer is not None:
            top.defer_grads, top.defer_loss = self._flags
            self._outer = self._flags = None
            if not top.defer_grads and top.partials and exc[0] is None:
                n = len(top.partials)
                top.partials.clear()
                raise GaeHipError(f"{n} gradient(s) were left as partial sums: deferred gradient reductions need "
                                  "gae_dgl_amd.optim.Adam.step() inside the block, after the backward pass")
            if not top.defer_loss and exc[0] is None:
                top.flush_loss_tails()
            return
        n = len(self.partials)
        self.partials.clear()
        if exc[0] is None:
            self.flush_loss_tails()
            if n:
                raise GaeHipError(f"{n} gradient(s) were left as partial sums: deferred gradient reductions need "
                                  "gae_dgl_amd.optim.Adam.step() inside the block, after the backward pass")
        else:
            self.tails.clear()


# The stack of open contexts is process-wide, not thread-local: autograd executes the backward nodes of a step on its
# own worker thread (one per device), and those nodes must find the context the training loop's thread opened.  One
# training step is in flight per process at a time (a second thread that trains concurrently needs its own process,
# like its own GPU).
_STEP_STACK = []
_STEP_LOCK = threading.Lock()
_NO_STEP = StepContext()           # outside every context: nothing is deferred (only the prepare request of an eager
#                                    step passes through it)


class _StackView:
    """push / pop under the lock (reads of the top need none: list indexing is atomic in CPython)"""

    def __bool__(self):
        return bool(_STEP_STACK)

    def __getitem__(self, i):
        return _STEP_STACK[i]

    def append(self, x):
        with _STEP_LOCK:
            _STEP_STACK.append(x)

    def pop(self):
        with _STEP_LOCK:
            return _STEP_STACK.pop()


def _step_stack():
    return _StackView()


def current_step():
    """the innermost open StepContext (one that defers nothing when there is none)"""
    try:
        return _STEP_STACK[-1]
    except IndexError:
        return _NO_STEP


def deferred_grad_reductions():
    """``with ops.deferred_grad_reductions():`` = a StepContext that defers the weight gradients' reductions"""
    return StepContext(defer_grads=True)


def deferred_loss_finalize():
    """``with ops.deferred_loss_finalize():`` = a StepContext that defers the loss's final reduction"""
    return StepContext(defer_loss=True)


def pending_loss_tail():
    """(BceTail, keep-alive) of a loss of the current step whose final reduction was deferred (removed from the
    context), or None"""
    return current_step().take_loss_tail()


def pending_partials(grad):
    """(keep-alive, ptr, n_partials, partial_stride, row_len, row_pitch) of a gradient of the current step whose
    reduction was deferred (removed from the context), or None"""
    return current_step().take_partials(grad)


def _workspace(nbytes, device):
    """scratch buffer for one call.  Buffers are cached per (device, stream, size class): every launch that
    uses one is ordered on that stream, so the next call may reuse it (no allocator round trip per call)."""
    nbytes = max(int(nbytes), 16)
    size = 1 << (nbytes - 1).bit_length()
    key = (device, _stream_handle(device.index), size)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(size, dtype=torch.uint8, device=device)   # graph-private pool owns captured scratch
    ws = _WS_CACHE.get(key)
    if ws is None:
        ws = _WS_CACHE[key] = torch.empty(size, dtype=torch.uint8, device=device)
    return ws


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the guard costs microseconds)"""
    __slots__ = ("guard",)

    def __init__(self, dev):
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.guard = None if idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *a):
        if self.guard is not None:
            self.guard.__exit__(*a)


def device_info(device=0):
    info = _lib.DeviceInfo()
    _lib.call("gae_device_info_get", int(device), ctypes.byref(info))
    return {f: (getattr(info, f).decode() if f == "name" else getattr(info, f)) for f, _ in info._fields_}


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
        F, ldo, odt = batch_feature_ld(feat, n_feat)
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
    F, ldo, odt = batch_feature_ld(feat, n_feat)
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


# ------------------------------------------------------------------ profiling hook
class EventProfiler:
    """Records a HIP event pair (on the launch stream) around selected C-ABI
    calls; used by bench.py for the live per-kernel roofline numbers."""

    def __init__(self):
        self.records = {}

    def wrap(self, key, fn):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.records.setdefault(key, []).append((e0, e1))
        return out

    def summary(self):
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) * 1e-3 for a, b in v] for k, v in self.records.items()}


profiler = None  # set to an EventProfiler to time launches


# ------------------------------------------------------------------ raw kernels
SKEW_THRESHOLD = 8       # rows with more in-edges than this go through the segment kernels
SKEW_SEGMENT = 512       # edges per segment (multiple of 64; RMAT s24: 256 -> 6.81 ms, 512 -> 6.65, 1024 -> 6.59)
SKEW_MIN_MAXDEG = 64     # graphs whose longest row is shorter need no plan (3 extra launches would not pay)
# ... and graphs small enough for a packed neighbour table (ELL_MAX_ROWS) none up to this row length: the table kernels
# take a row that outgrows its 16 slots with the whole wave (spmm_ell.hip: ell_long_row; 64 column ids per load, 8 rows
# in flight per lane), the fused loss's edge kernel likewise -- the real Planetoid graphs (longest rows 168 / 99 / 171)
# run the same launches as the uniform synthetic ones (tools/r05/hubs.sh: step + 7 .. 12 %).  GAE_TABLE_MAXDEG overrides.
TABLE_MAX_ROW = int(os.environ.get("GAE_TABLE_MAXDEG", 1024))


TILE_MIN_F = 64          # XCD feature tiles are only considered for rows wider than one lane group (16 vectors)
SCATTER_L2_BYTES = 2 << 20   # half of one XCD's 4 MiB L2: the window of H rows a row block can expect to find cached
ELL_MAX_ROWS = 1 << 18   # packed neighbour table only for graphs whose launches are latency-bound, not byte-bound


LIGHT_LIST = True        # skew plans carry the list of their rows with 1 .. threshold edges (gae_spmm_plan::light_desc)
INT32_MAX = 2 ** 31 - 1


class SpmmPlan:
    """per-CSR acceleration data of gae_spmm_csr (gae_spmm_plan in include/gae_hip.h): the degree-skew plan (light-row
    list, mid rows cut into segments, XCD-pinned very long rows -- built on the device by csrc/plan_build.hip) and / or
    the packed neighbour table.  ``parts``: dict of the device arrays by their field name in gae_spmm_plan."""

    def __init__(self, threshold, segment, parts=None, ell=None, ell_width=None, n_virtual=0, mid_tagged=False):
        parts = dict(parts or {})
        g = parts.get
        self.parts = parts
        self.threshold, self.segment = int(threshold), int(segment)
        self.ell = ell
        self.ell_width = (ell_width or _lib.SPMM_ELL_WIDTH) if ell is not None else 0
        self.n_heavy = 0 if g("heavy_rows") is None else int(g("heavy_rows").numel())
        self.n_segments = 0 if g("seg_heavy") is None else int(g("seg_heavy").numel())
        self.seg_desc = g("seg_desc")
        self.light_desc = g("light_desc")
        self.n_light = 0 if self.light_desc is None else int(self.light_desc.shape[0])
        self.mid_ids = g("mid_indices")
        self.hot_indices = self.mid_ids if mid_tagged else None       # (the tagged ids of the segment kernel, if any)
        self.homed = None
        if g("vh_rows") is not None:
            self.homed = dict(rows=g("vh_rows"), cols=g("vh_indices"), desc=g("vh_desc"), part_ptr=g("vh_part_ptr"),
                              part_pos=g("vh_part_pos"), n_edges=int(g("vh_indices").numel()), n_virtual=int(n_virtual))
        # heavy_rows first: tests read tensors[:3]; everything listed here is kept alive and counted as plan bytes
        self.tensors = tuple(parts.get(k) for k in ("heavy_rows", "heavy_seg_base", "seg_heavy", "seg_desc", "light_desc",
                                                    "mid_indices", "vh_rows", "vh_indices", "vh_desc", "vh_part_ptr",
                                                    "vh_part_pos")) + (ell,)
        ptr = lambda t: None if t is None else t.data_ptr()
        c = _lib.SpmmPlan()
        c.threshold, c.segment_edges = self.threshold, self.segment
        c.n_heavy, c.n_segments = self.n_heavy, self.n_segments
        c.heavy_rows, c.heavy_seg_base, c.seg_heavy = ptr(g("heavy_rows")), ptr(g("heavy_seg_base")), ptr(g("seg_heavy"))
        c.ell, c.ell_width = ptr(ell), self.ell_width
        c.seg_desc = ptr(self.seg_desc)
        c.light_desc, c.n_light = ptr(self.light_desc), self.n_light
        c.mid_indices, c.mid_tagged = ptr(self.mid_ids), 1 if (mid_tagged and self.mid_ids is not None) else 0
        if self.homed is not None:
            c.vh_n_rows, c.vh_n_virtual = int(g("vh_rows").numel()), int(n_virtual)
            c.vh_rows, c.vh_indices, c.vh_desc = ptr(g("vh_rows")), ptr(g("vh_indices")), ptr(g("vh_desc"))
            c.vh_part_ptr, c.vh_part_pos = ptr(g("vh_part_ptr")), ptr(g("vh_part_pos"))
        self.c = c

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.tensors if t is not None)

    def set_skip_rows(self, mask, covers_all_empty=False):
        """uint8 [n_rows] (or None): rows WITHOUT edges that a product launched with ``skip_dead=True`` need not write
        (gae_spmm_plan::skip_rows, GAE_SPMM_SKIP_ROWS) -- the caller's consumers treat them as zero without reading them.
        ``covers_all_empty``: the mask marks EVERY row without edges of this CSR (the empty-row stream is then not launched)"""
        self.skip_rows = mask
        self.c.skip_rows = None if mask is None else mask.data_ptr()
        self.c.reserved2 = 1 if (mask is not None and covers_all_empty) else 0


def ell_width_for(max_deg):
    """narrowest packed-table width (4, 8 or 16 slots) that holds every row of a graph whose longest (light) row
    has ``max_deg`` edges; rows longer than 16 continue from the CSR arrays"""
    return 4 if max_deg <= 4 else 8 if max_deg <= 8 else _lib.SPMM_ELL_WIDTH


ELL_OVERFLOW_SHARE = 0.01     # rows that may continue from the CSR arrays (they take a slower path in the kernel)


def ell_width_for_degrees(deg, cap=None):
    """narrowest table width that holds all but ELL_OVERFLOW_SHARE of the rows (``deg``: device tensor of row
    lengths; rows above ``cap`` are heavy rows of a skew plan and do not count): a molecule set whose atoms have at
    most 4 bonds except for a handful gets 4 slots = 16 bytes per row instead of 64.  One host read-back."""
    if deg.numel() == 0:
        return 4
    d = deg if cap is None else deg[deg <= cap]
    if d.numel() == 0:
        return 4
    over = torch.stack([(d > 4).float().mean(), (d > 8).float().mean()]).tolist()
    return 4 if over[0] <= ELL_OVERFLOW_SHARE else 8 if over[1] <= ELL_OVERFLOW_SHARE else _lib.SPMM_ELL_WIDTH


def table_plan(table, ell_width):
    """plan that only carries an already-built packed neighbour table (no heavy rows)"""
    return SpmmPlan(SKEW_THRESHOLD, SKEW_SEGMENT, None, table, ell_width)


HOT_COLUMNS = 65536          # columns tagged hot for the mid rows of a skew plan (RMAT s24, F = 32: 2 k 9.2 ms, 8 k 7.8, 16 k 7.2,
#                              32 k 6.9, 64 k 6.8, 256 k 7.2; untagged 7.8 -- streaming the moderately hot rows hurts)
HOT_MIN_EDGES = 1 << 22      # graphs with fewer edges fit the caches anyway

HOMED_MIN_DEGREE = 256       # rows with more in-edges than this are gathered XCD-pinned ("homed") ...
HOMED_MIN_EDGES = 1 << 24    # ... when they hold at least this many edges together (RMAT s24: 175 M of 268 M)
# (round 4: the pinned part carries NO hot tags -- with every column going through one L2, streaming loads for the
#  other columns cost more than they save: RMAT s24 pinned launch 1.90 ms tagged, 1.83 ms plain)


def column_home(cols):
    """XCD (0..7) through whose L2 a column is gathered in the pinned part of a plan: a multiplicative hash (the low
    bits of the hub ids of an R-MAT graph are all zero: ``col % 8`` would put 44 % of the edges on one XCD).  The
    device builder (csrc/plan_build.hip) uses the same function."""
    return ((cols.to(torch.int64) * 2654435761) >> 13) & 7


def spmm_plan(indptr, threshold=None, segment=None, indices=None, ell=None, ell_width=None, hot=None, n_cols=None, homed=None):
    """Build the plan of a CSR on the device (gae_spmm_plan_sizes / _build_rows / _build_pinned; no torch kernels), or
    None when it needs none.  Skew part: with the default threshold only for graphs whose longest row has more than
    SKEW_MIN_MAXDEG edges (TABLE_MAX_ROW for graphs that get a packed neighbour table).  Packed neighbour table: when ``indices`` is given and the graph has at most ELL_MAX_ROWS
    rows (``ell`` = True / False overrides); ``ell_width`` 4 / 8 / 16 slots per row, default: the narrowest that holds
    the longest light row.  ``hot`` (default: skew plans of graphs with at least HOT_MIN_EDGES edges when ``indices``
    is given; ``n_cols`` = columns of the CSR, default: its rows): the mid rows read a compact copy of their column
    ids with the sign bit on the HOT_COLUMNS most gathered columns, which lets the kernel stream the rarely gathered
    rows past the L2.  ``homed``: XCD-pinned regrouping of the rows with more than HOMED_MIN_DEGREE edges (default:
    when they hold at least HOMED_MIN_EDGES edges; True / False forces)."""
    auto = threshold is None
    threshold = SKEW_THRESHOLD if threshold is None else int(threshold)
    segment = SKEW_SEGMENT if segment is None else int(segment)
    _gpu(indptr, "indptr")
    dev = indptr.device
    n = indptr.numel() - 1
    want_ell = (indices is not None and 0 < n <= ELL_MAX_ROWS) if ell is None else bool(ell)
    if want_ell and indices is None:
        raise GaeHipError("spmm_plan: the packed neighbour table needs `indices`")
    if n <= 0:
        return None
    lib = _lib.load()
    with _on_device(dev):
        tiny = torch.empty(256, dtype=torch.uint8, device=dev)
        sizes = (ctypes.c_int64 * 8)()
        t2 = max(HOMED_MIN_DEGREE, threshold) if (indices is not None and homed is not False) else INT32_MAX
        _lib.call("gae_spmm_plan_sizes", _ptr(indptr), n, threshold, t2, segment, sizes, _ptr(tiny), tiny.numel(), _stream())
        nl, nm, sm, em, npin, epin, max_deg = (int(sizes[k]) for k in range(7))
        heavy = (nm + npin) > 0 and not (auto and max_deg <= (TABLE_MAX_ROW if want_ell else SKEW_MIN_MAXDEG))
        if not heavy and not want_ell:
            return None
        pin = heavy and npin > 0 and t2 != INT32_MAX and (bool(homed) or epin >= HOMED_MIN_EDGES)
        if heavy and npin > 0 and not pin:           # the long rows stay ordinary segmented rows
            t2 = INT32_MAX
            _lib.call("gae_spmm_plan_sizes", _ptr(indptr), n, threshold, t2, segment, sizes, _ptr(tiny), tiny.numel(), _stream())
            nl, nm, sm, em, npin, epin, max_deg = (int(sizes[k]) for k in range(7))
        parts, n_virtual, tagged = {}, 0, False
        if heavy:
            n_edges = int(indices.numel()) if indices is not None else 0
            tagged = ((n_edges >= HOT_MIN_EDGES) if hot is None else bool(hot)) and indices is not None and em > 0
            nc = int(n_cols) if n_cols is not None else n
            i32 = lambda *shape: torch.empty(*shape, dtype=torch.int32, device=dev)
            if LIGHT_LIST and nl:
                parts["light_desc"] = i32(nl, 4)
            if nm:
                parts.update(heavy_rows=i32(nm), heavy_seg_base=i32(nm), seg_heavy=i32(sm), seg_desc=i32(sm, 4))
                if tagged:
                    parts["mid_indices"] = i32(em)
            if npin:
                parts.update(vh_rows=i32(npin), vh_part_ptr=i32(npin + 1))
            sb = int(lib.gae_spmm_plan_scratch_bytes(n, nc, npin, epin, segment))
            scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
            ph = (ctypes.c_int64 * 4)()
            g = parts.get
            _lib.call("gae_spmm_plan_build_rows", _ptr(indptr), _ptr(indices), n, nc, threshold, t2, segment, sizes,
                      int(HOT_COLUMNS if tagged else 0), _ptr(g("light_desc")), _ptr(g("heavy_rows")),
                      _ptr(g("heavy_seg_base")), _ptr(g("seg_heavy")), _ptr(g("seg_desc")), _ptr(g("mid_indices")),
                      _ptr(g("vh_rows")), _ptr(g("vh_part_ptr")), _ptr(scratch), sb, ph, _stream())
            if npin:
                parts["vh_indices"] = i32(epin)
                _lib.call("gae_spmm_plan_build_pinned", _ptr(indptr), _ptr(indices), n, nc, segment, sizes, _ptr(g("vh_rows")),
                          _ptr(parts["vh_indices"]), None, None, _ptr(scratch), sb, ph, _stream())
                n_virtual = int(ph[1])
                parts["vh_desc"] = i32(n_virtual, 4)
                parts["vh_part_pos"] = i32(int(ph[0]))
                _lib.call("gae_spmm_plan_build_pinned", _ptr(indptr), _ptr(indices), n, nc, segment, sizes, _ptr(g("vh_rows")),
                          _ptr(parts["vh_indices"]), _ptr(parts["vh_desc"]), _ptr(parts["vh_part_pos"]), _ptr(scratch), sb, ph,
                          _stream())
            del scratch
        table = None
        if want_ell:
            if ell_width is None:
                ell_width = ell_width_for_degrees(indptr[1:] - indptr[:-1], threshold if heavy else None)
            table = torch.empty(n * ell_width, dtype=torch.int32, device=dev)
            _lib.call("gae_spmm_ell_build", _ptr(indptr), _ptr(indices), n, ell_width,
                      threshold if heavy else 2 ** 31 - 1, _ptr(table), _stream())
    return SpmmPlan(threshold, segment, parts, table, ell_width, n_virtual, tagged)


def gather_distance(indptr, indices):
    """median |column id - row id| over the edges (0 for an edge-less graph).  One host read-back; computed once
    per graph."""
    n = indptr.numel() - 1
    e = indices.numel()
    if n == 0 or e == 0:
        return 0
    deg = (indptr[1:] - indptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n, device=indptr.device), deg, output_size=e)
    return int((indices.to(torch.int64) - rows).abs().median())


def gather_scattered(indptr, indices, row_bytes, distance=None):
    """True when most neighbour rows lie further from their row than half an XCD's L2 holds (SCATTER_L2_BYTES /
    row_bytes rows): concurrently processed rows then share nothing in L1 / L2 and wide launches do better with
    XCD feature tiles (GAE_SPMM_TILE)."""
    d = gather_distance(indptr, indices) if distance is None else distance
    return d * row_bytes > SCATTER_L2_BYTES


BLOCKDIAG_GRAPHS = 4          # member graphs per thread block of the block-diagonal kernel
BLOCKDIAG_MIN_BLOCKS = 8192   # fewer blocks (a 4096-molecule batch has 1024) run faster on the row-group kernel
BLOCKDIAG_MAX_EDGES = 1024    # index slice staged in LDS per block (more edges are read from global memory)


class BlockDiag:
    """row runs closed under adjacency (whole member graphs) for gae_spmm_csr_blockdiag.  The number of member
    graphs per thread block follows the row width (about three rows per lane group: F = 39 -> 2 molecules,
    F = 32 -> 4, F <= 16 -> 8); the cuts are built on first use and cached."""

    def __init__(self, node_ptr_host, device, graphs_per_block=None):
        import numpy as np
        self.node_ptr = np.asarray(node_ptr_host, dtype=np.int64)
        self.device = device
        self.fixed = graphs_per_block
        self.min_blocks = BLOCKDIAG_MIN_BLOCKS
        self.max_edges = BLOCKDIAG_MAX_EDGES
        self._cuts = {}
        self._eptr = {}

    def _graphs_per_block(self, F):
        return self.fixed or (2 if F > 32 else 4 if F > 16 else BLOCKDIAG_GRAPHS * 2)

    def cuts(self, F):
        """(block_ptr int32 device tensor, n_blocks, max_rows) for feature width F"""
        import numpy as np
        g = self._graphs_per_block(F)
        if g not in self._cuts:
            c = self.node_ptr[::g]
            if c[-1] != self.node_ptr[-1]:
                c = np.append(c, self.node_ptr[-1])
            nb = len(c) - 1
            self._cuts[g] = (torch.from_numpy(c.astype(np.int32)).to(self.device), nb,
                             int(np.diff(c).max()) if nb else 0)
        return self._cuts[g]

    def eptr(self, indptr, block_ptr):
        """edge offset of every block for this CSR (forward and transposed structures differ)"""
        key = (indptr.data_ptr(), block_ptr.data_ptr())
        if key not in self._eptr:
            self._eptr[key] = indptr.index_select(0, block_ptr.to(torch.int64)).contiguous()
        return self._eptr[key]

    def usable(self, H, F, ldh, ldm):
        if H.dtype != torch.float32 or F > 256 or ldh % 4 or ldm % 4 or len(self.node_ptr) < 2:
            return False
        g = self._graphs_per_block(F)
        if (len(self.node_ptr) - 2) // g + 1 < self.min_blocks:    # too few blocks: decided without building the cuts
            return False
        _, nb, max_rows = self.cuts(F)
        if nb < self.min_blocks or max_rows > 511 or max_rows * ldh > 16 * 256 * 4 or self.max_edges > 1024:
            return False
        return _lib.load().gae_spmm_blockdiag_lds_bytes(max_rows, self.max_edges, ldh) <= 160 * 1024


def spmm_raw(indptr, indices, H, n_rows, row_scale=None, col_scale=None, out=None, plan=None, blockdiag=None,
             out_padded=False, scattered=False, accumulate=False, skip_dead=False):
    """M = diag(row_scale) A diag(col_scale) H  (K1/K2).  ``out_padded``: the caller's ``out`` is a view of a
    row-padded buffer whose pad columns may be overwritten (always true for the buffer allocated here).
    ``scattered``: the graph's column ids lie far from the row ids (GAE_SPMM_TILE).
    ``accumulate``: ``out += ...`` (GAE_SPMM_ACCUMULATE; needs ``out``)."""
    H, ldh = _rowmajor(H, "H")
    _gpu(indptr, "indptr")
    n_cols, F = H.shape
    if out is None:
        # rows padded to 16 / 128 bytes: keeps the vector path for any F (the pad columns are never read as data)
        out = torch.empty(n_rows, padded_ld(F, H.dtype), dtype=H.dtype, device=H.device)[:, :F]
        out_padded = True
    if accumulate and out is None:
        raise GaeHipError("spmm: accumulate=True adds to `out`")
    flags = (_lib.SPMM_STORE_PAD if out_padded else 0) | (_lib.SPMM_TILE if scattered else 0) | \
        (_lib.SPMM_ACCUMULATE if accumulate else 0)
    if skip_dead and plan is not None and getattr(plan, "skip_rows", None) is not None:
        flags |= _lib.SPMM_SKIP_ROWS        # the rows marked in the plan's mask stay UNWRITTEN (see SpmmPlan.set_skip_rows)
    if accumulate:
        blockdiag = None
    out2, ldm = _rowmajor(out, "out")
    if out2 is not out:
        raise GaeHipError("spmm: `out` must be row-major with unit inner stride")
    if blockdiag is not None and n_rows == n_cols and H.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 \
            and blockdiag.usable(H, F, ldh, ldm):
        with _on_device(H.device):
            block_ptr, n_blocks, max_rows = blockdiag.cuts(F)
            eptr = blockdiag.eptr(indptr, block_ptr)

            def launch_bd():
                _lib.call("gae_spmm_csr_blockdiag", _ptr(indptr), _ptr(indices), _ptr(block_ptr), _ptr(eptr), n_blocks,
                          max_rows, blockdiag.max_edges, n_rows, _ptr(H), ldh, _ptr(out), ldm, F, _ptr(row_scale),
                          _ptr(col_scale), flags, _stream())
            if profiler is not None:
                profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch_bd)
            else:
                launch_bd()
        return out
    with _on_device(H.device):
        pc, ws, ws_bytes = None, None, 0
        if plan is not None:
            pc = ctypes.byref(plan.c)
            ws_bytes = _lib.load().gae_spmm_workspace_bytes(pc, F)
            ws = _workspace(ws_bytes, H.device)

        def launch():
            _lib.call("gae_spmm_csr", _ptr(indptr), _ptr(indices), n_rows, n_cols, _ptr(H), ldh, _ptr(out), ldm, F,
                      _dtype_code(H), _ptr(row_scale), _ptr(col_scale), pc, _ptr(ws), ws_bytes, flags, _stream())
        if profiler is not None:
            profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch)
        else:
            launch()
    return out


def spmm_ep_raw(indptr, indices, H, n_rows, plan=None, bias=None, act=ACT_IDENTITY, row_scale=None, col_scale=None,
                out=None, accumulate=False):
    """M = act(diag(rs) A diag(cs) H (+ out, ``accumulate``) + bias): gae_spmm_csr with a store-time epilogue for ANY
    plan (gae_spmm_csr_ep) -- the sparse half of a layer evaluated as act(A (H W^T) + b) on graphs whose plans carry
    degree-skew segments / XCD-pinned rows.  fp32."""
    H, ldh = _rowmajor(_f32(_gpu(H, "H"), "spmm_ep: H"), "H")
    n_cols, F = H.shape
    _f32(bias, "spmm_ep: bias")
    if accumulate and out is None:
        raise GaeHipError("spmm_ep: accumulate=True adds to `out`")
    padded = out is None
    if out is None:
        out = torch.empty(n_rows, padded_ld(F, torch.float32), dtype=torch.float32, device=H.device)[:, :F]
    out2, ldm = _rowmajor(out, "out")
    if out2 is not out:
        raise GaeHipError("spmm_ep: `out` must be row-major with unit inner stride")
    flags = (_lib.SPMM_STORE_PAD if padded else 0) | (_lib.SPMM_ACCUMULATE if accumulate else 0)
    with _on_device(H.device):
        pc, ws, ws_bytes = None, None, 0
        if plan is not None:
            pc = ctypes.byref(plan.c)
            ws_bytes = _lib.load().gae_spmm_workspace_bytes(pc, F)
            ws = _workspace(ws_bytes, H.device)

        def launch():
            _lib.call("gae_spmm_csr_ep", _ptr(indptr), _ptr(indices), n_rows, n_cols, _ptr(H), ldh, _ptr(out), ldm, F,
                      _ptr(row_scale), _ptr(col_scale), pc, _ptr(ws), ws_bytes, flags, _ptr(bias), int(act), _stream())
        if profiler is not None:
            profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch)
        else:
            launch()
    return out


def linear2_usable(A, f_mid, f_out):
    """can gae_linear2_fwd / gae_gcn2_bwd_dense take this operand?  fp32 rows of whole 16-byte vectors, widths <= 32"""
    return (isinstance(A, torch.Tensor) and A.is_cuda and A.dtype == torch.float32 and A.dim() == 2 and A.shape[0] > 0
            and 1 <= A.shape[1] <= 32 and 1 <= f_mid <= 32 and 1 <= f_out <= 32)


def _dead_mask(t, n, what):
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 1 and t.numel() == n
            and t.is_contiguous()):
        raise GaeHipError(f"{what}: a dead-row mask is a contiguous uint8 device tensor with one entry per row")
    return t


def linear2_fwd_raw(A, W1, b1, act1, W2, want_y1=True, a_dead=None, rows=None, fill=None):
    """(Y1, T): Y1 = act1(A W1^T + b1), T = Y1 W2^T in ONE pass over A (gae_linear2_fwd); Y1 None when not wanted.
    Both outputs have rows of whole 16-byte vectors.  ``a_dead`` (uint8 [n]): rows of A that are zero and were never
    written (spmm_raw(skip_dead=True)) -- not read.  ``rows`` (int32, ascending; with ``a_dead`` marking all the others):
    list mode -- the two products run on the listed rows only, the other rows of T get their common value
    act1(b1) W2^T (gae_linear2_fill_dead); Y1 is not available then.  ``fill`` (uint8 [n], default ``a_dead``): the
    rows that receive that value -- a caller who knows that some dead rows of T are never read (T as the gather operand
    of the next aggregation: nodes without out-edges) leaves them unwritten."""
    A, lda = _rowmajor(_f32(_gpu(A, "A"), "linear2: A"), "A")
    if lda % 4 or A.data_ptr() % 16:
        A = pad_rows(A); lda = A.stride(0)
    W1 = _f32(_gpu(W1, "W1"), "linear2: W1"); W2 = _f32(_gpu(W2, "W2"), "linear2: W2")
    W1 = W1 if W1.stride(1) == 1 else W1.contiguous()
    W2 = W2 if W2.stride(1) == 1 else W2.contiguous()
    _f32(b1, "linear2: b1")
    n, f_in = A.shape
    f_mid, f_out = W1.shape[0], W2.shape[0]
    if W1.shape[1] != f_in or W2.shape[1] != f_mid:
        raise GaeHipError("linear2: weight shapes do not chain")
    dev = A.device
    ld1, ld2 = (f_mid + 3) // 4 * 4, (f_out + 3) // 4 * 4
    Y1 = torch.empty(n, ld1, dtype=torch.float32, device=dev)[:, :f_mid] if want_y1 else None
    T = torch.empty(n, ld2, dtype=torch.float32, device=dev)[:, :f_out]
    a_dead = _dead_mask(a_dead, n, "linear2")
    if rows is not None:
        if a_dead is None or want_y1 or rows.dtype != torch.int32 or not rows.is_cuda or not rows.is_contiguous():
            raise GaeHipError("linear2: list mode takes an int32 device list, the dead-row mask of all other rows, and no Y1")
    with _on_device(dev):
        def launch():
            if rows is not None:
                _lib.call("gae_linear2_fwd", _ptr(A), lda, n, f_in, _ptr(W1), W1.stride(0), _ptr(b1), f_mid, int(act1),
                          _ptr(W2), W2.stride(0), f_out, None, ld1, _ptr(T), ld2, None, _ptr(rows), int(rows.numel()),
                          _stream())
                _lib.call("gae_linear2_fill_dead", _ptr(b1), f_mid, int(act1), _ptr(W2), W2.stride(0), f_out,
                          _ptr(a_dead if fill is None else _dead_mask(fill, n, "linear2")), n, _ptr(T), ld2, _stream())
                return
            _lib.call("gae_linear2_fwd", _ptr(A), lda, n, f_in, _ptr(W1), W1.stride(0), _ptr(b1), f_mid, int(act1),
                      _ptr(W2), W2.stride(0), f_out, _ptr(Y1), ld1, _ptr(T), ld2, _ptr(a_dead), None, 0, _stream())
        if profiler is not None:
            profiler.wrap(("linear2", n, f_in, f_mid, f_out), launch)
        else:
            launch()
    return Y1, T


def gcn2_bwd_dense_raw(G, dZ, Y1, act1, M1, W2, W1=None, b1=None, m1_dead=None, g_dead=None, rows=None,
                       g_dead_listed=None):
    """(dW1, db1, dW2, db2) of a two-layer encoder from G = A^T dZ in ONE pass over G, dZ, Y1, M1 (gae_gcn2_bwd_dense):
    dW2 = G^T Y1, db2 = colsum(dZ), dY1 = (G W2) (.) act1'(Y1), dW1 = dY1^T M1, db1 = colsum(dY1).  Inside
    ``deferred_grad_reductions()`` the four gradients stay per-block partial sums for optim.Adam.step().
    ``Y1`` None: the pass recomputes Y1 = act1(M1 W1^T + b1) itself (bit-identical to linear2_fwd_raw's), ``W1`` / ``b1``
    needed; then ``m1_dead`` / ``g_dead`` (uint8 [n]) mark rows of M1 / G that are zero and were never written: not read.
    ``rows`` (int32, ascending = the rows that have an M1 row; ``m1_dead`` marks exactly the others; ``g_dead_listed`` =
    g_dead at the listed rows): list mode -- the pass visits the listed rows, the share of the others (a rank-one term of
    the column sums of their G / dZ rows) is one more partial of the list."""
    G, ldg = _rowmajor(_f32(_gpu(G, "G"), "gcn2_bwd: G"), "G")
    dZ, lddz = _rowmajor(_f32(_gpu(dZ, "dZ"), "gcn2_bwd: dZ"), "dZ")
    if ldg % 4 or G.data_ptr() % 16:
        G = pad_rows(G); ldg = G.stride(0)
    if lddz % 4 or dZ.data_ptr() % 16:
        dZ = pad_rows(dZ); lddz = dZ.stride(0)
    M1, ldm1 = _rowmajor(_f32(M1, "gcn2_bwd: M1"), "M1")
    W2 = _f32(W2, "gcn2_bwd: W2")
    W2 = W2 if W2.stride(1) == 1 else W2.contiguous()
    n, f_out = G.shape
    f_mid, f_in = W2.shape[1], M1.shape[1]
    ldy1 = ldw1 = 0
    if Y1 is not None:
        Y1, ldy1 = _rowmajor(_f32(Y1, "gcn2_bwd: Y1"), "Y1")
        if Y1.shape != (n, f_mid):
            raise GaeHipError("gcn2_bwd: operand shapes do not match")
    else:
        if W1 is None:
            raise GaeHipError("gcn2_bwd: recomputing Y1 needs W1")
        W1 = _f32(_gpu(W1, "W1"), "gcn2_bwd: W1")
        W1 = W1 if W1.stride(1) == 1 else W1.contiguous()
        _f32(b1, "gcn2_bwd: b1")
        ldw1 = W1.stride(0)
        if W1.shape != (f_mid, f_in):
            raise GaeHipError("gcn2_bwd: W1 does not match the operands")
        if ldm1 % 4 or M1.data_ptr() % 16:
            M1 = pad_rows(M1); ldm1 = M1.stride(0)
    if dZ.shape != G.shape or W2.shape[0] != f_out or M1.shape[0] != n:
        raise GaeHipError("gcn2_bwd: operand shapes do not match")
    dev = G.device
    m1_dead, g_dead = _dead_mask(m1_dead, n, "gcn2_bwd"), _dead_mask(g_dead, n, "gcn2_bwd")
    if Y1 is not None and (m1_dead is not None or g_dead is not None or rows is not None):
        raise GaeHipError("gcn2_bwd: dead-row masks / row lists go with the recomputing form (Y1 = None)")
    if rows is not None:
        if m1_dead is None or rows.dtype != torch.int32 or not rows.is_cuda or not rows.is_contiguous():
            raise GaeHipError("gcn2_bwd: list mode takes an int32 device list and the dead-row mask of all other rows")
        g_dead_listed = _dead_mask(g_dead_listed, int(rows.numel()), "gcn2_bwd")
    dW1 = torch.empty(f_mid, f_in, dtype=torch.float32, device=dev)
    db1 = torch.empty(f_mid, dtype=torch.float32, device=dev)
    dW2 = torch.empty(f_out, f_mid, dtype=torch.float32, device=dev)
    db2 = torch.empty(f_out, dtype=torch.float32, device=dev)
    with _on_device(dev):
        nbytes = _lib.load().gae_gcn2_bwd_dense_workspace_bytes(n, f_in, f_mid, f_out)
        if nbytes < 0:
            _lib.check(int(nbytes), "gae_gcn2_bwd_dense_workspace_bytes")
        defer = current_step().defer_grads
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev) if defer else _workspace(nbytes, dev)
        lay = (ctypes.c_int64 * 5)()

        def launch():
            _lib.call("gae_gcn2_bwd_dense", _ptr(G), ldg, _ptr(dZ), lddz, _ptr(Y1), ldy1, int(act1), _ptr(M1), ldm1,
                      _ptr(W2), W2.stride(0), n, f_in, f_mid, f_out, _ptr(dW1), _ptr(db1), _ptr(dW2), _ptr(db2), _ptr(ws),
                      ws.numel(), lay if defer else None, _ptr(W1) if Y1 is None else None, ldw1,
                      _ptr(b1) if Y1 is None else None, _ptr(m1_dead), _ptr(g_dead), _ptr(rows),
                      0 if rows is None else int(rows.numel()), _ptr(g_dead_listed), _stream())
        if profiler is not None:
            profiler.wrap(("gcn2_bwd", n, f_in, f_mid, f_out), launch)
        else:
            launch()
    if defer:
        base = ws.data_ptr()
        for t, off, ne in ((dW1, 0, f_mid * f_in), (db1, lay[2], f_mid), (dW2, lay[3], f_out * f_mid), (db2, lay[4], f_out)):
            current_step().add_partials(t, (ws, base + 4 * off, lay[0], lay[1], ne, ne))
    return dW1, db1, dW2, db2


def linear_fwd_raw(M, W, b, act):
    M, ldm = _rowmajor(_f32(M, "linear: M"), "M")
    W = _f32(_gpu(W, "W"), "linear: W").contiguous()
    _f32(b, "linear: b")
    n, f_in = M.shape
    f_out = W.shape[0]
    Y = torch.empty(n, f_out, dtype=torch.float32, device=M.device)
    with _on_device(M.device):
        nbytes = _lib.load().gae_linear_fwd_workspace_bytes(n, f_in, f_out)
        ws = _workspace(nbytes, M.device) if nbytes > 0 else None
        _lib.call("gae_linear_fwd", _ptr(M), ldm, n, f_in, _ptr(W), _ptr(b), f_out, act, _ptr(Y), max(f_out, 1),
                  _ptr(ws), ws.numel() if ws is not None else 0, _stream())
    return Y


def linear_bwd_raw(dY, Y, act, M, W, need_dW=True, need_db=True, need_dM=True, f_out=None):
    dY, lddy = _rowmajor(_f32(dY, "linear backward: dY"), "dY")
    M, ldm = _rowmajor(_f32(M, "linear backward: M"), "M")
    if W is not None:
        W = _f32(W, "linear backward: W").contiguous()
    elif need_dM or f_out is None:
        raise GaeHipError("linear backward: dM needs W (and f_out must be given without it)")
    _f32(Y, "linear backward: Y")
    n, f_in = M.shape
    f_out = W.shape[0] if W is not None else int(f_out)
    dev = M.device
    dW = torch.empty(f_out, f_in, dtype=torch.float32, device=dev) if need_dW else None
    db = torch.empty(f_out, dtype=torch.float32, device=dev) if need_db else None
    dM = torch.empty(n, f_in, dtype=torch.float32, device=dev) if need_dM else None
    ldy = 0
    if Y is not None:
        Y, ldy = _rowmajor(Y, "Y")
    with _on_device(dev):
        if current_step().defer_grads and n > 0 and f_in > 0 and (dW is not None or db is not None):
            # (dW, db) stay partial sums for the optimiser launch; dM, if wanted, comes from the ordinary entry point
            wsp = torch.empty(_lib.load().gae_linear_bwd_workspace_bytes(n, f_in, f_out), dtype=torch.uint8, device=dev)
            lay = (ctypes.c_int64 * 4)()
            _lib.call("gae_x_linear_bwd_partials", _ptr(dY), lddy, _ptr(Y), ldy, act, _ptr(M), ldm, n, f_in, f_out,
                      int(dW is not None), int(db is not None), _ptr(wsp), wsp.numel(), lay, _stream())
            if dW is not None:
                current_step().add_partials(dW, (wsp, wsp.data_ptr(), lay[0], lay[1], f_out * f_in, f_out * f_in))
            if db is not None:
                current_step().add_partials(db, (wsp, wsp.data_ptr() + 4 * lay[2], lay[0], lay[1], f_out, f_out))
            if dM is None:
                return dW, db, dM
            ws = _workspace(_lib.load().gae_linear_bwd_workspace_bytes(n, f_in, f_out), dev)
            _lib.call("gae_linear_bwd", _ptr(dY), lddy, _ptr(Y), ldy, act, _ptr(M), ldm, _ptr(W), n, f_in, f_out,
                      None, None, _ptr(dM), max(f_in, 1), _ptr(ws), ws.numel(), _stream())
            return dW, db, dM
        ws = _workspace(_lib.load().gae_linear_bwd_workspace_bytes(n, f_in, f_out), dev)
        _lib.call("gae_linear_bwd", _ptr(dY), lddy, _ptr(Y), ldy, act, _ptr(M), ldm, _ptr(W), n, f_in, f_out,
                  _ptr(dW), _ptr(db), _ptr(dM), max(f_in, 1), _ptr(ws), ws.numel(), _stream())
    return dW, db, dM


def dropout_mask(shape, p, seed, offset=0, device="cuda", draw_counter=None):
    """inverted-dropout multiplier; ``draw_counter`` (int64 device tensor [1]) selects the
    draw on the device so that captured HIP graphs advance the stream between replays"""
    mask = torch.empty(shape, dtype=torch.float32, device=device)
    with _on_device(mask.device):
        _lib.call("gae_dropout_mask", _ptr(mask), mask.numel(), float(p), int(seed) & (2 ** 64 - 1),
                  int(offset) & (2 ** 64 - 1), _ptr(draw_counter), _stream())
    return mask


def normal_noise(shape, seed, offset=0, device="cuda", draw_counter=None):
    """eps ~ N(0, 1) from the library's Philox + Box-Muller generator (VGAE reparameterisation)"""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    with _on_device(out.device):
        _lib.call("gae_normal_noise", _ptr(out), out.numel(), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
                  _ptr(draw_counter), _stream())
    return out


class VGAEHeadFunction(torch.autograd.Function):
    """z = mu + eps exp(logstd) and the KL term of Kipf & Welling's VGAE, fused (gae_vgae_head_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, mu, logstd, eps):
        mu = _gpu(mu, "mu").contiguous(); logstd = logstd.contiguous(); eps = eps.contiguous()
        n, d = mu.shape
        z = torch.empty_like(mu)
        kl = torch.empty(1, dtype=torch.float32, device=mu.device)
        with _on_device(mu.device):
            ws = _workspace(_lib.load().gae_vgae_head_workspace_bytes(n * d), mu.device)
            _lib.call("gae_vgae_head_fwd", _ptr(mu), _ptr(logstd), d, _ptr(eps), n, d, _ptr(z), _ptr(kl), _ptr(ws),
                      ws.numel(), _stream())
        ctx.save_for_backward(mu, logstd, eps)
        return z, kl.reshape(())

    @staticmethod
    def backward(ctx, dz, dkl):
        mu, logstd, eps = ctx.saved_tensors
        n, d = mu.shape
        dmu = torch.empty_like(mu); dls = torch.empty_like(mu)
        dz = None if dz is None else dz.contiguous()
        gkl = (torch.zeros(1, device=mu.device) if dkl is None else dkl.reshape(1).float().contiguous())
        with _on_device(mu.device):
            _lib.call("gae_vgae_head_bwd", _ptr(dz), _ptr(mu), _ptr(logstd), d, _ptr(eps), _ptr(gkl), n, d, _ptr(dmu),
                      _ptr(dls), _stream())
        return dmu, dls, None


class VGAEPackedHeadFunction(torch.autograd.Function):
    """the same on both heads PACKED in one [n, 2 d] matrix [mu | logstd] (what GCNTwoHeadFunction produces): one
    gradient matrix goes back, no slicing / concatenation kernels in between"""

    @staticmethod
    def forward(ctx, ml, eps):
        ml = _gpu(ml, "ml").contiguous(); eps = eps.contiguous()
        n, d2 = ml.shape
        d = d2 // 2
        z = torch.empty(n, d, dtype=torch.float32, device=ml.device)
        kl = torch.empty(1, dtype=torch.float32, device=ml.device)
        with _on_device(ml.device):
            ws = _workspace(_lib.load().gae_vgae_head_workspace_bytes(n * d), ml.device)
            _lib.call("gae_vgae_head_fwd", _ptr(ml), _vp(ml.data_ptr() + 4 * d), d2, _ptr(eps), n, d, _ptr(z), _ptr(kl),
                      _ptr(ws), ws.numel(), _stream())
        ctx.save_for_backward(ml, eps)
        return z, kl.reshape(())

    @staticmethod
    def backward(ctx, dz, dkl):
        ml, eps = ctx.saved_tensors
        n, d2 = ml.shape
        d = d2 // 2
        dml = torch.empty_like(ml)
        dz = None if dz is None else dz.contiguous()
        gkl = (torch.zeros(1, device=ml.device) if dkl is None else dkl.reshape(1).float().contiguous())
        with _on_device(ml.device):
            _lib.call("gae_vgae_head_bwd", _ptr(dz), _ptr(ml), _vp(ml.data_ptr() + 4 * d), d2, _ptr(eps), _ptr(gkl), n, d,
                      _ptr(dml), _vp(dml.data_ptr() + 4 * d), _stream())
        return dml, None


def vgae_head_packed(ml, eps):
    return VGAEPackedHeadFunction.apply(ml, eps)


VGAE_FUSED_LOSS = os.environ.get("GAE_VGAE_FUSED_LOSS", "1") != "0"


class VGAEHeadLossFunction(torch.autograd.Function):
    """The VGAE head, the KL term and the fused reconstruction loss on the PACKED heads [mu | logstd] (d = 16) as
    three launches: gae_x_vgae_head_prep (noise of this draw, z, KL partials, the loss's prepare step), then the dense
    and the edge kernel of gae_x_decoder_bce_prepared; the scalar rec + KL comes out of the loss's final reduction,
    which also adds the KL partials (gae_bce_tail::kl_*) -- inside ``deferred_loss_finalize()`` as one block of the
    optimiser launch.  Replaces gae_normal_noise, gae_vgae_head_fwd (2 launches), the prepare and final-reduction
    launches of the loss, the draw-counter increment and the ``rec + kl`` addition: 5 launches instead of 12.
    Returns (loss, z, kl, rec, eps); only ``loss`` carries a gradient (to ``ml``)."""

    @staticmethod
    def forward(ctx, ml, graph, eps, noise):
        ml = _f32(_gpu(ml, "ml"), "vgae loss: ml").contiguous()
        n, d2 = ml.shape
        d = d2 // 2
        dev = ml.device
        draw = eps is None
        seed, offset, draws = noise if noise is not None else (0, 0, None)
        eps_t = torch.empty(n, d, dtype=torch.float32, device=dev) if draw else _f32(_gpu(eps, "eps"), "eps").contiguous()
        z = torch.empty(n, d, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad[0]
        nnz = graph.number_of_edges()
        pw = (float(n) * float(n) - float(nnz)) / float(nnz)
        indptr, indices = graph.csr()
        t_indptr, t_indices = graph.csc() if need else (None, None)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        kl = torch.empty(1, dtype=torch.float32, device=dev)
        rec = torch.empty(1, dtype=torch.float32, device=dev)
        dZ = torch.empty(n, d, dtype=torch.float32, device=dev) if need else None
        with _on_device(dev):
            nbytes = _lib.load().gae_decoder_bce_workspace_bytes(n, n, d)
            if nbytes < 0:
                _lib.check(int(nbytes), "gae_decoder_bce_workspace_bytes")
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            klp = torch.empty((n + 63) // 64, dtype=torch.float64, device=dev)
            lay = _lib.BcePrep()
            _lib.call("gae_x_decoder_bce_prep_layout", n, d, _ptr(ws), ws.numel(), ctypes.byref(lay))
            blocks = ctypes.c_int64(0)
            if current_step().tails:
                current_step().flush_loss_tails()
            _lib.call("gae_x_vgae_head_prep", _ptr(ml), _vp(ml.data_ptr() + 4 * d), d2, _ptr(eps_t), 1 if draw else 0,
                      int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _ptr(draws) if draw else None, n, d, _ptr(z),
                      ctypes.byref(lay), _ptr(klp), klp.numel(), ctypes.byref(blocks), _stream())
            tail = _lib.BceTail()
            _lib.call("gae_x_decoder_bce_defer_finalize", ctypes.byref(tail))
            try:
                STATS["prepared_losses"] += 1
                _lib.call("gae_x_decoder_bce_prepared", None, d, n, d, _ptr(indptr), _ptr(indices), _ptr(t_indptr),
                          _ptr(t_indices), float(pw), None, 0.0, None, int(blocks.value), _ptr(loss), _ptr(dZ), d, _ptr(ws),
                          ws.numel(), _stream())
            except Exception:
                _lib.call("gae_x_decoder_bce_defer_finalize", None)
                raise
            tail.kl_partial = klp.data_ptr(); tail.n_kl = int(blocks.value); tail.kl_scale = -0.5 / (float(n) * float(n))
            tail.kl_out = kl.data_ptr(); tail.rec_out = rec.data_ptr()
            if draw and draws is not None:
                tail.bump_draw = draws.data_ptr()          # the noise counter advances with the loss's last block
            keep = (loss, ws, klp, kl, rec, draws)
            if current_step().defer_loss and need:
                current_step().tails.append((tail, keep))
            else:
                _lib.call("gae_x_decoder_bce_finalize", ctypes.byref(tail), _stream())
        ctx.save_for_backward(ml, eps_t, dZ)
        kl0, rec0 = kl.reshape(()), rec.reshape(())
        ctx.mark_non_differentiable(z, kl0, rec0, eps_t)     # (the returned objects themselves: autograd would otherwise
        ctx.set_materialize_grads(False)                      #  fill a zero gradient for each of them in every backward)
        return loss.reshape(()), z, kl0, rec0, eps_t

    @staticmethod
    def backward(ctx, g, *unused):
        ml, eps, dZ = ctx.saved_tensors
        n, d2 = ml.shape
        d = d2 // 2
        dml = torch.empty_like(ml)
        unit = _is_unit(g)
        dz = dZ if unit else dZ * g
        gkl = g.reshape(1).float().contiguous()              # d loss / d kl = the upstream gradient
        with _on_device(ml.device):
            _lib.call("gae_vgae_head_bwd", _ptr(dz), _ptr(ml), _vp(ml.data_ptr() + 4 * d), d2, _ptr(eps), _ptr(gkl), n, d,
                      _ptr(dml), _vp(dml.data_ptr() + 4 * d), _stream())
        return dml, None, None, None


def vgae_head_loss(ml, graph, eps=None, noise=None):
    """(loss, z, kl, rec, eps) -- see VGAEHeadLossFunction; None when the fused form does not apply (d != 16, no
    edges, fixed-capacity batch)"""
    if (not VGAE_FUSED_LOSS or not isinstance(ml, torch.Tensor) or not ml.is_cuda or ml.dim() != 2 or ml.shape[1] != 32
            or ml.dtype != torch.float32 or graph.number_of_edges() == 0
            or getattr(graph, "batch_counts", None) is not None or ml.shape[0] != graph.number_of_nodes()):
        return None
    return VGAEHeadLossFunction.apply(ml, graph, eps, noise)


def vgae_head(mu, logstd, eps):
    return VGAEHeadFunction.apply(mu, logstd, eps)


def decoder_dense_raw(Z, mask=None):
    Z, ldz = _rowmajor(_f32(Z, "decoder: Z"), "Z")
    _f32(mask, "decoder: mask")
    if mask is not None:
        mask = _gpu(mask, "mask").contiguous()
        if Z.stride(0) != mask.stride(0) and Z.shape[0] > 1:
            Z = Z.contiguous(); ldz = max(Z.shape[1], 1)
    n, d = Z.shape
    out = torch.empty(n, n, dtype=torch.float32, device=Z.device)
    with _on_device(Z.device):
        _lib.call("gae_decoder_dense", _ptr(Z), _ptr(mask), ldz, n, d, _ptr(out), max(n, 1), _stream())
    return out


def decoder_dense_bwd_raw(G, Z, mask=None):
    G, ldg = _rowmajor(_f32(G, "decoder backward: G"), "G")
    Z = _f32(_gpu(Z, "Z"), "decoder backward: Z").contiguous()
    _f32(mask, "decoder backward: mask")
    if mask is not None:
        mask = mask.contiguous()
    n, d = Z.shape
    dZ = torch.empty(n, d, dtype=torch.float32, device=Z.device)
    with _on_device(Z.device):
        ws = _workspace(_lib.load().gae_decoder_dense_bwd_workspace_bytes(n, d), Z.device)
        _lib.call("gae_decoder_dense_bwd", _ptr(G), ldg, _ptr(Z), _ptr(mask), max(d, 1), n, d, _ptr(dZ), max(d, 1),
                  _ptr(ws), ws.numel(), _stream())
    return dZ


def decoder_bce_raw(Z, mask, csr, csc, pos_weight, want_grad=True, row_begin=0, n_local=None, dropout=None,
                    counts=None, defer_ok=False, prepared=None):
    """fused decoder + weighted BCE (mean): returns (loss[1], dZ or None).
    ``counts`` (int64[2] on the device: true {nodes, edges}) = Z / csr are a fixed-capacity batch
    (gae_decoder_bce_padded): pos_weight and the mean come from the counts, ``pos_weight`` is ignored.
    ``row_begin/n_local`` select a row window (row-sharded form): Z/mask stay the
    full [n, d] arrays, csr/csc are the window's local row blocks.
    ``dropout`` = (p, seed, offset, draw_counter): the mask of this draw is generated inside the launch, written
    to ``mask`` (an [n, d] output buffer then) and the device draw counter is advanced by the library.
    ``defer_ok``: inside ``deferred_loss_finalize()`` the final reduction may be left to the optimiser launch -- the
    returned scalar is then NOT valid before ``optim.Adam.step()`` (or the end of the block) has run.
    ``prepared``: token of gcn_layer_fused_prep_raw -- the producer of Z already ran the prepare step (mask drawn,
    workspace filled): mask / dropout / counts are the token's."""
    Z = _f32(_gpu(Z, "Z"), "decoder_bce: Z").contiguous()
    if prepared is not None:
        if prepared["z_ptr"] != Z.data_ptr() or tuple(Z.shape) != (prepared["n"], prepared["d"]) or row_begin or \
                (n_local is not None and n_local != Z.shape[0]):
            raise GaeHipError("decoder_bce: the prepared workspace belongs to another embedding")
        mask, dropout, counts = prepared["mask"], prepared["dropout"], prepared["counts"]
    if mask is not None:
        mask = _f32(_gpu(mask, "mask"), "decoder_bce: mask").contiguous()
    p_drop, seed, offset, draws = dropout if dropout is not None else (0.0, 0, 0, None)
    if p_drop and (mask is None or mask.shape != Z.shape):
        raise GaeHipError("decoder_bce: in-kernel dropout needs an [n, d] mask output buffer")
    n, d = Z.shape
    n_local = n if n_local is None else int(n_local)
    dev = Z.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dZ = torch.empty(n_local, d, dtype=torch.float32, device=dev) if want_grad else None
    indptr, indices = csr
    t_indptr, t_indices = csc if csc is not None else (None, None)
    with _on_device(dev):
        if prepared is not None:
            ws = prepared["ws"]
        else:
            nbytes = _lib.load().gae_decoder_bce_workspace_bytes(n, n_local, d)
            if nbytes < 0:
                _lib.check(int(nbytes), "gae_decoder_bce_workspace_bytes")
            # a deferred final reduction reads the partial sums in the optimiser launch: they must not sit in the
            # per-stream scratch cache, which any launch in between (dM of gae_linear_bwd, a weight-gradient
            # reduction) may hand out again
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev) if (current_step().defer_loss and defer_ok and want_grad) \
                else _workspace(nbytes, dev)

        def launch():
            tail = None
            if current_step().tails:
                current_step().flush_loss_tails()                 # an earlier loss nobody took: its partial sums may live in the
                                                   # cached workspace this call is about to reuse
            if current_step().defer_loss and defer_ok and want_grad:
                tail = _lib.BceTail()
                _lib.call("gae_x_decoder_bce_defer_finalize", ctypes.byref(tail))
            try:
                launch_kernels()
            except Exception:
                if tail is not None:
                    _lib.call("gae_x_decoder_bce_defer_finalize", None)
                raise
            if tail is not None:
                current_step().tails.append((tail, (loss, ws, draws, counts)))

        def launch_kernels():
            if prepared is not None:
                STATS["prepared_losses"] += 1
                _lib.call("gae_x_decoder_bce_prepared", _ptr(mask), max(d, 1), n, d, _ptr(indptr), _ptr(indices),
                          _ptr(t_indptr), _ptr(t_indices), float(pos_weight), _ptr(counts), float(p_drop), _ptr(draws),
                          prepared["blocks"], _ptr(loss), _ptr(dZ), max(d, 1), _ptr(ws), ws.numel(), _stream())
                return
            if counts is not None:
                if row_begin or n_local != n:
                    raise GaeHipError("decoder_bce: a fixed-capacity batch has no row window")
                _lib.call("gae_decoder_bce_padded", _ptr(Z), _ptr(mask), max(d, 1), n, d, _ptr(indptr),
                          _ptr(indices), _ptr(t_indptr), _ptr(t_indices), _ptr(counts), float(p_drop),
                          int(seed) & (2 ** 64 - 1), int(offset), _ptr(draws), _ptr(loss), _ptr(dZ), max(d, 1),
                          _ptr(ws), ws.numel(), _stream())
                return
            _lib.call("gae_decoder_bce_rows", _ptr(Z), _ptr(mask), max(d, 1), n, d, int(row_begin), n_local,
                      _ptr(indptr), _ptr(indices), _ptr(t_indptr), _ptr(t_indices), float(pos_weight), float(p_drop),
                      int(seed) & (2 ** 64 - 1), int(offset), _ptr(draws), _ptr(loss), _ptr(dZ), max(d, 1), _ptr(ws),
                      ws.numel(), _stream())
        if profiler is not None:
            profiler.wrap(("decoder_bce", n, d, want_grad), launch)
        else:
            launch()
    return loss, dZ


# ------------------------------------------------------------------ autograd glue
def _scattered(graph, H):
    return H.shape[1] > TILE_MIN_F and graph.scattered(H.shape[1] * H.element_size())


class SpMMFunction(torch.autograd.Function):
    """update_all(copy_src, sum) with its backward  dH = A^T dM  (gae.py:28).
    The backward's operands (CSR of A^T, plan, norm) are taken from the graph in forward(): the autograd node must
    not hold the graph itself -- the graph holds the output (``g.ndata['h']``), whose grad_fn would hold the graph
    again, a cycle only the garbage collector can free (batches of an eager epoch would pile up in HBM until it
    runs, and their AccumulateGrad nodes would stay bound to the stream of a long-finished iteration)."""

    @staticmethod
    def forward(ctx, H, graph, use_norm):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        sc = _scattered(graph, H)
        if ctx.needs_input_grad[0]:
            ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True), graph.block_diag, sc)
        return spmm_raw(indptr, indices, H, n, norm, norm, plan=graph.spmm_plan(False),
                        blockdiag=graph.block_diag, scattered=sc)

    @staticmethod
    def backward(ctx, dM):
        (t_indptr, t_indices), n, norm, plan_t, blockdiag, sc = ctx.bwd
        return spmm_raw(t_indptr, t_indices, dM, n, norm, norm, plan=plan_t, blockdiag=blockdiag,
                        scattered=sc), None, None


FUSED_LAYER_MAX_IN, FUSED_LAYER_MAX_OUT = 64, 32      # gae_gcn_layer_fused: whole row in one lane group, <= 32 outputs


def gcn_layer_fused_usable(H, n_out, plan):
    """can gae_gcn_layer_fused run this layer?  fp32 rows of <= 64 features made of whole 16-byte vectors, <= 32
    outputs, a plan that carries a packed neighbour table and neither heavy nor XCD-pinned rows (their table rows
    are skip markers: the fused kernel would leave them unwritten)"""
    return (plan is not None and plan.ell is not None and plan.n_heavy == 0 and plan.homed is None
            and H.dtype == torch.float32
            and H.dim() == 2 and 1 <= H.shape[1] <= FUSED_LAYER_MAX_IN and 1 <= n_out <= FUSED_LAYER_MAX_OUT
            and H.shape[0] > 0 and H.stride(1) == 1 and H.stride(0) % 4 == 0 and H.data_ptr() % 16 == 0
            and H.shape[0] * H.stride(0) * 4 + (1 << 16) < (1 << 32))


def gcn_layer_fused_raw(indptr, indices, H, n_rows, plan, W, bias, act, row_scale=None, col_scale=None,
                        w_transposed=False, want_m=True):
    """(M or None, Y): the aggregation of spmm_raw and Y = act(M W^T + b) in one launch (gae_gcn_layer_fused).
    ``w_transposed``: use W^T, i.e. Y = M W for W [F, J] stored as nn.Linear keeps it ([out = F][in = J]) -- the
    backward form dH = (A^T dY) W."""
    H, ldh = _rowmajor(_f32(_gpu(H, "H"), "gcn_layer_fused: H"), "H")
    W = _f32(_gpu(W, "W"), "gcn_layer_fused: W")
    if W.stride(1) != 1:
        W = W.contiguous()
    n_cols, F = H.shape
    if w_transposed:
        J, so, sk = W.shape[1], 1, W.stride(0)
        if W.shape[0] != F:
            raise GaeHipError("gcn_layer_fused: weight shape does not match the features")
    else:
        J, so, sk = W.shape[0], W.stride(0), 1
        if W.shape[1] != F:
            raise GaeHipError("gcn_layer_fused: weight shape does not match the features")
    _f32(bias, "gcn_layer_fused: bias")
    M = torch.empty(n_rows, padded_ld(F, torch.float32), dtype=torch.float32, device=H.device)[:, :F] if want_m else None
    Y = torch.empty(n_rows, J, dtype=torch.float32, device=H.device)
    with _on_device(H.device):
        def launch():
            _lib.call("gae_gcn_layer_fused", _ptr(indptr), _ptr(indices), n_rows, n_cols, _ptr(H), ldh, _ptr(M),
                      M.stride(0) if M is not None else 0, F, _ptr(row_scale), _ptr(col_scale), ctypes.byref(plan.c),
                      _ptr(W), so, sk, _ptr(bias), J, int(act), _ptr(Y), max(J, 1), _stream())
        if profiler is not None:
            profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch)
        else:
            launch()
    return M, Y


# ---- the loss's prepare step in the epilogue of the layer that produces Z --------------------------------------
# ``with loss_prepare_request(graph, d, mask, dropout) as req:`` around the LAST encoder layer: if that layer runs as
# the fused launch (GCNLayerFusedFunction, identity activation, <= 16 outputs), the launch also writes Zt / hi / lo /
# column sums (and draws the dropout mask) into a loss workspace, and ``req.token`` describes it for
# ``decoder_bce(..., prepared=req.token)`` -- the loss then starts at its dense kernel (one kernel node fewer per step).
FUSE_LOSS_PREPARE = os.environ.get("GAE_FUSE_LOSS_PREPARE", "1") != "0"
STATS = {"prepared_losses": 0,      # losses that started at the dense kernel (tests read this)
         "xw_fwd": 0, "xw_wgrad": 0}  # launches of the one-pass layer-1 kernels (transform-first order)


class loss_prepare_request:
    def __init__(self, graph, d, mask, dropout):
        """``mask``: a given [n, d] multiplier (or None); ``dropout`` = (p, seed, offset, draw_counter) to draw one"""
        self.graph, self.d, self.mask, self.dropout, self.token = graph, int(d), mask, dropout, None

    def __enter__(self):
        self.step = current_step()             # the request belongs to the step it was made in
        self.prev = self.step.prep_req
        self.step.prep_req = self if FUSE_LOSS_PREPARE and self.d <= 16 else None
        return self

    def __exit__(self, *exc):
        self.step.prep_req = self.prev


def gcn_layer_fused_prep_raw(indptr, indices, H, n, plan, W, bias, row_scale, req, want_m=True):
    """(M or None, Z, token): gae_x_gcn_layer_fused_prep -- the fused layer (identity activation) with the prepare step
    of the loss in its epilogue; ``token`` goes to decoder_bce_raw(prepared=...)"""
    H, ldh = _rowmajor(_f32(_gpu(H, "H"), "gcn_layer_fused_prep: H"), "H")
    W = _f32(_gpu(W, "W"), "gcn_layer_fused_prep: W")
    if W.stride(1) != 1:
        W = W.contiguous()
    _f32(bias, "gcn_layer_fused_prep: bias")
    F, J = H.shape[1], W.shape[0]
    dev = H.device
    M = torch.empty(n, padded_ld(F, torch.float32), dtype=torch.float32, device=dev)[:, :F] if want_m else None
    Z = torch.empty(n, J, dtype=torch.float32, device=dev)
    p_drop, seed, offset, draws = req.dropout if req.dropout is not None else (0.0, 0, 0, None)
    mask = req.mask
    if p_drop:
        mask = torch.empty(n, J, dtype=torch.float32, device=dev)
    elif mask is not None:
        mask = _f32(_gpu(mask, "mask"), "gcn_layer_fused_prep: mask").contiguous()
    counts = getattr(req.graph, "batch_counts", None)
    with _on_device(dev):
        nbytes = _lib.load().gae_decoder_bce_workspace_bytes(n, n, J)
        if nbytes < 0:
            _lib.check(int(nbytes), "gae_decoder_bce_workspace_bytes")
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)       # lives until the loss has run: not the scratch cache
        lay = _lib.BcePrep()
        _lib.call("gae_x_decoder_bce_prep_layout", n, J, _ptr(ws), ws.numel(), ctypes.byref(lay))
        blocks = ctypes.c_int64(0)

        def launch():
            _lib.call("gae_x_gcn_layer_fused_prep", _ptr(indptr), _ptr(indices), n, _ptr(H), ldh, _ptr(M),
                      M.stride(0) if M is not None else 0, F, _ptr(row_scale), _ptr(row_scale), ctypes.byref(plan.c),
                      _ptr(W), W.stride(0), 1, _ptr(bias), J, _ptr(Z), J, ctypes.byref(lay), _ptr(mask), J, float(p_drop),
                      int(seed) & (2 ** 64 - 1), int(offset), _ptr(draws), _ptr(counts), ctypes.byref(blocks), _stream())
        if profiler is not None:
            profiler.wrap(("spmm", n, n, F, str(H.dtype)), launch)
        else:
            launch()
    token = {"ws": ws, "blocks": int(blocks.value), "mask": mask, "z_ptr": Z.data_ptr(), "n": n, "d": J,
             "dropout": req.dropout, "counts": counts}
    return M, Z, token


def gcn_layer_fused_wgrad_raw(t_indptr, t_indices, dY, n, plan_t, W, M, norm, want_dW=True, want_db=True):
    """(dH, dW, db): the identity-activation backward of the fused layer in one launch (gae_x_gcn_layer_fused_wgrad).
    Inside ``deferred_grad_reductions()`` dW / db are left as per-block partial sums for optim.Adam.step()."""
    dY, lddy = _rowmajor(_f32(_gpu(dY, "dY"), "gcn_layer_fused_wgrad: dY"), "dY")
    W = W if W.stride(1) == 1 else W.contiguous()
    f_out, f_in = W.shape
    if dY.shape[1] != f_out or M.shape[1] != f_in or M.stride(1) != 1:
        raise GaeHipError("gcn_layer_fused_wgrad: operand shapes do not match the weight")
    dev = dY.device
    dH = torch.empty(n, f_in, dtype=torch.float32, device=dev)
    dW = torch.empty(f_out, f_in, dtype=torch.float32, device=dev) if want_dW else None
    db = torch.empty(f_out, dtype=torch.float32, device=dev) if want_db else None
    defer = current_step().defer_grads and (want_dW or want_db)
    with _on_device(dev):
        nbytes = _lib.load().gae_x_gcn_layer_fused_wgrad_workspace_bytes(n, f_out, f_in)
        if nbytes < 0:
            _lib.check(int(nbytes), "gae_x_gcn_layer_fused_wgrad_workspace_bytes")
        # deferred partials outlive the call: they must not sit in the per-stream scratch cache
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev) if defer else _workspace(nbytes, dev)
        lay = (ctypes.c_int64 * 3)()

        def launch():
            _lib.call("gae_x_gcn_layer_fused_wgrad", _ptr(t_indptr), _ptr(t_indices), n, _ptr(dY), lddy, f_out, _ptr(norm),
                      _ptr(norm), ctypes.byref(plan_t.c), _ptr(W), W.stride(0), f_in, _ptr(dH), f_in, _ptr(M),
                      M.stride(0), None if defer else _ptr(dW), None if defer else _ptr(db), _ptr(ws), ws.numel(), lay,
                      _stream())
        if profiler is not None:
            profiler.wrap(("spmm", n, n, f_out, str(dY.dtype)), launch)
        else:
            launch()
    if defer:
        if dW is not None:
            current_step().add_partials(dW, (ws, ws.data_ptr(), lay[0], lay[1], f_out * f_in, f_out * f_in))
        if db is not None:
            current_step().add_partials(db, (ws, ws.data_ptr() + 4 * lay[2], lay[0], lay[1], f_out, f_out))
    return dH, dW, db


FUSED_LAYER_WGRAD = os.environ.get("GAE_FUSED_LAYER_WGRAD", "1") != "0"      # False: the fused layer's backward keeps its separate weight-gradient launch (experiments)


class GCNLayerFusedFunction(torch.autograd.Function):
    """GCN.forward (gae.py:26-31) as one launch: Y = act((A H) W^T + b).  Backward: dW = dYm^T M and db from the
    stored aggregate (gae_linear_bwd), and dH = A^T (dYm W) -- for an identity activation as ONE launch of the
    same kernel on the CSR of A^T, dH = (A^T dY) W (the aggregation then runs at the output width)."""

    @staticmethod
    def forward(ctx, H, W, b, graph, use_norm, act):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        need_w = ctx.needs_input_grad[1] or (b is not None and ctx.needs_input_grad[2])
        req = current_step().prep_req
        if (req is not None and req.token is None and req.graph is graph and act == ACT_IDENTITY
                and W.shape[0] == req.d and H.shape[0] == n and n > 0):
            # the last encoder layer of a training step: the loss's prepare step rides in this launch's epilogue
            M, Y, req.token = gcn_layer_fused_prep_raw(indptr, indices, H, n, graph.spmm_plan(False), W, b, norm, req,
                                                       want_m=need_w)
        else:
            M, Y = gcn_layer_fused_raw(indptr, indices, H, n, graph.spmm_plan(False), W, b, act, norm, norm,
                                       want_m=need_w)
        ctx.act, ctx.has_bias = act, b is not None
        if ctx.needs_input_grad[0]:
            ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True), graph.block_diag, _scattered(graph, H))
        ctx.save_for_backward(M, W, Y if act == ACT_RELU else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        M, W, Y = ctx.saved_tensors
        need_dH, need_dW = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        dW = db = dH = None
        fused_bwd = need_dH and ctx.act == ACT_IDENTITY and gcn_layer_fused_usable(dY.contiguous(), W.shape[1],
                                                                                 ctx.bwd[3])
        if fused_bwd and FUSED_LAYER_WGRAD and need_dW and M is not None and W.shape[0] <= 32 and W.shape[1] <= 32:
            # dH, dW and db from ONE launch: the blocks of the backward gather also add up dY^T M over their own rows
            (t_indptr, t_indices), n, norm, plan_t, _, _ = ctx.bwd
            dH, dW, db = gcn_layer_fused_wgrad_raw(t_indptr, t_indices, dY.contiguous(), n, plan_t, W, M, norm,
                                                   True, need_db)
            return dH, dW, db, None, None, None
        if need_dW or need_db or (need_dH and not fused_bwd):
            dW, db, dM = linear_bwd_raw(dY, Y, ctx.act, M if M is not None else dY.new_zeros(dY.shape[0], W.shape[1]),
                                        W, need_dW, need_db, need_dH and not fused_bwd)
        if need_dH:
            (t_indptr, t_indices), n, norm, plan_t, blockdiag, sc = ctx.bwd
            if fused_bwd:
                _, dH = gcn_layer_fused_raw(t_indptr, t_indices, dY.contiguous(), n, plan_t, W, None, ACT_IDENTITY,
                                            norm, norm, w_transposed=True, want_m=False)
            else:
                dH = spmm_raw(t_indptr, t_indices, dM, n, norm, norm, plan=plan_t, blockdiag=blockdiag, scattered=sc)
        return dH, dW, db, None, None, None


def _split_pending(t, rows):
    """a deferred gradient ``t`` [R, ...] handed out as the two row blocks t[:rows], t[rows:]: register the partial
    lists of the halves (same workspace, second one offset)"""
    step = current_step()
    ent = step.take_partials(t)
    if ent is None:
        return
    ws, ptr, n_part, stride, _, _ = ent
    per_row = t[0].numel() if t.dim() > 1 else 1
    n0, n1 = rows * per_row, (t.shape[0] - rows) * per_row
    current_step().add_partials(t[:rows], (ws, ptr, n_part, stride, max(n0, 1), max(n0, 1)))
    current_step().add_partials(t[rows:], (ws, ptr + 4 * n0, n_part, stride, max(n1, 1), max(n1, 1)))


class GCNTwoHeadFunction(torch.autograd.Function):
    """two identity-activation GCN layers on the same input (VGAE's mu and log sigma heads) as ONE fused launch
    (gae_x_gcn_layer_fused2): ML = [(A H) W1^T + b1 | (A H) W2^T + b2].  Backward: one dW launch for both heads
    (dML^T M, split by rows), one fused launch dH = (A^T dML) [W1; W2] -- instead of two of each plus an add."""

    @staticmethod
    def forward(ctx, H, W1, b1, W2, b2, graph, use_norm):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        plan = graph.spmm_plan(False)
        Hc, ldh = _rowmajor(_f32(_gpu(H, "H"), "two heads: H"), "H")
        W1 = W1 if W1.stride(1) == 1 else W1.contiguous()
        W2 = W2 if W2.stride(1) == 1 and W2.stride(0) == W1.stride(0) else W2.contiguous()
        F, d1, d2 = Hc.shape[1], W1.shape[0], W2.shape[0]
        need_w = any(ctx.needs_input_grad[1:5])
        M = torch.empty(n, padded_ld(F, torch.float32), dtype=torch.float32, device=Hc.device)[:, :F] if need_w else None
        Y = torch.empty(n, d1 + d2, dtype=torch.float32, device=Hc.device)
        with _on_device(Hc.device):
            _lib.call("gae_x_gcn_layer_fused2", _ptr(indptr), _ptr(indices), n, Hc.shape[0], _ptr(Hc), ldh, _ptr(M),
                      M.stride(0) if M is not None else 0, F, _ptr(norm), _ptr(norm), ctypes.byref(plan.c), _ptr(W1),
                      _ptr(W2), d1, 0, W1.stride(0), 1, _ptr(b1), _ptr(b2), d1 + d2, ACT_IDENTITY, _ptr(Y), d1 + d2,
                      _stream())
        ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True), d1, d2, b1 is not None)
        ctx.save_for_backward(M, W1, W2)
        return Y

    @staticmethod
    def backward(ctx, dY):
        M, W1, W2 = ctx.saved_tensors
        (t_indptr, t_indices), n, norm, plan_t, d1, d2, has_bias = ctx.bwd
        need_dH = ctx.needs_input_grad[0]
        need_dW = ctx.needs_input_grad[1] or ctx.needs_input_grad[3]
        need_db = has_bias and (ctx.needs_input_grad[2] or ctx.needs_input_grad[4])
        dW1 = db1 = dW2 = db2 = dH = None
        dYc = dY.contiguous()
        if (FUSED_LAYER_WGRAD and need_dH and need_dW and M is not None and d1 + d2 <= 32 and W1.shape[1] <= 32
                and M.stride(1) == 1):
            # dH, dW and db of both heads from ONE launch (gae_x_gcn_layer_fused2_wgrad: side work of the gather's blocks)
            f_out, f_in = d1 + d2, W1.shape[1]
            dev = dYc.device
            dH = torch.empty(n, f_in, dtype=torch.float32, device=dev)
            dW = torch.empty(f_out, f_in, dtype=torch.float32, device=dev)
            db = torch.empty(f_out, dtype=torch.float32, device=dev) if need_db else None
            defer = current_step().defer_grads
            with _on_device(dev):
                nbytes = _lib.load().gae_x_gcn_layer_fused_wgrad_workspace_bytes(n, f_out, f_in)
                if nbytes < 0:
                    _lib.check(int(nbytes), "gae_x_gcn_layer_fused_wgrad_workspace_bytes")
                ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev) if defer else _workspace(nbytes, dev)
                lay = (ctypes.c_int64 * 3)()
                _lib.call("gae_x_gcn_layer_fused2_wgrad", _ptr(t_indptr), _ptr(t_indices), n, _ptr(dYc), f_out, f_out,
                          _ptr(norm), _ptr(norm), ctypes.byref(plan_t.c), _ptr(W1), _ptr(W2), d1, W1.stride(0), f_in,
                          _ptr(dH), f_in, _ptr(M), M.stride(0), None if defer else _ptr(dW), None if defer else _ptr(db),
                          _ptr(ws), ws.numel(), lay, _stream())
            if defer:
                current_step().add_partials(dW, (ws, ws.data_ptr(), lay[0], lay[1], f_out * f_in, f_out * f_in))
                if db is not None:
                    current_step().add_partials(db, (ws, ws.data_ptr() + 4 * lay[2], lay[0], lay[1], f_out, f_out))
            _split_pending(dW, d1)
            dW1, dW2 = dW[:d1], dW[d1:]
            if db is not None:
                _split_pending(db, d1)
                db1, db2 = db[:d1], db[d1:]
            return dH, dW1, db1, dW2, db2, None, None
        if need_dW or need_db:
            dW, db, _ = linear_bwd_raw(dYc, None, ACT_IDENTITY, M, None, need_dW, need_db, False, f_out=d1 + d2)
            if dW is not None:
                _split_pending(dW, d1)
                dW1, dW2 = dW[:d1], dW[d1:]
            if db is not None:
                _split_pending(db, d1)
                db1, db2 = db[:d1], db[d1:]
        if need_dH:
            F = W1.shape[1]
            dH = torch.empty(n, F, dtype=torch.float32, device=dYc.device)
            with _on_device(dYc.device):
                # dH = (A^T dY) [W1; W2]: the stacked matrix addressed transposed (element (o, k) at row k, column o)
                _lib.call("gae_x_gcn_layer_fused2", _ptr(t_indptr), _ptr(t_indices), n, n, _ptr(dYc), d1 + d2, None, 0,
                          d1 + d2, _ptr(norm), _ptr(norm), ctypes.byref(plan_t.c), _ptr(W1), _ptr(W2), d1, 1, 1,
                          W1.stride(0), None, None, F, ACT_IDENTITY, _ptr(dH), F, _stream())
        return dH, dW1, db1, dW2, db2, None, None


def gcn_two_heads(graph, H, lin1, lin2, use_norm=False):
    """[head1 | head2] of two identity-activation GCN layers (nn.Linear modules lin1, lin2) on ``H`` in one launch, or
    None when the shapes / the graph do not allow the fused layer"""
    if not isinstance(H, torch.Tensor) or not H.is_cuda or H.dtype != torch.float32 or graph.number_of_edges() == 0:
        return None
    Hc, _ = _rowmajor(H, "H")
    if Hc.stride(0) % 4 or Hc.data_ptr() % 16:
        Hc = pad_rows(Hc)
    d1, d2 = lin1.weight.shape[0], lin2.weight.shape[0]
    plan, plan_t = graph.spmm_plan(False), graph.spmm_plan(True)
    if lin1.weight.shape[1] != lin2.weight.shape[1] or (lin1.bias is None) != (lin2.bias is None):
        return None
    if not gcn_layer_fused_usable(Hc, d1 + d2, plan) or not _table_only(plan_t) or d1 + d2 > FUSED_LAYER_MAX_IN:
        return None
    # the backward is gae_x_gcn_layer_fused2(_wgrad) on the CSR of A^T with dML [n, d1 + d2] as the gathered operand and
    # the heads' INPUT width as its output: rows of whole 16-byte vectors and <= 32 outputs, or the two separate
    # layers (which have their own fallbacks) must run instead
    if lin1.weight.shape[1] > FUSED_LAYER_MAX_OUT or (d1 + d2) % 4 != 0:
        return None
    bd = graph.block_diag
    if bd is not None and bd.usable(Hc, Hc.shape[1], Hc.stride(0), Hc.stride(0)):
        return None
    return GCNTwoHeadFunction.apply(Hc, lin1.weight, lin1.bias, lin2.weight, lin2.bias, graph, use_norm)


def gcn_layer(graph, H, W, b, act, use_norm=False):
    """one GCN layer on ``graph``: fused launch when the shapes allow it (gcn_layer_fused_usable), None otherwise
    (the caller then runs update_all + apply_nodes as two launches)"""
    if not isinstance(H, torch.Tensor) or not H.is_cuda or H.dtype != torch.float32:
        return None
    if graph.number_of_edges() == 0:
        return None
    Hc, _ = _rowmajor(H, "H")
    if Hc.stride(0) % 4 or Hc.data_ptr() % 16:
        Hc = pad_rows(Hc)
    bd = graph.block_diag
    if bd is not None and bd.usable(Hc, Hc.shape[1], Hc.stride(0), Hc.stride(0)):
        return None                  # whole-set molecule launches: the LDS-staged block-diagonal kernel is faster
    plan = graph.spmm_plan(False)
    if not gcn_layer_fused_usable(Hc, W.shape[0], plan):
        return None
    return GCNLayerFusedFunction.apply(Hc, W, b, graph, use_norm, act)


# ------------------------------------------------------------------ transform-first GCN layer (wide in, narrow out)
def xw_usable(X, n_out):
    """can gae_xw_fwd / gae_xw_wgrad take this operand?  (fp32 or bf16 rows of whole 16-byte vectors, f_in >= 193,
    f_out <= 32, X below 3.5 GiB)"""
    if not isinstance(X, torch.Tensor) or not X.is_cuda or X.dim() != 2 or X.dtype not in (torch.float32, torch.bfloat16):
        return False
    if X.shape[0] == 0 or X.stride(1) != 1:
        return False
    return bool(_lib.load().gae_xw_usable(_ptr(X), X.stride(0), _dtype_code(X), X.shape[0], X.shape[1], int(n_out)))


def xw_fwd_raw(X, W, b, act, keep_splits=False):
    """P = act(X W^T + b) with X read once and W stationary in registers (gae_xw_fwd); X fp32 or bf16 storage.
    ``keep_splits`` (b None, identity): when the library splits a long f_in over thread blocks, return the partial
    products [splits, n, f_out] instead of their sum (a consumer adds them: spmm_epilogue_raw) -- returns (P, 1) when
    there is no split"""
    W = _f32(_gpu(W, "W"), "xw_fwd: W")
    if W.stride(1) != 1:
        W = W.contiguous()
    _f32(b, "xw_fwd: b")
    n, f_in = X.shape
    f_out = W.shape[0]
    code = _dtype_code(X)
    lib = _lib.load()
    STATS["xw_fwd"] += 1
    splits = int(lib.gae_xw_fwd_splits(n, f_in, f_out, code)) if keep_splits else 1
    keep = keep_splits and splits > 1
    with _on_device(X.device):
        nbytes = lib.gae_xw_fwd_workspace_bytes(n, f_in, f_out, code)
        if nbytes < 0:
            _lib.check(int(nbytes), "gae_xw_fwd_workspace_bytes")
        if keep:       # the partials ARE the result: a buffer of their own, not the shared scratch
            parts = torch.empty(splits, n, f_out, dtype=torch.float32, device=X.device)
            ws, ws_n, P = parts, parts.numel() * 4, None
        else:
            ws = _workspace(nbytes, X.device) if nbytes > 0 else None
            ws_n = ws.numel() if ws is not None else 0
            P = torch.empty(n, f_out, dtype=torch.float32, device=X.device)

        def launch():
            _lib.call("gae_xw_fwd", _ptr(X), X.stride(0), code, n, f_in, _ptr(W), W.stride(0), _ptr(b), f_out, int(act),
                      _ptr(P), max(f_out, 1), _ptr(ws), ws_n, 1 if keep else 0, _stream())
        if profiler is not None:
            profiler.wrap(("xw_fwd", n, f_in, f_out, str(X.dtype)), launch)
        else:
            launch()
    if keep_splits:
        return (parts, splits) if keep else (P, 1)
    return P


def xw_wgrad_raw(X, G, Gmask, D, Dmask, f_out, need_dW=True, need_db=True):
    """(dW [f_out, f_in] = (G (.) [Gmask > 0])^T X, db [f_out] = colsum(D (.) [Dmask > 0])) in one pass over X
    (gae_xw_wgrad); masks may be None"""
    n, f_in = X.shape
    code = _dtype_code(X)
    dev = X.device
    STATS["xw_wgrad"] += 1
    G, ldg = _rowmajor(_f32(G, "xw_wgrad: G"), "G")
    ldgm = ldd = lddm = 0
    if Gmask is not None:
        Gmask, ldgm = _rowmajor(_f32(Gmask, "xw_wgrad: Gmask"), "Gmask")
    if D is not None:
        D, ldd = _rowmajor(_f32(D, "xw_wgrad: D"), "D")
    if Dmask is not None:
        Dmask, lddm = _rowmajor(_f32(Dmask, "xw_wgrad: Dmask"), "Dmask")
    dW = torch.empty(f_out, f_in, dtype=torch.float32, device=dev) if need_dW else None
    db = torch.empty(f_out, dtype=torch.float32, device=dev) if need_db and D is not None else None
    if current_step().defer_grads and (dW is not None or db is not None):
        with _on_device(dev):
            ws = torch.empty(_lib.load().gae_xw_wgrad_workspace_bytes(n, f_in, code), dtype=torch.uint8, device=dev)
            lay = (ctypes.c_int64 * 8)()
            _lib.call("gae_x_xw_wgrad_partials", _ptr(X), X.stride(0), code, n, f_in, _ptr(G), ldg, _ptr(Gmask), ldgm,
                      _ptr(D), ldd, _ptr(Dmask), lddm, int(f_out), int(dW is not None), int(db is not None), _ptr(ws),
                      ws.numel(), lay, _stream())
        if dW is not None:
            current_step().add_partials(dW, (ws, ws.data_ptr(), lay[0], lay[1], f_in, lay[2]))
        if db is not None:
            current_step().add_partials(db, (ws, ws.data_ptr() + 4 * lay[3], lay[4], lay[5], f_out, f_out))
        return dW, db
    with _on_device(dev):
        ws = _workspace(_lib.load().gae_xw_wgrad_workspace_bytes(n, f_in, code), dev)

        def launch():
            _lib.call("gae_xw_wgrad", _ptr(X), X.stride(0), code, n, f_in, _ptr(G), ldg, _ptr(Gmask), ldgm, _ptr(D), ldd,
                      _ptr(Dmask), lddm, int(f_out), _ptr(dW), max(f_in, 1), _ptr(db), _ptr(ws), ws.numel(), _stream())
        if profiler is not None:
            profiler.wrap(("xw_wgrad", n, f_in, f_out, str(X.dtype)), launch)
        else:
            launch()
    return dW, db


def spmm_epilogue_raw(indptr, indices, H, n_rows, plan, bias=None, act=ACT_IDENTITY, Hmask=None, row_scale=None,
                      col_scale=None):
    """Y = act(diag(rs) A diag(cs) (H (.) [Hmask > 0]) + bias) in one launch of the packed-table kernel
    (gae_spmm_csr_epilogue); H fp32 [n_cols, F <= 64] with rows of whole 16-byte vectors -- or a contiguous stack
    [splits, n_cols, F] of partial matrices (xw_fwd_raw(keep_splits=True)): a gathered row is then the sum of its
    partial rows in split order"""
    n_splits, split_stride = 1, 0
    if H.dim() == 3:
        if Hmask is not None or not H.is_contiguous() or H.shape[2] % 4 or H.data_ptr() % 16:
            H = H.sum(0)                                    # (not reached by the library's own callers)
        else:
            n_splits, split_stride = int(H.shape[0]), int(H.shape[1] * H.shape[2])
            H = H[0]
    H, ldh = _rowmajor(_f32(_gpu(H, "H"), "spmm_epilogue: H"), "H")
    if ldh % 4 or H.data_ptr() % 16:
        H = pad_rows(H); ldh = H.stride(0)
    n_cols, F = H.shape
    if Hmask is not None:
        Hmask, ldk = _rowmajor(_f32(_gpu(Hmask, "Hmask"), "spmm_epilogue: Hmask"), "Hmask")
        if ldk != ldh or Hmask.data_ptr() % 16:
            buf = torch.empty(n_cols, ldh, dtype=torch.float32, device=H.device)[:, :F]
            buf.copy_(Hmask)
            Hmask = buf
    _f32(bias, "spmm_epilogue: bias")
    ldy = (F + 3) // 4 * 4
    Y = torch.empty(n_rows, ldy, dtype=torch.float32, device=H.device)[:, :F]
    with _on_device(H.device):
        def launch():
            _lib.call("gae_spmm_csr_epilogue", _ptr(indptr), _ptr(indices), n_rows, n_cols, _ptr(H), ldh, _ptr(Hmask),
                      _ptr(Y), ldy, F, _ptr(row_scale), _ptr(col_scale), ctypes.byref(plan.c), _ptr(bias), int(act),
                      n_splits, split_stride, _stream())
        if profiler is not None:
            profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch)
        else:
            launch()
    return Y


def _table_only(plan):
    return plan is not None and plan.ell is not None and plan.n_heavy == 0 and plan.homed is None


def gcn_transform_first_usable(graph, H, n_out):
    """can GCNTransformFirstFunction run this layer?  A layer that narrows wide features (gae_xw_usable), on a graph
    whose plans carry a packed neighbour table and no heavy rows (every citation / molecule graph)"""
    if not xw_usable(H, n_out) or graph.number_of_edges() == 0:
        return False
    n = graph.number_of_nodes()
    if n != H.shape[0] or n * ((n_out + 3) // 4 * 4) * 4 + (1 << 16) >= (1 << 32):
        return False
    return _table_only(graph.spmm_plan(False)) and _table_only(graph.spmm_plan(True))


class GCNTransformFirstFunction(torch.autograd.Function):
    """GCN.forward (gae.py:26-31) of a layer that narrows its features, evaluated as Y = act(A (H W^T) + b) -- the
    value of the reference's act((A H) W^T + b) up to fp32 rounding -- in three launches forward (gae_xw_fwd,
    gae_spmm_csr_epilogue) and backward (gae_spmm_csr_epilogue on A^T with the ReLU gate in the gather,
    gae_xw_wgrad): H is read once per direction and nothing of the input width is written."""

    @staticmethod
    def forward(ctx, H, W, b, graph, use_norm, act):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        # (a long f_in is split over thread blocks: the aggregation adds the partial rows itself, no reduction launch)
        # (gae_spmm_csr_epilogue addresses the stack of partials through one raw buffer: splits * n * f_out * 4 < 2^27 bytes;
        # larger operands let gae_xw_fwd reduce the splits itself)
        f_out = W.shape[0]
        splits = int(_lib.load().gae_xw_fwd_splits(H.shape[0], H.shape[1], f_out, _dtype_code(H))) if f_out % 4 == 0 else 1
        keep = f_out % 4 == 0 and splits > 1 and splits * H.shape[0] * f_out * 4 < (1 << 27)
        P = xw_fwd_raw(H, W, None, ACT_IDENTITY, keep_splits=keep)
        if keep:
            P = P[0]
        Y = spmm_epilogue_raw(indptr, indices, P, n, graph.spmm_plan(False), b, act, None, norm, norm)
        ctx.act, ctx.has_bias = act, b is not None
        ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True))
        ctx.save_for_backward(H, W, Y if act == ACT_RELU else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        H, W, Y = ctx.saved_tensors
        (t_indptr, t_indices), n, norm, plan_t = ctx.bwd
        need_dH, need_dW = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        dW = db = dH = None
        dYc, _ = _rowmajor(_f32(dY, "dY"), "dY")
        G = None
        if need_dH or need_dW:
            G = spmm_epilogue_raw(t_indptr, t_indices, dYc, n, plan_t, None, ACT_IDENTITY, Y, norm, norm)   # A^T dYm
        if need_dW or need_db:
            dW, db = xw_wgrad_raw(H, G if need_dW else dYc, None, dYc if need_db else None, Y, W.shape[0],
                                  need_dW=need_dW, need_db=need_db)
        if need_dH:
            if H.dtype != torch.float32:
                raise GaeHipError("transform-first layer: a bf16-stored input cannot receive a gradient")
            _, _, dH = linear_bwd_raw(G, None, ACT_IDENTITY, H, W, False, False, True)      # dH = G W
        return dH, dW, db, None, None, None


def gcn_layer_transform_first(graph, H, W, b, act, use_norm=False):
    """the layer as GCNTransformFirstFunction, or None when the shapes / the graph do not allow it"""
    if not isinstance(H, torch.Tensor) or not H.is_cuda:
        return None
    if H.dtype == torch.float32 and (H.stride(1) != 1 or H.stride(0) % 4 or H.data_ptr() % 16):
        H = pad_rows(H)
    if not gcn_transform_first_usable(graph, H, W.shape[0]):
        return None
    return GCNTransformFirstFunction.apply(H, W, b, graph, use_norm, act)


# ------------------------------------------------------------------ layer 1 on sparse input features (opt-in)
def spx_fwd_raw(sf, W):
    """P = X W^T from the compressed rows of X (gae_spx_fwd); sf: sparse.SparseFeatures"""
    W = _f32(_gpu(W, "W"), "spx_fwd: W")
    if W.stride(1) != 1:
        W = W.contiguous()
    n, K = sf.shape
    J = W.shape[0]
    ldp = (J + 3) // 4 * 4
    P = torch.empty(n, ldp, dtype=torch.float32, device=W.device)[:, :J]
    with _on_device(W.device):
        ws = _workspace(K * 32 * 4 + 256, W.device)

        def launch():
            _lib.call("gae_spx_fwd", _ptr(sf.rowptr), _ptr(sf.col), _ptr(sf.val), n, K, _ptr(W), W.stride(0), J, _ptr(P),
                      ldp, _ptr(ws), ws.numel(), _stream())
        if profiler is not None:
            profiler.wrap(("spx_fwd", n, K, J, sf.nnz), launch)
        else:
            launch()
    return P


def spx_wgrad_raw(sf, G, D, Dmask, f_out, need_dW=True, need_db=True):
    """(dW = G^T X from the compressed rows of X^T, db = colsum(D (.) [Dmask > 0])) -- gae_spx_wgrad; inside
    deferred_grad_reductions() the sums stay partial lists for the optimiser launch"""
    n, K = sf.shape
    dev = G.device
    G, ldg = _rowmajor(_f32(G, "spx_wgrad: G"), "G")
    ldd = lddm = 0
    if D is not None:
        D, ldd = _rowmajor(_f32(D, "spx_wgrad: D"), "D")
    if Dmask is not None:
        Dmask, lddm = _rowmajor(_f32(Dmask, "spx_wgrad: Dmask"), "Dmask")
    dW = torch.empty(f_out, K, dtype=torch.float32, device=dev) if need_dW else None
    db = torch.empty(f_out, dtype=torch.float32, device=dev) if need_db and D is not None else None
    lay = (ctypes.c_int64 * 5)()
    _lib.call("gae_spx_wgrad_layout", n, K, sf.max_segments, lay)
    defer = current_step().defer_grads and (dW is not None or db is not None)
    with _on_device(dev):
        ws = torch.empty(lay[4], dtype=torch.uint8, device=dev) if defer else _workspace(lay[4], dev)

        def launch():
            _lib.call("gae_spx_wgrad", _ptr(sf.t_rowptr), _ptr(sf.t_row), _ptr(sf.t_val), _ptr(sf.seg_feat),
                      _ptr(sf.seg_e0), _ptr(sf.seg_slot), sf.seg_feat.numel(), sf.max_segments, n, K, _ptr(G), ldg,
                      _ptr(D), ldd, _ptr(Dmask), lddm, int(f_out), _ptr(dW), max(K, 1), _ptr(db), 0 if defer else 1,
                      _ptr(ws), ws.numel(), _stream())
        if profiler is not None:
            profiler.wrap(("spx_wgrad", n, K, f_out, sf.nnz), launch)
        else:
            launch()
    if defer:
        if dW is not None:
            current_step().add_partials(dW, (ws, ws.data_ptr(), lay[0], lay[1], f_out * K, f_out * K))
        if db is not None:
            current_step().add_partials(db, (ws, ws.data_ptr() + 4 * lay[2], lay[3], 32, f_out, f_out))
    return dW, db


class GCNSparseInputFunction(torch.autograd.Function):
    """GCNTransformFirstFunction on compressed input features: P = X W^T and dW = G^T X come from the non-zeros of X
    (gae_spx_fwd / gae_spx_wgrad); the sparse halves (gae_spmm_csr_epilogue) are the same launches"""

    @staticmethod
    def forward(ctx, W, b, sf, graph, use_norm, act):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        P = spx_fwd_raw(sf, W)
        Y = spmm_epilogue_raw(indptr, indices, P, n, graph.spmm_plan(False), b, act, None, norm, norm)
        ctx.act, ctx.has_bias, ctx.sf, ctx.f_out = act, b is not None, sf, W.shape[0]
        ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True))
        ctx.save_for_backward(Y if act == ACT_RELU else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        (Y,) = ctx.saved_tensors
        (t_indptr, t_indices), n, norm, plan_t = ctx.bwd
        need_dW = ctx.needs_input_grad[0]
        need_db = ctx.has_bias and ctx.needs_input_grad[1]
        dW = db = None
        dYc, _ = _rowmajor(_f32(dY, "dY"), "dY")
        if need_dW or need_db:
            G = spmm_epilogue_raw(t_indptr, t_indices, dYc, n, plan_t, None, ACT_IDENTITY, Y, norm, norm) \
                if need_dW else dYc
            dW, db = spx_wgrad_raw(ctx.sf, G, dYc if need_db else None, Y, ctx.f_out, need_dW=need_dW, need_db=need_db)
        return dW, db, None, None, None, None


def sparse_input_usable(graph, n, K, f_out):
    """can GCNSparseInputFunction run a K -> f_out layer on ``graph`` with compressed [n, K] features?  (the check
    sparse.SparseFeatures.maybe_from_dense makes BEFORE compressing)"""
    if graph.number_of_nodes() != n or f_out > 32 or graph.number_of_edges() == 0:
        return False
    if n * ((f_out + 3) // 4 * 4) * 4 + (1 << 16) >= (1 << 32):
        return False
    return _table_only(graph.spmm_plan(False)) and _table_only(graph.spmm_plan(True))


def gcn_layer_sparse_input(graph, sf, W, b, act, use_norm=False):
    """the layer on sparse.SparseFeatures input, or None when graph / widths do not allow it (the caller then
    densifies, once: SparseFeatures.to_dense(cache=True))"""
    if W.shape[1] != sf.shape[1] or not sparse_input_usable(graph, sf.shape[0], sf.shape[1], W.shape[0]):
        return None
    return GCNSparseInputFunction.apply(W, b, sf, graph, use_norm, act)


class LinearFunction(torch.autograd.Function):
    """NodeApplyModule: act(M W^T + b)  (gae.py:13-16)."""

    @staticmethod
    def forward(ctx, M, W, b, act):
        Y = linear_fwd_raw(M, W, b, act)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.save_for_backward(M, W, Y if act == ACT_RELU else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        M, W, Y = ctx.saved_tensors
        need_dM, need_dW = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        dW, db, dM = linear_bwd_raw(dY, Y, ctx.act, M, W, need_dW, need_db, need_dM)
        return dM, dW, db, None


class DecoderDenseFunction(torch.autograd.Function):
    """InnerProductDecoder: (Z m)(Z m)^T  (gae.py:70-71)."""

    @staticmethod
    def forward(ctx, Z, mask):
        ctx.save_for_backward(Z, mask)
        return decoder_dense_raw(Z, mask)

    @staticmethod
    def backward(ctx, G):
        Z, mask = ctx.saved_tensors
        return decoder_dense_bwd_raw(G, Z, mask), None


class DecoderBCEFunction(torch.autograd.Function):
    """train_inductive.py:44-48 fused: label from the graph's CSR, pos_weight,
    (Z m)(Z m)^T, BCE-with-logits mean.  The gradient w.r.t. Z is produced by
    the same launch sequence as the loss (flash-style) and scaled in backward."""

    @staticmethod
    def forward(ctx, Z, mask, graph, dropout=None, prepared=None):
        n = graph.number_of_nodes()
        nnz = graph.number_of_edges()
        counts = getattr(graph, "batch_counts", None)             # fixed-capacity batch: true sizes on the device
        pw = 0.0 if counts is not None else (float(n) * float(n) - float(nnz)) / float(nnz)  # train_inductive.py:46
        need = ctx.needs_input_grad[0]
        loss, dZ = decoder_bce_raw(Z, mask, graph.csr(), graph.csc() if need else None, pw, want_grad=need,
                                   dropout=dropout, counts=counts, defer_ok=True, prepared=prepared)
        ctx.save_for_backward(dZ)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dZ,) = ctx.saved_tensors
        if _is_unit(g):
            return dZ, None, None, None, None    # upstream gradient is the cached constant 1 (ops.backward)
        return dZ * g, None, None, None, None


def bce_logits_raw(logits, labels, pos_weight, want_grad=True):
    """F.binary_cross_entropy_with_logits(logits, labels, pos_weight) (mean) on materialised [n, m] fp32 matrices:
    returns (loss[1], dLoss/dLogits or None).  The gradient is written over ``logits``."""
    X, ldx = _rowmajor(_f32(logits, "bce_logits: logits"), "logits")
    Y, ldy = _rowmajor(_f32(labels, "bce_logits: labels"), "labels")
    if X.shape != Y.shape:
        raise GaeHipError("bce_logits: logits / labels shape mismatch")
    n, m = X.shape
    loss = torch.empty(1, dtype=torch.float32, device=X.device)
    with _on_device(X.device):
        ws = _workspace(_lib.load().gae_bce_logits_workspace_bytes(), X.device)
        _lib.call("gae_bce_logits", _ptr(X), ldx, _ptr(Y), ldy, n, m, float(pos_weight), _ptr(loss),
                  _ptr(X) if want_grad else None, ldx, _ptr(ws), ws.numel(), _stream())
    return loss, (X if want_grad else None)


class DecoderDenseBCEFunction(torch.autograd.Function):
    """train_inductive.py:44-48 in the reference's own shape -- dense label, N x N logits, weighted BCE (mean) -- as a
    chain of HIP kernels (gae_decoder_dense, gae_csr_to_dense, gae_bce_logits, gae_decoder_dense_bwd).  Used for
    embedding widths above FUSED_MAX_D, which the never-materialised kernel does not take; O(N^2) memory."""

    @staticmethod
    def forward(ctx, Z, mask, graph):
        n, nnz = graph.number_of_nodes(), graph.number_of_edges()
        pw = (float(n) * float(n) - float(nnz)) / float(nnz)      # train_inductive.py:46
        need = ctx.needs_input_grad[0]
        logits = decoder_dense_raw(Z, mask)
        loss, G = bce_logits_raw(logits, graph.dense_adjacency(), pw, want_grad=need)
        ctx.save_for_backward(G, Z, mask)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        G, Z, mask = ctx.saved_tensors
        dZ = decoder_dense_bwd_raw(G, Z, mask)
        return (dZ if _is_unit(g) else dZ * g), None, None


class BCELogitsFunction(torch.autograd.Function):
    """F.binary_cross_entropy_with_logits(logits, label, pos_weight=pw) (mean) on materialised matrices -- the
    reference's own loss call (train_inductive.py:48) -- by gae_bce_logits: loss and dLoss/dLogits in one pass."""

    @staticmethod
    def forward(ctx, logits, label, pos_weight):
        need = ctx.needs_input_grad[0]
        X, ldx = _rowmajor(_f32(_gpu(logits, "logits"), "bce_logits: logits"), "logits")
        Y, ldy = _rowmajor(_f32(_gpu(label, "label"), "bce_logits: labels"), "labels")
        if X.shape != Y.shape:
            raise GaeHipError("bce_with_logits: logits / labels shape mismatch")
        n, m = X.shape
        loss = torch.empty(1, dtype=torch.float32, device=X.device)
        G = torch.empty(n, m, dtype=torch.float32, device=X.device) if need else None
        with _on_device(X.device):
            ws = _workspace(_lib.load().gae_bce_logits_workspace_bytes(), X.device)
            _lib.call("gae_bce_logits", _ptr(X), ldx, _ptr(Y), ldy, n, m, float(pos_weight), _ptr(loss), _ptr(G),
                      max(m, 1), _ptr(ws), ws.numel(), _stream())
        ctx.save_for_backward(G)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (G,) = ctx.saved_tensors
        return (G if _is_unit(g) else G * g), None, None


def bce_with_logits(logits, label, pos_weight):
    """drop-in for ``BCELoss(logits, label, pos_weight=pw)`` of the reference's Trainer (train_inductive.py:48) on the
    HIP kernel; ``pos_weight`` may be a 0-dim / 1-element tensor like the reference's (read back once per call)"""
    return BCELogitsFunction.apply(logits, label, float(pos_weight))


FUSED_MAX_D = 64     # widest embedding of the fused decoder + BCE kernels (gae_decoder_bce)

_UNIT = {}


def _is_unit(g):
    one = _UNIT.get(g.device)
    return one is not None and g.dim() == 0 and g.data_ptr() == one.data_ptr()


def backward(loss, params=None):
    """``loss.backward()`` with the upstream gradient 1 handed over as a cached device constant: the fused loss
    recognises it and returns its stored gradient as is (saves the fill and the multiply launch of a plain
    ``loss.backward()``; 9 us of a 190 us Cora step).
    ``params``: write the gradients of exactly these tensors through ``torch.autograd.grad`` (p.grad is REPLACED, not
    accumulated).  No AccumulateGrad node takes part then -- those remember the stream of the iteration that
    created them, which breaks a HIP-graph capture that follows eager steps on another stream."""
    one = _UNIT.get(loss.device)
    if one is None:
        one = _UNIT[loss.device] = torch.ones((), dtype=loss.dtype, device=loss.device)
    if params is None:
        loss.backward(gradient=one)
        return
    params = [p for p in params if p.requires_grad]
    for p, g in zip(params, torch.autograd.grad(loss, params, grad_outputs=one, allow_unused=True)):
        p.grad = g


def decoder_bce(Z, mask, graph, dropout=None, prepared=None):
    """``dropout`` = (p, seed, offset, draw_counter): draw the mask inside the fused launch into ``mask``.
    ``prepared``: token of a producer launch that already ran the prepare step (loss_prepare_request).
    Embeddings wider than FUSED_MAX_D take the dense HIP chain (same value, O(N^2) memory)."""
    if prepared is not None:
        return DecoderBCEFunction.apply(Z, prepared["mask"], graph, prepared["dropout"], prepared)
    if Z.shape[1] > FUSED_MAX_D:
        if getattr(graph, "batch_counts", None) is not None:
            raise GaeHipError(f"fixed-capacity batches need an embedding width <= {FUSED_MAX_D}")
        if dropout is not None and dropout[0]:
            p_drop, seed, offset, draws = dropout
            mask.copy_(dropout_mask(tuple(Z.shape), p_drop, seed, offset, Z.device, draw_counter=draws))
            if draws is not None:
                draws += 1
        return DecoderDenseBCEFunction.apply(Z, mask, graph)
    return DecoderBCEFunction.apply(Z, mask, graph, dropout)


def gram_raw(M):
    """M^T M ([f, f], fp32-grade products) through gae_linear_bwd's weight-gradient kernel (dW = dY^T M with dY = M);
    never deferred to an optimiser launch (it is a value of the forward pass, not a parameter gradient)"""
    M, ldm = _rowmajor(_f32(_gpu(M, "M"), "gram: M"), "M")
    n, f = M.shape
    G = torch.empty(f, f, dtype=torch.float32, device=M.device)
    with _on_device(M.device):
        ws = _workspace(_lib.load().gae_linear_bwd_workspace_bytes(n, f, f), M.device)
        _lib.call("gae_linear_bwd", _ptr(M), ldm, None, 0, ACT_IDENTITY, _ptr(M), ldm, None, n, f, f,
                  _ptr(G), None, None, max(f, 1), _ptr(ws), ws.numel(), _stream())
    return G


class DecoderMSEFunction(torch.autograd.Function):
    """The criterion of the reference's hyper-parameter search, optuna_gae.py:16,21: ``nn.MSELoss()(model.forward(g),
    g.adjacency_matrix().to_dense())`` -- the mean over all N^2 ordered pairs of (s_ij - a_ij)^2, s = Zt Zt^T,
    Zt = Z (.) mask -- WITHOUT the N x N matrices:
        sum_ij (s_ij - a_ij)^2 = ||Zt^T Zt||_F^2 - 2 <Zt, A Zt> + sum_ij a_ij^2
        dL/dZt = (2 / N^2) (2 Zt (Zt^T Zt) - A Zt - A^T Zt)
    (oracle/gae_oracle.py: mse_closed_form): O(N d^2 + E d) work from launches the library already has --
    gae_spmm_csr on A and on A^T, gae_linear_bwd for the d x d Gram matrix (fp32-grade products), gae_linear_fwd for
    Zt G -- and two sums of N d products, taken in fp64."""

    @staticmethod
    def forward(ctx, Z, mask, graph):
        if getattr(graph, "batch_counts", None) is not None:
            raise GaeHipError("decoder_mse: fixed-capacity batches are not supported (the BCE loss's captured step only)")
        n = graph.number_of_nodes()
        Z = _f32(_gpu(Z, "Z"), "decoder_mse: Z")
        if Z.shape[0] != n:
            raise GaeHipError(f"decoder_mse: Z has {Z.shape[0]} rows, the graph {n} nodes")
        d = Z.shape[1]
        Zt = pad_rows(Z if mask is None else Z * mask)
        ip, ix = graph.csr()
        AZ = spmm_raw(ip, ix, Zt, n, plan=graph.spmm_plan(False))
        G = gram_raw(Zt)                                                      # Zt^T Zt
        inv = 1.0 / (float(n) * float(n))
        loss = ((G.double() ** 2).sum() - 2.0 * (Zt.double() * AZ.double()).sum() + graph.adjacency_sq_sum()) * inv
        if ctx.needs_input_grad[0]:
            tp, tx = graph.csc()
            AtZ = spmm_raw(tp, tx, Zt, n, plan=graph.spmm_plan(True))
            ZG = linear_fwd_raw(Zt, G, None, ACT_IDENTITY)                   # Zt G (G is symmetric)
            dZ = (2.0 * ZG - AZ - AtZ) * (2.0 * inv)
            if mask is not None:
                dZ = dZ * mask
            ctx.save_for_backward(dZ)
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (dZ,) = ctx.saved_tensors
        return (dZ if _is_unit(g) else dZ * g), None, None


def decoder_mse(Z, mask, graph):
    """mean((Zt Zt^T - A)^2) over all N^2 ordered pairs, Zt = Z (.) mask (optuna_gae.py:16,21), never materialised"""
    return DecoderMSEFunction.apply(Z, mask, graph)


class ShardedDecoderBCEFunction(torch.autograd.Function):
    """Row block of the fused loss on a row-sharded graph (parallel.ShardedGraph):
    Zt = Z (.) mask is all-gathered (N x d, small), each rank evaluates its rows
    against all columns, the partial means are summed with a scalar all-reduce."""

    @staticmethod
    def forward(ctx, z_local, mask_local, sg, n_edges_global):
        p = sg.part
        zt_local = z_local if mask_local is None else z_local * mask_local
        full = sg.allgather_rows(zt_local)
        n = p.n
        pw = (float(n) * float(n) - float(n_edges_global)) / float(n_edges_global)
        need = ctx.needs_input_grad[0]
        # labels are looked up by GLOBAL column id, whatever the exchange mode of the SpMM
        loss, dzt = decoder_bce_raw(full[:n], None, sg.csr_global("fwd"), sg.csr_global("bwd") if need else None, pw,
                                    want_grad=need, row_begin=p.r0, n_local=p.n_local)
        sg.allreduce_sum(loss)
        ctx.save_for_backward(dzt, mask_local)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dzt, mask_local = ctx.saved_tensors
        dz = dzt * g
        if mask_local is not None:
            dz = dz * mask_local
        return dz, None, None, None


def sharded_decoder_bce(z_local, mask_local, sg, n_edges_global=None):
    if n_edges_global is None:
        n_edges_global = sg.n_edges_global()          # (cached: no host read-back per step)
    return ShardedDecoderBCEFunction.apply(z_local, mask_local, sg, n_edges_global)


def spmm(graph, H, use_norm=False):
    return SpMMFunction.apply(H, graph, use_norm)


def linear(M, W, b, act=ACT_IDENTITY):
    return LinearFunction.apply(M, W, b, act)


def decoder_dense(Z, mask=None):
    return DecoderDenseFunction.apply(Z, mask)
