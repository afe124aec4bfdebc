"""Shared plumbing of the operator wrappers: device pointers and streams, row layout helpers, the step context
(what one training step defers to its optimiser launch), the workspace cache and the launch profiler.

Part of the package gae_dgl_amd.ops (one module until round 6).  Functions look each other up in the PACKAGE
namespace (`_ops.<name>`) when they run: setting a flag or replacing a function on `gae_dgl_amd.ops` reaches every caller."""
import ctypes
import threading

import torch

import gae_dgl_amd.ops as _ops
from .. import _lib
from .._lib import BF16, F32, GaeHipError

__all__ = [
    '_vp', '_raw_stream', '_stream_handle', '_stream', '_ptr', '_gpu', 'row_quantum', 'padded_ld', 'pad_rows',
    'float_rows', '_rowmajor', '_f32', '_dtype_code', '_WS_CACHE', 'StepContext', '_STEP_STACK', '_STEP_LOCK',
    '_NO_STEP', '_StackView', '_step_stack', 'current_step', 'deferred_grad_reductions', 'deferred_loss_finalize',
    'pending_loss_tail', 'pending_partials', '_workspace', '_on_device', 'device_info', 'EventProfiler', 'profiler',
]


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
    q = _ops.row_quantum(f, dtype)
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
    q = multiple or _ops.row_quantum(f, t.dtype)
    if t.stride(1) == 1 and t.stride(0) % q == 0 and t.stride(0) >= f and t.data_ptr() % (q * t.element_size()) == 0:
        return t
    ld = (f + q - 1) // q * q if multiple else _ops.padded_ld(f, t.dtype)
    buf = torch.zeros(n, ld, dtype=t.dtype, device=t.device)
    buf[:, :f] = t
    return buf[:, :f]


def float_rows(t):
    """fp32 copy of a bf16-stored [N, F] operand in a row-padded buffer (see pad_rows): the Linear and dW kernels
    that read it next then take their aligned 16-byte paths (a plain ``.float()`` of F = 3703 has rows of 14812
    bytes: scalar loads; Citeseer VGAE layer-1 Linear 31 -> 22.5 us, dW 37.9 -> 14.9 us)"""
    n, f = t.shape
    ld = _ops.padded_ld(f, torch.float32)
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
    return _ops.StepContext(defer_grads=True)


def deferred_loss_finalize():
    """``with ops.deferred_loss_finalize():`` = a StepContext that defers the loss's final reduction"""
    return _ops.StepContext(defer_loss=True)


def pending_loss_tail():
    """(BceTail, keep-alive) of a loss of the current step whose final reduction was deferred (removed from the
    context), or None"""
    return _ops.current_step().take_loss_tail()


def pending_partials(grad):
    """(keep-alive, ptr, n_partials, partial_stride, row_len, row_pitch) of a gradient of the current step whose
    reduction was deferred (removed from the context), or None"""
    return _ops.current_step().take_partials(grad)


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
