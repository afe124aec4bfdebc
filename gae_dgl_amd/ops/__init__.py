"""Torch-facing wrappers of the C ABI (include/gae_hip.h).

Tensors are only carriers of device pointers here: every op below launches
hand-written HIP kernels from libgae_hip.so on PyTorch's current HIP stream.
There is no CPU / eager fallback -- a CPU tensor raises.

A package since round 6 (one 2 355-line module before): `_base` (plumbing, step context), `structure`, `aggregate`, `dense`,
`vgae_heads`, `loss`, `layers`, `wide`.  Every name of the former module is re-exported here; the submodules resolve each
other through THIS namespace at call time, so `ops.FLAG = ...` and patched functions behave as they did."""
from .. import _lib
from .._lib import ACT_IDENTITY, ACT_RELU, BF16, F32, GaeHipError
from ._base import *  # noqa: F401,F403
from .structure import *  # noqa: F401,F403
from .aggregate import *  # noqa: F401,F403
from .dense import *  # noqa: F401,F403
from .vgae_heads import *  # noqa: F401,F403
from .loss import *  # noqa: F401,F403
from .layers import *  # noqa: F401,F403
from .wide import *  # noqa: F401,F403
from ._base import _vp, _raw_stream, _stream_handle, _stream, _ptr, _gpu, _rowmajor, _f32, _dtype_code, _WS_CACHE, _STEP_STACK, _STEP_LOCK, _NO_STEP, _StackView, _step_stack, _workspace, _on_device  # noqa: F401
from .aggregate import _scattered  # noqa: F401
from .dense import _dead_mask  # noqa: F401
from .loss import _UNIT, _is_unit  # noqa: F401
from .layers import _split_pending  # noqa: F401
from .wide import _table_only  # noqa: F401
