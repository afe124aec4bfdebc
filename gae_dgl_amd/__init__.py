"""gae_dgl_amd -- MI355X-native GCN-encoder hot path of shionhonda/gae-dgl.

Host-side mirror of the reference's Python API (gae_dgl/gae.py and the DGL
calls it makes) on top of libgae_hip.so (hand-written HIP kernels for gfx950,
C ABI in include/gae_hip.h).  PyTorch-ROCm only provides device memory,
streams, autograd plumbing and torch.distributed."""
from . import function, init  # noqa: F401
from .graph import DGLGraph, Graph, batch, readout_nodes  # noqa: F401
from .gae import GAE, GCN, InnerProductDecoder, NodeApplyModule, gcn_msg, gcn_reduce  # noqa: F401
from .sparse import SparseFeatures  # noqa: F401

__version__ = "0.1.0"
