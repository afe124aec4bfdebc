"""The two dgl.function builtins the reference uses (gae_dgl/gae.py:18-19).
They are descriptors; Graph.update_all recognises the (copy_src, sum) pair and
runs it as one fused HIP SpMM."""
from collections import namedtuple

CopySrc = namedtuple("CopySrc", ["src", "out"])
SumReduce = namedtuple("SumReduce", ["msg", "out"])


def copy_src(src, out):
    return CopySrc(src, out)


def sum(msg, out):  # noqa: A001 - name fixed by the DGL API
    return SumReduce(msg, out)
