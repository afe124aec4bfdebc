"""dgl.init stand-in (gae_dgl/train_inductive.py:93-94).  The HIP SpMM always
writes zero rows for nodes that receive no message, so the initializer is a
marker only."""


def zero_initializer(shape, dtype, ctx, id_range=None):
    import torch
    return torch.zeros(shape, dtype=dtype, device=ctx)
