"""The collectives of the row-sharded path (parallel.py), by transport.

  * backend "nccl" (= RCCL over xGMI, one process per GPU): the product transport.  Device tensors go straight
    through; the work runs on RCCL's stream and ``wait()`` orders the caller's stream behind it.
  * backend "gloo", host tensors: the CPU host-logic tests.
  * backend "gloo", DEVICE tensors: several ranks SHARING one GPU (RCCL refuses two ranks on one device).  Every
    collective is staged through host memory: device -> host copy, gloo collective on the host buffers, host -> device
    copy into the tensor the caller passed.  This is a functional rehearsal of the N-rank path with the real HIP
    kernels on a box that has one GPU (tests/test_gpu_multiproc.py, ``bench.py --oversubscribe``) -- never a measurement
    of the exchange: a staged collective synchronises the stream and moves every byte over PCIe twice.

The entry points mirror torch.distributed's (same argument meaning); ``async_op=True`` returns an object with
``wait()``, in the staged case a handle whose ``wait()`` finishes the host collective and issues the copy back."""
import torch
import torch.distributed as dist


GROUPED_UNEVEN_ALLGATHER = True    # RCCL: one grouped all_gather into uneven views; False: one broadcast per owner
                                   # (bench.py's collective preflight switches it off when the grouped form fails)


def backend(group=None):
    return dist.get_backend(group)


def staged(t, group=None):
    """does a collective on ``t`` have to go through host memory?  (gloo has no device all-to-all / all-gather-base)"""
    return t.is_cuda and backend(group) != "nccl"


class _Staged:
    """pending staged collective: ``work`` on host buffers, then ``pairs`` of (device tensor, host tensor) to copy back"""

    def __init__(self, work, pairs):
        self.work, self.pairs = work, pairs

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        for dst, src in self.pairs:
            dst.copy_(src)
        self.pairs = []
        return True


class _Done:
    def wait(self):
        return True


def _finish(h, async_op):
    if async_op:
        return h
    h.wait()
    return None


def all_reduce(t, op=None, group=None, async_op=False):
    op = dist.ReduceOp.SUM if op is None else op
    if not staged(t, group):
        return dist.all_reduce(t, op=op, group=group, async_op=async_op)
    h = t.detach().cpu()
    return _finish(_Staged(dist.all_reduce(h, op=op, group=group, async_op=True), [(t, h)]), async_op)


def all_to_all_single(out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
    if not staged(out, group):
        return dist.all_to_all_single(out, inp, output_split_sizes=output_split_sizes,
                                      input_split_sizes=input_split_sizes, group=group, async_op=async_op)
    oh = torch.empty(out.shape, dtype=out.dtype)
    ih = inp.detach().cpu().contiguous()
    w = dist.all_to_all_single(oh, ih, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes,
                               group=group, async_op=True)
    return _finish(_Staged(w, [(out, oh)]), async_op)


def all_gather_into_tensor(full, mine, group=None, async_op=False):
    if not staged(full, group):
        return dist.all_gather_into_tensor(full, mine, group=group, async_op=async_op)
    fh = torch.empty(full.shape, dtype=full.dtype)
    w = dist.all_gather_into_tensor(fh, mine.detach().cpu().contiguous(), group=group, async_op=True)
    return _finish(_Staged(w, [(full, fh)]), async_op)


def broadcast(t, src, group=None, async_op=False):
    """``src``: GLOBAL rank of the owner (torch.distributed's convention)"""
    if not staged(t, group):
        return dist.broadcast(t, src=src, group=group, async_op=async_op)
    mine = dist.get_rank() == src
    h = t.detach().cpu().contiguous() if mine else torch.empty(t.shape, dtype=t.dtype)
    w = dist.broadcast(h, src=src, group=group, async_op=True)
    return _finish(_Staged(w, [] if mine else [(t, h)]), async_op)


def all_gather_uneven(views, t_local, rank, group=None):
    """every rank's block into ``views[q]`` (blocks of different lengths; ``views[rank]`` receives ``t_local``).
    RCCL: ONE grouped all_gather into the views; gloo has no uneven all_gather: one broadcast per owner.  Returns the
    list of pending works."""
    if backend(group) == "nccl" and GROUPED_UNEVEN_ALLGATHER:
        return [dist.all_gather(views, t_local.contiguous(), group=group, async_op=True)]
    views[rank].copy_(t_local)
    works = []
    for q, v in enumerate(views):
        if v.numel() == 0:
            continue
        src = dist.get_global_rank(group, q) if group is not None else q
        works.append(broadcast(v, src, group, async_op=True) or _Done())
    return works


def barrier(group=None):
    dist.barrier(group=group)
