"""HIP-graph capture of the transductive training step.

The full-graph step (encoder forward, fused decoder+BCE, backward, Adam) is a
fixed sequence of ~30 launches on static buffers, so it is captured once in a
hipGraph (through torch.cuda.CUDAGraph, which records every launch on the
capture stream -- including the ctypes launches of libgae_hip.so, which use
PyTorch's current stream) and replayed per epoch: no Python / launch overhead
between kernels.  The decoder's dropout mask still changes every replay
because its Philox draw counter lives in device memory (gae_decoder_bce), and so does Adam's step counter
(gae_dgl_amd.optim.Adam / torch.optim.Adam(capturable=True))."""
import torch

from . import ops


class CapturedTrainStep:
    """Construct it BEFORE running eager steps on the default stream, or after dropping every reference to their
    losses / outputs: a live autograd graph of an earlier step keeps the parameters' AccumulateGrad nodes bound to
    the default stream, and replaying them inside the capture breaks it (PyTorch warns "AccumulateGrad node's
    stream does not match", then hipStreamEndCapture fails).  The warm-up steps of this class run on a side
    stream for that reason."""

    def __init__(self, model, optimizer, graph, features, loss_fn=None, warmup=3):
        self.model, self.opt, self.g, self.x = model, optimizer, graph, features
        self.loss_fn = loss_fn or (lambda m, g: m.reconstruction_loss(g))
        for group in optimizer.param_groups:           # Adam must keep its step counter on the device
            if "capturable" in group and not group["capturable"]:
                raise ValueError("build the optimizer with capturable=True to capture its step")
        # structure is static: build it outside the capture (CSR build sorts and reads back a status word)
        graph.csr(); graph.csc(); graph.spmm_plan(False); graph.spmm_plan(True); graph.scattered()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        self._capture()

    def _hyper(self):
        """optimiser hyper-parameters that a launch takes by value (frozen into the captured graph)"""
        return tuple((g["lr"], tuple(g.get("betas", ())), g.get("eps"), g.get("weight_decay")) for g in self.opt.param_groups)

    def _capture(self):
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._fwd_bwd_step()
        self._captured_hyper = self._hyper()

    def _fwd_bwd_step(self):
        self.g.ndata['h'] = self.x
        loss = self.loss_fn(self.model, self.g)
        ops.backward(loss)
        self.opt.step()
        return loss.detach()

    def _eager_step(self):
        self.opt.zero_grad(set_to_none=True)
        return self._fwd_bwd_step()

    def __call__(self):
        """one training step; returns the (static) loss tensor of this replay.  A changed learning rate (scheduler,
        manual edit of param_groups) is picked up by capturing the step again."""
        if self._hyper() != self._captured_hyper:
            self._capture()
        self.graph.replay()
        return self.loss
