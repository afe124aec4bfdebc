"""HIP-graph capture of the transductive training step.

The full-graph step (encoder forward, fused decoder+BCE, backward, Adam) is a
fixed sequence of ~30 launches on static buffers, so it is captured once in a
hipGraph (through torch.cuda.CUDAGraph, which records every launch on the
capture stream -- including the ctypes launches of libgae_hip.so, which use
PyTorch's current stream) and replayed per epoch: no Python / launch overhead
between kernels.  The decoder's dropout mask still changes every replay
because its Philox draw counter lives in device memory (gae_decoder_bce), and so does Adam's step counter
(gae_dgl_amd.optim.Adam / torch.optim.Adam(capturable=True))."""
import gc
import os

import torch

from . import ops


def _defer(opt, loss_too=False, grads=True):
    """the step's context (ops.StepContext): the weight gradients' reductions -- and, ``loss_too``, the loss's final one
    -- are left to the optimiser launch when the optimiser is the library's Adam (which consumes them); a context that
    defers nothing for any other optimiser.  ``loss_too``: the step's loss is the fused decoder + BCE scalar itself and
    nothing reads it before the optimiser launch (a loss_fn of the caller's may do arithmetic on it: those steps keep
    the reduction launch).  ``grads`` False: the gradients must be finished before the optimiser launch (data-parallel
    replicas all-reduce them in between)."""
    from .optim import Adam
    ours = isinstance(opt, Adam)
    return ops.StepContext(defer_grads=ours and grads and DEFER_GRAD_REDUCTIONS,
                           defer_loss=ours and loss_too and DEFER_LOSS_FINALIZE)


def _state_outside_capture(opt, steps_taken=0):
    """optimiser state (moments, step counters) must exist BEFORE the capture: created lazily inside it, it would sit in
    the graph's private pool and be zero-filled again by every replay.  The library's Adam creates it on request; any
    other optimiser must have taken a step (warm-up or the caller's own eager steps).  ``steps_taken``: warm-up steps
    the caller itself just ran -- an optimiser that still has no state after a step keeps none (plain SGD), which is
    fine."""
    from .optim import Adam
    if isinstance(opt, Adam):
        opt.materialize_state()
    elif steps_taken == 0 and any(p.requires_grad and len(opt.state.get(p, {})) == 0
                                  for g in opt.param_groups for p in g["params"]):
        raise ValueError("capture needs the optimiser's state in place: run at least one step (warmup >= 1) before it")


FUSED_COLLATE_MAX_GRAPHS = 1024   # batches up to this size collate in one launch (gae_x_batch_gather_next); 0 = never
DEFER_GRAD_REDUCTIONS = True      # False: captured steps keep the separate reduction launches (experiments)
DEFER_LOSS_FINALIZE = os.environ.get("GAE_DEFER_LOSS_FINALIZE", "1") != "0"   # False: the fused loss keeps its own final-reduction launch (experiments)


class CapturedTrainStep:
    """The warm-up steps run on a side stream and the captured step takes its gradients through
    ``torch.autograd.grad`` (``ops.backward(loss, params)``): an AccumulateGrad node of an earlier eager iteration
    stays bound to that iteration's stream, and replaying it inside a capture invalidates the capture (PyTorch warns
    "AccumulateGrad node's stream does not match", then hipStreamEndCapture fails).  A garbage collection in front
    of the warm-up frees autograd graphs that only reference cycles keep alive.  The warm-up steps DO train the
    model (``warmup`` real steps); use ``warmup=0`` after eager steps of your own if that matters."""

    def __init__(self, model, optimizer, graph, features, loss_fn=None, warmup=3, defer_loss=None):
        self.model, self.opt, self.g, self.x = model, optimizer, graph, features
        self._params = [p for group in optimizer.param_groups for p in group["params"]]
        # defer_loss: may the loss's final reduction ride in the optimiser launch?  Only if nothing reads the scalar
        # inside the step.  None = yes for the default reconstruction loss, no for a caller's loss_fn; True = the
        # caller vouches for its loss_fn (e.g. VGAE.loss, which returns the fused scalar itself)
        self._defer_loss = (loss_fn is None and hasattr(model, "reconstruction_loss")) if defer_loss is None \
            else bool(defer_loss)
        self.loss_fn = loss_fn or (lambda m, g: m.reconstruction_loss(g))
        for group in optimizer.param_groups:           # Adam must keep its step counter on the device
            if "capturable" in group and not group["capturable"]:
                raise ValueError("build the optimizer with capturable=True to capture its step")
        # structure is static: build it outside the capture (CSR build sorts and reads back a status word)
        graph.csr(); graph.csc(); graph.spmm_plan(False); graph.spmm_plan(True); graph.scattered()
        gc.collect()       # autograd graphs of earlier eager steps that only a collection frees (see the class note)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        _state_outside_capture(optimizer, warmup)
        for m in model.modules():          # device-side draw counters (dropout mask, VGAE noise): same rule
            if getattr(m, "_draws", False) is None:
                m._draws = torch.zeros(1, dtype=torch.int64, device=features.device)
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
        # the optimiser launch follows the backward pass directly: the library's Adam adds the weight gradients'
        # partial sums itself (two reduction launches less per step) and finishes the loss scalar in an extra block
        # (one more)
        with _defer(self.opt, self._defer_loss):
            loss = self.loss_fn(self.model, self.g)
            ops.backward(loss, self._params)      # autograd.grad: no AccumulateGrad nodes (stream-bound) in the capture
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


class CapturedInductiveStep:
    """The mini-batch training step of train_inductive.py:88-99 -- collate (dgl.batch), forward, loss, backward, Adam
    -- as ONE captured HIP graph, replayed per batch.

    Batches of molecules all differ in size; a captured graph must launch the same shapes every time.  The step
    therefore runs on a FIXED-CAPACITY batch: ``batch_size`` graphs are gathered from the device-resident dataset
    into static buffers of ``cap_nodes`` rows / ``cap_edges`` edges (gae_batch_gather pads the rows behind the batch
    with isolated zero-feature nodes and leaves the true sizes in device memory), the encoder runs on all
    ``cap_nodes`` rows (padding rows have no edges: they never reach a real row), and the loss reads the true
    N and E on the device (gae_decoder_bce_padded: pos_weight, the mean, zero gradient for padding rows).  The graph
    ids come from an epoch order uploaded once and a device-side cursor (gae_batch_select).  Nothing crosses the
    host <-> device boundary per batch and the host issues one graph launch per step.

    The capacities are the largest full batch of the epoch orders seen so far (+ ``margin``); an epoch whose largest
    batch does not fit is captured again with larger buffers.  The ragged last batch of an epoch (fewer graphs)
    runs through the ordinary eager path.

        runner = CapturedInductiveStep(model, optimizer, dataset, batch_size=128)
        for epoch in range(n_epochs):
            for loss in runner.epoch(rng.permutation(dataset.ids)):   # device scalars; .item() only when needed
                ...
    """

    def __init__(self, model, optimizer, dataset, batch_size, warmup=2, margin=1.005, group=None, replicas=False):
        """``replicas`` (with ``group``, default process group when None): data-parallel replicas -- every rank replays
        its own batches; the parameter gradients are averaged over the ranks by ONE all-reduce (RCCL, part of the
        captured graph) between the backward pass and the optimiser launch, and the batch capacities are agreed on
        (all-reduce MAX) so that every rank captures at the same steps.  Every rank must run the same number of
        steps per epoch (dataset.shard_order)."""
        self.group, self.replicas = group, bool(replicas)
        if self.replicas:
            import torch.distributed as dist
            from . import transport
            if not dist.is_initialized():
                raise ValueError("replicas=True needs an initialised process group")
            if transport.backend(group) != "nccl":
                raise RuntimeError("a captured data-parallel step needs RCCL (backend nccl): collectives staged through "
                                   "host memory synchronise the stream, which a capture forbids -- use the eager step")
        if not dataset.ell_width or not dataset.no_heavy_rows:
            raise ops.GaeHipError("CapturedInductiveStep needs a dataset of low-degree graphs (packed neighbour "
                                  "table); skewed graphs take the eager path")
        self.model, self.opt, self.ds = model, optimizer, dataset
        self._params = [p for group in optimizer.param_groups for p in group["params"]]
        self.B = int(batch_size)
        if self.B < 1 or self.B > len(dataset):
            raise ValueError("batch_size must be in [1, len(dataset)]")
        self.warmup, self.margin = warmup, margin
        for group in optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                raise ValueError("build the optimizer with capturable=True to capture its step")
        if any(getattr(m, "cache_aggregate", False) for m in model.modules()):
            raise ValueError("cache_first_aggregate keys its cache on tensor identity; the static batch buffers of a "
                             "captured step change content, not identity")
        dev = dataset.device
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self.d_order = torch.zeros(len(dataset), dtype=torch.int64, device=dev)
        self.cap_nodes = self.cap_edges = 0
        self.graph = None
        self.captures = 0
        self._order = None
        self._done = self._n_full = 0

    # ---------------------------------------------------------------- static buffers
    def _allocate(self, cap_nodes, cap_edges):
        from .graph import Graph
        ds, dev, B, W = self.ds, self.ds.device, self.B, self.ds.ell_width
        self.cap_nodes, self.cap_edges = cap_nodes, cap_edges
        self.gids = torch.zeros(B, dtype=torch.int64, device=dev)
        self.ptrs = torch.zeros(2 if ds.symmetric else 3, B + 1, dtype=torch.int64, device=dev)
        self.counts = torch.zeros(4, dtype=torch.int64, device=dev)   # {nodes, edges, graphs dropped by the guard, ticket}
        F, ldo, odt = ops.batch_feature_ld(ds.feat, ds.n_feat)
        ip = torch.zeros(cap_nodes + 1, dtype=torch.int32, device=dev)
        ix = torch.zeros(cap_edges, dtype=torch.int32, device=dev)
        feat = torch.zeros(cap_nodes, ldo, dtype=odt, device=dev)
        table = torch.full((cap_nodes * W,), -1, dtype=torch.int32, device=dev)
        self.fwd = (ip, ix, feat, table)
        if ds.symmetric:
            self.bwd = None
            tp, tx, t_table = ip, ix, table
        else:
            tp = torch.zeros_like(ip); tx = torch.zeros_like(ix); t_table = torch.full_like(table, -1)
            self.bwd = (tp, tx, None, t_table)
        g = Graph(device=dev)
        g._n = cap_nodes
        g._src = g._dst = None
        g.set_csr(ip, ix, tp, tx)
        g.ndata['h'] = feat[:, :F]
        g.no_heavy_rows = True
        g.block_diag = None                     # its cuts are host-side and batch-specific: the table kernels run here
        g._cache["plan"] = ops.table_plan(table, W)
        g._cache["plan_t"] = ops.table_plan(t_table, W)
        g._cache["graph_ptr"] = self.ptrs[0]
        g.batch_counts = self.counts
        self.g, self.x = g, feat[:, :F]

    def _step_body(self):
        """select + plan -> gather -> forward -> loss -> backward -> Adam, all on static buffers"""
        ds, g = self.ds, self.g
        if ds.symmetric and self.B <= FUSED_COLLATE_MAX_GRAPHS:
            # small batches are bound by the number of kernel nodes: select + plan + gather as ONE launch
            ops.batch_gather_next(ds.graph_ptr, ds.indptr, ds.indices, ds.feat, self.d_order, self.cursor, self.gids,
                                  self.ptrs, self.cap_nodes, self.cap_edges, self.fwd, self.counts,
                                  ell_width=ds.ell_width, n_feat=ds.n_feat)
            node_ptr = edge_ptr = t_edge_ptr = None
        else:
            node_ptr, edge_ptr, t_edge_ptr = ops.batch_plan_next(ds.graph_ptr, ds.indptr,
                                                                 None if ds.symmetric else ds.t_indptr, self.d_order,
                                                                 self.cursor, self.gids, self.ptrs)
            ops.batch_gather(ds.graph_ptr, ds.indptr, ds.indices, ds.feat, self.gids, node_ptr, edge_ptr,
                             self.cap_nodes, self.cap_edges, ell_width=ds.ell_width, n_feat=ds.n_feat, out=self.fwd,
                             pad_to_capacity=True, counts=self.counts)
        if self.bwd is not None:
            ops.batch_gather(ds.graph_ptr, ds.t_indptr, ds.t_indices, None, self.gids, node_ptr, t_edge_ptr,
                             self.cap_nodes, self.cap_edges, ell_width=ds.ell_width, out=self.bwd,
                             pad_to_capacity=True)
        for key in ("deg", "norm"):             # degree norms (norm="both" models) follow the batch: recompute
            g._cache.pop(key, None)
        g.ndata.clear()
        g.ndata['h'] = self.x
        with _defer(self.opt, True, grads=not self.replicas):
            loss = self.model.reconstruction_loss(g)
            ops.backward(loss, self._params)      # autograd.grad: no AccumulateGrad nodes (stream-bound) in the capture
            if self.replicas:                     # 1 808 floats for the 39 -> 32 -> 16 model: one small all-reduce
                from .parallel import allreduce_grads
                allreduce_grads(self._params, self.group, average=True)
            self.opt.step()
        return loss.detach()

    # ---------------------------------------------------------------- capture
    def _state(self):
        """name -> tensor of everything a warm-up step changes: parameters, optimiser state, device-side counters.
        Keyed, not positional: a warm-up step may CREATE tensors (lazy optimiser state after a load_state_dict,
        the decoder's draw counter), so the lists before and after differ in length and order."""
        ts = {f"param/{k}": p.data for k, p in self.model.named_parameters()}
        for i, (p, st) in enumerate(self.opt.state.items()):
            pid = next((k for k, q in self.model.named_parameters() if q is p), f"#{i}")
            for name, v in st.items():
                if isinstance(v, torch.Tensor):
                    ts[f"opt/{pid}/{name}"] = v
        for key, c in getattr(self.opt, "_counters", {}).items():
            ts[("counter",) + tuple(key)] = c
        for k, m in self.model.named_modules():
            if getattr(m, "_draws", None) is not None:
                ts[f"draws/{k}"] = m._draws
        return ts

    def _capture(self, cap_nodes, cap_edges, warm_batch=0):
        """``warm_batch``: index of a FULL batch of the uploaded order that is known to fit the capacities; every
        warm-up step runs on it (the cursor is put back in front of it each time).  Warm-up steps must not walk on:
        the batch behind the last full one is the ragged tail padded with a repeated graph and was never checked
        against the capacities (the kernel's own guard would drop graphs, see gae_batch_gather)."""
        self.graph = None
        self._allocate(cap_nodes, cap_edges)
        # the warm-up steps (allocator, lazy optimiser state, autograd streams) must not train the model: snapshot
        # every tensor they change BY NAME, run them, restore; tensors the warm-up created get their start values
        gc.collect()       # autograd graphs of earlier eager steps that only a collection frees keep AccumulateGrad
        #                    nodes bound to the default stream, which would invalidate the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            before = {k: t.clone() for k, t in self._state().items()}
            resume = getattr(self.opt, "_resume_steps", None) or []
            for _ in range(max(self.warmup, 1)):
                self.cursor.fill_(int(warm_batch))
                self.opt.zero_grad(set_to_none=True)
                self._step_body()
            for k, t in self._state().items():
                if k in before:
                    t.copy_(before[k])
                elif isinstance(k, tuple):          # step counter created by the warm-up: back to the resume point
                    t.zero_()                       # (word 0 = steps taken; zeros elsewhere = no cached beta^t)
                    if k[1] < len(resume):
                        t[0] = int(resume[k[1]])
                else:                               # lazily created optimiser moments, a new draw counter
                    t.zero_()
            self.counts.zero_()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._step_body()
        self.g.ndata.clear()            # drops the captured iteration's autograd graph (kept alive by the embedding)
        self._captured_hyper = self._hyper()
        self.captures += 1

    def _hyper(self):
        return tuple((g["lr"], tuple(g.get("betas", ())), g.get("eps"), g.get("weight_decay"))
                     for g in self.opt.param_groups)

    # ---------------------------------------------------------------- epoch interface
    def begin_epoch(self, order):
        """upload the epoch's order of graph ids (``DataLoader(shuffle=True)``'s permutation) and reset the cursor;
        returns the number of full batches"""
        import numpy as np
        order = np.ascontiguousarray(order, dtype=np.int64)
        if len(order) > self.d_order.numel():
            raise ValueError("epoch order longer than the dataset")
        n_full = len(order) // self.B
        self._order = order
        if n_full:
            starts = np.arange(0, n_full * self.B, self.B)
            head = order[:n_full * self.B]
            need_n = int(np.add.reduceat(self.ds.sizes_host[head], starts).max())
            need_e = int(max(np.add.reduceat(self.ds.edges_host[head], starts).max(),
                             np.add.reduceat(self.ds.t_edges_host[head], starts).max()))
            if self.replicas:      # the replicas must (re)capture together: the warm-up steps hold collectives
                import torch.distributed as dist
                from . import transport
                need = torch.tensor([need_n, need_e], dtype=torch.int64, device=self.ds.device)
                transport.all_reduce(need, op=dist.ReduceOp.MAX, group=self.group)
                need_n, need_e = (int(v) for v in need.tolist())
            if self.graph is None or need_n > self.cap_nodes or need_e > self.cap_edges:
                # round the capacities up (whole row blocks of the kernels, headroom for later epochs)
                cap_n = max(-(-int(need_n * self.margin) // 64) * 64, self.cap_nodes)
                cap_e = max(-(-int(need_e * self.margin) // 256) * 256, self.cap_edges, 1)
                self.d_order[:len(order)].copy_(torch.from_numpy(order))   # the warm-up steps read batch 0 of it
                self._capture(cap_n, cap_e, warm_batch=0)
        self.d_order[:len(order)].copy_(torch.from_numpy(order))
        self.cursor.zero_()
        self.counts.zero_()
        self._done, self._n_full = 0, n_full
        return n_full

    def step(self):
        """next full batch of the epoch: one graph launch; returns the static loss tensor of this replay"""
        if self._done >= self._n_full:
            raise ops.GaeHipError("CapturedInductiveStep.step: no full batch left in this epoch (begin_epoch() "
                                  "returns their number; the ragged tail goes through tail_step())")
        if self._hyper() != self._captured_hyper:
            self._recapture()
        self.graph.replay()
        self._done += 1
        return self.loss

    def _recapture(self):
        """same capacities, new hyper-parameters (a learning-rate scheduler stepping per batch): the warm-up runs
        on the batch that comes next -- a full, validated batch, because step() checked that one is left"""
        self._capture(self.cap_nodes, self.cap_edges, warm_batch=self._done)
        self.cursor.fill_(self._done)

    def dropped_graphs(self):
        """graphs the device-side guard of gae_batch_gather left out since begin_epoch() (host read-back; 0 unless the
        buffers were too small for a batch -- a bug in the capacity bookkeeping, never silent: epoch() raises)"""
        return int(self.counts[2].item())

    def tail_step(self):
        """the ragged last batch (len(order) % batch_size graphs), eagerly; None when the epoch has no tail"""
        order, lo = self._order, (len(self._order) // self.B) * self.B
        if lo == len(order):
            return None
        bg = self.ds._assemble(self.d_order[lo:len(order)], order[lo:])
        self.opt.zero_grad(set_to_none=True)
        loss = self.model.reconstruction_loss(bg)
        ops.backward(loss)
        if self.replicas:
            from .parallel import allreduce_grads
            allreduce_grads(self._params, self.group, average=True)
        self.opt.step()
        return loss.detach()

    def epoch(self, order):
        """iterate over the losses of one epoch (device scalars; the full-batch ones alias one static tensor: read or
        clone before the next step)"""
        n_full = self.begin_epoch(order)
        for _ in range(n_full):
            yield self.step()
        if n_full and self.dropped_graphs():
            raise ops.GaeHipError(f"CapturedInductiveStep: {self.dropped_graphs()} graphs did not fit the static "
                                  f"batch buffers ({self.cap_nodes} rows / {self.cap_edges} edges) and were left out")
        tail = self.tail_step()
        if tail is not None:
            yield tail

    def batch_sizes(self):
        """true (nodes, edges) of the batch the last replay trained on (host read-back; debugging / tests)"""
        n, e = self.counts[:2].tolist()
        return int(n), int(e)
