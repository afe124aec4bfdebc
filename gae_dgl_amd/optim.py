"""Adam on the library's one-launch HIP kernel (gae_adam_step, include/gae_hip.h).

Same update rule, defaults and constructor as ``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` -- the
optimiser of the reference's Trainer (train_inductive.py:40,50-52; train_transductive.py:43,66-68) -- without
amsgrad / maximize.  One step counter per parameter group (torch keeps one per tensor; the two agree whenever
every parameter receives a gradient on every step, as in the reference's loops).  The counter lives on the device,
so the step can be captured in a HIP graph (``capture.CapturedTrainStep``) like ``torch.optim.Adam(capturable=True)``;
lr / betas / eps / weight_decay are launch arguments and therefore frozen into a captured graph --
``CapturedTrainStep`` re-captures when it sees them change (an LR scheduler keeps working).
``state_dict()`` / ``load_state_dict()`` use torch.optim.Adam's layout (per-parameter ``step``, ``exp_avg``,
``exp_avg_sq``), so checkpoints move between the two optimisers.  PyTorch's fused Adam takes two
multi-tensor launches (about 17 us on an MI355X) for the 16 k parameters of a GAE; this one takes one short launch."""
import ctypes

import torch

from . import _lib, ops
from .ops import _stream, _on_device


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("Adam: hyper-parameter out of range")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                      capturable=True))
        self._counters = {}      # (group index, chunk) -> device int64[6]: steps taken, ticket, cached beta^t

    def _moments(self, p):
        st = self.state[p]
        if not st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st["exp_avg"], st["exp_avg_sq"]

    def materialize_state(self):
        """create the moments and the device-side step counters of every parameter NOW.  A HIP-graph capture must
        find them in place: state that step() creates lazily INSIDE a capture lives in the graph's private pool and
        its zero fill becomes a captured node -- every replay would restart the moments and the step count at zero."""
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.requires_grad]
            for p in ps:
                self._moments(p)
            for c in range((len(ps) + _lib.ADAM_MAX_TENSORS - 1) // _lib.ADAM_MAX_TENSORS):
                self._counter(gi, c, ps[0].device)

    def _counter(self, gi, chunk, dev):
        key = (gi, chunk)
        if key not in self._counters:
            self._counters[key] = torch.zeros(_lib.ADAM_STATE_WORDS, dtype=torch.int64, device=dev)
            resume = getattr(self, "_resume_steps", None)
            if resume is not None and gi < len(resume):
                self._counters[key][0] = resume[gi]
        return self._counters[key]

    def steps_taken(self, group=0, chunk=0):
        """number of steps the device-side counter of a parameter group has seen (host read-back)"""
        c = self._counters.get((group, chunk))
        if c is not None:
            return int(c[0])
        resume = getattr(self, "_resume_steps", None)      # loaded, no step taken yet: the checkpoint's count
        return int(resume[group]) if resume is not None and group < len(resume) else 0

    def hyper_params(self):
        """the values a captured launch has frozen in"""
        return tuple((g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]) for g in self.param_groups)

    def state_dict(self):
        sd = super().state_dict()
        for gi, group in enumerate(sd["param_groups"]):
            step = torch.tensor(float(self.steps_taken(gi)))
            for idx in group["params"]:
                if idx in sd["state"]:
                    sd["state"][idx]["step"] = step.clone()      # torch.optim.Adam keeps one per tensor
        return sd

    def load_state_dict(self, state_dict):
        steps = []
        for group in state_dict["param_groups"]:
            seen = [float(state_dict["state"][i]["step"]) for i in group["params"]
                    if i in state_dict["state"] and "step" in state_dict["state"][i]]
            steps.append(int(max(seen)) if seen else 0)
        # moment tensors that exist stay where they are, too: a captured HIP graph has their addresses baked in, so the
        # loaded values are copied INTO them (torch.optim.Optimizer.load_state_dict would bind new tensors, and the
        # replays would go on updating the old ones with the checkpoint's step count)
        old = {p: (st.get("exp_avg"), st.get("exp_avg_sq")) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            st.pop("step", None)
            for key, t_old in zip(("exp_avg", "exp_avg_sq"), old.get(p, (None, None))):
                t_new = st.get(key)
                if t_old is not None and t_new is not None and t_new is not t_old and t_old.shape == t_new.shape:
                    t_old.copy_(t_new)
                    st[key] = t_old
        # counters that exist stay where they are (a captured HIP graph holds their addresses): set them to the
        # checkpoint's count and invalidate their cached beta^t (zeros = "recompute with pow", see optim.hip);
        # counters of groups that have not stepped yet are created at the resume point by step()
        for (gi, _), c in self._counters.items():
            c.zero_()
            if gi < len(steps):
                c[0] = steps[gi]
        self._resume_steps = steps

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            for c0 in range(0, len(ps), _lib.ADAM_MAX_TENSORS):
                chunk = ps[c0:c0 + _lib.ADAM_MAX_TENSORS]
                dev = chunk[0].device
                arr = (_lib.AdamTensor * len(chunk))()
                keep = []
                for k, p in enumerate(chunk):
                    if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                        raise _lib.GaeHipError("gae_dgl_amd.optim.Adam: parameters must be contiguous fp32 tensors "
                                               "on one AMD GPU")
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    if g.dtype != torch.float32 or g.is_sparse:
                        raise _lib.GaeHipError("gae_dgl_amd.optim.Adam: dense fp32 gradients only")
                    m, v = self._moments(p)
                    keep.append(g)
                    pend = ops.current_step().take_partials(p.grad)       # deferred reduction (ops.StepContext)
                    if pend is not None:
                        keep.append(pend[0])
                        arr[k] = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                                 pend[1], pend[2], pend[3], pend[4], pend[5])
                    else:
                        arr[k] = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                                 None, 0, 0, 1, 1)
                key = (gi, c0 // _lib.ADAM_MAX_TENSORS)        # one device counter pair per chunk of 16 tensors
                self._counter(key[0], key[1], dev)
                counters = self._counters
                # the step's loss may have left its final reduction to this launch (ops.deferred_loss_finalize)
                step = ops.current_step()
                pend_tail = step.take_loss_tail()
                if pend_tail is not None and pend_tail[1][0].device != dev:
                    step.tails.append(pend_tail)
                    pend_tail = None
                with _on_device(dev):
                    _lib.call("gae_x_adam_step_tail", arr, len(chunk), float(group["lr"]), float(group["betas"][0]),
                              float(group["betas"][1]), float(group["eps"]), float(group["weight_decay"]),
                              ctypes.c_void_p(counters[key].data_ptr()),
                              ctypes.byref(pend_tail[0]) if pend_tail is not None else None, _stream())
        return loss
