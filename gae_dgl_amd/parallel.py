"""Row-sharded multi-GPU execution of the hot path.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo"
on CPU for the host-logic tests).  1-D row-block partition: rank p owns the
node block [p*B, (p+1)*B) -- its rows of A (forward structure), its rows of
A^T (backward structure), its rows of H, M and Z.

  forward   M_p  = A_p   * H     H    = exchange(H_p)      (one collective)
  backward  dH_p = A^T_p * dM    dM   = exchange(dM_p)     (one collective)

so no reduce-scatter and no float atomics are needed and every rank's result
is bit-identical to the rows the single-GPU kernel would produce.  Two exchange
modes:
  * "allgather": RCCL all-gather of the whole feature matrix (power-law graphs
    such as RMAT, where nearly every column is a boundary column);
  * "boundary":  all-to-all-v of only the remote rows a rank's block actually
    references (citation-like graphs with locality; zero traffic for
    block-diagonal molecule batches).  The local buffer keeps ascending global
    row order, so column order inside every CSR row is unchanged.
Linear weights are replicated; their gradients are all-reduced (sum).  The
fused decoder+BCE loss is evaluated per row block against the all-gathered Z
(N x 16: small) and summed with a scalar all-reduce."""
import numpy as np
import torch
import torch.distributed as dist


def block_bounds(n, world):
    """equal row blocks (last one may be short): returns int64[world+1]"""
    b = (n + world - 1) // world
    return np.minimum(np.arange(world + 1, dtype=np.int64) * b, n)


def nnz_balanced_bounds(n, src, dst, world):
    """contiguous row blocks with (almost) equal in+out edge counts: power-law graphs put most edges on
    few rows, so equal ROW blocks would leave one rank with half of the work (RMAT s24, 8 ranks: 44 %)."""
    src = torch.as_tensor(src).to(torch.int64).reshape(-1); dst = torch.as_tensor(dst).to(torch.int64).reshape(-1)
    w = torch.bincount(dst, minlength=n) + torch.bincount(src, minlength=n) + 1      # +1: rows cost something too
    c = torch.cumsum(w, 0)
    targets = (torch.arange(1, world, device=c.device, dtype=torch.float64) * (float(c[-1]) / world)).to(c.dtype)
    cuts = torch.searchsorted(c, targets).clamp(max=n).cpu().numpy().astype(np.int64)
    b = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


class LocalGroup:
    """Single-process stand-in for a process group: `world` virtual ranks whose
    blocks live in one process (validates the sharded kernels and the
    partition/exchange plans on ONE GPU or on the CPU).  The caller publishes
    the matrix a collective would assemble with ``publish(full)``; the virtual
    ranks' exchanges then read their share of it."""

    def __init__(self, world):
        self.world = world
        self.full = None

    def publish(self, full):
        self.full = full


class RowPartition:
    """Host-side plan of one rank's share of a graph (pure index bookkeeping,
    torch ops on whatever device the edge list lives on)."""

    def __init__(self, n, src, dst, rank, world, mode="allgather", balance="rows"):
        assert mode in ("allgather", "boundary") and balance in ("rows", "nnz")
        self.n, self.rank, self.world, self.mode = int(n), int(rank), int(world), mode
        self.bounds = block_bounds(self.n, self.world) if balance == "rows" else \
            nnz_balanced_bounds(self.n, src, dst, self.world)
        self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.n_local = self.r1 - self.r0
        sizes = np.diff(self.bounds)
        # equal blocks (the last may be short) allow one all_gather_into_tensor on a padded buffer
        self.block = int(sizes[0]) if world > 0 else self.n
        self.uniform = bool(np.all(sizes[:-1] == self.block) and sizes[-1] <= self.block)
        src = torch.as_tensor(src).to(torch.int64).reshape(-1)
        dst = torch.as_tensor(dst).to(torch.int64).reshape(-1)
        fwd = (dst >= self.r0) & (dst < self.r1)        # in-edges of my rows  -> A_p   (rows dst, cols src)
        bwd = (src >= self.r0) & (src < self.r1)        # out-edges of my rows -> A^T_p (rows src, cols dst)
        self.fwd_rows, self.fwd_cols = dst[fwd] - self.r0, src[fwd]
        self.bwd_rows, self.bwd_cols = src[bwd] - self.r0, dst[bwd]
        self.n_cols = {"fwd": self.padded_n, "bwd": self.padded_n}
        self.need = {}
        if mode == "boundary":
            for k, cols in (("fwd", self.fwd_cols), ("bwd", self.bwd_cols)):
                remote = cols[(cols < self.r0) | (cols >= self.r1)]
                need = torch.unique(remote)             # sorted global ids of remote rows referenced
                self.need[k] = need
                # local buffer = [remote rows below my block | own rows | remote rows above], i.e. ascending
                # GLOBAL order: the remapped CSR keeps the column order of the global CSR, so the row sums
                # are bit-identical to the single-GPU kernel's
                n_before = int((need < self.r0).sum())
                self.n_before = getattr(self, "n_before", {})
                self.n_before[k] = n_before
                pos = torch.searchsorted(need, cols)
                own = (cols >= self.r0) & (cols < self.r1)
                newc = torch.where(own, cols - self.r0 + n_before, torch.where(cols < self.r0, pos, pos + self.n_local))
                if k == "fwd":
                    self.fwd_cols = newc
                else:
                    self.bwd_cols = newc
                self.n_cols[k] = self.n_local + int(need.numel())

    @property
    def padded_n(self):
        """rows of the assembled matrix an exchange returns in all-gather mode"""
        return self.block * self.world if self.uniform else self.n


class ShardedGraph:
    """One rank's row block with its device CSRs and the exchange plan."""

    def __init__(self, n, src, dst, rank=None, world=None, group=None, mode="allgather", device=None,
                 balance="rows"):
        if isinstance(group, LocalGroup):
            assert rank is not None
            world = group.world
        else:
            rank = dist.get_rank(group) if rank is None else rank
            world = dist.get_world_size(group) if world is None else world
        self.group = group
        self.part = RowPartition(n, src, dst, rank, world, mode, balance)
        self.device = torch.device(device) if device is not None else torch.as_tensor(src).device
        self._csr = {}
        self._plan = {}
        self._a2a = {}
        if mode == "boundary":
            self._setup_boundary()

    # ------------------------------------------------------------------ structure (HIP)
    def csr(self, which="fwd"):
        if which not in self._csr:
            from . import ops
            p = self.part
            rows, cols = (p.fwd_rows, p.fwd_cols) if which == "fwd" else (p.bwd_rows, p.bwd_cols)
            self._csr[which] = ops.csr_from_coo(rows.to(self.device), cols.to(self.device), p.n_local,
                                                p.n_cols[which])
        return self._csr[which]

    def plan(self, which="fwd"):
        if which not in self._plan:
            from . import ops
            # explicit threshold: whether a row is segmented must depend on that row only, so that every
            # sharding of a graph (and the 1-rank case) produces bit-identical sums
            self._plan[which] = ops.spmm_plan(self.csr(which)[0], threshold=ops.SKEW_THRESHOLD)
        return self._plan[which]

    def n_edges(self, which="fwd"):
        p = self.part
        return int((p.fwd_rows if which == "fwd" else p.bwd_rows).numel())

    # ------------------------------------------------------------------ exchange (collectives)
    def _setup_boundary(self):
        """tell every owner which of its rows I need (one all-to-all of counts + one of ids)"""
        p = self.part
        for k in ("fwd", "bwd"):
            need = p.need[k]
            ends = torch.as_tensor(p.bounds[1:], device=need.device)
            owner = torch.searchsorted(ends, need, right=True)
            counts = torch.bincount(owner, minlength=p.world).to(torch.int64)
            if isinstance(self.group, LocalGroup):
                self._a2a[k] = dict(need=need, recv_counts=counts.tolist())
                continue
            dev = need.device
            send_counts = torch.zeros(p.world, dtype=torch.int64, device=dev)
            dist.all_to_all_single(send_counts, counts, group=self.group)
            req = torch.empty(int(send_counts.sum()), dtype=torch.int64, device=dev)
            dist.all_to_all_single(req, need, output_split_sizes=send_counts.tolist(),
                                   input_split_sizes=counts.tolist(), group=self.group)
            self._a2a[k] = dict(need=need, recv_counts=counts.tolist(), send_counts=send_counts.tolist(),
                                send_idx=req - p.r0)

    def exchange(self, h_local, which="fwd"):
        """feature rows this rank's SpMM reads: [padded_n, F] (allgather) or
        [n_local + n_needed, F] (boundary).  Collective."""
        p = self.part
        F = h_local.shape[1]
        if isinstance(self.group, LocalGroup):
            full = self.group.full
            assert full is not None and torch.equal(full[p.r0:p.r1], h_local), "publish() the assembled matrix first"
            if p.mode == "allgather":
                pad = p.padded_n - full.shape[0]
                return full if pad <= 0 else torch.cat([full, full.new_zeros(pad, F)])
            rem = full.index_select(0, self._a2a[which]["need"].to(full.device))
            nb = p.n_before[which]
            return torch.cat([rem[:nb], h_local, rem[nb:]])
        if p.mode == "allgather":
            return self.allgather_rows(h_local)
        a = self._a2a[which]
        send = h_local.index_select(0, a["send_idx"].to(h_local.device))
        recv = h_local.new_empty(sum(a["recv_counts"]), F)
        dist.all_to_all_single(recv, send, output_split_sizes=a["recv_counts"], input_split_sizes=a["send_counts"],
                               group=self.group)
        nb = p.n_before[which]          # rows owned by lower ranks arrive first (ascending global id)
        return torch.cat([recv[:nb], h_local, recv[nb:]])

    def allgather_rows(self, t_local):
        """[padded_n, F] matrix of every rank's row block (all-gather; rows >= n are zero padding)"""
        p = self.part
        if isinstance(self.group, LocalGroup):
            full = self.group.full
            pad = p.padded_n - full.shape[0]
            return full if pad <= 0 else torch.cat([full, full.new_zeros(pad, full.shape[1])])
        if p.uniform:
            pad = p.block - p.n_local
            mine = t_local if pad == 0 else torch.cat([t_local, t_local.new_zeros(pad, t_local.shape[1])])
            full = t_local.new_empty(p.padded_n, t_local.shape[1])
            dist.all_gather_into_tensor(full, mine.contiguous(), group=self.group)
            return full
        # nnz-balanced (uneven) blocks: one broadcast per owner straight into its slice of the result
        full = t_local.new_empty(p.n, t_local.shape[1])
        full[p.r0:p.r1] = t_local
        works = [dist.broadcast(full[int(p.bounds[q]):int(p.bounds[q + 1])], src=dist.get_global_rank(self.group, q)
                                if self.group is not None else q, group=self.group, async_op=True)
                 for q in range(p.world) if p.bounds[q + 1] > p.bounds[q]]
        for w in works:
            w.wait()
        return full

    def allreduce_sum(self, t):
        if not isinstance(self.group, LocalGroup):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def exchange_bytes(self, F, elem=4, which="fwd"):
        """bytes this rank RECEIVES per exchange"""
        p = self.part
        if p.mode == "allgather":
            return (p.n - p.n_local) * F * elem
        return int(p.need[which].numel()) * F * elem

    # ------------------------------------------------------------------ compute
    def spmm(self, h_local):
        return ShardedSpMMFunction.apply(h_local, self)


class ShardedSpMMFunction(torch.autograd.Function):
    """M_p = A_p exchange(H_p);  dH_p = A^T_p exchange(dM_p)"""

    @staticmethod
    def forward(ctx, h_local, sg):
        from . import ops
        ctx.sg = sg
        full = sg.exchange(h_local, "fwd")
        ip, ix = sg.csr("fwd")
        return ops.spmm_raw(ip, ix, full, sg.part.n_local, plan=sg.plan("fwd"))

    @staticmethod
    def backward(ctx, dm_local):
        from . import ops
        sg = ctx.sg
        full = sg.exchange(dm_local.contiguous(), "bwd")
        ip, ix = sg.csr("bwd")
        return ops.spmm_raw(ip, ix, full, sg.part.n_local, plan=sg.plan("bwd")), None


def sharded_encode(model, sg, x_local):
    """GAE.encode on a row block: same layers, aggregation through the sharded SpMM"""
    from . import ops
    from .gae import _act_code
    h = x_local
    for conv in model.layers:
        lin = conv.apply_mod.linear
        code = _act_code(conv.apply_mod.activation)
        m = sg.spmm(h)
        h = ops.linear(m, lin.weight, lin.bias, code if code is not None else 0)
        if code is None:
            h = conv.apply_mod.activation(h)
    return h


def sharded_loss(model, sg, x_local, mask_local=None):
    """train_inductive.py:44-48 on a row-sharded graph: each rank evaluates its
    row block of the N x N loss against the all-gathered Z and the partial
    sums are all-reduced.  Returns the global mean loss (same on every rank)."""
    from . import ops
    z_local = sharded_encode(model, sg, x_local)
    return ops.sharded_decoder_bce(z_local, mask_local, sg)


def allreduce_grads(params, group=None):
    """replicated Linear weights: sum the row-block gradients (bucketed into one flat all-reduce)"""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
