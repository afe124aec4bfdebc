"""Row-sharded multi-GPU execution of the hot path.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo"
on CPU for the host-logic tests).  1-D row-block partition: rank p owns the
node block [p*B, (p+1)*B) -- its rows of A (forward structure), its rows of
A^T (backward structure), its rows of H, M and Z.

  forward   M_p  = A_p   * H     H    = exchange(H_p)      (one collective)
  backward  dH_p = A^T_p * dM    dM   = exchange(dM_p)     (one collective)

so no reduce-scatter and no float atomics are needed and every rank's result
is bit-identical to the rows the single-GPU kernel would produce (exception:
rows of more than ops.HOMED_MIN_DEGREE edges in a shard of at least
ops.HOMED_MIN_EDGES such edges, whose XCD-pinned gather adds the row in (home,
chunk) order -- deterministic, another association, <= 2e-6 of the scale
measured on RMAT s24 shards, tools/r02/rmat_shard.py).  Two exchange
modes:
  * "allgather": RCCL all-gather of the whole feature matrix (power-law graphs
    such as RMAT, where nearly every column is a boundary column);
  * "boundary":  all-to-all-v of only the remote rows a rank's block actually
    references (citation-like graphs with locality; zero traffic for
    block-diagonal molecule batches).  The local buffer keeps ascending global
    row order, so column order inside every CSR row is unchanged.
Linear weights are replicated; their gradients are all-reduced (sum).  The
fused decoder+BCE loss is evaluated per row block against the all-gathered Z
(N x 16: small) and summed with a scalar all-reduce.

``overlap=True`` splits a rank's structure by COLUMN owner: A_p = [A_own | A_remote].  The exchange is started
asynchronously, ``A_own * H_p`` runs on the rank's own rows while it is in flight, then ``M_p += A_remote * recv``
(GAE_SPMM_ACCUMULATE) reads the received rows where the collective put them -- no assembled [below | own | above]
copy exists.  Every partial is a CSR-order sum and the two are added in a fixed order, so the result is
deterministic; it differs from the single-GPU sum by fp32 rounding (a different association), where the default
``overlap=False`` is bit-identical to it."""
import numpy as np
import torch
import torch.distributed as dist

from . import transport as comm


def block_bounds(n, world):
    """equal row blocks (last one may be short): returns int64[world+1]"""
    b = (n + world - 1) // world
    return np.minimum(np.arange(world + 1, dtype=np.int64) * b, n)


# Cost of one row of a block in units of one edge (a row's in- and out-edges count 1 each).  Measured, RMAT s24 on 8
# virtual ranks, compute of one encoder step without the collectives (tools/r02/rmat_rank_step.py): a row costs what
# 16-24 edge visits cost (its share of the dense layers, the activations and the products' outputs: ~1.5 KB of HBM
# traffic per step).  Slowest rank 7.56 ms at cost 1 (7.1 M rows of the tail; kernels of that day), with today's
# kernels 4.84 ms at 8, 4.20 at 12, 3.76 at 16, 3.30 at 24 (the mean is 2.97); the hub-owning ranks receive more
# boundary rows the higher the cost (rank 0: 0.92 -> 1.08 GB per step from 1 to 24 when X is sent every step, half
# of that with cache_constant_inputs), which at 150-300 GB/s of all-to-all bandwidth evens the ranks out between 16
# and 24.
# The same figure from bytes (round 6, no virtual-rank timing involved): per step a row with in-edges costs its M1 row
# read twice and written once (3 x 128 B), its T / Z / G / dZ rows (4 x 64 B written, 4 x 64 B read) and its CSR pointers
# in four structures (~32 B): ~1.4 KB; an edge visit costs what the PMC passes measure on the RMAT s24 product,
# 23.5 GB / 2^28 edges = 88 B (profiles/pmc_traffic_r05.json: the gathered row is re-fetched 4.3 x more often than the
# compulsory count).  1.4 KB / 88 B = 16.  bench.py --row-cost overrides it for a sweep on real xGMI ranks.
ROW_COST = 16
# RCCL fast path of the boundary exchange: dist.all_to_all on views of the assembled receive buffer (no cat).  bench.py's
# collective preflight sets it to False when that primitive misbehaves on the node (all_to_all_single + cat instead).
A2A_RECEIVE_VIEWS = True


def nnz_balanced_bounds(n, src, dst, world, row_cost=None):
    """contiguous row blocks with (almost) equal cost = in+out edge counts + ``row_cost`` per row: power-law graphs
    put most edges on few rows, so equal ROW blocks would leave one rank with half of the work (RMAT s24, 8 ranks:
    44 %); a row costs something too (its share of the dense layers, of the product's output and of the exchange)."""
    src = torch.as_tensor(src).to(torch.int64).reshape(-1); dst = torch.as_tensor(dst).to(torch.int64).reshape(-1)
    w = torch.bincount(dst, minlength=n) + torch.bincount(src, minlength=n) + int(ROW_COST if row_cost is None else row_cost)
    return _weights_to_bounds(w, world)


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


def _weights_to_bounds(w, world):
    """contiguous blocks of (almost) equal total weight: int64[world + 1]"""
    n = int(w.numel())
    c = torch.cumsum(w, 0)
    targets = (torch.arange(1, world, device=c.device, dtype=torch.float64) * (float(c[-1]) / world)).to(c.dtype)
    cuts = torch.searchsorted(c, targets).clamp(max=n).cpu().numpy().astype(np.int64)
    b = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


class RowPartition:
    """Host-side plan of one rank's share of a graph (pure index bookkeeping,
    torch ops on whatever device the edge list lives on).

    ``RowPartition(n, src, dst, rank, world, ...)`` filters the rank's edges out of the WHOLE edge list (virtual ranks,
    small graphs); ``RowPartition.from_edge_slice`` builds the same plan when every rank holds only a slice of the
    list (any split of the edges over the ranks): the degree histogram is all-reduced for the block bounds and every
    edge travels once per direction to the owner of its row."""

    def __init__(self, n, src, dst, rank, world, mode="allgather", balance="rows", overlap=False, _edges=None,
                 _bounds=None):
        assert mode in ("allgather", "boundary") and balance in ("rows", "nnz")
        self.overlap = bool(overlap)
        self.n, self.rank, self.world, self.mode = int(n), int(rank), int(world), mode
        if _bounds is not None:
            self.bounds = _bounds
        else:
            self.bounds = block_bounds(self.n, self.world) if balance == "rows" else \
                nnz_balanced_bounds(self.n, src, dst, self.world)
        self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.n_local = self.r1 - self.r0
        sizes = np.diff(self.bounds)
        # equal blocks (the last may be short) allow one all_gather_into_tensor on a padded buffer
        self.block = int(sizes[0]) if world > 0 else self.n
        self.uniform = bool(np.all(sizes[:-1] == self.block) and sizes[-1] <= self.block)
        if _edges is not None:         # (from_edge_slice) the rank's edges are already here
            f_dst, f_src, b_src, b_dst = _edges
            self.fwd_rows, self.fwd_cols = f_dst - self.r0, f_src
            self.bwd_rows, self.bwd_cols = b_src - self.r0, b_dst
        else:
            src = torch.as_tensor(src).to(torch.int64).reshape(-1)
            dst = torch.as_tensor(dst).to(torch.int64).reshape(-1)
            fwd = (dst >= self.r0) & (dst < self.r1)        # in-edges of my rows  -> A_p   (rows dst, cols src)
            bwd = (src >= self.r0) & (src < self.r1)        # out-edges of my rows -> A^T_p (rows src, cols dst)
            self.fwd_rows, self.fwd_cols = dst[fwd] - self.r0, src[fwd]
            self.bwd_rows, self.bwd_cols = src[bwd] - self.r0, dst[bwd]
        # GLOBAL column ids of the local rows (the fused loss labels pairs by global id, whatever the exchange mode)
        self.cols_global = {"fwd": self.fwd_cols, "bwd": self.bwd_cols}
        self.n_cols = {"fwd": self.padded_n, "bwd": self.padded_n}
        self.need = {}
        self.split = {}            # overlap: k -> dict(own=(rows, cols), remote=(rows, cols), n_remote_cols)
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
        if self.overlap:
            for k, rows in (("fwd", self.fwd_rows), ("bwd", self.bwd_rows)):
                cols = self.cols_global[k]
                own = (cols >= self.r0) & (cols < self.r1)
                if mode == "boundary":       # remote ids -> position in the receive buffer (ascending global id)
                    rcols, n_rc = torch.searchsorted(self.need[k], cols[~own]), int(self.need[k].numel())
                else:                        # the gathered matrix is indexed by global id
                    rcols, n_rc = cols[~own], self.padded_n
                self.split[k] = dict(own=(rows[own], cols[own] - self.r0), remote=(rows[~own], rcols),
                                     n_remote_cols=n_rc)

    @classmethod
    def from_edge_slice(cls, n, src_slice, dst_slice, group=None, mode="allgather", balance="rows", overlap=False,
                        row_cost=None):
        """The plan of this rank when the ranks hold DISJOINT slices of the edge list (their union is the graph; any
        assignment of edges to slices).  Collective over ``group``:
          1. block bounds: equal rows, or -- "nnz" -- from the all-reduced in+out degree histogram (one int64[n]
             all-reduce; the same bounds on every rank as nnz_balanced_bounds gives on the whole list);
          2. every edge goes to the owner of its destination row (forward structure) and to the owner of its source
             row (backward structure): two all-to-all-v of (src, dst) pairs.
        The CSR build sorts (row, column) keys, so the structure does not depend on which slice an edge came from: the
        plan equals RowPartition(n, all src, all dst, rank, world, ...)."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        src = torch.as_tensor(src_slice).to(torch.int64).reshape(-1)
        dst = torch.as_tensor(dst_slice).to(torch.int64).reshape(-1)
        n = int(n)
        if balance == "nnz":
            w = torch.bincount(dst, minlength=n) + torch.bincount(src, minlength=n)
            if world > 1:
                comm.all_reduce(w, group=group)
            bounds = _weights_to_bounds(w + int(ROW_COST if row_cost is None else row_cost), world)
            del w
        else:
            bounds = block_bounds(n, world)
        if world == 1:
            edges = (dst, src, src, dst)
        else:
            ends = torch.as_tensor(bounds[1:], device=src.device)

            def to_owner(key):
                """(src, dst) of the edges whose ``key`` row this rank owns, from every rank's slice"""
                owner = torch.searchsorted(ends, key.contiguous(), right=True)
                order = torch.argsort(owner, stable=True)
                counts = torch.bincount(owner, minlength=world)
                recv_counts = torch.empty_like(counts)
                comm.all_to_all_single(recv_counts, counts, group=group)
                out = []
                for t in (src, dst):
                    recv = torch.empty(int(recv_counts.sum()), dtype=torch.int64, device=t.device)
                    comm.all_to_all_single(recv, t[order].contiguous(), output_split_sizes=recv_counts.tolist(),
                                           input_split_sizes=counts.tolist(), group=group)
                    out.append(recv)
                return out
            f_src, f_dst = to_owner(dst)
            b_src, b_dst = to_owner(src)
            edges = (f_dst, f_src, b_src, b_dst)
        return cls(n, None, None, rank, world, mode, balance, overlap, _edges=edges, _bounds=bounds)

    @property
    def padded_n(self):
        """rows of the assembled matrix an exchange returns in all-gather mode"""
        return self.block * self.world if self.uniform else self.n


class ShardedGraph:
    """One rank's row block with its device CSRs and the exchange plan."""

    def __init__(self, n, src, dst, rank=None, world=None, group=None, mode="allgather", device=None,
                 balance="rows", overlap=False, part=None):
        if isinstance(group, LocalGroup):
            assert rank is not None
            world = group.world
        elif part is None:
            rank = dist.get_rank(group) if rank is None else rank
            world = dist.get_world_size(group) if world is None else world
        self.group = group
        self.part = part if part is not None else RowPartition(n, src, dst, rank, world, mode, balance, overlap)
        self.timers = None         # set to {} to collect HIP-event pairs of the exchange / SpMM parts (bench.py)
        self.device = torch.device(device) if device is not None else self.part.fwd_rows.device
        self._n_edges_global = None
        self._csr = {}
        self._plan = {}
        self._a2a = {}
        # Input features are constant across training steps (a transductive graph's X, requires_grad False): the
        # remote rows of such an operand need to travel once, not once per step.  True: the forward product of an
        # operand that needs no gradient reuses the exchanged rows while the same tensor (storage, shape, version
        # counter) comes back; the product itself is evaluated every time.
        self.cache_constant_inputs = False
        self._xcache = {}
        if mode == "boundary":
            self._setup_boundary()

    @classmethod
    def from_edge_slice(cls, n, src_slice, dst_slice, group=None, mode="allgather", device=None, balance="rows",
                        overlap=False):
        """every rank passes ITS slice of the edge list (see RowPartition.from_edge_slice); collective"""
        part = RowPartition.from_edge_slice(n, src_slice, dst_slice, group, mode, balance, overlap)
        return cls(n, None, None, group=group, mode=mode, device=device, part=part)

    def n_edges_global(self):
        """edges of the whole graph (one scalar all-reduce, then cached: the loss's pos_weight needs it every step)"""
        if self._n_edges_global is None:
            t = self.allreduce_sum(torch.tensor([self.n_edges("fwd")], dtype=torch.int64, device=self.device))
            self._n_edges_global = int(t)
        return self._n_edges_global

    # ------------------------------------------------------------------ structure (HIP)
    def csr(self, which="fwd", part=None):
        """device CSR of this rank's rows: the whole structure (``part`` None; column ids as the exchange buffer
        orders them), or -- overlap -- its "own" / "remote" column part"""
        key = which if part is None else (which, part)
        if key not in self._csr:
            from . import ops
            p = self.part
            if part is None:
                rows, cols = (p.fwd_rows, p.fwd_cols) if which == "fwd" else (p.bwd_rows, p.bwd_cols)
                n_cols = p.n_cols[which]
            else:
                rows, cols = p.split[which][part]
                n_cols = p.n_local if part == "own" else p.split[which]["n_remote_cols"]
            self._csr[key] = ops.csr_from_coo(rows.to(self.device), cols.to(self.device), p.n_local, max(n_cols, 1))
        return self._csr[key]

    def csr_global(self, which="fwd"):
        """CSR of this rank's rows with GLOBAL column ids (what the fused loss reads; in all-gather mode this is
        csr(which) itself)"""
        if self.part.mode == "allgather" and which in self._csr:
            return self._csr[which]
        key = (which, "global")
        if key not in self._csr:
            from . import ops
            p = self.part
            rows = p.fwd_rows if which == "fwd" else p.bwd_rows
            self._csr[key] = ops.csr_from_coo(rows.to(self.device), p.cols_global[which].to(self.device), p.n_local,
                                              p.n)
        return self._csr[key]

    def plan(self, which="fwd", part=None):
        key = which if part is None else (which, part)
        if key not in self._plan:
            from . import ops
            # explicit threshold: whether a row is segmented must depend on that row only, so that every
            # sharding of a graph (and the 1-rank case) produces bit-identical sums (below the size at which
            # ops.spmm_plan pins the very long rows to XCDs, see the module docstring)
            p = self.part
            n_cols = p.n_cols[which] if part is None else p.n_local if part == "own" else \
                p.split[which]["n_remote_cols"]
            ip, ix = self.csr(which, part)
            # (hot-column tags: cache hints for the heavy-row kernel, values unaffected)
            self._plan[key] = ops.spmm_plan(ip, threshold=ops.SKEW_THRESHOLD, indices=ix, ell=False,
                                            n_cols=max(int(n_cols), 1))
            if self._plan[key] is not None and ip.is_cuda:
                dead = self.dead_rows(which)
                all_empty = bool(((ip[1:] == ip[:-1]) == dead.bool()).all())     # (one-off read-back at plan time)
                self._plan[key].set_skip_rows(dead, covers_all_empty=all_empty)
        return self._plan[key]

    def dead_rows(self, which="fwd"):
        """uint8 [n_local]: 1 for the rows of this rank's block without ANY edge in ``which`` direction (in the whole
        structure: own and remote columns) -- rows whose aggregate is zero whatever the operand.  The one-pass encoder
        (ShardedEncoder2Function) neither writes nor reads them (power-law graphs: most rows)."""
        key = ("dead", which)
        if key not in self._csr:
            if self.part.overlap:
                ipo, ipr = self.csr(which, "own")[0], self.csr(which, "remote")[0]
                dead = (ipo[1:] == ipo[:-1]) & (ipr[1:] == ipr[:-1])
            else:
                ip = self.csr(which)[0]
                dead = ip[1:] == ip[:-1]
            self._csr[key] = dead.to(torch.uint8).contiguous()
        return self._csr[key]

    def read_dead_rows(self):
        """uint8 [n_local]: rows without in-edges that HAVE out-edges (their row of a forward operand is gathered by
        somebody, so it must hold the value all zero-aggregate rows share)"""
        key = ("dead", "read")
        if key not in self._csr:
            self._csr[key] = (self.dead_rows("fwd").bool() & ~self.dead_rows("bwd").bool()).to(torch.uint8).contiguous()
        return self._csr[key]

    def live_rows(self, which="fwd"):
        """(int32 list of the rows WITH edges in ``which`` direction, ascending; the other direction's dead mask at those
        rows) -- the rows the dense passes of the one-pass encoder visit (list mode)"""
        key = ("live", which)
        if key not in self._csr:
            other = "bwd" if which == "fwd" else "fwd"
            rows = torch.nonzero(self.dead_rows(which) == 0).reshape(-1).to(torch.int32).contiguous()
            self._csr[key] = (rows, self.dead_rows(other)[rows.long()].contiguous())
        return self._csr[key]

    def _timed(self, name, fn):
        if self.timers is None:
            return fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record()
        self.timers.setdefault(name, []).append((e0, e1))
        return out

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
            comm.all_to_all_single(send_counts, counts, group=self.group)
            req = torch.empty(int(send_counts.sum()), dtype=torch.int64, device=dev)
            comm.all_to_all_single(req, need, output_split_sizes=send_counts.tolist(),
                                   input_split_sizes=counts.tolist(), group=self.group)
            self._a2a[k] = dict(need=need, recv_counts=counts.tolist(), send_counts=send_counts.tolist(),
                                send_idx=(req - p.r0).to(self.device))

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
        nb = p.n_before[which]          # rows owned by lower ranks arrive first (ascending global id)
        if h_local.is_cuda and h_local.dtype == torch.float32 and comm.backend(self.group) == "nccl" and A2A_RECEIVE_VIEWS:
            # HIP pack kernels + receive views: the rows to send go straight into the send buffer, the own rows
            # straight into their slot of the assembled buffer, and every peer's rows land where the local CSR
            # indexes them -- no ATen index_select / cat between two products
            from . import ops
            send = ops.rows_pack(h_local, a["send_idx"])
            full = h_local.new_empty(p.n_local + sum(a["recv_counts"]), F)
            ops.rows_pack(h_local, None, out=full[nb:nb + p.n_local])
            outs, ins, so, ro = [], [], 0, 0
            for q in range(p.world):
                ins.append(send[so:so + a["send_counts"][q]]); so += a["send_counts"][q]
                at = ro if ro < nb else ro + p.n_local                   # peers above this rank: behind the own rows
                outs.append(full[at:at + a["recv_counts"][q]]); ro += a["recv_counts"][q]
            dist.all_to_all(outs, ins, group=self.group)
            return full
        send = h_local.index_select(0, a["send_idx"].to(h_local.device))
        recv = h_local.new_empty(sum(a["recv_counts"]), F)
        comm.all_to_all_single(recv, send, output_split_sizes=a["recv_counts"], input_split_sizes=a["send_counts"],
                               group=self.group)
        return torch.cat([recv[:nb], h_local, recv[nb:]])          # (gloo: CPU host-logic tests, ranks sharing a GPU)

    def exchange_start(self, h_local, which="fwd"):
        """overlap: start the exchange of the REMOTE rows and return (buffer the remote CSR indexes, wait()).  The
        collective runs on RCCL's stream; wait() orders the caller's stream behind it."""
        p = self.part
        F = h_local.shape[1]
        if isinstance(self.group, LocalGroup):
            full = self.group.full
            assert full is not None and torch.equal(full[p.r0:p.r1], h_local), "publish() the assembled matrix first"
            if p.mode == "allgather":
                pad = p.padded_n - full.shape[0]
                return (full if pad <= 0 else torch.cat([full, full.new_zeros(pad, F)])), (lambda: None)
            return full.index_select(0, self._a2a[which]["need"].to(full.device)), (lambda: None)
        if p.mode == "allgather":
            full, works = self._allgather_async(h_local)
            return full, (lambda: [w.wait() for w in works])
        a = self._a2a[which]
        if h_local.is_cuda and h_local.dtype == torch.float32:      # (gae_rows_pack moves fp32 rows; bf16-stored features
            from . import ops                                         #  take the index_select below)
            send = ops.rows_pack(h_local, a["send_idx"])             # HIP gather straight into the send buffer
        else:
            send = h_local.index_select(0, a["send_idx"].to(h_local.device))
        recv = h_local.new_empty(sum(a["recv_counts"]), F)
        work = comm.all_to_all_single(recv, send, output_split_sizes=a["recv_counts"],
                                      input_split_sizes=a["send_counts"], group=self.group, async_op=True)
        return recv, work.wait

    def _allgather_async(self, t_local):
        """(full [padded_n or n, F], list of Work): equal blocks -> one all_gather_into_tensor; uneven (nnz-balanced)
        blocks -> ONE grouped all_gather into views of the result (RCCL: a ncclGroup of broadcasts); the gloo
        backend of the CPU tests has no uneven all_gather: one broadcast per owner there"""
        p = self.part
        if p.uniform:
            pad = p.block - p.n_local
            if pad == 0:
                mine = t_local
            elif t_local.is_cuda and t_local.dtype == torch.float32:
                from . import ops
                mine = ops.rows_pack(t_local, None, n_out_rows=p.block)       # own rows + zero pad rows, one HIP launch
            else:
                mine = torch.cat([t_local, t_local.new_zeros(pad, t_local.shape[1])])
            full = t_local.new_empty(p.padded_n, t_local.shape[1])
            return full, [comm.all_gather_into_tensor(full, mine.contiguous(), group=self.group, async_op=True)]
        full = t_local.new_empty(p.n, t_local.shape[1])
        views = [full[int(p.bounds[q]):int(p.bounds[q + 1])] for q in range(p.world)]
        return full, comm.all_gather_uneven(views, t_local, p.rank, self.group)

    def allgather_rows(self, t_local):
        """[padded_n, F] matrix of every rank's row block (all-gather; rows >= n are zero padding)"""
        p = self.part
        if isinstance(self.group, LocalGroup):
            full = self.group.full
            pad = p.padded_n - full.shape[0]
            return full if pad <= 0 else torch.cat([full, full.new_zeros(pad, full.shape[1])])
        full, works = self._allgather_async(t_local)
        for w in works:
            w.wait()
        return full

    def allreduce_sum(self, t):
        if not isinstance(self.group, LocalGroup):
            comm.all_reduce(t, group=self.group)
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


def _rank_product(sg, t_local, which, constant=False, bias=None, act=0, skip_dead=False):
    """this rank's rows of  act(A exchange(t) + bias)  (``which`` = "fwd": A_p, "bwd": A^T_p).  ``constant``: the
    operand does not change between steps (input features): its exchanged rows are kept (sg.cache_constant_inputs).
    ``skip_dead``: rows without any edge (sg.dead_rows) may stay unwritten -- the caller's consumers do not read them.
    The epilogue (bias, activation: ops.spmm_ep_raw) rides in the launch that finishes a row -- with the overlapped
    exchange that is the remote-column product, which adds to the own-column one."""
    from . import ops
    n_local = sg.part.n_local
    epi = bias is not None or act != 0
    key = (t_local.data_ptr(), tuple(t_local.shape), t_local._version) if constant else None
    hit = sg._xcache.get(which) if constant else None
    hit = hit[1] if hit is not None and hit[0] == key else None

    def product(ip, ix, H, plan, out=None, accumulate=False, last=True):
        if epi and last:
            return ops.spmm_ep_raw(ip, ix, H, n_local, plan, bias, act, out=out, accumulate=accumulate)
        return ops.spmm_raw(ip, ix, H, n_local, out=out, plan=plan, accumulate=accumulate, skip_dead=skip_dead)

    if not sg.part.overlap:
        full = hit if hit is not None else sg._timed("exchange", lambda: sg.exchange(t_local, which))
        if constant:
            sg._xcache[which] = (key, full)
        ip, ix = sg.csr(which)
        return sg._timed("spmm", lambda: product(ip, ix, full, sg.plan(which)))
    # own columns while the remote rows travel, then M += A_remote * received
    if hit is not None:
        recv, wait = hit, (lambda: None)
    else:
        recv, wait = sg._timed("exchange_start", lambda: sg.exchange_start(t_local, which))
    oip, oix = sg.csr(which, "own")
    rip, rix = sg.csr(which, "remote")
    remote = rix.numel() > 0
    out = sg._timed("spmm_own", lambda: product(oip, oix, t_local, sg.plan(which, "own"), last=not remote))
    if hit is None:
        sg._timed("exchange_wait", wait)
        if constant:
            sg._xcache[which] = (key, recv)
    if remote:
        sg._timed("spmm_remote", lambda: product(rip, rix, recv, sg.plan(which, "remote"), out=out, accumulate=True))
    return out


class ShardedSpMMFunction(torch.autograd.Function):
    """M_p = A_p exchange(H_p);  dH_p = A^T_p exchange(dM_p)"""

    _product = staticmethod(_rank_product)

    @staticmethod
    def forward(ctx, h_local, sg):
        ctx.sg = sg
        constant = bool(sg.cache_constant_inputs) and not ctx.needs_input_grad[0]
        return _rank_product(sg, h_local.contiguous(), "fwd", constant)

    @staticmethod
    def backward(ctx, dm_local):
        return _rank_product(ctx.sg, dm_local.contiguous(), "bwd"), None


LIST_MODE = True      # the dense passes of ShardedEncoder2Function visit only the rows that have in-edges (False: all rows)


class ShardedEncoder2Function(torch.autograd.Function):
    """The two-layer encoder of gae.py:36-45 on a row block, Z = A act1((A X) W1^T + b1) W2^T + b2, with the second
    layer evaluated as A (H1 W2^T) + b2 and the backward pass from G = A^T dZ (csrc/tall.hip):

        forward    M1 = A X                       product at the input width (exchange of X: once, if it is constant)
                   T = gae_linear2_fwd(M1)        one pass: H1 = act1(M1 W1^T + b1) (registers only), T = H1 W2^T
                   Z = A T + b2                   product at the OUTPUT width, bias in its epilogue (gae_spmm_csr_ep)
        backward   G = A^T dZ                     product at the output width
                   dW1, db1, dW2, db2 = gae_gcn2_bwd_dense(G, dZ, M1)     one pass; H1 recomputed (same bits), dY1 never stored

    Three products and two dense passes per step instead of three products at the hidden width and five dense
    launches; A H1 is never formed, so two of the three exchanges between ranks move the output width.  Values equal the
    reference's order up to fp32 rounding (tests/test_gpu_tall.py against the oracle).  X receives no gradient (input
    features: train_transductive.py:38)."""

    @staticmethod
    def forward(ctx, x_local, W1, b1, W2, b2, sg, act1):
        from . import ops
        ctx.sg, ctx.act1 = sg, act1
        constant = bool(sg.cache_constant_inputs)
        # rows of the block without in-edges: M1 = 0 there.  They are neither written by the product nor read by the
        # two dense passes (R-MAT s24: 70 % of the rows, 1.5 GB written and 3 GB read per step otherwise)
        dead = sg.dead_rows("fwd") if x_local.is_cuda else None
        M1 = _rank_product(sg, x_local.contiguous(), "fwd", constant, skip_dead=dead is not None)
        need = any(ctx.needs_input_grad[1:5])
        # H1 is not stored: the backward pass recomputes it, bit for bit, from the tile of M1 it reads anyway
        rows = sg.live_rows("fwd")[0] if (dead is not None and LIST_MODE) else None
        # T is the gather operand of the next aggregation and nothing else: of the rows without in-edges only those WITH
        # out-edges are ever read (here or, after the exchange, on another rank) -- isolated nodes stay unwritten
        fill = sg.read_dead_rows() if rows is not None else None
        _, T = sg._timed("dense_fwd", lambda: ops.linear2_fwd_raw(M1, W1, b1, act1, W2, want_y1=False, a_dead=dead,
                                                                  rows=rows, fill=fill))
        Z = _rank_product(sg, T, "fwd", bias=b2)
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        if need:
            ctx.save_for_backward(M1, W1, b1, W2)
        return Z

    @staticmethod
    def backward(ctx, dZ):
        from . import ops
        M1, W1, b1, W2 = ctx.saved_tensors
        sg = ctx.sg
        dZc = dZ.contiguous()
        G = _rank_product(sg, dZc, "bwd", skip_dead=True)       # rows without out-edges: G = 0, not written, not read
        dW1, db1, dW2, db2 = sg._timed("dense_bwd",
                                       lambda: ops.gcn2_bwd_dense_raw(G, dZc, None, ctx.act1, M1, W2, W1=W1, b1=b1,
                                                                      m1_dead=sg.dead_rows("fwd"),
                                                                      g_dead=sg.dead_rows("bwd"),
                                                                      rows=sg.live_rows("fwd")[0] if LIST_MODE else None,
                                                                      g_dead_listed=sg.live_rows("fwd")[1] if LIST_MODE else None))
        return None, dW1, (db1 if ctx.has_b1 else None), dW2, (db2 if ctx.has_b2 else None), None, None


def encoder2_usable(model, x_local):
    """can ShardedEncoder2Function run this model?  Two layers, widths <= 32, fused-epilogue activations, fp32 input"""
    from . import ops
    from .gae import _act_code
    if len(model.layers) != 2:
        return False
    l1, l2 = (m.apply_mod for m in model.layers)
    if _act_code(l1.activation) is None or _act_code(l2.activation) != 0:
        return False
    if x_local.requires_grad:           # the one-pass encoder returns no dX (input features: train_transductive.py:38)
        return False
    return ops.linear2_usable(x_local, l1.linear.weight.shape[0], l2.linear.weight.shape[0])


def sharded_encode(model, sg, x_local, transform_first=False):
    """GAE.encode on a row block: same layers, aggregation through the sharded SpMM.
    ``transform_first``: evaluate the LAST layer as ``A (H W^T) + b`` -- the value of the reference's
    ``(A H) W^T + b`` up to fp32 rounding -- through ShardedEncoder2Function (two-layer encoders of widths <= 32, e.g.
    BASELINE config 4: 32 -> 32 -> 16): the second aggregation, its backward and both of their exchanges move the
    output width, and the dense halves of the step are two one-pass kernels.  Other models keep the reference's order."""
    from . import ops
    from .gae import _act_code
    if transform_first and encoder2_usable(model, x_local):
        l1, l2 = (m.apply_mod for m in model.layers)
        return ShardedEncoder2Function.apply(x_local, l1.linear.weight, l1.linear.bias, l2.linear.weight, l2.linear.bias,
                                             sg, _act_code(l1.activation))
    h = x_local
    for conv in model.layers:
        lin = conv.apply_mod.linear
        code = _act_code(conv.apply_mod.activation)
        m = sg.spmm(h)
        h = ops.linear(m, lin.weight, lin.bias, code if code is not None else 0)
        if code is None:
            h = conv.apply_mod.activation(h)
    return h


def sharded_loss(model, sg, x_local, mask_local=None, transform_first=False):
    """train_inductive.py:44-48 on a row-sharded graph: each rank evaluates its
    row block of the N x N loss against the all-gathered Z and the partial
    sums are all-reduced.  Returns the global mean loss (same on every rank)."""
    from . import ops
    z_local = sharded_encode(model, sg, x_local, transform_first)
    return ops.sharded_decoder_bce(z_local, mask_local, sg)


def allreduce_grads(params, group=None, average=False):
    """replicated Linear weights: sum the gradients of all ranks (bucketed into ONE flat all-reduce: 1 808 floats for
    the 39 -> 32 -> 16 model).  Row-sharded graphs add their row blocks' shares (sum); data-parallel replicas
    (``average``) divide by the number of replicas: the step then equals one process whose loss is the mean of the
    replicas' batch losses (train_inductive.py:44-53 on N GPUs).  Every rank ends with the same bits."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    comm.all_reduce(flat, group=group)
    if average:
        flat /= float(dist.get_world_size(group))
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


class ShardedTrainStep:
    """One data-parallel training step of train_transductive.py:56-66 on a row-sharded graph: sharded encoder + row
    block of the fused loss (scalar all-reduce), backward, all-reduce of the replicated weights' gradients, optimiser.
    Every rank runs the same sequence and ends with the same weights.

    ``capture=True``: after ``warmup`` eager steps (they create the communicators, the structures, the optimiser
    state) the step -- collectives included -- is captured into one HIP graph and replayed; RCCL's work is then part
    of the graph, the host issues one launch per step.  ``mask_local``: the rank's rows of a fixed dropout mask (None
    = no dropout)."""

    def __init__(self, model, optimizer, sg, x_local, mask_local=None, transform_first=True, capture=False, warmup=2):
        self.model, self.opt, self.sg, self.x, self.mask = model, optimizer, sg, x_local, mask_local
        self.transform_first = transform_first
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        self.graph = None
        self.loss = None
        sg.n_edges_global()                                # one host read-back, outside the steps
        if capture:
            self._capture(warmup)

    def _step(self):
        from . import ops
        loss = sharded_loss(self.model, self.sg, self.x, self.mask, self.transform_first)
        ops.backward(loss, self.params)                    # autograd.grad: no stream-bound AccumulateGrad nodes
        if not isinstance(self.sg.group, LocalGroup):      # (a 1-rank process group still runs the collective)
            allreduce_grads(self.params, self.sg.group)
        self.opt.step()
        return loss.detach()

    def _capture(self, warmup):
        import gc
        if not isinstance(self.sg.group, LocalGroup) and comm.staged(self.x, self.sg.group):
            raise RuntimeError("ShardedTrainStep(capture=True) needs RCCL (backend nccl): a collective staged through "
                               "host memory synchronises the stream, which a stream capture forbids")
        from .capture import _state_outside_capture
        for w in ("fwd", "bwd"):                           # structures and plans: built outside the capture
            for part in (("own", "remote") if self.sg.part.overlap else (None,)):
                self.sg.csr(w, part); self.sg.plan(w, part)
        self.sg.csr_global("fwd"); self.sg.csr_global("bwd")
        gc.collect()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):           # >= 1: RCCL creates its communicator on first use
                self.opt.zero_grad(set_to_none=True)
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        _state_outside_capture(self.opt, max(int(warmup), 1))
        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog thread polls events while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = self._step()

    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
            return self.loss
        self.opt.zero_grad(set_to_none=True)
        self.loss = self._step()
        return self.loss
