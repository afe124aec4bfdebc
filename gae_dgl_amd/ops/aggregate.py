"""The aggregation M = A H (K1 / K2): launch plans (packed neighbour table, skew plan, XCD-pinned rows, block-diagonal
cuts), the products themselves and their autograd Function.

Part of the package gae_dgl_amd.ops (one module until round 6).  Functions look each other up in the PACKAGE
namespace (`_ops.<name>`) when they run: setting a flag or replacing a function on `gae_dgl_amd.ops` reaches every caller."""
import ctypes
import os

import torch

import gae_dgl_amd.ops as _ops
from .. import _lib
from .._lib import ACT_IDENTITY, GaeHipError
from ._base import _dtype_code, _f32, _gpu, _on_device, _ptr, _rowmajor, _stream, _workspace

__all__ = [
    'SKEW_THRESHOLD', 'SKEW_SEGMENT', 'SKEW_MIN_MAXDEG', 'TABLE_MAX_ROW', 'TILE_MIN_F', 'SCATTER_L2_BYTES',
    'ELL_MAX_ROWS', 'LIGHT_LIST', 'INT32_MAX', 'SpmmPlan', 'ell_width_for', 'ELL_OVERFLOW_SHARE',
    'ell_width_for_degrees', 'table_plan', 'HOT_COLUMNS', 'HOT_MIN_EDGES', 'HOMED_MIN_DEGREE', 'HOMED_MIN_EDGES',
    'column_home', 'spmm_plan', 'gather_distance', 'gather_scattered', 'BLOCKDIAG_GRAPHS', 'BLOCKDIAG_MIN_BLOCKS',
    'BLOCKDIAG_MAX_EDGES', 'BlockDiag', 'spmm_raw', 'spmm_ep_raw', '_scattered', 'SpMMFunction',
    'FUSED_LAYER_MAX_IN', 'FUSED_LAYER_MAX_OUT', 'spmm',
]


# ------------------------------------------------------------------ raw kernels
SKEW_THRESHOLD = 8       # rows with more in-edges than this go through the segment kernels


SKEW_SEGMENT = 512       # edges per segment (multiple of 64; RMAT s24: 256 -> 6.81 ms, 512 -> 6.65, 1024 -> 6.59)


SKEW_MIN_MAXDEG = 64     # graphs whose longest row is shorter need no plan (3 extra launches would not pay)


# ... and graphs small enough for a packed neighbour table (ELL_MAX_ROWS) none up to this row length: the table kernels
# take a row that outgrows its 16 slots with the whole wave (spmm_ell.hip: ell_long_row; 64 column ids per load, 8 rows
# in flight per lane), the fused loss's edge kernel likewise -- the real Planetoid graphs (longest rows 168 / 99 / 171)
# run the same launches as the uniform synthetic ones (tools/r05/hubs.sh: step + 7 .. 12 %).  GAE_TABLE_MAXDEG overrides.
TABLE_MAX_ROW = int(os.environ.get("GAE_TABLE_MAXDEG", 1024))


TILE_MIN_F = 64          # XCD feature tiles are only considered for rows wider than one lane group (16 vectors)


SCATTER_L2_BYTES = 2 << 20   # half of one XCD's 4 MiB L2: the window of H rows a row block can expect to find cached


ELL_MAX_ROWS = 1 << 18   # packed neighbour table only for graphs whose launches are latency-bound, not byte-bound


LIGHT_LIST = True        # skew plans carry the list of their rows with 1 .. threshold edges (gae_spmm_plan::light_desc)


INT32_MAX = 2 ** 31 - 1


class SpmmPlan:
    """per-CSR acceleration data of gae_spmm_csr (gae_spmm_plan in include/gae_hip.h): the degree-skew plan (light-row
    list, mid rows cut into segments, XCD-pinned very long rows -- built on the device by csrc/plan_build.hip) and / or
    the packed neighbour table.  ``parts``: dict of the device arrays by their field name in gae_spmm_plan."""

    def __init__(self, threshold, segment, parts=None, ell=None, ell_width=None, n_virtual=0, mid_tagged=False):
        parts = dict(parts or {})
        g = parts.get
        self.parts = parts
        self.threshold, self.segment = int(threshold), int(segment)
        self.ell = ell
        self.ell_width = (ell_width or _lib.SPMM_ELL_WIDTH) if ell is not None else 0
        self.n_heavy = 0 if g("heavy_rows") is None else int(g("heavy_rows").numel())
        self.n_segments = 0 if g("seg_heavy") is None else int(g("seg_heavy").numel())
        self.seg_desc = g("seg_desc")
        self.light_desc = g("light_desc")
        self.n_light = 0 if self.light_desc is None else int(self.light_desc.shape[0])
        self.mid_ids = g("mid_indices")
        self.hot_indices = self.mid_ids if mid_tagged else None       # (the tagged ids of the segment kernel, if any)
        self.homed = None
        if g("vh_rows") is not None:
            self.homed = dict(rows=g("vh_rows"), cols=g("vh_indices"), desc=g("vh_desc"), part_ptr=g("vh_part_ptr"),
                              part_pos=g("vh_part_pos"), n_edges=int(g("vh_indices").numel()), n_virtual=int(n_virtual))
        # heavy_rows first: tests read tensors[:3]; everything listed here is kept alive and counted as plan bytes
        self.tensors = tuple(parts.get(k) for k in ("heavy_rows", "heavy_seg_base", "seg_heavy", "seg_desc", "light_desc",
                                                    "mid_indices", "vh_rows", "vh_indices", "vh_desc", "vh_part_ptr",
                                                    "vh_part_pos")) + (ell,)
        ptr = lambda t: None if t is None else t.data_ptr()
        c = _lib.SpmmPlan()
        c.threshold, c.segment_edges = self.threshold, self.segment
        c.n_heavy, c.n_segments = self.n_heavy, self.n_segments
        c.heavy_rows, c.heavy_seg_base, c.seg_heavy = ptr(g("heavy_rows")), ptr(g("heavy_seg_base")), ptr(g("seg_heavy"))
        c.ell, c.ell_width = ptr(ell), self.ell_width
        c.seg_desc = ptr(self.seg_desc)
        c.light_desc, c.n_light = ptr(self.light_desc), self.n_light
        c.mid_indices, c.mid_tagged = ptr(self.mid_ids), 1 if (mid_tagged and self.mid_ids is not None) else 0
        if self.homed is not None:
            c.vh_n_rows, c.vh_n_virtual = int(g("vh_rows").numel()), int(n_virtual)
            c.vh_rows, c.vh_indices, c.vh_desc = ptr(g("vh_rows")), ptr(g("vh_indices")), ptr(g("vh_desc"))
            c.vh_part_ptr, c.vh_part_pos = ptr(g("vh_part_ptr")), ptr(g("vh_part_pos"))
        self.c = c

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.tensors if t is not None)

    def set_skip_rows(self, mask, covers_all_empty=False):
        """uint8 [n_rows] (or None): rows WITHOUT edges that a product launched with ``skip_dead=True`` need not write
        (gae_spmm_plan::skip_rows, GAE_SPMM_SKIP_ROWS) -- the caller's consumers treat them as zero without reading them.
        ``covers_all_empty``: the mask marks EVERY row without edges of this CSR (the empty-row stream is then not launched)"""
        self.skip_rows = mask
        self.c.skip_rows = None if mask is None else mask.data_ptr()
        self.c.reserved2 = 1 if (mask is not None and covers_all_empty) else 0


def ell_width_for(max_deg):
    """narrowest packed-table width (4, 8 or 16 slots) that holds every row of a graph whose longest (light) row
    has ``max_deg`` edges; rows longer than 16 continue from the CSR arrays"""
    return 4 if max_deg <= 4 else 8 if max_deg <= 8 else _lib.SPMM_ELL_WIDTH


ELL_OVERFLOW_SHARE = 0.01     # rows that may continue from the CSR arrays (they take a slower path in the kernel)


def ell_width_for_degrees(deg, cap=None):
    """narrowest table width that holds all but ELL_OVERFLOW_SHARE of the rows (``deg``: device tensor of row
    lengths; rows above ``cap`` are heavy rows of a skew plan and do not count): a molecule set whose atoms have at
    most 4 bonds except for a handful gets 4 slots = 16 bytes per row instead of 64.  One host read-back."""
    if deg.numel() == 0:
        return 4
    d = deg if cap is None else deg[deg <= cap]
    if d.numel() == 0:
        return 4
    over = torch.stack([(d > 4).float().mean(), (d > 8).float().mean()]).tolist()
    return 4 if over[0] <= _ops.ELL_OVERFLOW_SHARE else 8 if over[1] <= _ops.ELL_OVERFLOW_SHARE else _lib.SPMM_ELL_WIDTH


def table_plan(table, ell_width):
    """plan that only carries an already-built packed neighbour table (no heavy rows)"""
    return _ops.SpmmPlan(_ops.SKEW_THRESHOLD, _ops.SKEW_SEGMENT, None, table, ell_width)


HOT_COLUMNS = 65536          # columns tagged hot for the mid rows of a skew plan (RMAT s24, F = 32: 2 k 9.2 ms, 8 k 7.8, 16 k 7.2,


#                              32 k 6.9, 64 k 6.8, 256 k 7.2; untagged 7.8 -- streaming the moderately hot rows hurts)
HOT_MIN_EDGES = 1 << 22      # graphs with fewer edges fit the caches anyway


HOMED_MIN_DEGREE = 256       # rows with more in-edges than this are gathered XCD-pinned ("homed") ...


HOMED_MIN_EDGES = 1 << 24    # ... when they hold at least this many edges together (RMAT s24: 175 M of 268 M)


# (round 4: the pinned part carries NO hot tags -- with every column going through one L2, streaming loads for the
#  other columns cost more than they save: RMAT s24 pinned launch 1.90 ms tagged, 1.83 ms plain)


def column_home(cols):
    """XCD (0..7) through whose L2 a column is gathered in the pinned part of a plan: a multiplicative hash (the low
    bits of the hub ids of an R-MAT graph are all zero: ``col % 8`` would put 44 % of the edges on one XCD).  The
    device builder (csrc/plan_build.hip) uses the same function."""
    return ((cols.to(torch.int64) * 2654435761) >> 13) & 7


def spmm_plan(indptr, threshold=None, segment=None, indices=None, ell=None, ell_width=None, hot=None, n_cols=None, homed=None):
    """Build the plan of a CSR on the device (gae_spmm_plan_sizes / _build_rows / _build_pinned; no torch kernels), or
    None when it needs none.  Skew part: with the default threshold only for graphs whose longest row has more than
    SKEW_MIN_MAXDEG edges (TABLE_MAX_ROW for graphs that get a packed neighbour table).  Packed neighbour table: when ``indices`` is given and the graph has at most ELL_MAX_ROWS
    rows (``ell`` = True / False overrides); ``ell_width`` 4 / 8 / 16 slots per row, default: the narrowest that holds
    the longest light row.  ``hot`` (default: skew plans of graphs with at least HOT_MIN_EDGES edges when ``indices``
    is given; ``n_cols`` = columns of the CSR, default: its rows): the mid rows read a compact copy of their column
    ids with the sign bit on the HOT_COLUMNS most gathered columns, which lets the kernel stream the rarely gathered
    rows past the L2.  ``homed``: XCD-pinned regrouping of the rows with more than HOMED_MIN_DEGREE edges (default:
    when they hold at least HOMED_MIN_EDGES edges; True / False forces)."""
    auto = threshold is None
    threshold = _ops.SKEW_THRESHOLD if threshold is None else int(threshold)
    segment = _ops.SKEW_SEGMENT if segment is None else int(segment)
    _gpu(indptr, "indptr")
    dev = indptr.device
    n = indptr.numel() - 1
    want_ell = (indices is not None and 0 < n <= _ops.ELL_MAX_ROWS) if ell is None else bool(ell)
    if want_ell and indices is None:
        raise GaeHipError("spmm_plan: the packed neighbour table needs `indices`")
    if n <= 0:
        return None
    lib = _lib.load()
    with _on_device(dev):
        tiny = torch.empty(256, dtype=torch.uint8, device=dev)
        sizes = (ctypes.c_int64 * 8)()
        t2 = max(_ops.HOMED_MIN_DEGREE, threshold) if (indices is not None and homed is not False) else _ops.INT32_MAX
        _lib.call("gae_spmm_plan_sizes", _ptr(indptr), n, threshold, t2, segment, sizes, _ptr(tiny), tiny.numel(), _stream())
        nl, nm, sm, em, npin, epin, max_deg = (int(sizes[k]) for k in range(7))
        heavy = (nm + npin) > 0 and not (auto and max_deg <= (_ops.TABLE_MAX_ROW if want_ell else _ops.SKEW_MIN_MAXDEG))
        if not heavy and not want_ell:
            return None
        pin = heavy and npin > 0 and t2 != _ops.INT32_MAX and (bool(homed) or epin >= _ops.HOMED_MIN_EDGES)
        if heavy and npin > 0 and not pin:           # the long rows stay ordinary segmented rows
            t2 = _ops.INT32_MAX
            _lib.call("gae_spmm_plan_sizes", _ptr(indptr), n, threshold, t2, segment, sizes, _ptr(tiny), tiny.numel(), _stream())
            nl, nm, sm, em, npin, epin, max_deg = (int(sizes[k]) for k in range(7))
        parts, n_virtual, tagged = {}, 0, False
        if heavy:
            n_edges = int(indices.numel()) if indices is not None else 0
            tagged = ((n_edges >= _ops.HOT_MIN_EDGES) if hot is None else bool(hot)) and indices is not None and em > 0
            nc = int(n_cols) if n_cols is not None else n
            i32 = lambda *shape: torch.empty(*shape, dtype=torch.int32, device=dev)
            if _ops.LIGHT_LIST and nl:
                parts["light_desc"] = i32(nl, 4)
            if nm:
                parts.update(heavy_rows=i32(nm), heavy_seg_base=i32(nm), seg_heavy=i32(sm), seg_desc=i32(sm, 4))
                if tagged:
                    parts["mid_indices"] = i32(em)
            if npin:
                parts.update(vh_rows=i32(npin), vh_part_ptr=i32(npin + 1))
            sb = int(lib.gae_spmm_plan_scratch_bytes(n, nc, npin, epin, segment))
            scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
            ph = (ctypes.c_int64 * 4)()
            g = parts.get
            _lib.call("gae_spmm_plan_build_rows", _ptr(indptr), _ptr(indices), n, nc, threshold, t2, segment, sizes,
                      int(_ops.HOT_COLUMNS if tagged else 0), _ptr(g("light_desc")), _ptr(g("heavy_rows")),
                      _ptr(g("heavy_seg_base")), _ptr(g("seg_heavy")), _ptr(g("seg_desc")), _ptr(g("mid_indices")),
                      _ptr(g("vh_rows")), _ptr(g("vh_part_ptr")), _ptr(scratch), sb, ph, _stream())
            if npin:
                parts["vh_indices"] = i32(epin)
                _lib.call("gae_spmm_plan_build_pinned", _ptr(indptr), _ptr(indices), n, nc, segment, sizes, _ptr(g("vh_rows")),
                          _ptr(parts["vh_indices"]), None, None, _ptr(scratch), sb, ph, _stream())
                n_virtual = int(ph[1])
                parts["vh_desc"] = i32(n_virtual, 4)
                parts["vh_part_pos"] = i32(int(ph[0]))
                _lib.call("gae_spmm_plan_build_pinned", _ptr(indptr), _ptr(indices), n, nc, segment, sizes, _ptr(g("vh_rows")),
                          _ptr(parts["vh_indices"]), _ptr(parts["vh_desc"]), _ptr(parts["vh_part_pos"]), _ptr(scratch), sb, ph,
                          _stream())
            del scratch
        table = None
        if want_ell:
            if ell_width is None:
                ell_width = _ops.ell_width_for_degrees(indptr[1:] - indptr[:-1], threshold if heavy else None)
            table = torch.empty(n * ell_width, dtype=torch.int32, device=dev)
            _lib.call("gae_spmm_ell_build", _ptr(indptr), _ptr(indices), n, ell_width,
                      threshold if heavy else 2 ** 31 - 1, _ptr(table), _stream())
    return _ops.SpmmPlan(threshold, segment, parts, table, ell_width, n_virtual, tagged)


def gather_distance(indptr, indices):
    """median |column id - row id| over the edges (0 for an edge-less graph).  One host read-back; computed once
    per graph."""
    n = indptr.numel() - 1
    e = indices.numel()
    if n == 0 or e == 0:
        return 0
    deg = (indptr[1:] - indptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n, device=indptr.device), deg, output_size=e)
    return int((indices.to(torch.int64) - rows).abs().median())


def gather_scattered(indptr, indices, row_bytes, distance=None):
    """True when most neighbour rows lie further from their row than half an XCD's L2 holds (SCATTER_L2_BYTES /
    row_bytes rows): concurrently processed rows then share nothing in L1 / L2 and wide launches do better with
    XCD feature tiles (GAE_SPMM_TILE)."""
    d = _ops.gather_distance(indptr, indices) if distance is None else distance
    return d * row_bytes > _ops.SCATTER_L2_BYTES


BLOCKDIAG_GRAPHS = 4          # member graphs per thread block of the block-diagonal kernel


BLOCKDIAG_MIN_BLOCKS = 8192   # fewer blocks (a 4096-molecule batch has 1024) run faster on the row-group kernel


BLOCKDIAG_MAX_EDGES = 1024    # index slice staged in LDS per block (more edges are read from global memory)


class BlockDiag:
    """row runs closed under adjacency (whole member graphs) for gae_spmm_csr_blockdiag.  The number of member
    graphs per thread block follows the row width (about three rows per lane group: F = 39 -> 2 molecules,
    F = 32 -> 4, F <= 16 -> 8); the cuts are built on first use and cached."""

    def __init__(self, node_ptr_host, device, graphs_per_block=None):
        import numpy as np
        self.node_ptr = np.asarray(node_ptr_host, dtype=np.int64)
        self.device = device
        self.fixed = graphs_per_block
        self.min_blocks = _ops.BLOCKDIAG_MIN_BLOCKS
        self.max_edges = _ops.BLOCKDIAG_MAX_EDGES
        self._cuts = {}
        self._eptr = {}

    def _graphs_per_block(self, F):
        return self.fixed or (2 if F > 32 else 4 if F > 16 else _ops.BLOCKDIAG_GRAPHS * 2)

    def cuts(self, F):
        """(block_ptr int32 device tensor, n_blocks, max_rows) for feature width F"""
        import numpy as np
        g = self._graphs_per_block(F)
        if g not in self._cuts:
            c = self.node_ptr[::g]
            if c[-1] != self.node_ptr[-1]:
                c = np.append(c, self.node_ptr[-1])
            nb = len(c) - 1
            self._cuts[g] = (torch.from_numpy(c.astype(np.int32)).to(self.device), nb,
                             int(np.diff(c).max()) if nb else 0)
        return self._cuts[g]

    def eptr(self, indptr, block_ptr):
        """edge offset of every block for this CSR (forward and transposed structures differ)"""
        key = (indptr.data_ptr(), block_ptr.data_ptr())
        if key not in self._eptr:
            self._eptr[key] = indptr.index_select(0, block_ptr.to(torch.int64)).contiguous()
        return self._eptr[key]

    def usable(self, H, F, ldh, ldm):
        if H.dtype != torch.float32 or F > 256 or ldh % 4 or ldm % 4 or len(self.node_ptr) < 2:
            return False
        g = self._graphs_per_block(F)
        if (len(self.node_ptr) - 2) // g + 1 < self.min_blocks:    # too few blocks: decided without building the cuts
            return False
        _, nb, max_rows = self.cuts(F)
        if nb < self.min_blocks or max_rows > 511 or max_rows * ldh > 16 * 256 * 4 or self.max_edges > 1024:
            return False
        return _lib.load().gae_spmm_blockdiag_lds_bytes(max_rows, self.max_edges, ldh) <= 160 * 1024


def spmm_raw(indptr, indices, H, n_rows, row_scale=None, col_scale=None, out=None, plan=None, blockdiag=None,
             out_padded=False, scattered=False, accumulate=False, skip_dead=False):
    """M = diag(row_scale) A diag(col_scale) H  (K1/K2).  ``out_padded``: the caller's ``out`` is a view of a
    row-padded buffer whose pad columns may be overwritten (always true for the buffer allocated here).
    ``scattered``: the graph's column ids lie far from the row ids (GAE_SPMM_TILE).
    ``accumulate``: ``out += ...`` (GAE_SPMM_ACCUMULATE; needs ``out``)."""
    H, ldh = _rowmajor(H, "H")
    _gpu(indptr, "indptr")
    n_cols, F = H.shape
    if out is None:
        # rows padded to 16 / 128 bytes: keeps the vector path for any F (the pad columns are never read as data)
        out = torch.empty(n_rows, _ops.padded_ld(F, H.dtype), dtype=H.dtype, device=H.device)[:, :F]
        out_padded = True
    if accumulate and out is None:
        raise GaeHipError("spmm: accumulate=True adds to `out`")
    flags = (_lib.SPMM_STORE_PAD if out_padded else 0) | (_lib.SPMM_TILE if scattered else 0) | \
        (_lib.SPMM_ACCUMULATE if accumulate else 0)
    if skip_dead and plan is not None and getattr(plan, "skip_rows", None) is not None:
        flags |= _lib.SPMM_SKIP_ROWS        # the rows marked in the plan's mask stay UNWRITTEN (see SpmmPlan.set_skip_rows)
    if accumulate:
        blockdiag = None
    out2, ldm = _rowmajor(out, "out")
    if out2 is not out:
        raise GaeHipError("spmm: `out` must be row-major with unit inner stride")
    if blockdiag is not None and n_rows == n_cols and H.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 \
            and blockdiag.usable(H, F, ldh, ldm):
        with _on_device(H.device):
            block_ptr, n_blocks, max_rows = blockdiag.cuts(F)
            eptr = blockdiag.eptr(indptr, block_ptr)

            def launch_bd():
                _lib.call("gae_spmm_csr_blockdiag", _ptr(indptr), _ptr(indices), _ptr(block_ptr), _ptr(eptr), n_blocks,
                          max_rows, blockdiag.max_edges, n_rows, _ptr(H), ldh, _ptr(out), ldm, F, _ptr(row_scale),
                          _ptr(col_scale), flags, _stream())
            if _ops.profiler is not None:
                _ops.profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch_bd)
            else:
                launch_bd()
        return out
    with _on_device(H.device):
        pc, ws, ws_bytes = None, None, 0
        if plan is not None:
            pc = ctypes.byref(plan.c)
            ws_bytes = _lib.load().gae_spmm_workspace_bytes(pc, F)
            ws = _workspace(ws_bytes, H.device)

        def launch():
            _lib.call("gae_spmm_csr", _ptr(indptr), _ptr(indices), n_rows, n_cols, _ptr(H), ldh, _ptr(out), ldm, F,
                      _dtype_code(H), _ptr(row_scale), _ptr(col_scale), pc, _ptr(ws), ws_bytes, flags, _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch)
        else:
            launch()
    return out


def spmm_ep_raw(indptr, indices, H, n_rows, plan=None, bias=None, act=ACT_IDENTITY, row_scale=None, col_scale=None,
                out=None, accumulate=False):
    """M = act(diag(rs) A diag(cs) H (+ out, ``accumulate``) + bias): gae_spmm_csr with a store-time epilogue for ANY
    plan (gae_spmm_csr_ep) -- the sparse half of a layer evaluated as act(A (H W^T) + b) on graphs whose plans carry
    degree-skew segments / XCD-pinned rows.  fp32."""
    H, ldh = _rowmajor(_f32(_gpu(H, "H"), "spmm_ep: H"), "H")
    n_cols, F = H.shape
    _f32(bias, "spmm_ep: bias")
    if accumulate and out is None:
        raise GaeHipError("spmm_ep: accumulate=True adds to `out`")
    padded = out is None
    if out is None:
        out = torch.empty(n_rows, _ops.padded_ld(F, torch.float32), dtype=torch.float32, device=H.device)[:, :F]
    out2, ldm = _rowmajor(out, "out")
    if out2 is not out:
        raise GaeHipError("spmm_ep: `out` must be row-major with unit inner stride")
    flags = (_lib.SPMM_STORE_PAD if padded else 0) | (_lib.SPMM_ACCUMULATE if accumulate else 0)
    with _on_device(H.device):
        pc, ws, ws_bytes = None, None, 0
        if plan is not None:
            pc = ctypes.byref(plan.c)
            ws_bytes = _lib.load().gae_spmm_workspace_bytes(pc, F)
            ws = _workspace(ws_bytes, H.device)

        def launch():
            _lib.call("gae_spmm_csr_ep", _ptr(indptr), _ptr(indices), n_rows, n_cols, _ptr(H), ldh, _ptr(out), ldm, F,
                      _ptr(row_scale), _ptr(col_scale), pc, _ptr(ws), ws_bytes, flags, _ptr(bias), int(act), _stream())
        if _ops.profiler is not None:
            _ops.profiler.wrap(("spmm", n_rows, n_cols, F, str(H.dtype)), launch)
        else:
            launch()
    return out


# ------------------------------------------------------------------ autograd glue
def _scattered(graph, H):
    return H.shape[1] > _ops.TILE_MIN_F and graph.scattered(H.shape[1] * H.element_size())


class SpMMFunction(torch.autograd.Function):
    """update_all(copy_src, sum) with its backward  dH = A^T dM  (gae.py:28).
    The backward's operands (CSR of A^T, plan, norm) are taken from the graph in forward(): the autograd node must
    not hold the graph itself -- the graph holds the output (``g.ndata['h']``), whose grad_fn would hold the graph
    again, a cycle only the garbage collector can free (batches of an eager epoch would pile up in HBM until it
    runs, and their AccumulateGrad nodes would stay bound to the stream of a long-finished iteration)."""

    @staticmethod
    def forward(ctx, H, graph, use_norm):
        indptr, indices = graph.csr()
        norm = graph.norm() if use_norm else None
        n = graph.number_of_nodes()
        sc = _scattered(graph, H)
        if ctx.needs_input_grad[0]:
            ctx.bwd = (graph.csc(), n, norm, graph.spmm_plan(True), graph.block_diag, sc)
        return _ops.spmm_raw(indptr, indices, H, n, norm, norm, plan=graph.spmm_plan(False),
                        blockdiag=graph.block_diag, scattered=sc)

    @staticmethod
    def backward(ctx, dM):
        (t_indptr, t_indices), n, norm, plan_t, blockdiag, sc = ctx.bwd
        return _ops.spmm_raw(t_indptr, t_indices, dM, n, norm, norm, plan=plan_t, blockdiag=blockdiag,
                        scattered=sc), None, None


FUSED_LAYER_MAX_IN, FUSED_LAYER_MAX_OUT = 64, 32      # gae_gcn_layer_fused: whole row in one lane group, <= 32 outputs


def spmm(graph, H, use_norm=False):
    return _ops.SpMMFunction.apply(H, graph, use_norm)
