"""Seeded synthetic workloads with the shapes of the reference's datasets
(SURVEY.md section 8(d)); the real Planetoid / ZINC-250k files are not
available offline.  Generators return host numpy arrays (citation, zinc) or
device tensors (rmat)."""
import numpy as np
import torch

CITATION = {  # N, E (directed, symmetric), F
    "cora": (2708, 10556, 1433),
    "citeseer": (3327, 9228, 3703),
    "pubmed": (19717, 88651, 500),
}


PLANETOID_MAX_DEGREE = {"cora": 168, "citeseer": 99, "pubmed": 171}     # longest rows of the real Planetoid graphs


def _planetoid_pairs(n, half, dmax, rng):
    """``half`` undirected pairs whose degree sequence is heavy-tailed like the real citation graphs': degrees drawn
    from P(d) ~ (d + d0)^-3.5 on 1 .. dmax (d0 fitted to the mean degree: about 40 % of the nodes have one neighbour,
    0.1 - 0.2 % more than 64 -- Cora's real graph: 168, 78, 74, 65 ...), the largest forced to dmax, stubs paired at
    random (configuration model); self-pairs and repeated pairs are redrawn uniformly."""
    d = np.arange(1, dmax + 1, dtype=np.float64)
    target = 2.0 * half / n
    lo, hi = -0.99, 50.0
    for _ in range(60):                                   # the mean degree grows with the shift
        d0 = 0.5 * (lo + hi)
        w = (d + d0) ** -3.5
        lo, hi = (d0, hi) if (w * d).sum() / w.sum() < target else (lo, d0)
    w = (d + d0) ** -3.5
    deg = rng.choice(np.arange(1, dmax + 1), size=n, p=w / w.sum())
    deg[rng.integers(0, n)] = dmax
    stubs = np.repeat(np.arange(n), deg)
    rng.shuffle(stubs)
    m = min(half, stubs.size // 2)
    a, b = stubs[:m].copy(), stubs[m:2 * m].copy()
    if m < half:                                          # (the draw came out short: the rest uniformly)
        a = np.concatenate([a, rng.integers(0, n, half - m)]); b = np.concatenate([b, rng.integers(0, n, half - m)])
    for _ in range(8):
        key = np.minimum(a, b) * n + np.maximum(a, b)
        _, first = np.unique(key, return_index=True)
        bad = np.ones(half, bool); bad[first] = False
        bad |= a == b
        if not bad.any():
            break
        a[bad] = rng.integers(0, n, int(bad.sum())); b[bad] = rng.integers(0, n, int(bad.sum()))
    return a, b


def citation_graph(name, seed=0, degrees="uniform"):
    """E/2 undirected pairs, mirrored, no self-loops; X = sparse non-negative rows normalised to sum 1 (as DGL's
    citation loader).  ``degrees``: "uniform" (pairs sampled uniformly: longest row ~ 20) or "planetoid" (heavy-tailed
    degree sequence with the real graph's longest row: a few hubs of 100 - 170 neighbours)."""
    n, e, f = CITATION[name]
    rng = np.random.default_rng(seed)
    half = e // 2
    if degrees not in ("uniform", "planetoid"):
        raise ValueError(f"degrees: 'uniform' or 'planetoid', not {degrees!r}")
    a = rng.integers(0, n, half); b = rng.integers(0, n, half)
    same = a == b
    b[same] = (b[same] + 1 + rng.integers(0, n - 1, int(same.sum()))) % n
    if degrees == "planetoid":      # (its own generator: X below is the same matrix for both degree profiles)
        prng = np.random.default_rng(seed + 7919)
        a, b = _planetoid_pairs(n, half, PLANETOID_MAX_DEGREE[name], prng)
        same = a == b
        b[same] = (b[same] + 1 + prng.integers(0, n - 1, int(same.sum()))) % n
    src = np.concatenate([a, b]); dst = np.concatenate([b, a])
    if e % 2:  # Pubmed's directed count is odd: one extra directed edge
        src = np.append(src, a[0]); dst = np.append(dst, (a[0] + 1) % n)
    nnz_per_row = max(1, int(0.0127 * f) if name != "pubmed" else 50)
    X = np.zeros((n, f), np.float32)
    cols = rng.integers(0, f, (n, nnz_per_row))
    vals = rng.random((n, nnz_per_row)).astype(np.float32) + 0.05
    np.put_along_axis(X, cols, vals, axis=1)
    X /= X.sum(1, keepdims=True)
    return n, src.astype(np.int64), dst.astype(np.int64), X


def zinc_like(n_graphs=249455, seed=0):
    """ZINC-250k-shaped molecule set as one block-diagonal dataset:
    n ~ clip(round(N(23.2, 4.6)), 6, 38) atoms; a random tree whose parents lie
    1..3 positions back (degree <= 4) plus Poisson(2.75) ring-closing bonds to
    the atom 5 back; both bond directions (prepare_data.py:56-65); 39 one-hot
    style binary features (prepare_data.py:14-16,31-36).
    Returns graph_ptr[int64 G+1], src, dst (GLOBAL node ids, int64), X[N,39]."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.rint(rng.normal(23.2, 4.6, n_graphs)), 6, 38).astype(np.int64)
    gptr = np.zeros(n_graphs + 1, np.int64)
    np.cumsum(sizes, out=gptr[1:])
    N = int(gptr[-1])
    gid = np.repeat(np.arange(n_graphs), sizes)
    local = np.arange(N) - gptr[gid]
    # tree bonds
    child = np.nonzero(local > 0)[0]
    back = np.minimum(rng.integers(1, 4, child.size), local[child])
    parent = child - back
    # ring closures
    n_ring = np.minimum(rng.poisson(2.75, n_graphs), np.maximum(sizes - 5, 0))
    rg = np.repeat(np.arange(n_graphs), n_ring)
    rl = 5 + (rng.random(rg.size) * (sizes[rg] - 5)).astype(np.int64)
    ra = gptr[rg] + rl
    rb = ra - 5
    a = np.concatenate([child, ra]); b = np.concatenate([parent, rb])
    # interleave both directions per bond like prepare_data.py:61-64
    src = np.stack([a, b], 1).reshape(-1); dst = np.stack([b, a], 1).reshape(-1)
    X = np.zeros((N, 39), np.float32)
    for lo, w in ((0, 23), (23, 6), (29, 5), (34, 4)):
        X[np.arange(N), lo + rng.integers(0, w, N)] = 1.0
    X[:, 38] = rng.random(N) < 0.3
    return gptr, src.astype(np.int64), dst.astype(np.int64), X


def rmat_edges(scale=24, edge_factor=16, abcd=(0.57, 0.19, 0.19, 0.05), seed=0, device="cuda",
               chunk=1 << 22, part=None):
    """R-MAT edge list on the device (directed, duplicates kept): (src, dst) int64.

    The list is made of chunks of ``chunk`` edges, each drawn from its own generator (seed, chunk index), so any
    range of chunks can be produced without the ones before it.  ``part`` = (index, count): only the index-th of
    ``count`` contiguous slices of the list (whole chunks; the slices of all indices concatenate to the full list) --
    what one rank of ``count`` generates when no process holds the whole edge list (parallel.ShardedGraph.
    from_edge_slice).  ``part`` None: all edge_factor * 2^scale edges."""
    n_edges = edge_factor << scale
    a, b, c, _ = abcd
    n_chunks = (n_edges + chunk - 1) // chunk
    if part is None:
        c0, c1 = 0, n_chunks
    else:
        idx, cnt = part
        c0, c1 = n_chunks * idx // cnt, n_chunks * (idx + 1) // cnt
    lo_all, hi_all = c0 * chunk, min(c1 * chunk, n_edges)
    src = torch.empty(max(hi_all - lo_all, 0), dtype=torch.int64, device=device)
    dst = torch.empty_like(src)
    gen = torch.Generator(device=device)
    for ci in range(c0, c1):
        lo = ci * chunk
        m = min(chunk, n_edges - lo)
        gen.manual_seed((int(seed) * 0x9E3779B1 + ci) & 0x7FFFFFFFFFFFFFFF)
        s = torch.zeros(m, dtype=torch.int64, device=device)
        d = torch.zeros(m, dtype=torch.int64, device=device)
        for _ in range(scale):
            r = torch.rand(m, device=device, generator=gen)
            sbit = (r >= a + b).to(torch.int64)                       # quadrants c, d -> row bit 1
            dbit = ((r >= a) & (r < a + b) | (r >= a + b + c)).to(torch.int64)  # quadrants b, d -> col bit 1
            s = (s << 1) | sbit
            d = (d << 1) | dbit
        src[lo - lo_all:lo - lo_all + m] = s; dst[lo - lo_all:lo - lo_all + m] = d
    return src, dst


def spmm_alg_bytes(n_rows, n_cols, nnz, F, elem=4, scaled=False):
    """compulsory bytes of one SpMM launch (int32 CSR)."""
    b = 4 * (n_rows + 1) + 4 * nnz + elem * F * n_cols + elem * F * n_rows
    if scaled:
        b += 4 * n_rows + 4 * n_cols
    return b
