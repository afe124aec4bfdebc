"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle
and the committed golden vectors.  Integer/index work is bit-exact; fp32 work
is checked to the north-star tolerance 1e-5 (relative to the output scale)."""
import numpy as np
import pytest
import torch

from conftest import golden_params, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return torch.device("cuda:0")


def O():
    from oracle import gae_oracle
    return gae_oracle


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / b.abs().max().clamp(min=1.0))


def t(x, dev, dtype=None):
    x = torch.as_tensor(np.asarray(x))
    return x.to(device=dev, dtype=dtype) if dtype else x.to(dev)


def rand_graph(rng, n, e, dup=True, hub=False):
    src = rng.integers(0, n, e); dst = rng.integers(0, n, e)
    if hub and n > 2:
        dst[: e // 3] = 1          # one heavy row
        src[e // 3: e // 2] = 2    # one heavy column
    if dup and e > 4:
        src[:2] = src[2:4]; dst[:2] = dst[2:4]
    return src.astype(np.int64), dst.astype(np.int64)


# ----------------------------------------------------------------- structure
def test_csr_exact_golden(golden, dev):
    from gae_dgl_amd import ops
    g = golden; n = int(g["n"])
    ip, ix = O().csr_from_coo(g["src"], g["dst"], n)
    dip, dix = ops.csr_from_coo(t(g["dst"], dev), t(g["src"], dev), n, n)
    assert dip.dtype == torch.int32 and dix.dtype == torch.int32
    assert np.array_equal(dip.cpu().numpy(), ip) and np.array_equal(dix.cpu().numpy(), ix)
    tp, tx = O().csc_from_coo(g["src"], g["dst"], n)
    dtp, dtx = ops.csr_from_coo(t(g["src"], dev), t(g["dst"], dev), n, n)
    assert np.array_equal(dtp.cpu().numpy(), tp) and np.array_equal(dtx.cpu().numpy(), tx)
    deg, norm = ops.degree_norm(dip)
    assert np.array_equal(deg.cpu().numpy().astype(np.int64), g["in_degrees"])
    assert np.array_equal(norm.cpu().numpy().reshape(-1, 1), g["norm"])
    assert np.array_equal(ops.csr_to_dense(dip, dix, n, n).cpu().numpy(), g["adj"])


@pytest.mark.parametrize("n,e", [(1, 0), (1, 3), (5, 0), (7, 1), (64, 64), (1000, 5000), (4097, 70001), (200000, 1500000)])
def test_csr_exact_random(n, e, dev):
    from gae_dgl_amd import ops
    rng = np.random.default_rng(n * 31 + e)
    src, dst = rand_graph(rng, n, e, hub=True)
    ip, ix = O().csr_from_coo(src, dst, n)
    dip, dix = ops.csr_from_coo(t(dst, dev), t(src, dev), n, n)
    assert np.array_equal(dip.cpu().numpy(), ip) and np.array_equal(dix.cpu().numpy(), ix)


def test_csr_rectangular_and_bad_ids(dev):
    from gae_dgl_amd import ops
    from gae_dgl_amd._lib import GaeHipError
    rng = np.random.default_rng(5)
    row = rng.integers(0, 37, 500); col = rng.integers(0, 1000, 500)
    ip, ix = O().csr_from_coo(col, row, 37, 1000)
    dip, dix = ops.csr_from_coo(t(row, dev), t(col, dev), 37, 1000)
    assert np.array_equal(dip.cpu().numpy(), ip) and np.array_equal(dix.cpu().numpy(), ix)
    with pytest.raises(GaeHipError):
        ops.csr_from_coo(t(np.array([0, 99]), dev), t(np.array([0, 1]), dev), 10, 10)


# ----------------------------------------------------------------- K1/K2 SpMM
@pytest.mark.parametrize("F", [1, 3, 4, 16, 32, 39, 40, 64, 100, 256, 500, 1433])
@pytest.mark.parametrize("scaled", [False, True])
def test_spmm_fp32(F, scaled, dev):
    from gae_dgl_amd import ops
    rng = np.random.default_rng(F)
    n, e = 777, 6000
    src, dst = rand_graph(rng, n, e, hub=True)
    dst[dst == 5] = 6  # row 5 has zero in-degree
    H = rng.standard_normal((n, F)).astype(np.float32)
    ip, ix = O().csr_from_coo(src, dst, n)
    norm = O().norm_from_in_degrees(O().in_degrees(dst, n)).numpy() if scaled else None
    ref = O().spmm_csr(ip, ix, torch.from_numpy(H).double(), norm, norm)
    out = ops.spmm_raw(t(ip, dev), t(ix, dev), t(H, dev), n, t(norm, dev) if scaled else None,
                       t(norm, dev) if scaled else None)
    assert rel_err(out, ref) < TOL
    assert float(out[5].abs().max()) == 0.0
    if not scaled:
        # same summation order as the C oracle => bit-exact for the un-normalised sum
        from oracle import c_oracle
        assert np.array_equal(out.cpu().numpy(), c_oracle.spmm_csr(ip, ix, H))


@pytest.mark.parametrize("F,dtype", [(32, torch.float32), (39, torch.float32), (16, torch.float32),
                                      (500, torch.float32), (64, torch.bfloat16)])
@pytest.mark.parametrize("thr,seg", [(1, 64), (8, 128), (64, 512)])
def test_spmm_skew_plan(F, dtype, thr, seg, dev):
    """heavy rows cut into segments (power-law graphs): same result as the oracle, deterministic"""
    from gae_dgl_amd import ops
    rng = np.random.default_rng(F + thr)
    n, e = 3000, 60000
    src = rng.integers(0, n, e)
    dst = (rng.integers(0, n, e).astype(np.float64) ** 3 / n ** 2).astype(np.int64)   # heavy head, long tail
    dst[:5000] = 7                                                                    # one 5000+-edge hub
    H = rng.standard_normal((n, F)).astype(np.float32)
    ip, ix = O().csr_from_coo(src, dst, n)
    norm = O().norm_from_in_degrees(O().in_degrees(dst, n)).numpy()
    dip, dix = t(ip, dev), t(ix, dev)
    plan = ops.spmm_plan(dip, threshold=thr, segment=seg)
    assert plan is not None and plan.n_heavy > 0 and plan.n_segments >= plan.n_heavy
    deg = np.diff(ip)
    assert plan.n_heavy == int((deg > thr).sum())
    assert plan.n_segments == int(np.ceil(deg[deg > thr] / seg).sum())
    Hd = t(H, dev).to(dtype)
    tol = TOL if dtype == torch.float32 else 1e-2
    for scaled in (False, True):
        sc = t(norm, dev) if scaled else None
        ref = O().spmm_csr(ip, ix, Hd.float().cpu().double(), norm if scaled else None, norm if scaled else None)
        out = ops.spmm_raw(dip, dix, Hd, n, sc, sc, plan=plan)
        assert rel_err(out.float(), ref) < tol
        out2 = ops.spmm_raw(dip, dix, Hd, n, sc, sc, plan=plan)
        assert torch.equal(out, out2)                         # bit-stable run to run
        base = ops.spmm_raw(dip, dix, Hd, n, sc, sc)          # no plan: CSR-order sums
        assert rel_err(out.float(), base.float()) < tol
    assert ops.spmm_plan(dip, threshold=10 ** 6) is None      # nothing heavy -> no plan


@pytest.mark.parametrize("F,dtype", [(16, torch.float32), (32, torch.float32), (39, torch.float32),
                                      (500, torch.float32), (1433, torch.float32), (64, torch.bfloat16)])
def test_spmm_packed_table_bit_identical(F, dtype, dev):
    """packed neighbour table (gae_spmm_ell_build): one load instead of the indptr -> indices chain, same CSR-order
    sums -- rows longer than the table continue from the CSR arrays, heavy rows of a skew plan are skipped"""
    from gae_dgl_amd import ops, _lib
    rng = np.random.default_rng(F)
    n, e = 2500, 12000
    src, dst = rand_graph(rng, n, e)
    dst[:40] = 11; dst[40:57] = 12; dst[57:73] = 13; dst[73:88] = 14     # rows of 40+, 17+, 16+, 15+ edges
    dst[dst == 5] = 6                                                     # an empty row
    H = rng.standard_normal((n, F)).astype(np.float32)
    ip, ix = O().csr_from_coo(src, dst, n)
    norm = O().norm_from_in_degrees(O().in_degrees(dst, n)).numpy()
    dip, dix = t(ip, dev), t(ix, dev)
    Hd = ops.pad_rows(t(H, dev).to(dtype))
    table = ops.spmm_plan(dip, indices=dix)                               # auto: no heavy rows, table only
    assert table is not None and table.n_heavy == 0 and table.ell is not None
    W = _lib.SPMM_ELL_WIDTH
    assert table.ell_width == W                                           # longest row > 8 edges: 16 slots
    tab = table.ell.cpu().numpy().reshape(n, W)
    deg = np.diff(ip)
    for r in (5, 11, 12, 13, 14, 100):
        k = min(int(deg[r]), W if deg[r] <= W else W - 1)
        assert np.array_equal(tab[r, :k], ix[ip[r]:ip[r] + k])
        assert (tab[r, k:] == (-2 if deg[r] > W else -1)).all() or (deg[r] > W and tab[r, W - 1] == -2)
    both = ops.spmm_plan(dip, threshold=8, segment=64, indices=dix)       # heavy rows + table
    assert both.n_heavy == int((deg > 8).sum()) and both.ell is not None
    assert both.ell_width == 8                                            # light rows have at most 8 edges
    assert int(both.ell.view(n, 8)[11, 0]) == -3
    narrow = ops.spmm_plan(dip, indices=dix, ell_width=4)                 # most rows overflow into the CSR arrays

    def same(out, base, width):
        """bit-identical on the rows that fit the table; a row that outgrows it is gathered by the whole wave (round 5,
        spmm_ell.hip: ell_long_row): the same terms added in another order"""
        o, b = out.float().cpu().numpy(), base.float().cpu().numpy()
        fits = deg <= width
        assert np.array_equal(o[fits], b[fits])
        tol = 1e-2 if dtype == torch.bfloat16 else 2e-6
        assert np.abs(o[~fits] - b[~fits]).max() <= tol * np.abs(b).max()

    for scaled in (False, True):
        sc = t(norm, dev) if scaled else None
        base = ops.spmm_raw(dip, dix, Hd, n, sc, sc)
        for rpg in (1, 2):
            _lib.call("gae_tuning_set", b"spmm_rpg", rpg)
            try:
                same(ops.spmm_raw(dip, dix, Hd, n, sc, sc, plan=table), base, W)
                same(ops.spmm_raw(dip, dix, Hd, n, sc, sc, plan=narrow), base, 4)
                for ell_kernels in (2, 1):        # 2 = row-group kernel reads the table, 1 = spmm_ell.hip kernels
                    _lib.call("gae_tuning_set", b"spmm_ell", ell_kernels)
                    _lib.call("gae_tuning_set", b"spmm_ell_rpg", rpg)
                    out = ops.spmm_raw(dip, dix, Hd, n, sc, sc, plan=table)
                    if ell_kernels == 2:
                        assert torch.equal(out, base)             # (CSR order throughout)
                    else:
                        same(out, base, W)
                _lib.call("gae_tuning_set", b"spmm_ell_rpg", 0)
                heavy = ops.spmm_raw(dip, dix, Hd, n, sc, sc, plan=ops.spmm_plan(dip, threshold=8, segment=64))
                assert torch.equal(ops.spmm_raw(dip, dix, Hd, n, sc, sc, plan=both), heavy)
            finally:
                _lib.call("gae_tuning_set", b"spmm_rpg", 0)
                _lib.call("gae_tuning_set", b"spmm_ell", 1)
    if dtype == torch.float32:
        from oracle import c_oracle
        same(ops.spmm_raw(dip, dix, Hd, n, plan=table), torch.from_numpy(c_oracle.spmm_csr(ip, ix, H)), W)


@pytest.mark.parametrize("F", [100, 500, 1433, 3703])
@pytest.mark.parametrize("n", [300, 5000])
def test_spmm_feature_tiles_bit_identical(F, n, dev):
    """GAE_SPMM_TILE: XCD-owned feature tiles on rows made of whole 128-byte lines == the plain row-group launch"""
    from gae_dgl_amd import ops, _lib
    rng = np.random.default_rng(F + n)
    src, dst = rand_graph(rng, n, 5 * n, hub=True)
    H = rng.standard_normal((n, F)).astype(np.float32)
    ip, ix = O().csr_from_coo(src, dst, n)
    dip, dix = t(ip, dev), t(ix, dev)
    Hp = ops.pad_rows(t(H, dev))
    assert Hp.stride(0) % 32 == 0 if F * 4 >= 512 else Hp.stride(0) % 4 == 0
    base = ops.spmm_raw(dip, dix, Hp, n)
    from oracle import c_oracle
    assert np.array_equal(base.cpu().numpy(), c_oracle.spmm_csr(ip, ix, H))
    plan = ops.spmm_plan(dip, indices=dix, threshold=ops.SKEW_THRESHOLD)   # hub row -> skew plan (segment sums: own order) + packed table
    assert plan.n_heavy > 0 and plan.ell is not None
    planned = ops.spmm_raw(dip, dix, Hp, n, plan=plan)
    assert rel_err(planned, base.double().cpu()) < TOL
    assert torch.equal(ops.spmm_raw(dip, dix, Hp, n, scattered=True), base)
    assert torch.equal(ops.spmm_raw(dip, dix, Hp, n, scattered=True, plan=plan), planned)
    norm = t(O().norm_from_in_degrees(O().in_degrees(dst, n)).numpy(), dev)
    assert torch.equal(ops.spmm_raw(dip, dix, Hp, n, norm, norm, scattered=True),
                       ops.spmm_raw(dip, dix, Hp, n, norm, norm))
    for tv in (8, 16, 40, 64, 136):          # forced tile widths, also on rows that are NOT whole lines
        _lib.call("gae_tuning_set", b"spmm_tile_vecs", tv)
        try:
            assert torch.equal(ops.spmm_raw(dip, dix, Hp, n, plan=plan), planned)
            Hu = t(H, dev)                                           # ld = F: unaligned rows, scalar or vector path
            assert torch.equal(ops.spmm_raw(dip, dix, Hu, n), base)
        finally:
            _lib.call("gae_tuning_set", b"spmm_tile_vecs", 0)


def test_gather_scattered_statistic(dev):
    from gae_dgl_amd import ops, workloads as W
    import gae_dgl_amd as G
    rng = np.random.default_rng(0)
    n = 20000
    dst = rng.integers(0, n, 5 * n); src = rng.integers(0, n, 5 * n)
    ip, ix = ops.csr_from_coo(t(dst, dev), t(src, dev), n, n)
    d = ops.gather_distance(ip, ix)
    assert d == int(np.median(np.abs(src - dst)))
    assert ops.gather_scattered(ip, ix, 2000) and not ops.gather_scattered(ip, ix, 128)
    near = np.clip(dst + rng.integers(-32, 33, dst.size), 0, n - 1)
    assert not ops.gather_scattered(*ops.csr_from_coo(t(dst, dev), t(near, dev), n, n), 2000)
    gp, s2, d2, _ = W.zinc_like(2000, seed=1)
    assert not ops.gather_scattered(*ops.csr_from_coo(t(d2, dev), t(s2, dev), int(gp[-1]), int(gp[-1])), 2000)
    assert ops.gather_distance(*ops.csr_from_coo(t(dst[:0], dev), t(src[:0], dev), 10, 10)) == 0
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    assert g.scattered(2000) and not g.scattered(64)


@pytest.mark.parametrize("F,ld", [(32, 32), (39, 40), (16, 16), (8, 8), (130, 132)])
@pytest.mark.parametrize("gpb", [1, 4, 16])
def test_spmm_blockdiag_bit_identical(F, ld, gpb, dev):
    """LDS-staged block-diagonal kernel == row-group kernel, bit for bit (same CSR-order sums)"""
    from gae_dgl_amd import ops, workloads as W
    gp, src, dst, _ = W.zinc_like(500, seed=F + gpb)
    n = int(gp[-1])
    ip, ix = ops.csr_from_coo(t(dst, dev), t(src, dev), n, n)
    gen = torch.Generator(device=dev).manual_seed(1)
    H = torch.randn(n, ld, device=dev, generator=gen)[:, :F]
    norm = ops.degree_norm(ip)[1]
    bd = ops.BlockDiag(gp, dev, graphs_per_block=gpb)
    bd.min_blocks = 0                                        # force the LDS-staged kernel at this small size
    for sc in (None, norm):
        ref = ops.spmm_raw(ip, ix, H, n, sc, sc)
        out = ops.spmm_raw(ip, ix, H, n, sc, sc, blockdiag=bd)
        assert torch.equal(out, ref)
    assert ops.BlockDiag(gp, dev).cuts(F)[1] > 0           # width-dependent cuts
    # index slice smaller than a block's edge count: overflow edges come from global memory
    bd.max_edges = 16
    assert torch.equal(ops.spmm_raw(ip, ix, H, n, blockdiag=bd), ops.spmm_raw(ip, ix, H, n))
    # oracle
    assert rel_err(ops.spmm_raw(ip, ix, H, n, blockdiag=bd), O().spmm_csr(ip.cpu().numpy(), ix.cpu().numpy(), H.cpu())) < TOL


def test_spmm_padded_ld_and_views(dev):
    from gae_dgl_amd import ops
    rng = np.random.default_rng(1)
    n, e, F = 300, 2000, 39
    src, dst = rand_graph(rng, n, e)
    H = rng.standard_normal((n, F)).astype(np.float32)
    ip, ix = O().csr_from_coo(src, dst, n)
    ref = O().spmm_csr(ip, ix, H)
    Hp = torch.full((n, 40), float("nan"), device=dev)
    Hp[:, :F] = t(H, dev)
    outp = torch.full((n, 40), 7.0, device=dev)
    ops.spmm_raw(t(ip, dev), t(ix, dev), Hp[:, :F], n, out=outp[:, :F])   # vector path, ld = 40
    assert rel_err(outp[:, :F], ref) < TOL
    assert float((outp[:, F:] - 7.0).abs().max()) == 0.0                  # padding untouched (no STORE_PAD flag)
    ops.spmm_raw(t(ip, dev), t(ix, dev), Hp[:, :F], n, out=outp[:, :F], out_padded=True)   # caller allows the pad
    assert rel_err(outp[:, :F], ref) < TOL
    out = ops.spmm_raw(t(ip, dev), t(ix, dev), t(H, dev), n)              # scalar path, ld = 39
    assert rel_err(out, ref) < TOL


def test_spmm_bf16(dev):
    from gae_dgl_amd import ops
    rng = np.random.default_rng(2)
    n, e = 500, 4000
    src, dst = rand_graph(rng, n, e)
    ip, ix = O().csr_from_coo(src, dst, n)
    for F in (16, 39, 128, 3703):
        H = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).bfloat16()
        ref = O().spmm_csr(ip, ix, H.float())
        out = ops.spmm_raw(t(ip, dev), t(ix, dev), H.to(dev), n)
        assert out.dtype == torch.bfloat16
        assert rel_err(out.float(), ref) < 1e-2  # bf16 storage: 8 mantissa bits


def test_spmm_rectangular_and_empty(dev):
    from gae_dgl_amd import ops
    rng = np.random.default_rng(3)
    row = rng.integers(0, 50, 400); col = rng.integers(0, 300, 400)
    ip, ix = O().csr_from_coo(col, row, 50, 300)
    H = rng.standard_normal((300, 32)).astype(np.float32)
    assert rel_err(ops.spmm_raw(t(ip, dev), t(ix, dev), t(H, dev), 50), O().spmm_csr(ip, ix, H)) < TOL
    # graph without edges: all-zero aggregate
    ip0 = torch.zeros(11, dtype=torch.int32, device=dev); ix0 = torch.zeros(0, dtype=torch.int32, device=dev)
    out = ops.spmm_raw(ip0, ix0, torch.randn(10, 32, device=dev), 10)
    assert float(out.abs().max()) == 0.0
    # zero rows / zero features
    assert ops.spmm_raw(torch.zeros(1, dtype=torch.int32, device=dev), ix0, torch.randn(0, 8, device=dev), 0).shape == (0, 8)


def test_spmm_properties_large(dev):
    """size-independent properties at a size the oracle would not finish quickly:
    A 1 = in-degree, linearity, transpose identity <A x, y> = <x, A^T y>."""
    from gae_dgl_amd import ops
    n, e, F = 1 << 20, 1 << 24, 32
    gen = torch.Generator(device=dev).manual_seed(0)
    src = torch.randint(0, n, (e,), device=dev, generator=gen)
    dst = (torch.randint(0, n, (e,), device=dev, generator=gen) ** 2 // n)  # skewed degrees
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    tp, tx = ops.csr_from_coo(src, dst, n, n)
    assert int(ip[-1]) == e and int(tp[-1]) == e
    assert bool((ip[1:] >= ip[:-1]).all())
    deg, _ = ops.degree_norm(ip)
    ones = torch.ones(n, F, device=dev)
    out = ops.spmm_raw(ip, ix, ones, n)
    assert torch.equal(out[:, 0], deg.float()) and torch.equal(out[:, F - 1], deg.float())
    x = torch.randn(n, F, device=dev, generator=gen); y = torch.randn(n, F, device=dev, generator=gen)
    plan, tplan = ops.spmm_plan(ip), ops.spmm_plan(tp)
    assert plan is not None                                    # skewed in-degrees
    assert torch.equal(ops.spmm_raw(ip, ix, ones, n, plan=plan)[:, 3], deg.float())
    ax, ay = ops.spmm_raw(ip, ix, x, n, plan=plan), ops.spmm_raw(ip, ix, y, n, plan=plan)
    axy = ops.spmm_raw(ip, ix, 2 * x + y, n, plan=plan)
    assert rel_err(axy, 2 * ax + ay) < TOL
    assert rel_err(ax, ops.spmm_raw(ip, ix, x, n)) < TOL       # plan vs CSR-order sums
    aty = ops.spmm_raw(tp, tx, y, n, plan=tplan)
    lhs = float((ax.double() * y.double()).sum()); rhs = float((x.double() * aty.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))


def test_spmm_baseline_full_sizes(dev):
    """BASELINE.json sizes through size-independent properties: the ZINC-250k whole-set launch (block-diagonal
    LDS kernel) and the Pubmed layer-1 launch (XCD feature tiles + packed table) are bit-identical to the plain
    row-group launch, A 1 = in-degree, and no edge leaves its molecule"""
    from gae_dgl_amd import ops, workloads as W
    gptr, src, dst, _ = W.zinc_like(249455, seed=0)
    n = int(gptr[-1])
    ip, ix = ops.csr_from_coo(t(dst, dev), t(src, dev), n, n)
    assert int(ip[-1]) == len(src)
    deg, _ = ops.degree_norm(ip)
    gen = torch.Generator(device=dev).manual_seed(1)
    bd = ops.BlockDiag(gptr, dev)
    for F in (39, 32):
        H = ops.pad_rows(torch.randn(n, F, device=dev, generator=gen))
        plain = ops.spmm_raw(ip, ix, H, n)
        assert torch.equal(ops.spmm_raw(ip, ix, H, n, blockdiag=bd), plain)
        ones = ops.pad_rows(torch.ones(n, F, device=dev))
        assert torch.equal(ops.spmm_raw(ip, ix, ones, n, blockdiag=bd)[:, F - 1], deg.float())
    # block-diagonal: the molecule of every column id is the molecule of its row
    mol = torch.repeat_interleave(torch.arange(len(gptr) - 1, device=dev),
                                  torch.as_tensor(np.diff(gptr), device=dev))
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg.long())
    assert torch.equal(mol[ix.long()], mol[rows])
    del H, plain, ones, mol, rows
    n, src, dst, X = W.citation_graph("pubmed", seed=0)
    ip, ix = ops.csr_from_coo(t(dst, dev), t(src, dev), n, n)
    Xd = ops.pad_rows(t(X, dev))
    assert Xd.stride(0) == 544 and ops.gather_scattered(ip, ix, 2000)      # 17 lines per row (odd: ops.padded_ld)
    plan = ops.spmm_plan(ip, indices=ix)
    plain = ops.spmm_raw(ip, ix, Xd, n)
    assert torch.equal(ops.spmm_raw(ip, ix, Xd, n, plan=plan, scattered=True), plain)
    deg, _ = ops.degree_norm(ip)
    ones = ops.pad_rows(torch.ones(n, 500, device=dev))
    assert torch.equal(ops.spmm_raw(ip, ix, ones, n, plan=plan, scattered=True)[:, 499], deg.float())


@pytest.mark.parametrize("seed", range(40))
def test_spmm_dispatch_fuzz(seed, dev):
    """random (graph, width, leading dimension, dtype, scales, plan / packed table / feature tiles / rows-per-group /
    store-pad) combinations: every dispatch branch of gae_spmm_csr gives the CSR-order sums -- bit-identical to the
    C oracle for un-normalised fp32, bit-identical across the variants otherwise (rows that outgrow a packed table: to
    rounding, the table kernels gather them with the whole wave)"""
    from gae_dgl_amd import ops, _lib
    from oracle import c_oracle
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 7, 64, 333, 1500, 6000]))
    e = int(rng.integers(0, 8 * n + 1))
    F = int(rng.choice([1, 2, 5, 8, 16, 31, 32, 33, 39, 64, 65, 100, 129, 256, 500, 777, 1433]))
    hub = bool(rng.integers(0, 2)) and n > 2
    src, dst = rand_graph(rng, n, e, hub=hub) if e else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    dtype = torch.bfloat16 if rng.integers(0, 4) == 0 else torch.float32
    scaled = bool(rng.integers(0, 3) == 0)
    ip, ix = O().csr_from_coo(src, dst, n)
    dip, dix = t(ip, dev), t(ix, dev)
    H32 = rng.standard_normal((n, F)).astype(np.float32)
    pad_mode = int(rng.integers(0, 3))                 # 0: contiguous, 1: pad_rows, 2: odd leading dimension
    Hd = t(H32, dev).to(dtype)
    if pad_mode == 1:
        Hd = ops.pad_rows(Hd)
    elif pad_mode == 2:
        Hd = torch.cat([Hd, torch.full((n, 3), 7.0, device=dev, dtype=dtype)], dim=1)[:, :F]
    norm = t(O().norm_from_in_degrees(O().in_degrees(dst, n)).numpy(), dev) if scaled else None
    base = ops.spmm_raw(dip, dix, Hd, n, norm, norm)
    if dtype == torch.float32 and not scaled:
        assert np.array_equal(base.cpu().numpy(), c_oracle.spmm_csr(ip, ix, H32))
    else:
        ref = O().spmm_csr(ip, ix, Hd.float().cpu().double(), None if norm is None else norm.cpu().numpy(),
                           None if norm is None else norm.cpu().numpy())
        assert rel_err(base.float(), ref) < (TOL if dtype == torch.float32 else 1e-2)
    table = ops.spmm_plan(dip, indices=dix, ell=True, threshold=10 ** 6,
                          ell_width=[None, 4, 8, 16][int(rng.integers(0, 4))]) if n else None
    heavy = ops.spmm_plan(dip, threshold=int(rng.choice([1, 4, 8])), segment=64)
    for trial in range(4):
        rpg = int(rng.integers(0, 3)); tv = int(rng.choice([0, 0, 8, 16, 40]))
        use_table = bool(rng.integers(0, 2)) and table is not None
        scattered = bool(rng.integers(0, 2))
        out = torch.full((n, Hd.stride(0) if n > 1 else max(F, 1)), 3.0, device=dev, dtype=dtype)[:, :F] \
            if rng.integers(0, 2) else None
        _lib.call("gae_tuning_set", b"spmm_rpg", rpg); _lib.call("gae_tuning_set", b"spmm_tile_vecs", tv)
        _lib.call("gae_tuning_set", b"spmm_ell_rpg", rpg); _lib.call("gae_tuning_set", b"spmm_ell", int(rng.integers(1, 3)))
        try:
            got = ops.spmm_raw(dip, dix, Hd, n, norm, norm, out=out, plan=table if use_table else None,
                               scattered=scattered, out_padded=out is not None and pad_mode == 1)
        finally:
            _lib.call("gae_tuning_set", b"spmm_rpg", 0); _lib.call("gae_tuning_set", b"spmm_tile_vecs", 0)
            _lib.call("gae_tuning_set", b"spmm_ell_rpg", 0); _lib.call("gae_tuning_set", b"spmm_ell", 1)
        what = (n, e, F, dtype, scaled, pad_mode, rpg, tv, use_table, scattered)
        if use_table:      # rows beyond the table: gathered by the whole wave in the table kernels, same terms, other order
            fits = t(np.diff(ip) <= table.ell_width, dev)
            assert torch.equal(got[fits], base[fits]), what
            assert float((got.float() - base.float()).abs().max()) <= (2e-6 if dtype == torch.float32 else 1e-2) * \
                max(float(base.float().abs().max()), 1e-30), what
        else:
            assert torch.equal(got, base), what
    if heavy is not None and F > 12:                  # segment sums have their own (fixed) order: tolerance, stable
        a = ops.spmm_raw(dip, dix, Hd, n, norm, norm, plan=heavy)
        assert rel_err(a.float(), base.float().double().cpu()) < (TOL if dtype == torch.float32 else 2e-2)
        assert torch.equal(a, ops.spmm_raw(dip, dix, Hd, n, norm, norm, plan=heavy, scattered=True))


# ----------------------------------------------------------------- K3-K5 linear
@pytest.mark.parametrize("n,fin,fout", [(1, 1, 1), (5, 7, 3), (200, 39, 32), (333, 32, 16), (1000, 500, 32),
                                         (513, 1433, 32), (129, 100, 200), (4099, 16, 256)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_fwd_bwd(n, fin, fout, act, dev):
    from gae_dgl_amd import ops
    rng = np.random.default_rng(n + fin)
    M = rng.standard_normal((n, fin)).astype(np.float32)
    W = (rng.standard_normal((fout, fin)) / np.sqrt(fin)).astype(np.float32)
    b = rng.standard_normal(fout).astype(np.float32)
    dY = rng.standard_normal((n, fout)).astype(np.float32)
    Mt = torch.tensor(M, dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    Yref = Mt @ Wt.t() + bt
    if act:
        Yref = torch.relu(Yref)
    Yref.backward(torch.tensor(dY, dtype=torch.float64))
    Md = t(M, dev).requires_grad_(True); Wd = t(W, dev).requires_grad_(True); bd = t(b, dev).requires_grad_(True)
    Y = ops.linear(Md, Wd, bd, act)
    assert rel_err(Y, Yref) < TOL
    # compare gradients where the ReLU mask agrees (fp32 vs fp64 sign flips at ~0 are measure-zero)
    Y.backward(t(dY, dev))
    assert rel_err(Wd.grad, Wt.grad) < 5 * TOL
    assert rel_err(bd.grad, bt.grad) < 5 * TOL
    assert rel_err(Md.grad, Mt.grad) < 5 * TOL


@pytest.mark.parametrize("n,fin,fout,pad", [(1, 32, 1, False), (65, 39, 32, True), (200, 500, 32, True),
                                             (513, 1433, 32, True), (513, 1433, 17, False), (300, 3703, 32, True),
                                             (4099, 64, 16, False), (700, 2048, 32, True)])
def test_linear_fwd_weight_slices_in_lds(n, fin, fout, pad, dev):
    """linear_fwd_wlds_kernel (weight slices staged in LDS, 64-row blocks, split-K): forced on every shape it
    accepts, and on its default shape (few rows, f_in >= 2048), against fp64 and against gemm_stream_kernel"""
    from gae_dgl_amd import ops, _lib
    rng = np.random.default_rng(n + fin + fout)
    M = rng.standard_normal((n, fin)).astype(np.float32)
    W = (rng.standard_normal((fout, fin)) / np.sqrt(fin)).astype(np.float32)
    b = rng.standard_normal(fout).astype(np.float32)
    ref = torch.relu(torch.tensor(M, dtype=torch.float64) @ torch.tensor(W, dtype=torch.float64).t()
                     + torch.tensor(b, dtype=torch.float64))
    Md = ops.pad_rows(t(M, dev)) if pad else t(M, dev)
    out = {}
    _lib.call("gae_tuning_set", b"linear_bf16", 0)          # this test is about the fp32-MFMA kernels
    try:
        for mode in (2, 0, 1):
            _lib.call("gae_tuning_set", b"linear_wlds", mode)
            try:
                out[mode] = ops.linear_fwd_raw(Md, t(W, dev), t(b, dev), 1)
            finally:
                _lib.call("gae_tuning_set", b"linear_wlds", 1)
            assert rel_err(out[mode], ref) < TOL
        assert rel_err(out[2], out[0].double().cpu()) < TOL
        assert torch.equal(out[2], ops.linear_fwd_raw(Md, t(W, dev), t(b, dev), 1)) or fin < 2048   # default = forced
    finally:
        _lib.call("gae_tuning_set", b"linear_bf16", 0)
    # bf16 x 3 kernel (64-byte row pieces, split-K; opt-in), forced on every shape it accepts incl. unaligned rows of W
    _lib.call("gae_tuning_set", b"linear_bf16", 2)
    try:
        forced = ops.linear_fwd_raw(Md, t(W, dev), t(b, dev), 1)
    finally:
        _lib.call("gae_tuning_set", b"linear_bf16", 0)
    assert rel_err(forced, ref) < TOL


@pytest.mark.parametrize("seed", range(16))
def test_linear_fuzz(seed, dev):
    """random NodeApplyModule shapes (rows, widths, padded / odd leading dimensions, bias, activation, which
    gradients are wanted) against fp64: stream / tiled / split-K / LDS-slice forward, 8-wave dW kernel, dM"""
    from gae_dgl_amd import ops
    rng = np.random.default_rng(900 + seed)
    n = int(rng.choice([1, 31, 32, 33, 64, 65, 257, 1000, 5000]))
    fin = int(rng.choice([1, 3, 31, 32, 33, 39, 64, 100, 500, 1433, 2049]))
    fout = int(rng.choice([1, 7, 16, 32, 33, 64, 130]))
    act = int(rng.integers(0, 2)); bias = bool(rng.integers(0, 4))
    M = rng.standard_normal((n, fin)).astype(np.float32)
    W = (rng.standard_normal((fout, fin)) / np.sqrt(fin)).astype(np.float32)
    b = rng.standard_normal(fout).astype(np.float32) if bias else None
    dY = rng.standard_normal((n, fout)).astype(np.float32)
    Mt = torch.tensor(M, dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    Yref = Mt @ Wt.t()
    if bias:
        bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
        Yref = Yref + bt
    if act:
        Yref = torch.relu(Yref)
    Yref.backward(torch.tensor(dY, dtype=torch.float64))
    Md = t(M, dev)
    mode = int(rng.integers(0, 3))
    if mode == 1:
        Md = ops.pad_rows(Md)
    elif mode == 2:
        Md = torch.cat([Md, torch.full((n, 5), 9.0, device=dev)], dim=1)[:, :fin]
    Md = Md.requires_grad_(bool(rng.integers(0, 2)) or True)
    Wd = t(W, dev).requires_grad_(True)
    bd = t(b, dev).requires_grad_(True) if bias else None
    Y = ops.linear(Md, Wd, bd, act)
    assert rel_err(Y, Yref) < TOL, (n, fin, fout, act, bias, mode)
    Y.backward(t(dY, dev))
    assert rel_err(Wd.grad, Wt.grad) < 5 * TOL, (n, fin, fout, act, bias, mode)
    assert rel_err(Md.grad, Mt.grad) < 5 * TOL
    if bias:
        assert rel_err(bd.grad, bt.grad) < 5 * TOL


@pytest.mark.parametrize("n,fin,fout,mode", [(1, 32, 32, 0), (33, 32, 16, 0), (1000, 39, 32, 1), (777, 64, 64, 0),
                                             (4100, 16, 32, 2), (513, 7, 5, 2), (300000, 32, 32, 0),
                                             (262144 + 17, 32, 16, 0)])
def test_linear_short_rows_wave_per_tile(n, fin, fout, mode, dev, tuning):
    """gemm_rows_kernel (f_in <= 64: a wave owns whole 32-row tiles, the weights stay in registers): forced on
    every shape it accepts (knob gemm_rows = 2; padded, odd and aligned leading dimensions), on its default shapes
    (>= 2^18 rows), forward with bias + ReLU and the masked dM of the backward, against fp64 and against
    gemm_stream_kernel"""
    from gae_dgl_amd import ops
    rng = np.random.default_rng(n + fin + fout)
    M = rng.standard_normal((n, fin)).astype(np.float32)
    W = (rng.standard_normal((fout, fin)) / np.sqrt(fin)).astype(np.float32)
    b = rng.standard_normal(fout).astype(np.float32)
    dY = rng.standard_normal((n, fout)).astype(np.float32)
    Mt = torch.tensor(M, dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    pre = Mt @ Wt.t() + bt
    # millions of pre-activations: a few lie within fp32 rounding of 0, where the ReLU mask of an fp32 kernel may
    # differ from fp64's -- no gradient flows into those elements in this test
    dY[(pre.detach().abs() < 1e-4).numpy()] = 0.0
    Yref = torch.relu(pre)
    Yref.backward(torch.tensor(dY, dtype=torch.float64))
    Md = t(M, dev)
    if mode == 1:
        Md = ops.pad_rows(Md)
    elif mode == 2:
        Md = torch.cat([Md, torch.full((n, 5), 9.0, device=dev)], dim=1)[:, :fin]
    res = {}
    for knob in (2, 0, 1):
        tuning("gemm_rows", knob)
        Mg = Md.detach().requires_grad_(True)
        Wd = t(W, dev).requires_grad_(True); bd = t(b, dev).requires_grad_(True)
        Y = ops.linear(Mg, Wd, bd, 1)
        Y.backward(t(dY, dev))
        res[knob] = (Y.detach(), Mg.grad)
        assert rel_err(Y, Yref) < TOL and rel_err(Mg.grad, Mt.grad) < 5 * TOL
        assert rel_err(Wd.grad, Wt.grad) < 5 * TOL and rel_err(bd.grad, bt.grad) < 5 * TOL
    assert rel_err(res[2][0], res[0][0].double().cpu()) < TOL and rel_err(res[2][1], res[0][1].double().cpu()) < TOL
    assert torch.equal(res[1][0], res[2 if n >= (1 << 18) else 0][0])        # default: only very tall operands


def test_linear_odd_ld_and_no_bias(dev):
    from gae_dgl_amd import ops
    M = torch.randn(70, 45, device=dev)[:, :39]  # ld 45: scalar staging path
    W = torch.randn(32, 39, device=dev)
    Y = ops.linear(M, W, None, 1)
    assert rel_err(Y, torch.relu(M.double().cpu() @ W.double().cpu().t())) < TOL


# ----------------------------------------------------------------- K6/K7 decoder
@pytest.mark.parametrize("n,d", [(1, 1), (6, 3), (200, 16), (1000, 16), (515, 48), (300, 130)])
def test_decoder_dense_fwd_bwd(n, d, dev):
    from gae_dgl_amd import ops
    rng = np.random.default_rng(n)
    Z = rng.standard_normal((n, d)).astype(np.float32)
    mask = ((rng.random((n, d)) >= 0.1) / 0.9).astype(np.float32)
    G = rng.standard_normal((n, n)).astype(np.float32)  # non-symmetric upstream gradient
    for m in (None, mask):
        Zt = torch.tensor(Z, dtype=torch.float64, requires_grad=True)
        zz = Zt if m is None else Zt * torch.tensor(m, dtype=torch.float64)
        ref = zz @ zz.t()
        ref.backward(torch.tensor(G, dtype=torch.float64))
        Zd = t(Z, dev).requires_grad_(True)
        out = ops.decoder_dense(Zd, None if m is None else t(m, dev))
        assert rel_err(out, ref) < TOL
        assert rel_err(out, O().decoder_logits(Z, m)) < TOL
        out.backward(t(G, dev))
        assert rel_err(Zd.grad, Zt.grad) < 5 * TOL


def test_dropout_mask_statistics_and_reproducibility(dev):
    from gae_dgl_amd import ops
    m1 = ops.dropout_mask((100000, 16), 0.1, seed=7, offset=0, device=dev)
    m2 = ops.dropout_mask((100000, 16), 0.1, seed=7, offset=0, device=dev)
    m3 = ops.dropout_mask((100000, 16), 0.1, seed=8, offset=0, device=dev)
    assert torch.equal(m1, m2) and not torch.equal(m1, m3)
    vals = torch.unique(m1).cpu().numpy()
    assert len(vals) == 2 and vals[0] == 0.0 and np.isclose(vals[1], 1 / 0.9)
    keep = float((m1 > 0).float().mean())
    assert abs(keep - 0.9) < 2e-3
    # counter-based: a shifted offset continues the same stream
    a = ops.dropout_mask((4096,), 0.3, seed=1, offset=0, device=dev)
    b = ops.dropout_mask((2048,), 0.3, seed=1, offset=512, device=dev)
    assert torch.equal(a[2048:], b)
    assert float(ops.dropout_mask((1000,), 0.0, 1, 0, dev).min()) == 1.0


# ----------------------------------------------------------------- module-level parity (golden vectors)
def build_model(g, dev):
    import gae_dgl_amd as G
    hidden = [int(h) for h in g["hidden"]]
    model = G.GAE(g["X"].shape[1], hidden)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    model.load_state_dict(sd)  # reference checkpoint keys round-trip
    return model.to(dev)


def fresh_graph(g, dev):
    import gae_dgl_amd as G
    gr = G.DGLGraph()
    gr.add_nodes(int(g["n"]))
    gr.add_edges(g["src"], g["dst"])
    gr.to(dev)
    gr.ndata['h'] = t(g["X"], dev)
    return gr


def test_gae_encode_forward_side_effects(golden, dev):
    g = golden
    model = build_model(g, dev)
    gr = fresh_graph(g, dev)
    Z = model.encode(gr)
    assert rel_err(Z, g["Z"]) < TOL
    assert 'h' not in gr.ndata                      # A9: encode pops 'h'
    model.decoder.dropout = 0.0
    gr = fresh_graph(g, dev)
    logits = model(gr)
    assert rel_err(logits, g["logits_p0"]) < TOL
    assert rel_err(gr.ndata['h'], g["Z"]) < TOL     # A9: forward leaves Z in ndata['h']
    # injected reference mask (always-on dropout, also in eval mode: A8)
    model.decoder.dropout = 0.1
    model.decoder.mask = t(g["mask"], dev)
    model.eval()
    assert rel_err(model(fresh_graph(g, dev)), g["logits_p01"]) < TOL
    model.decoder.mask = None
    l1, l2 = model(fresh_graph(g, dev)), model(fresh_graph(g, dev))
    if g["Z"].size > 40:
        assert not torch.equal(l1, l2)              # dropout stays on in eval mode


def test_gae_loss_and_grads(golden, dev):
    import torch.nn.functional as F
    g = golden
    for tag, mask in (("p0", None), ("p01", g["mask"])):
        model = build_model(g, dev)
        model.decoder.dropout = 0.0 if mask is None else 0.1
        model.decoder.mask = None if mask is None else t(mask, dev)
        gr = fresh_graph(g, dev)
        adj = gr.adjacency_matrix().to_dense()
        assert np.array_equal(adj.cpu().numpy(), g["adj"])
        assert torch.equal(adj, gr.dense_adjacency())
        pw = (adj.shape[0] * adj.shape[0] - adj.sum()) / adj.sum()
        assert rel_err(pw, g["pos_weight"]) < 1e-6
        loss = F.binary_cross_entropy_with_logits(model(gr), adj, pos_weight=pw)
        assert rel_err(loss, g["loss_" + tag]) < TOL
        loss.backward()
        for k, p in model.named_parameters():
            assert rel_err(p.grad, g[f"grad_{tag}/{k}"]) < 5 * TOL, k


def test_gae_three_adam_steps(golden, dev):
    import torch.nn.functional as F
    g = golden
    model = build_model(g, dev)
    model.decoder.dropout = 0.0
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    adj = t(g["adj"], dev); pw = t(g["pos_weight"], dev)
    losses = []
    for _ in range(3):
        loss = F.binary_cross_entropy_with_logits(model(fresh_graph(g, dev)), adj, pos_weight=pw)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g["adam3_losses"], rtol=5e-5)
    for k, v in model.state_dict().items():
        assert rel_err(v, g["sd_after3/" + k]) < 1e-4, k


@pytest.mark.parametrize("n,captured", [(1500, False), (8300, True)])
def test_training_trajectory_matches_cpu_reference_step(n, captured, dev):
    """12 full training steps (encoder, fused loss -- full-square below 5120 rows, symmetric above --, backward,
    one-launch Adam; eagerly or replayed from the captured HIP graph) against oracle.CpuReferenceStep, the
    reference step in plain PyTorch CPU with torch.optim.Adam: same loss trajectory, same final weights"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    rng = np.random.default_rng(n)
    src, dst = rand_graph(rng, n, 5 * n, hub=False)
    X = rng.standard_normal((n, 70)).astype(np.float32)
    ref = O().CpuReferenceStep(src, dst, n, X, 70, [32, 16], lr=1e-2, seed=0, dropout=0.0)
    model = G.GAE(70, [32, 16]).to(dev)
    model.decoder.dropout = 0.0
    sd = {f"layers.{i}.apply_mod.linear.{k}": getattr(l, k).detach().clone() for i, l in enumerate(ref.layers)
          for k in ("weight", "bias")}
    model.load_state_dict(sd)
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Xd = ops.pad_rows(t(X, dev))
    opt = Adam(model.parameters(), lr=1e-2)
    steps = 12
    want = [ref.step() for _ in range(steps)]

    def eager_step():       # no reference to the loss survives the call (see CapturedTrainStep's docstring)
        g.ndata['h'] = Xd
        loss = model.reconstruction_loss(g)
        opt.zero_grad(); ops.backward(loss); opt.step()
        return float(loss.detach())

    got = []
    step = None
    for k in range(steps):
        if captured and k == 1:
            step = CapturedTrainStep(model, opt, g, Xd, warmup=0)
        got.append(float(step()) if step is not None else eager_step())
    np.testing.assert_allclose(got, want, rtol=2e-4)
    # Adam normalises every gradient entry by its own running magnitude, so entries with a tiny gradient amplify
    # the last-bit differences of the kernels: the weights agree to a few 1e-3 of their scale after 12 steps
    for i, l in enumerate(ref.layers):
        assert rel_err(model.layers[i].apply_mod.linear.weight, l.weight.detach()) < 5e-3
        assert rel_err(model.layers[i].apply_mod.linear.bias, l.bias.detach()) < 5e-3


def test_norm_both_matches_oracle(dev):
    import gae_dgl_amd as G
    g = load_golden("sym200")
    Ws, bs = golden_params(g)
    n = int(g["n"])
    ip, ix = O().csr_from_coo(g["src"], g["dst"], n)
    norm = g["norm"].ravel()
    ref = O().gae_encode(ip, ix, g["X"], Ws, bs, norm)
    model = G.GAE(39, [32, 16], norm="both")
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")})
    model.to(dev)
    gr = fresh_graph(g, dev)
    assert rel_err(model.encode(gr), ref) < TOL
    assert np.array_equal(gr.in_degrees().cpu().numpy(), g["in_degrees"])


def test_batch_on_device_matches_golden(dev):
    import gae_dgl_amd as G
    parts = load_golden("mol8_parts"); whole = load_golden("mol8")
    gs = []
    for i in range(int(parts["n_graphs"])):
        gr = G.DGLGraph()
        gr.add_nodes(int(parts[f"g{i}/n"])); gr.add_edges(parts[f"g{i}/src"], parts[f"g{i}/dst"])
        gr.ndata['h'] = torch.from_numpy(parts[f"g{i}/X"])
        gs.append(gr.to(dev))
    bg = G.batch(gs)
    model = build_model(whole, dev)
    assert rel_err(model.encode(bg), whole["Z"]) < TOL


# ----------------------------------------------------------------- K7+K8+K9 fused decoder + BCE
def test_fused_loss_matches_golden(golden, dev):
    g = golden
    for tag, mask in (("p0", None), ("p01", g["mask"])):
        model = build_model(g, dev)
        model.decoder.dropout = 0.0 if mask is None else 0.1
        model.decoder.mask = None if mask is None else t(mask, dev)
        gr = fresh_graph(g, dev)
        loss = model.reconstruction_loss(gr)
        assert rel_err(loss, g["loss_" + tag]) < TOL
        assert rel_err(gr.ndata['h'], g["Z"]) < TOL
        loss.backward()
        for k, p in model.named_parameters():
            assert rel_err(p.grad, g[f"grad_{tag}/{k}"]) < 5 * TOL, k


@pytest.mark.parametrize("n,d,e", [(1, 1, 1), (17, 3, 40), (130, 16, 900), (700, 16, 3000), (257, 32, 2000),
                                   (300, 48, 1000), (129, 64, 77), (5000, 16, 30000)])
def test_fused_loss_vs_oracle_random(n, d, e, dev):
    """directed multigraphs (duplicates, self loops), tails in every tile dimension"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(n * 7 + d)
    src, dst = rand_graph(rng, n, e, hub=n > 2)
    Z = (rng.standard_normal((n, d)) * 0.7).astype(np.float32)
    mask = ((rng.random((n, d)) >= 0.1) / 0.9).astype(np.float32)
    adj = O().dense_adjacency(src, dst, n, dtype=torch.float64)
    pw = O().pos_weight_of(adj)
    Zt = torch.tensor(Z, dtype=torch.float64, requires_grad=True)
    ref = O().bce_with_logits_mean(O().decoder_logits(Zt, torch.tensor(mask, dtype=torch.float64)), adj, pw)
    ref.backward()
    gr = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Zd = t(Z, dev).requires_grad_(True)
    loss = ops.decoder_bce(Zd, t(mask, dev), gr)
    assert rel_err(loss, ref) < TOL
    (3.0 * loss).backward()
    assert rel_err(Zd.grad, 3.0 * Zt.grad) < 5 * TOL
    # loss-only mode (validation, train=False): same value, no gradient work
    with torch.no_grad():
        assert rel_err(ops.decoder_bce(t(Z, dev), t(mask, dev), gr), ref) < TOL


@pytest.mark.parametrize("n,d", [(512, 16), (513, 16), (700, 7), (1000, 3), (1025, 16), (4096, 16), (5000, 12),
                                 (8193, 16)])
@pytest.mark.parametrize("bal", [2, 0])
def test_fused_loss_symmetric_kernel(n, d, bal, dev):
    """symmetric dense kernel (tiles right of the block diagonal evaluated once, mirror product, strip reduction)
    == the full-square kernel == the oracle; panels / tiles / chunks with tails in every dimension"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, _lib
    rng = np.random.default_rng(n + d)
    src, dst = rand_graph(rng, n, 5 * n, hub=True)
    Z = (rng.standard_normal((n, d)) * 0.7).astype(np.float32)
    mask = ((rng.random((n, d)) >= 0.1) / 0.9).astype(np.float32)
    gr = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    out = {}
    for sym in (2, 0):                       # 2 = symmetric kernel from 512 rows on (default: from 5120)
        _lib.call("gae_tuning_set", b"bce_sym", sym)
        _lib.call("gae_tuning_set", b"bce_sym_ri", 4 if n % 2 else 2)      # both panel heights across the cases
        _lib.call("gae_tuning_set", b"bce_sym_bal", bal)                    # balanced schedule (round 6) / the 2-D grid
        try:
            Zd = t(Z, dev).requires_grad_(True)
            loss = ops.decoder_bce(Zd, t(mask, dev), gr)
            loss.backward()
            with torch.no_grad():
                lo = ops.decoder_bce(t(Z, dev), t(mask, dev), gr)          # loss-only mode
            again = ops.decoder_bce(t(Z, dev).requires_grad_(True), t(mask, dev), gr)
            out[sym] = (loss.detach(), Zd.grad.clone(), lo, again.detach())
        finally:
            _lib.call("gae_tuning_set", b"bce_sym", 1)
            _lib.call("gae_tuning_set", b"bce_sym_ri", 0)
            _lib.call("gae_tuning_set", b"bce_sym_bal", 1)
    assert rel_err(out[2][0], out[0][0].double().cpu()) < 2e-6
    assert rel_err(out[2][1], out[0][1].double().cpu()) < TOL
    assert rel_err(out[2][2], out[0][0].double().cpu()) < 2e-6
    assert torch.equal(out[2][0], out[2][3])                               # deterministic
    if n <= 1100:
        adj = O().dense_adjacency(src, dst, n, dtype=torch.float64)
        Zt = torch.tensor(Z, dtype=torch.float64, requires_grad=True)
        ref = O().bce_with_logits_mean(O().decoder_logits(Zt, torch.tensor(mask, dtype=torch.float64)), adj,
                                       O().pos_weight_of(adj))
        ref.backward()
        assert rel_err(out[2][0], ref) < TOL and rel_err(out[2][1], Zt.grad) < 5 * TOL


@pytest.mark.parametrize("seed", range(12))
def test_fused_loss_fuzz(seed, dev):
    """random sizes / widths / multigraphs / kernel choices of the fused loss against the fp64 oracle"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, _lib
    rng = np.random.default_rng(500 + seed)
    n = int(rng.choice([2, 9, 63, 64, 65, 127, 129, 511, 640, 1100, 1537]))
    d = int(rng.choice([1, 2, 7, 15, 16, 17, 32, 33, 48, 64]))
    e = int(rng.integers(1, 6 * n + 2))
    src, dst = rand_graph(rng, n, e, hub=bool(rng.integers(0, 2)) and n > 2)
    if rng.integers(0, 2):
        src[: min(e, 5)] = dst[: min(e, 5)]                       # self loops
    Z = (rng.standard_normal((n, d)) * rng.choice([0.1, 0.7, 3.0])).astype(np.float32)
    mask = ((rng.random((n, d)) >= 0.1) / 0.9).astype(np.float32) if rng.integers(0, 3) else None
    adj = O().dense_adjacency(src, dst, n, dtype=torch.float64)
    Zt = torch.tensor(Z, dtype=torch.float64, requires_grad=True)
    mk = None if mask is None else torch.tensor(mask, dtype=torch.float64)
    ref = O().bce_with_logits_mean(O().decoder_logits(Zt, mk), adj, O().pos_weight_of(adj))
    ref.backward()
    gr = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    sym = int(rng.choice([0, 1, 2])); sri = int(rng.choice([0, 2, 4])); bal = int(rng.choice([0, 1, 2]))
    _lib.call("gae_tuning_set", b"bce_sym", sym); _lib.call("gae_tuning_set", b"bce_sym_ri", sri)
    _lib.call("gae_tuning_set", b"bce_sym_bal", bal)
    try:
        Zd = t(Z, dev).requires_grad_(True)
        loss = ops.decoder_bce(Zd, None if mask is None else t(mask, dev), gr)
        loss.backward()
        with torch.no_grad():
            loss_only = ops.decoder_bce(t(Z, dev), None if mask is None else t(mask, dev), gr)
    finally:
        _lib.call("gae_tuning_set", b"bce_sym", 1); _lib.call("gae_tuning_set", b"bce_sym_ri", 0)
        _lib.call("gae_tuning_set", b"bce_sym_bal", 1)
    assert rel_err(loss, ref) < TOL and rel_err(loss_only, ref) < TOL, (n, d, e, sym, sri, bal)
    assert rel_err(Zd.grad, Zt.grad) < 5 * TOL, (n, d, e, sym, sri)


def test_fused_loss_equals_dense_path_large(dev):
    """Pubmed-sized: fused loss/grad == dense HIP decoder + torch BCE on the same Z"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    import torch.nn.functional as F
    n, src, dst, _ = W.citation_graph("pubmed", seed=1)
    gr = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    Z = (torch.randn(n, 16, device=dev, generator=gen) * 0.5)
    mask = ops.dropout_mask((n, 16), 0.1, seed=3, device=dev)
    Z1 = Z.clone().requires_grad_(True); Z2 = Z.clone().requires_grad_(True)
    adj = gr.dense_adjacency()
    pw = (n * n - adj.sum()) / adj.sum()
    ref = F.binary_cross_entropy_with_logits(ops.decoder_dense(Z1, mask), adj, pos_weight=pw)
    ref.backward()
    loss = ops.decoder_bce(Z2, mask, gr)
    loss.backward()
    assert rel_err(loss, ref) < TOL
    assert rel_err(Z2.grad, Z1.grad) < 5 * TOL
    l2 = ops.decoder_bce(Z.clone().requires_grad_(True), mask, gr)
    assert torch.equal(l2, loss.detach())  # deterministic reductions


@pytest.mark.parametrize("n,d", [(700, 16), (1300, 7), (257, 48)])
def test_fused_loss_draws_its_own_mask(n, d, dev):
    """dropout_p > 0: the mask of the draw is generated inside the fused launch -- same Philox stream as
    gae_dropout_mask, same loss / gradient bit for bit -- and the device draw counter advances by one per call"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(n)
    src, dst = rand_graph(rng, n, 6 * n)
    gr = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Z = t(rng.standard_normal((n, d)).astype(np.float32) * 0.5, dev)
    draws = torch.zeros(1, dtype=torch.int64, device=dev)
    masks = []
    for k in range(3):
        ref_mask = ops.dropout_mask((n, d), 0.1, seed=11, device=dev, draw_counter=torch.full_like(draws, k))
        Z1 = Z.clone().requires_grad_(True); Z2 = Z.clone().requires_grad_(True)
        ref = ops.decoder_bce(Z1, ref_mask, gr); ref.backward()
        mask = torch.full((n, d), -7.0, device=dev)
        loss = ops.decoder_bce(Z2, mask, gr, dropout=(0.1, 11, 0, draws)); loss.backward()
        assert int(draws) == k + 1
        assert torch.equal(mask, ref_mask)
        assert torch.equal(loss.detach(), ref.detach()) and torch.equal(Z2.grad, Z1.grad)
        masks.append(mask)
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])
    # module level: InnerProductDecoder.loss draws in-kernel, forward() with the separate mask kernel: same stream
    dec_a, dec_c = G.gae.InnerProductDecoder(seed=5), G.gae.InnerProductDecoder(seed=5)
    la = dec_a.loss(Z, gr)
    assert torch.equal(dec_a.last_mask, dec_c._draw_mask(Z)) and int(dec_a._draws) == 1 and int(dec_c._draws) == 1
    assert float(la) > 0


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_adam_matches_torch(wd, dev):
    """gae_adam_step == torch.optim.Adam (train_inductive.py:40) over a trajectory; 20 tensors -> two launches"""
    from gae_dgl_amd.optim import Adam
    gen = torch.Generator(device=dev).manual_seed(0)
    shapes = [(32, 500), (32,), (16, 32), (16,), (1,), (3, 1025)] + [(7, 5)] * 14
    pa = [torch.randn(s, device=dev, generator=gen).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = Adam(pa, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    for it in range(25):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, device=dev, generator=gen) * (1.0 + it)
            a.grad = gr.clone(); b.grad = gr.clone()
        oa.step(); ob.step()
    assert oa.steps_taken() == 25
    for a, b in zip(pa, pb):
        assert rel_err(a, b) < 2e-6
        assert torch.isfinite(a).all()
    # a tensor without a gradient is left alone (the step counter is per parameter group, not per tensor)
    before = pa[2].detach().clone()
    for a in pa:
        a.grad = torch.ones_like(a)
    pa[2].grad = None
    oa.step()
    assert torch.equal(pa[2].detach(), before) and not torch.equal(pa[0].detach(), pb[0].detach())


def test_adam_in_captured_step(dev):
    """device-side step counter: a replayed HIP graph keeps counting (bias correction follows the replays)"""
    import gae_dgl_amd as G
    from gae_dgl_amd import workloads as W, ops
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    n, src, dst, X = W.citation_graph("cora", seed=0)
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
    torch.manual_seed(0)
    ma = G.GAE(X.shape[1], [32, 16]).to(dev); ma.decoder.dropout = 0.0
    mb = G.GAE(X.shape[1], [32, 16]).to(dev); mb.load_state_dict(ma.state_dict()); mb.decoder.dropout = 0.0
    oa = Adam(ma.parameters(), lr=1e-2)
    ob = torch.optim.Adam(mb.parameters(), lr=1e-2)
    step = CapturedTrainStep(ma, oa, g, Xd, warmup=2)       # 2 eager warm-up steps, then replays
    la = [float(step()) for _ in range(6)]
    lb = []
    for _ in range(2 + 1 + 6):                              # warm-up + the captured (not executed... see below) + replays
        g.ndata['h'] = Xd
        l = mb.reconstruction_loss(g)
        ob.zero_grad(); l.backward(); ob.step()
        lb.append(float(l.detach()))
    # capture itself does not execute the step: the replays are steps 3..8 of the trajectory
    assert oa.steps_taken() == 2 + 6
    assert np.allclose(la, lb[2:8], rtol=2e-4, atol=0)


# ----------------------------------------------------------------- VGAE (BASELINE config 5)
def test_normal_noise_moments(dev):
    from gae_dgl_amd import ops
    e = ops.normal_noise((400000, 16), seed=3, device=dev)
    assert abs(float(e.mean())) < 3e-3 and abs(float(e.std()) - 1.0) < 3e-3
    assert abs(float((e ** 3).mean())) < 1e-2 and abs(float((e ** 4).mean()) - 3.0) < 3e-2
    assert torch.equal(e, ops.normal_noise((400000, 16), seed=3, device=dev))
    assert not torch.equal(e, ops.normal_noise((400000, 16), seed=4, device=dev))


@pytest.mark.parametrize("dtype,tol,n_small", [(torch.float32, 2e-5, 600), (torch.bfloat16, 2e-2, 600),
                                               (torch.float32, 2e-5, None), (torch.bfloat16, 2e-2, None)])
def test_vgae_matches_oracle(dtype, tol, n_small, dev):
    """Citeseer-shaped VGAE step (mu / logstd heads, sampled decoder, BCE + KL) vs the CPU restatement: a 600-node
    cut and the whole graph of BASELINE config 5 (3327 nodes, F = 3703; the oracle's dense N x N label is 44 MB)"""
    import gae_dgl_amd as G
    from gae_dgl_amd import workloads as W
    from gae_dgl_amd.vgae import VGAE
    n, src, dst, X = W.citation_graph("citeseer", seed=0)
    n_small = n if n_small is None else n_small            # oracle needs the dense N x N label
    keep = (src < n_small) & (dst < n_small)
    src, dst, X = src[keep], dst[keep], X[:n_small]
    torch.manual_seed(0)
    model = VGAE(X.shape[1], [32, 16], seed=11).to(dev)
    g = G.DGLGraph((src, dst), num_nodes=n_small).to(dev)
    Xd = torch.from_numpy(X).to(dev).to(dtype)
    g.ndata['h'] = Xd
    loss = model.loss(g)
    loss.backward()
    last = {k: v.detach().cpu() for k, v in model.last.items()}
    # oracle on the SAME (bf16-rounded) inputs and the same eps
    Xo = Xd.float().cpu()
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    ip, ix = O().csr_from_coo(src, dst, n_small)
    mu, ls, z = O().vgae_forward(ip, ix, Xo, P["shared.apply_mod.linear.weight"], P["shared.apply_mod.linear.bias"],
                                 P["mu_head.apply_mod.linear.weight"], P["mu_head.apply_mod.linear.bias"],
                                 P["logstd_head.apply_mod.linear.weight"], P["logstd_head.apply_mod.linear.bias"],
                                 last["eps"])
    adj = O().dense_adjacency(src, dst, n_small)
    rec = O().bce_with_logits_mean(z @ z.t(), adj, O().pos_weight_of(adj))
    kl = O().vgae_kl(mu, ls)
    (rec + kl).backward()
    assert rel_err(last["mu"], mu) < tol and rel_err(last["logstd"], ls) < tol and rel_err(last["z"], z) < tol
    assert rel_err(last["kl"], kl) < max(tol, 1e-5) and rel_err(last["rec"], rec) < max(tol, 1e-5)
    assert rel_err(loss, rec + kl) < max(tol, 1e-5)
    for k, p in model.named_parameters():
        assert rel_err(p.grad, P[k].grad) < 10 * tol, k
    # a few Adam steps reduce the loss
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    first = None
    for _ in range(20):
        g.ndata['h'] = Xd
        l = model.loss(g)
        opt.zero_grad(); l.backward(); opt.step()
        first = float(l.detach()) if first is None else first
    assert float(l.detach()) < first


def test_vgae_fused_heads_equal_two_layers(dev):
    """mu and log sigma heads as ONE fused launch (gae_x_gcn_layer_fused2 + the packed head kernels) == the two GCN layers
    they replace: forward values bit for bit (same aggregation order, same per-output product), every parameter gradient
    within the fp32 tolerance (one dW launch for both heads, another summation order for dH)"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, vgae as V, workloads as W
    n, src, dst, X = W.citation_graph("cora", seed=0)
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
    out = {}
    for fused in (True, False):
        V.FUSE_HEADS = fused
        try:
            torch.manual_seed(0)
            model = V.VGAE(X.shape[1], [32, 16], seed=5).to(dev)
            calls = []
            orig = ops.gcn_two_heads
            ops.gcn_two_heads = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            g.ndata['h'] = Xd
            loss = model.loss(g)
            ops.gcn_two_heads = orig
            assert len(calls) == (1 if fused else 0)
            loss.backward()
            out[fused] = (float(loss), {k: v.detach().clone() for k, v in model.last.items() if k in ("mu", "logstd", "z")},
                          {k: p.grad.detach().clone() for k, p in model.named_parameters()})
        finally:
            V.FUSE_HEADS = True
    assert out[True][0] == out[False][0]
    for k in ("mu", "logstd", "z"):
        assert torch.equal(out[True][1][k], out[False][1][k]), k
    for k in out[True][2]:
        assert rel_err(out[True][2][k], out[False][2][k]) < TOL, k


def test_vgae_captured_step_equals_eager_steps(dev):
    """the VGAE step (noise drawn from a device-side draw counter) as one captured HIP graph: the replayed losses and
    the trained parameters equal those of the same steps launched eagerly, bit for bit"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    from gae_dgl_amd.capture import CapturedTrainStep
    from gae_dgl_amd.optim import Adam
    from gae_dgl_amd.vgae import VGAE
    n, src, dst, X = W.citation_graph("cora", seed=0)
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
    runs = {}
    for mode in ("eager", "captured"):
        torch.manual_seed(0)
        model = VGAE(X.shape[1], [32, 16], seed=5).to(dev)
        opt = Adam(model.parameters(), lr=1e-2)
        params = list(model.parameters())
        losses = []

        def eager():
            g.ndata['h'] = Xd
            loss = model.loss(g)
            opt.zero_grad(set_to_none=True); ops.backward(loss, params); opt.step()
            model.last = {}
            return float(loss.detach())
        if mode == "eager":
            losses = [eager() for _ in range(8)]
        else:
            step = CapturedTrainStep(model, opt, g, Xd, loss_fn=lambda m, gg: m.loss(gg), warmup=2)   # 2 real steps
            losses = [None, None] + [float(step()) for _ in range(6)]
        runs[mode] = (losses, [p.detach().clone() for p in params])
    assert runs["eager"][0][2:] == runs["captured"][0][2:]
    assert runs["eager"][0][-1] < runs["eager"][0][0]
    for a, b in zip(runs["eager"][1], runs["captured"][1]):
        assert torch.equal(a, b)


# ----------------------------------------------------------------- graph-level readout (README.md:54)
def test_readout_golden_molecules(dev):
    """mean | sum | max per molecule of the golden 8-molecule batch == the oracle, through readout_nodes(batch)"""
    import gae_dgl_amd as G
    parts = load_golden("mol8_parts"); whole = load_golden("mol8")
    graphs = []
    for i in range(int(parts["n_graphs"])):
        g = G.DGLGraph((parts[f"g{i}/src"], parts[f"g{i}/dst"]), num_nodes=int(parts[f"g{i}/n"])).to(dev)
        g.ndata['h'] = t(parts[f"g{i}/X"], dev)
        graphs.append(g)
    bg = G.batch(graphs)
    gp = bg.graph_ptr().cpu().numpy()
    assert gp[-1] == int(whole["n"]) and len(gp) == len(graphs) + 1
    Z = t(whole["Z"], dev)
    out = G.readout_nodes(bg, Z)
    assert out.shape == (len(graphs), 3 * Z.shape[1])
    assert rel_err(out, O().segment_readout(whole["Z"], gp)) < TOL
    model = build_model(whole, dev)
    bg.ndata['h'] = t(whole["X"], dev)
    with torch.no_grad():
        model(bg)                                   # GAE.forward leaves Z in ndata['h'] (gae.py:53)
    assert rel_err(G.readout_nodes(bg), O().segment_readout(whole["Z"], gp)) < TOL


@pytest.mark.parametrize("d", [1, 3, 16, 48, 64, 100])
def test_readout_ragged(d, dev):
    """ragged segments incl. empty graphs, single-node graphs and one long graph; deterministic"""
    from gae_dgl_amd import ops
    rng = np.random.default_rng(d)
    sizes = np.concatenate([[0, 1, 1, 0, 700], rng.integers(0, 40, 300), [0]])
    gp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    Z = rng.standard_normal((int(gp[-1]), d)).astype(np.float32)
    Zd = ops.pad_rows(t(Z, dev))
    out = ops.segment_readout(Zd, t(gp, dev))
    ref = O().segment_readout(Z, gp)
    assert rel_err(out, ref) < TOL
    assert float(out[0].abs().max()) == 0.0 and float(out[-1].abs().max()) == 0.0      # empty graphs
    assert torch.equal(out[:, 2 * d:][1], Zd[0])                                        # max of a single node
    assert torch.equal(out, ops.segment_readout(Zd, t(gp, dev)))


# ----------------------------------------------------------------- ADVICE r01 items
@pytest.mark.parametrize("n,d", [(300, 128), (257, 65), (130, 256)])
def test_loss_wide_embedding_dense_chain(n, d, dev):
    """embedding widths above the fused kernel's 64 columns (optuna_gae.py:29-34 samples hidden dims up to 256) take
    the dense HIP chain (gae_decoder_dense -> gae_csr_to_dense -> gae_bce_logits -> gae_decoder_dense_bwd): loss and
    gradient vs the fp64 oracle, in-kernel dropout draws included, and a full model step with hidden dims 256 128"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(n + d)
    src, dst = rand_graph(rng, n, 5 * n, hub=True)
    Z = (rng.standard_normal((n, d)) * 0.3).astype(np.float32)
    mask = ((rng.random((n, d)) >= 0.1) / 0.9).astype(np.float32)
    adj = O().dense_adjacency(src, dst, n, dtype=torch.float64)
    Zt = torch.tensor(Z, dtype=torch.float64, requires_grad=True)
    ref = O().bce_with_logits_mean(O().decoder_logits(Zt, torch.tensor(mask, dtype=torch.float64)), adj,
                                   O().pos_weight_of(adj))
    ref.backward()
    gr = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    Zd = t(Z, dev).requires_grad_(True)
    loss = ops.decoder_bce(Zd, t(mask, dev), gr)
    assert rel_err(loss, ref) < TOL
    (2.0 * loss).backward()
    assert rel_err(Zd.grad, 2.0 * Zt.grad) < 5 * TOL
    with torch.no_grad():
        assert rel_err(ops.decoder_bce(t(Z, dev), t(mask, dev), gr), ref) < TOL
    # in-kernel dropout contract: the mask of this draw lands in the buffer, the device counter advances
    draws = torch.zeros(1, dtype=torch.int64, device=dev)
    mbuf = torch.empty(n, d, device=dev)
    l2 = ops.decoder_bce(t(Z, dev).requires_grad_(True), mbuf, gr, dropout=(0.1, 7, 0, draws))
    assert int(draws) == 1 and torch.equal(mbuf, ops.dropout_mask((n, d), 0.1, 7, device=dev))
    assert torch.equal(l2.detach(), ops.decoder_bce(t(Z, dev), mbuf, gr).detach())
    if d == 128:
        torch.manual_seed(1)
        model = G.GAE(39, [256, 128]).to(dev)          # the ADVICE example: --hidden_dims 256 128
        gr.ndata['h'] = t((rng.random((n, 39)) < 0.2).astype(np.float32), dev)
        lm = model.reconstruction_loss(gr)
        ops.backward(lm)
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())


def test_bce_logits_kernel(dev):
    """gae_bce_logits == F.binary_cross_entropy_with_logits (train_inductive.py:48) with pos_weight, mean reduction,
    and its gradient; ragged leading dimensions; deterministic"""
    from gae_dgl_amd import ops
    rng = np.random.default_rng(3)
    n, m = 301, 517
    x = rng.standard_normal((n, m)).astype(np.float32) * 4
    y = ((rng.random((n, m)) < 0.02) * rng.integers(1, 3, (n, m))).astype(np.float32)   # labels 0 / 1 / 2 (duplicate edges)
    pw = 37.5
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = O().bce_with_logits_mean(xt, torch.tensor(y, dtype=torch.float64), pw)
    ref.backward()
    xd = torch.zeros(n, m + 3, device=dev)[:, :m]; xd.copy_(t(x, dev))
    loss, g = ops.bce_logits_raw(xd, t(y, dev), pw)
    assert rel_err(loss.reshape(()), ref) < TOL and rel_err(g, xt.grad) < TOL
    loss2, g2 = ops.bce_logits_raw(t(x, dev), t(y, dev), pw, want_grad=False)
    assert g2 is None and torch.equal(loss2, loss)


def test_dense_kernels_reject_other_dtypes(dev):
    """fp32 kernels fed with fp64 / bf16 tensors raise instead of reinterpreting the memory"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    from gae_dgl_amd._lib import GaeHipError
    x = torch.randn(50, 8, device=dev); w = torch.randn(4, 8, device=dev); b = torch.randn(4, device=dev)
    for bad in (lambda: ops.linear_fwd_raw(x.double(), w, b, 0), lambda: ops.linear_fwd_raw(x, w.bfloat16(), b, 0),
                lambda: ops.linear_fwd_raw(x, w, b.double(), 0), lambda: ops.decoder_dense_raw(x.double()),
                lambda: ops.linear_bwd_raw(torch.randn(50, 4, device=dev).double(), None, 0, x, w)):
        with pytest.raises(GaeHipError):
            bad()
    g = G.DGLGraph((np.array([0, 1]), np.array([1, 0])), num_nodes=50).to(dev)
    with pytest.raises(GaeHipError):
        ops.decoder_bce(x.double(), None, g)
    model = G.GAE(8, [4, 2]).to(dev).double()
    g.ndata['h'] = x
    with pytest.raises(GaeHipError):
        model.encode(g)


def test_dropout_draws_are_disjoint_streams(dev):
    """the draw index selects a Philox stream of its own (high counter words): draw 1 over n elements is NOT the
    continuation of draw 0 over a longer tensor (it was, when the draw index scaled the element counter)"""
    from gae_dgl_amd import ops
    n = 4096
    one = torch.ones(1, dtype=torch.int64, device=dev)
    m0_long = ops.dropout_mask((2 * n,), 0.5, seed=9, device=dev)
    m1 = ops.dropout_mask((n,), 0.5, seed=9, device=dev, draw_counter=one)
    assert not torch.equal(m1, m0_long[n:]) and not torch.equal(m1, m0_long[:n])
    assert torch.equal(ops.dropout_mask((n,), 0.5, seed=9, device=dev), m0_long[:n])     # draw 0: prefix property
    assert torch.equal(m1, ops.dropout_mask((n,), 0.5, seed=9, device=dev, draw_counter=one))
    assert abs(float((m1 > 0).float().mean()) - 0.5) < 0.05


def test_adam_state_dict_and_captured_lr_change(dev):
    """optimizer.state_dict() has torch.optim.Adam's layout (checkpoints move between the two optimisers, steps
    included); a learning-rate change after the capture reaches the replayed step"""
    import gae_dgl_amd as G
    from gae_dgl_amd.optim import Adam
    from gae_dgl_amd.capture import CapturedTrainStep
    torch.manual_seed(0)
    ps = [torch.randn(7, 5, device=dev, requires_grad=True), torch.randn(5, device=dev, requires_grad=True)]
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    a, b = Adam(ps, lr=1e-2), torch.optim.Adam(qs, lr=1e-2)
    for k in range(3):
        for p, q in zip(ps, qs):
            gk = torch.randn_like(p); p.grad = gk.clone(); q.grad = gk.clone()
        a.step(); b.step()
    sd = a.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 3.0
    assert "_hip_steps" not in sd["param_groups"][0]
    rs = [p.detach().clone().requires_grad_(True) for p in ps]
    import copy        # load_state_dict keeps same-device tensors by reference: copy, or the optimisers share moments
    c = torch.optim.Adam(rs, lr=1e-2); c.load_state_dict(copy.deepcopy(sd))   # ours -> torch
    us = [p.detach().clone().requires_grad_(True) for p in ps]
    d = Adam(us, lr=1e-2); d.load_state_dict(copy.deepcopy(b.state_dict()))   # torch -> ours
    for p, q, r, u in zip(ps, qs, rs, us):
        gk = torch.randn_like(p)
        for x in (p, q, r, u):
            x.grad = gk.clone()
    for o in (a, b, c, d):
        o.step()
    assert d.steps_taken() == 4
    for p, q, r, u in zip(ps, qs, rs, us):
        assert rel_err(p, q) < 2e-6 and rel_err(r, q) < 2e-6 and rel_err(u, q) < 2e-6
    # captured step: lr is a launch argument -> the capture is redone when it changes
    rng = np.random.default_rng(0)
    n = 600
    src, dst = rand_graph(rng, n, 3000)
    g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    X = t(rng.standard_normal((n, 20)).astype(np.float32), dev)
    torch.manual_seed(2)
    model = G.GAE(20, [8, 4]).to(dev); model.decoder.dropout = 0.0
    opt = Adam(model.parameters(), lr=1e-2)
    step = CapturedTrainStep(model, opt, g, X)
    step(); torch.cuda.synchronize()
    w0 = model.layers[0].apply_mod.linear.weight.detach().clone()
    for grp in opt.param_groups:
        grp["lr"] = 0.0
    step(); torch.cuda.synchronize()
    assert torch.equal(model.layers[0].apply_mod.linear.weight.detach(), w0)       # lr = 0 took effect
    for grp in opt.param_groups:
        grp["lr"] = 1e-2
    step(); torch.cuda.synchronize()
    assert not torch.equal(model.layers[0].apply_mod.linear.weight.detach(), w0)


# ----------------------------------------------------------------- semantics decisions / opt-in layer options
def test_update_all_on_edgeless_graph(dev):
    """DGL 0.4 (the reference's API era) short-cuts update_all on a graph without any edge and leaves 'h' = the
    layer input (default here: drop-in); "zeros" is the mathematical aggregate, which is also what the raw SpMM
    kernel returns.  Graphs with edges: zero-in-degree rows aggregate to 0 in both modes."""
    import gae_dgl_amd as G
    from gae_dgl_amd import graph as gmod
    n = 9
    X = torch.randn(n, 5, device=dev)
    torch.manual_seed(0)
    layer = G.GCN(5, 3, torch.relu).to(dev)
    W, b = layer.apply_mod.linear.weight, layer.apply_mod.linear.bias
    empty = G.DGLGraph((np.zeros(0, np.int64), np.zeros(0, np.int64)), num_nodes=n).to(dev)
    assert gmod.ZERO_EDGE_UPDATE_ALL == "dgl04"
    with torch.no_grad():
        out = layer(empty, X)
        assert rel_err(out, torch.relu(X @ W.t() + b)) < TOL and 'h' not in empty.ndata
        gmod.ZERO_EDGE_UPDATE_ALL = "zeros"
        try:
            assert rel_err(layer(empty, X), torch.relu(b).expand(n, 3)) < TOL
        finally:
            gmod.ZERO_EDGE_UPDATE_ALL = "dgl04"
        one = G.DGLGraph((np.array([0]), np.array([1])), num_nodes=n).to(dev)        # a single edge 0 -> 1
        ref = torch.zeros(n, 5, device=dev); ref[1] = X[0]
        assert rel_err(layer(one, X), torch.relu(ref @ W.t() + b)) < TOL


@pytest.mark.parametrize("name", ["sym200", "deep3"])
def test_transform_first_option(name, dev):
    """opt-in A (H W^T) order of the narrowing layers: embeddings, loss and parameter gradients within the fp32
    tolerance of the reference-order goldens; state-dict keys unchanged"""
    import gae_dgl_amd as G
    g = load_golden(name)
    model = G.GAE(g["X"].shape[1], [int(v) for v in g["hidden"]], transform_first=True)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    assert sorted(model.state_dict()) == sorted(sd)                 # the option adds no checkpoint keys
    model.load_state_dict(sd)
    model = model.to(dev)
    assert any(l.transform_first for l in model.layers)          # only layers that narrow the features reorder
    model.decoder.dropout = 0.0
    gr = fresh_graph(g, dev)
    loss = model.reconstruction_loss(gr)
    assert rel_err(gr.ndata['h'], g["Z"]) < TOL and rel_err(loss, g["loss_p0"]) < TOL
    loss.backward()
    for k, p in model.named_parameters():
        assert rel_err(p.grad, g[f"grad_p0/{k}"]) < 5 * TOL, k


def test_cache_first_aggregate_option(dev):
    """opt-in reuse of A X across steps (the transductive loop aggregates the same features every epoch): identical
    results, the layer-1 SpMM runs once, an in-place change of X invalidates the cache"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops
    rng = np.random.default_rng(4)
    n = 500
    src, dst = rand_graph(rng, n, 2500)
    X = t(rng.standard_normal((n, 70)).astype(np.float32), dev)
    torch.manual_seed(3)
    plain = G.GAE(70, [16, 8]).to(dev)
    torch.manual_seed(3)
    cached = G.GAE(70, [16, 8], cache_first_aggregate=True).to(dev)
    g1 = G.DGLGraph((src, dst), num_nodes=n).to(dev); g2 = G.DGLGraph((src, dst), num_nodes=n).to(dev)
    for m in (plain, cached):
        m.decoder.dropout = 0.0
    prof = ops.EventProfiler(); ops.profiler = prof
    try:
        for step in range(3):
            g1.ndata['h'] = X; g2.ndata['h'] = X
            la, lb = plain.reconstruction_loss(g1), cached.reconstruction_loss(g2)
            assert torch.equal(la.detach(), lb.detach())
            la.backward(); lb.backward()
        wide = [k for k in prof.records if k[0] == "spmm" and k[3] == 70]
        assert sum(len(prof.records[k]) for k in wide) == 3 + 1                 # plain: every step, cached: once
        for pa, pb in zip(plain.parameters(), cached.parameters()):
            assert torch.equal(pa.grad, pb.grad)
        X.mul_(2.0)                                                             # version bump -> recompute
        g1.ndata['h'] = X; g2.ndata['h'] = X
        assert torch.equal(plain.reconstruction_loss(g1).detach(), cached.reconstruction_loss(g2).detach())
    finally:
        ops.profiler = None


@pytest.mark.parametrize("f_in,f_out,act,norm", [(32, 16, "identity", "none"), (39, 32, "relu", "none"),
                                                 (16, 32, "identity", "both"), (64, 7, "relu", "both"),
                                                 (5, 3, "identity", "none")])
def test_fused_gcn_layer_matches_two_launches(f_in, f_out, act, norm, dev):
    """gae_gcn_layer_fused (aggregation + Linear + bias + activation in one launch, backward of the identity layer
    as one launch on A^T) == update_all + apply_nodes: forward and all three gradients, with and without the
    D^-1/2 A D^-1/2 scales, widths with tails (39, 7, 5, 3), rows longer than the packed table"""
    import gae_dgl_amd as G
    from gae_dgl_amd import gae as GM
    rng = np.random.default_rng(f_in * 100 + f_out)
    n = 3000
    src, dst = rand_graph(rng, n, 5 * n, hub=False)
    src = np.concatenate([src, rng.integers(0, n, 40)]); dst = np.concatenate([dst, np.full(40, 7)])   # one 40+-edge row
    X = rng.standard_normal((n, f_in)).astype(np.float32)
    actf = GM.identity if act == "identity" else torch.relu
    out = {}
    for fused in (True, False):
        GM.FUSE_NARROW_LAYERS = fused
        try:
            torch.manual_seed(5)
            layer = GM.GCN(f_in, f_out, actf, norm=norm).to(dev)
            g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
            x = t(X, dev).requires_grad_(True)
            y = layer(g, x)
            w = t(rng.standard_normal((n, f_out)).astype(np.float32), dev) if fused else out["w"]
            out["w"] = w
            (y * w).sum().backward()
            out[fused] = (y.detach(), x.grad.clone(), layer.apply_mod.linear.weight.grad.clone(),
                          layer.apply_mod.linear.bias.grad.clone())
        finally:
            GM.FUSE_NARROW_LAYERS = True
    for a, b, tol in zip(out[True], out[False], (1e-5, 5e-5, 5e-5, 5e-5)):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6)
    # and against the oracle's layer
    from oracle import gae_oracle as O
    ip, ix = O.csr_from_coo(src, dst, n)
    layer_w = layer.apply_mod.linear.weight.detach().cpu().double().numpy()
    layer_b = layer.apply_mod.linear.bias.detach().cpu().double().numpy()
    nrm = O.norm_from_in_degrees(O.in_degrees(dst, n)) if norm == "both" else None
    ref = O.gcn_layer(ip, ix, torch.as_tensor(X).double(), torch.as_tensor(layer_w), torch.as_tensor(layer_b), act,
                      norm=None if nrm is None else torch.as_tensor(nrm).double())
    assert float((out[True][0].cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_spmm_hot_column_tags_are_only_cache_hints(dev, tuning):
    """hot-column tags of a device-built plan (csrc/plan_build.hip): the mid rows read a COMPACT copy of their column
    ids whose sign bit marks the most gathered columns (frequencies counted over that copy, threshold = frequency of
    the HOT_COLUMNS-th column, never below 2); a heavy-row SpMM is bit-identical with and without the tags (fp32 and
    bf16 storage, scaled)"""
    from gae_dgl_amd import ops
    rng = np.random.default_rng(21)
    n, e = 4000, 120000
    dst = (rng.integers(0, n, e).astype(np.float64) ** 3 / n ** 2).astype(np.int64)          # heavy rows
    src = (rng.integers(0, n, e).astype(np.float64) ** 4 / n ** 3).astype(np.int64)          # hub columns
    ip, ix = ops.csr_from_coo(t(dst, dev), t(src, dev), n, n)
    old = ops.HOT_COLUMNS
    ops.HOT_COLUMNS = 64
    try:
        plan_hot = ops.spmm_plan(ip, threshold=8, segment=64, indices=ix, ell=False, hot=True, n_cols=n)
    finally:
        ops.HOT_COLUMNS = old
    plan_off = ops.spmm_plan(ip, threshold=8, segment=64, indices=ix, ell=False, hot=False, n_cols=n)
    assert plan_hot.hot_indices is not None and plan_off.hot_indices is None and plan_hot.n_heavy > 0
    assert plan_off.mid_ids is None                               # untagged plans read the CSR's own ids: no copy
    # the compact copy: ids of the heavy rows in row order, tags by frequency
    ipn, ixh = ip.cpu().numpy(), ix.cpu().numpy()
    hr = plan_hot.tensors[0].cpu().numpy()
    want = np.concatenate([ixh[ipn[r]:ipn[r + 1]] for r in hr])
    tg = plan_hot.mid_ids.cpu().numpy()
    assert np.array_equal(tg & 0x7fffffff, want)
    freq = np.bincount(want, minlength=n)
    kth = max(np.sort(freq)[::-1][63], 2)
    assert np.array_equal(tg < 0, freq[want] >= kth) and 0 < (tg < 0).mean() < 1
    deg, norm = ops.degree_norm(ip)
    for dtype in (torch.float32, torch.bfloat16):
        H = t(rng.standard_normal((n, 40)).astype(np.float32), dev).to(dtype)
        for sc in (None, norm):
            a = ops.spmm_raw(ip, ix, H, n, sc, sc, plan=plan_hot)
            b = ops.spmm_raw(ip, ix, H, n, sc, sc, plan=plan_off)
            assert torch.equal(a, b)
            tuning("spmm_hot", 0)
            c = ops.spmm_raw(ip, ix, H, n, sc, sc, plan=plan_hot)
            tuning("spmm_hot", 1)
            assert torch.equal(a, c)


def test_spmm_segment_descriptors_match_the_index_chain(dev, tuning):
    """seg_desc of a device-built plan: {row, first edge, end edge, only segment} of every segment, rows ascending,
    segments of a row consecutive; offsets index the CSR (untagged plan) or the compact id copy (tagged plan), whose
    slices hold the same ids.  A heavy-row SpMM (plain, scaled, accumulating, with an XCD-pinned part) is bit-identical
    whether the kernel reads the descriptors or walks the index chain (knob spmm_desc; untagged plans only: the compact
    copy has no chain)"""
    from gae_dgl_amd import ops
    rng = np.random.default_rng(22)
    n, e, seg = 5000, 300000, 128
    dst = (rng.integers(0, n, e).astype(np.float64) ** 4 / n ** 3).astype(np.int64)
    src = (rng.integers(0, n, e).astype(np.float64) ** 2 / n).astype(np.int64)
    ip, ix = ops.csr_from_coo(t(dst, dev), t(src, dev), n, n)
    ipn, ixn = ip.cpu().numpy(), ix.cpu().numpy()
    deg_n = np.diff(ipn)
    for homed in (False, True):
        for hot in (False, True):
            plan = ops.spmm_plan(ip, threshold=8, segment=seg, indices=ix, ell=False, hot=hot, n_cols=n, homed=homed)
            assert plan.seg_desc is not None and plan.n_segments > 0 and (plan.homed is not None) == homed
            hr, hb, sh = (x.cpu().numpy() for x in plan.tensors[:3])
            hi = ops.HOMED_MIN_DEGREE if homed else 2 ** 31 - 1
            assert np.array_equal(hr, np.nonzero((deg_n > 8) & (deg_n <= hi))[0])           # ascending rows
            ns = (deg_n[hr] + seg - 1) // seg
            assert np.array_equal(hb, np.cumsum(ns) - ns) and np.array_equal(sh, np.repeat(np.arange(len(hr)), ns))
            d = plan.seg_desc.cpu().numpy()
            rows = hr[sh]
            k = np.arange(plan.n_segments) - hb[sh]
            assert np.array_equal(d[:, 0], rows)
            assert np.array_equal(d[:, 2] - d[:, 1], np.minimum(seg, deg_n[rows] - k * seg))
            assert np.array_equal(d[:, 3], ((k == 0) & (deg_n[rows] <= seg)).astype(np.int32))
            e0 = ipn[rows] + k * seg
            if not hot:
                assert plan.mid_ids is None and np.array_equal(d[:, 1], e0)
            else:
                ids = plan.mid_ids.cpu().numpy() & 0x7fffffff
                for sidx in rng.integers(0, plan.n_segments, 50):
                    assert np.array_equal(ids[d[sidx, 1]:d[sidx, 2]], ixn[e0[sidx]:e0[sidx] + d[sidx, 2] - d[sidx, 1]])
            deg, norm = ops.degree_norm(ip)
            for dtype in (torch.float32, torch.bfloat16):
                H = t(rng.standard_normal((n, 40)).astype(np.float32), dev).to(dtype)
                base = t(rng.standard_normal((n, 40)).astype(np.float32), dev).to(dtype)
                for sc in (None, norm):
                    out = {}
                    for knob in ((1, 0) if not hot else (1,)):
                        tuning("spmm_desc", knob)
                        acc = base.clone()
                        ops.spmm_raw(ip, ix, H, n, sc, sc, plan=plan, out=acc, accumulate=True)
                        out[knob] = (ops.spmm_raw(ip, ix, H, n, sc, sc, plan=plan), acc)
                    tuning("spmm_desc", 1)
                    if not hot:
                        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("dtype,F", [(torch.float32, 40), (torch.bfloat16, 40), (torch.float32, 300)])
def test_spmm_homed_rows_match_oracle(dtype, F, dev):
    """XCD-pinned ("homed") part of a device-built skew plan: the very long rows are evaluated from virtual rows grouped
    by the home of their columns -- same terms, another summation order: == the plain plan and the fp64 oracle to 1e-5
    of the scale (fp32), scaled and unscaled, with GAE_SPMM_ACCUMULATE, and the layout invariants of the plan hold"""
    from gae_dgl_amd import ops
    from oracle import c_oracle as C
    rng = np.random.default_rng(33)
    n, e = 6000, 400000
    dst = (rng.integers(0, n, e).astype(np.float64) ** 4 / n ** 3).astype(np.int64)          # rows of thousands of edges
    src = (rng.integers(0, n, e).astype(np.float64) ** 2 / n).astype(np.int64)
    ip, ix = ops.csr_from_coo(t(dst, dev), t(src, dev), n, n)
    plan = ops.spmm_plan(ip, threshold=8, segment=128, indices=ix, ell=False, hot=True, n_cols=n, homed=True)
    plain = ops.spmm_plan(ip, threshold=8, segment=128, indices=ix, ell=False, hot=False, n_cols=n, homed=False)
    hp = plan.homed
    assert hp is not None and plain.homed is None
    ipn, ixn = ip.cpu().numpy(), ix.cpu().numpy()
    deg = np.diff(ipn)
    vrows = np.nonzero(deg > ops.HOMED_MIN_DEGREE)[0]
    assert np.array_equal(hp["rows"].cpu().numpy(), vrows)
    if plan.n_heavy:
        hr = plan.tensors[0].cpu().numpy()
        assert ((deg[hr] > 8) & (deg[hr] <= ops.HOMED_MIN_DEGREE)).all()
    # ids: every pinned row's list regrouped by home, ascending columns inside a home (stable partition of the CSR row)
    cols = hp["cols"].cpu().numpy()
    assert len(cols) == hp["n_edges"] == deg[vrows].sum()
    home_of = lambda c: ops.column_home(torch.from_numpy(c.astype(np.int64))).numpy()
    at = 0
    for r in vrows[:40]:
        row = ixn[ipn[r]:ipn[r + 1]]
        want = np.concatenate([row[home_of(row) == h] for h in range(8)])
        assert np.array_equal(cols[at:at + len(row)], want)
        at += len(row)
    # virtual rows: position p is gathered by thread block p / 4 on XCD (p / 4) % 8 = the home of all its columns
    desc = hp["desc"].cpu().numpy()
    V = hp["n_virtual"]
    assert desc.shape == (V, 4) and V % 32 == 0
    lens = desc[:, 2] - desc[:, 1]
    live = np.nonzero(lens > 0)[0]
    assert lens.max() <= 128 and lens.min() >= 0 and lens.sum() == hp["n_edges"]
    assert np.array_equal(desc[live, 0], live)
    for p_ in live[rng.integers(0, len(live), 200)]:
        assert (home_of(cols[desc[p_, 1]:desc[p_, 2]]) == (p_ // 4) % 8).all()
    # the chunks tile the id array exactly once
    order = np.argsort(desc[live, 1])
    assert np.array_equal(desc[live, 1][order][1:], desc[live, 2][order][:-1]) and desc[live, 1].min() == 0
    # partial lists: row r's virtual rows in (home, chunk) order
    pp, pq = hp["part_ptr"].cpu().numpy(), hp["part_pos"].cpu().numpy()
    assert len(np.unique(pq)) == len(pq) == len(live) and pp[-1] == len(pq) and len(pp) == len(vrows) + 1
    starts = np.concatenate([[0], np.cumsum(deg[vrows])])
    for k_ in range(min(len(vrows), 40)):
        ps = pq[pp[k_]:pp[k_ + 1]]
        assert (desc[ps, 1] >= starts[k_]).all() and (desc[ps, 2] <= starts[k_ + 1]).all()
        assert np.array_equal(desc[ps, 1], np.sort(desc[ps, 1])) and lens[ps].sum() == deg[vrows[k_]]
    # inside a home the chunks are launched in the order of their first column
    for h in range(8):
        ph = live[(live // 4) % 8 == h]
        rank = (ph // 32) * 4 + ph % 4
        first = cols[desc[ph, 1]][np.argsort(rank)]
        assert np.array_equal(first, np.sort(first))
    deg_t, norm = ops.degree_norm(ip)
    H = t(rng.standard_normal((n, F)).astype(np.float32), dev).to(dtype)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for sc in (None, norm):
        a = ops.spmm_raw(ip, ix, H, n, sc, sc, plan=plan)
        b = ops.spmm_raw(ip, ix, H, n, sc, sc, plan=plain)
        Hn = H.float().cpu().numpy()
        if sc is None:
            ref = C.spmm_csr_acc64(ip.cpu().numpy(), ix.cpu().numpy(), Hn)
        else:       # diag(s) A diag(s) H = s * (A (s * H)) with the scaling done in fp64 around the oracle's sum
            sn = sc.cpu().numpy().astype(np.float64)
            ref = sn[:, None] * C.spmm_csr_acc64(ip.cpu().numpy(), ix.cpu().numpy(),
                                                 (sn[:, None] * Hn).astype(np.float32))
        scale = float(np.abs(ref).max())
        assert float((a.float() - b.float()).abs().max()) <= tol * scale
        assert float(np.abs(a.float().cpu().numpy() - ref).max()) <= tol * scale
        base = torch.randn(n, F, device=dev).to(dtype)
        acc = base.clone()
        ops.spmm_raw(ip, ix, H, n, sc, sc, plan=plan, out=acc, accumulate=True)
        # (bf16: one rounding of base + a, whose magnitude is that of the unseeded N(0, 1) base, not of the product)
        bound = 1e-6 * scale if dtype == torch.float32 else 2e-2 * max(scale, float(base.float().abs().max()))
        assert float((acc.float() - (base.float() + a.float())).abs().max()) <= bound
    # deterministic
    assert torch.equal(ops.spmm_raw(ip, ix, H, n, plan=plan), ops.spmm_raw(ip, ix, H, n, plan=plan))
