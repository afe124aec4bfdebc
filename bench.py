#!/usr/bin/env python3
"""Benchmark of the GCN-encoder hot path on MI355X (metric / configs: BASELINE.json).

  python bench.py --gpus N --steps K --warmup W [--workload pubmed|cora|citeseer|zinc|vgae|rmat]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started without a launcher (no WORLD_SIZE in the environment), `--gpus N` with N > 1 starts its N ranks itself (one
process per GPU, re-executed under torch.distributed.run) and exits with code 2 when fewer than N GPUs are visible;
under a launcher, WORLD_SIZE must equal --gpus.  `n_gpus` in the line is the size of the live process group.

A "step" is one pass of the hot path over one batch of synthetic input:

  pubmed / cora / citeseer (N=1 default: pubmed = BASELINE configs[1])
      one full-graph training step = one transductive epoch: encoder forward
      (2 HIP SpMM + 2 fp32-MFMA Linear), fused inner-product decoder + weighted
      BCE (never materialises N x N), backward (HIP SpMM on A^T, Linear
      backward), Adam.
  zinc    one inductive training step on a batch of 4096 ZINC-shaped molecules:
      on-device dgl.batch (HIP gather from the dataset CSR) + the same step.
  rmat    (default for N>1) RMAT scale 24 / edge factor 16, F=32, graph
      row-sharded over the N ranks with an RCCL all-gather of H before every
      SpMM: encoder forward + backward + gradient all-reduce + Adam.  The
      O(N^2) decoder is not part of this config (2.8e14 logits); the encoder's
      upstream gradient dZ is a fixed synthetic tensor.  Strong scaling.

value = SpMM edges aggregated per step (forward + backward launches, all ranks)
/ step time: whole-job, inputs resident in HBM, max over ranks.  `value_spmm_only`
= the same edges / the HIP-event time of the step's SpMM launches alone.
`roofline`  : the metric's kernel, the SpMM aggregation, HIP-event timed on the
              launch stream against SURVEY.md 8(d)'s B_alg.  Pubmed (default): the
              reference-order layer-1 product A X at F = 500 (what `--layer1 reference`
              runs inside the step), with `in_step` = the aggregation the default step
              runs (F = 32).  `frac` = back-to-back launches on one operand set (below
              256 MB: resident in the Infinity Cache); `frac_cold` = the same launch
              rotating over >= 512 MB of disjoint operand sets (every launch reads DRAM).
`roofline_dense`: the dense layer-1 pair of the default step (xw_fwd, xtg), same fields.
`roofline_step_dominant`: the fused decoder + BCE against the measured issue rate.
`cpu_baseline`: the CPU oracle's restatement of the same step on this host.
`extra`     : kernel-only SpMM numbers (warm and cold) for the other BASELINE shapes,
              whole steps of the other single-GPU configurations.
N > 1: `rccl_ranks`, `preflight` (every collective primitive checked against known
answers before the workload is built), `comm` (exchange vs SpMM time per step).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# VALU roofline of the fused decoder + BCE kernels (the step's dominant launch): fp32 lane-operations per second of the
# vector ALUs, 256 CUs x 4 SIMD-32 x 2.4 GHz (spec; tools/probes/valu_rate.hip sustains 5.55e13 = 71 % of it with
# v_fma_f32), a transcendental (v_exp / v_log / v_rcp) costs 4 plain operations (quarter rate; measured 3.6-3.9).
# ---- issue model of the fused decoder + BCE kernels (tools/probes/inst_cost.hip, profiles/r02_probe_inst_cost.txt).
# On gfx950 a SIMD's MFMA time and VALU time ADD (mfma + 4 v_fma: 11.2 ns against 7.4 + 5.7), so the bound of these
# kernels is the sum of their instructions' issue costs.  Costs in units of one v_fma_f32 (1.42 ns per wave64
# instruction and SIMD at 4 waves / SIMD): transcendental 2.5, v_pk_*_f32 1.8, v_cvt_pk_bf16_f32 / v_bfi_b32 1.7,
# v_perm_b32 1.4, v_lshlrev_b32 1.6, v_mfma_f32_16x16x{16,32}_{bf16,f16} 5.2.
SPEC_LANE_OPS = 157.3e12 / 2     # MI355X data sheet: 157.3 TFLOP/s fp32 vector = 78.6 T fused lane-operations per second
SLOT_NS = 1.42
ISSUE_PEAK_LANE_SLOTS = 256 * 4 * 64 / (SLOT_NS * 1e-9)      # lane-slots per second of the whole chip
SLOT_COST = {"plain": 1.0, "trans": 2.5, "pk": 1.8, "cvt_pk": 1.7, "bfi": 1.7, "perm": 1.4, "lshl": 1.6, "mfma": 5.2}
# instructions per wave and 64-column tile of a 32-row wave slice (= 32 logits per lane), counted in the ISA of the
# kernels' main loops (decoder_bce.hip, K = 32 MFMAs); third entry: share of the N^2 logits that is evaluated
LOSS_ISA = {
    # round 4: the symmetric kernel splits into fp16 pieces (v_cvt_pk_f16_f32 / v_cvt_pkrtz_f16_f32 counted as cvt_pk,
    # v_fma_mix_f32 as plain), the full-square kernel forms S from three bf16 pieces (24 + 12 MFMAs); since the
    # one-reciprocal-per-four-logits form 10 instead of 34 transcendentals behind the 32 v_exp (tile loops of the
    # gfx950 assembly, instruction classes counted by script: 424 / 344 vector instructions per tile)
    "symmetric": ({"trans": 42, "pk": 6, "cvt_pk": 68, "bfi": 32, "perm": 0, "lshl": 3, "plain": 217, "mfma": 56}, 0.5),
    # the 256-row-panel form of the symmetric kernel (from 32 k rows on; unrolled tile loop: 711 vector instructions per
    # 64 logits and lane, halved here)
    "symmetric256": ({"trans": 40, "pk": 3, "cvt_pk": 66, "bfi": 32, "perm": 0, "lshl": 1.5, "plain": 157, "mfma": 56}, 0.5),
    "full": ({"trans": 40, "pk": 14, "cvt_pk": 32, "bfi": 32, "perm": 0, "lshl": 22, "plain": 168, "mfma": 36}, 1.0),
}


# The ALGORITHMICALLY REQUIRED part of those instruction mixes: the S product and the two P V products
# (row side and mirror) at the chosen piece count, one exp per logit, one reciprocal per four, the log of the running
# product, and the fp32 arithmetic between them (|y| sum, 1 + e, product tree, R t, the fma that forms sigma - 1/2).
# Everything else the kernels issue -- splitting P into 16-bit pieces, copysign, re-laying P out for the mirror product
# (identity MFMAs + conversions), LDS traffic, loop overhead -- is emulation overhead of running fp32-grade products on
# the 16-bit matrix pipe.  The full-square kernel has no mirror product.
LOSS_ALG = {
    "symmetric": {"mfma": 16 + 12 + 12, "trans": 32 + 8 + 2, "plain": 32 + 32 + 32 + 16 + 32},
    "symmetric256": {"mfma": 16 + 12 + 12, "trans": 32 + 8 + 2, "plain": 32 + 32 + 32 + 16 + 32},
    "full": {"mfma": 24 + 12, "trans": 32 + 8 + 2, "plain": 32 + 32 + 32 + 16 + 32},
}


def loss_alg_slots_per_logit(kind):
    return sum(SLOT_COST[k] * v for k, v in LOSS_ALG[kind].items()) / 32.0


def loss_slots_per_logit(kind):
    isa, frac_eval = LOSS_ISA[kind]
    return sum(SLOT_COST[k] * v for k, v in isa.items()) / 32.0, frac_eval


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--loss", choices=["fused", "dense"], default="fused",
                    help="fused = HIP decoder+BCE kernel (never materialises N x N); dense = reference-shaped "
                         "N x N logits/labels with torch BCE")
    ap.add_argument("--rmat-scale", type=int, default=24)
    ap.add_argument("--batch-graphs", type=int, default=4096)
    ap.add_argument("--exchange", choices=["allgather", "boundary"], default="boundary",
                    help="rmat: rows exchanged before each SpMM -- boundary = all-to-all-v of the remote rows a rank's "
                         "block references (RMAT s24 / 8 ranks: 7-31 %% of the all-gather volume), allgather = all rows")
    ap.add_argument("--balance", choices=["rows", "nnz"], default="nnz",
                    help="rmat: row blocks of equal row count or of equal edge count (RMAT puts 44 %% of the edges "
                         "into the first of 8 equal row blocks)")
    ap.add_argument("--row-cost", type=int, default=None,
                    help="rmat, --balance nnz: cost of one ROW of a block in units of one edge visit (default: "
                         "parallel.ROW_COST = 16, from the byte model there: ~1.5 KB of dense-pass traffic per row and step "
                         "against ~90 B of measured HBM traffic per edge visit)")
    ap.add_argument("--layer-order", choices=["auto", "aggregate-first", "transform-first"], default="auto",
                    help="rmat workload: aggregate-first = every layer as (A H) W^T like gae.py:26-31; transform-first "
                         "(= auto) = the 32 -> 16 layer as A (H W^T) + b and its backward from G = A^T dZ (same values up to "
                         "fp32 rounding): two of the three aggregations and exchanges per step run at width 16, the dense "
                         "halves are two one-pass kernels (parallel.ShardedEncoder2Function)")
    ap.add_argument("--no-cache-input-exchange", action="store_true",
                    help="rmat, N > 1: exchange the remote rows of the constant input features X in every step "
                         "(default: once -- X does not change between steps, only its product A X is recomputed)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="rmat: exchange, then one SpMM over the assembled rows (bit-identical to one GPU) instead of "
                         "the default own-column SpMM under the exchange + accumulated remote-column SpMM")
    ap.add_argument("--torch-adam", action="store_true",
                    help="torch.optim.Adam(fused=True) instead of the library's one-launch Adam (same update rule)")
    ap.add_argument("--knobs", default="", help="comma-separated gae_tuning_set name=value pairs (experiments)")
    ap.add_argument("--layer1", choices=["transform-first", "reference"], default="transform-first",
                    help="citation workloads, layer 1 (500 / 1433 / 3703 -> 32): transform-first = act(A (X W^T) + b), the "
                         "library's default for layers that narrow wide features (X read once per direction, the 40 MB "
                         "aggregate A X never exists; value of gae.py:26-31 up to fp32 rounding); reference = "
                         "act((A X) W^T + b) in the reference's order (the F_in-wide SpMM is then the dominant launch)")
    ap.add_argument("--degrees", choices=["uniform", "planetoid"], default="uniform",
                    help="citation workloads, the synthetic graph's degree sequence: uniform = pairs sampled uniformly "
                         "(longest row ~ 16; every round's headline), planetoid = heavy-tailed with the real graph's "
                         "longest row (hubs of 100 - 170 neighbours: workloads.citation_graph)")
    ap.add_argument("--features", choices=["auto", "dense", "sparse"], default="auto",
                    help="citation workloads, the constant input features X: dense = as the reference holds them, a dense "
                         "FloatTensor (gae_xw_fwd / gae_xw_wgrad stream it once per direction); sparse = compressed once at "
                         "set-up (gae_dgl_amd.SparseFeatures: the non-zeros of the bag-of-words rows), layer 1 from the "
                         "non-zeros (gae_spx_fwd / gae_spx_wgrad, same values); auto (default) = SparseFeatures."
                         "maybe_from_dense: compressed only where that is faster (wide and very sparse X: Citeseer, Cora; "
                         "Pubmed -- the headline -- stays dense)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="N > 1 on a box with fewer than N GPUs: the ranks SHARE the visible device(s) (rank r on GPU "
                         "r mod #GPUs) and the collectives travel over gloo, staged through host memory (RCCL refuses "
                         "two ranks on one device).  A functional rehearsal of the N-rank launch with the real kernels; "
                         "the line says so in config.transport and its numbers are NOT a scaling measurement")
    ap.add_argument("--no-fused-layers", action="store_true",
                    help="run narrow GCN layers as two launches (update_all, apply_nodes) instead of gae_gcn_layer_fused")
    ap.add_argument("--no-hipgraph", action="store_true",
                    help="citation workloads: launch the step eagerly.  Default: the timed steps replay the step as "
                         "one captured HIP graph (the ~30 launches are host-bound otherwise); HIP events cannot be "
                         "recorded inside a replay, so the per-kernel event timings of `roofline` come from an eager "
                         "pass of the SAME steps immediately before the capture (with --no-hipgraph they sit inside "
                         "the timed region)")
    return ap.parse_args()


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic_r02.json, else _r01: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction)."""
    for name in ("pmc_traffic_r06.json", "pmc_traffic_r05.json", "pmc_traffic_r04.json", "pmc_traffic_r03.json", "pmc_traffic_r02.json", "pmc_traffic_r01.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                v = json.load(f).get(key, {}).get("hbm_bytes_per_launch")
            if v is not None:
                return v
        except OSError:
            pass
    return None


def time_launches(fn, iters=50, warmup=5):
    """average duration of `fn`'s launches between one HIP event pair on the launch stream.  Short launches
    (< 1 ms) are replayed from a HIP graph holding `iters` copies: issued one by one from Python they are
    bounded by the host cost of a ctypes launch (~10 us, more while the host is busy), not by the GPU."""
    for _ in range(warmup):
        fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    if t < 1e-3:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(iters):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            best = t
            for _ in range(3):
                e0.record(); g.replay(); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e-3 / iters)
            t = best
        except Exception as ex:   # capture not possible (e.g. a collective inside): keep the eager figure
            print(f"[bench] graph-replay timing unavailable: {ex}", file=sys.stderr)
            torch.cuda.synchronize()
    return t


COLD_BYTES = 512 << 20     # operand copies a cold measurement rotates over: twice the 256 MB Infinity Cache (MALL)


def time_rotation(fns, iters=None):
    """average duration of one launch when ``fns`` (the same kernel on DISJOINT operand copies) are launched in
    rotation: every launch finds operands that 511+ MB of other traffic have pushed out of the L2s and the Infinity
    Cache since it last touched them -- the MALL-cold figure (VERDICT r05 #2b).  Same method as time_launches: one HIP
    event pair on the launch stream around a HIP-graph replay of the whole rotation."""
    K = len(fns)
    iters = iters or max(3 * K, 48)
    iters = (iters + K - 1) // K * K
    for f in fns:
        f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            # (results kept alive through the capture: a launch that allocates its output gets a block of its own
            #  instead of the one the previous launch just freed -- which would be a warm write)
            keep = [fns[i % K]() for i in range(iters)]
        g.replay()
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3 / iters)
        return best
    except Exception as ex:
        print(f"[bench] graph-replay timing of the rotation unavailable: {ex}", file=sys.stderr)
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            fns[i % K]()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / iters


def cold_copies(working_set_bytes, cap=160):
    """how many disjoint operand copies a rotation needs so that >= COLD_BYTES pass between two uses of one copy"""
    return int(min(cap, max(2, -(-COLD_BYTES // max(int(working_set_bytes), 1)) + 1)))


def copy_bandwidth_cold(nbytes, dev):
    """copy_bandwidth on MALL-cold operands: the same copy rotating over disjoint buffer pairs that add up to
    >= COLD_BYTES (GB/s, read + write counted).  Working sets beyond the cache are cold anyway."""
    half = max(int(nbytes) // 2 // 16 * 16, 1 << 16)
    if 2 * half >= COLD_BYTES:
        return copy_bandwidth(nbytes, dev)
    K = cold_copies(2 * half)
    bufs = [(torch.zeros(half, dtype=torch.uint8, device=dev), torch.empty(half, dtype=torch.uint8, device=dev))
            for _ in range(K)]
    t = time_rotation([(lambda s=s_, d=d_: d.copy_(s)) for s_, d_ in bufs])
    return 2 * half / t / 1e9


def cold_fields(alg_bytes, t_cold, dev, copies):
    """the MALL-cold companions of a roofline entry"""
    c = copy_bandwidth_cold(alg_bytes, dev)
    return {"avg_launch_us_cold": t_cold * 1e6, "achieved_GBs_cold": alg_bytes / t_cold / 1e9,
            "frac_cold": alg_bytes / t_cold / 1e9 / HBM_PEAK_GBS, "copy_GBs_cold": c,
            "frac_of_copy_cold": alg_bytes / t_cold / 1e9 / c, "cold_copies": copies}


def copy_bandwidth(nbytes, dev):
    """measured device-copy rate (GB/s, read + write counted) of a copy that moves ``nbytes`` in total: SURVEY 8(d)'s
    second denominator -- what the memory system delivers to the simplest possible kernel at this size"""
    half = max(int(nbytes) // 2 // 16 * 16, 1 << 16)
    src = torch.empty(half, dtype=torch.uint8, device=dev)
    dst = torch.empty(half, dtype=torch.uint8, device=dev)
    src.zero_()
    t = time_launches(lambda: dst.copy_(src), iters=50 if nbytes < 1e9 else 10, warmup=5)
    return 2 * half / t / 1e9


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_spmm_baseline(src, dst, n, widths, seconds):
    """the SpMM metric on the host cores -- oracle/spmm_ref.c (OpenMP row-parallel CSR traversal, the
    algorithm of DGL's CPU backend) and torch.sparse.mm on CSR, forward (A H) and backward (A^T dM) separately,
    all cores and one thread, median of >= 10 runs after 3 warm-ups (bounded by `seconds` per entry)."""
    from oracle import c_oracle, gae_oracle as O
    ip, ix = O.csr_from_coo(src, dst, n)
    tp, tx = O.csc_from_coo(src, dst, n)
    E = len(ix)
    rng = np.random.default_rng(0)
    out = {"cpu_model": cpu_model_name(), "host_cpu_count": os.cpu_count(), "edges": E, "entries": []}
    all_threads = torch.get_num_threads()

    def median_time(fn, budget):
        for _ in range(3):
            fn()
        ts = []
        t_end = time.perf_counter() + budget
        while len(ts) < 10 or (time.perf_counter() < t_end and len(ts) < 50):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            if len(ts) >= 10 and time.perf_counter() >= t_end:
                break
        return float(np.median(ts)), len(ts)

    def sparse(ptr, idx):
        return torch.sparse_csr_tensor(torch.as_tensor(ptr.astype(np.int64)), torch.as_tensor(idx.astype(np.int64)),
                                       torch.ones(len(idx)), size=(n, n))

    A, At = sparse(ip, ix), sparse(tp, tx)
    for F in widths:
        H = rng.random((n, F), dtype=np.float32)
        Ht = torch.from_numpy(H)
        Mo = np.zeros((n, F), dtype=np.float32)        # preallocated and touched: no page faults in the timed calls
        for direction, (p_, x_, S) in (("fwd A*H", (ip, ix, A)), ("bwd A^T*dM", (tp, tx, At))):
            for threads in (all_threads, 1):
                os.environ["OMP_NUM_THREADS"] = str(threads)
                torch.set_num_threads(threads)
                c_oracle.set_num_threads(threads)
                tc, kc = median_time(lambda: c_oracle.spmm_csr(p_, x_, H, out=Mo), seconds)
                tt, kt = median_time(lambda: torch.sparse.mm(S, Ht), seconds)
                out["entries"].append({"F": F, "direction": direction, "threads": threads,
                                       "openmp_c_edges_per_s": E / tc, "openmp_c_ms": tc * 1e3, "openmp_c_runs": kc,
                                       "torch_sparse_mm_edges_per_s": E / tt, "torch_sparse_mm_ms": tt * 1e3,
                                       "torch_sparse_mm_runs": kt})
    torch.set_num_threads(all_threads)
    c_oracle.set_num_threads(all_threads)
    return out


def make_adam(params, lr, args, capturable=False):
    """train_inductive.py:40 / train_transductive.py:43: Adam with the reference's hyper-parameters"""
    if args.torch_adam:
        return torch.optim.Adam(params, lr=lr, fused=True, capturable=capturable), "torch.optim.Adam(fused)"
    from gae_dgl_amd.optim import Adam
    return Adam(params, lr=lr), "gae_adam_step (torch.optim.Adam rule, one launch)"


def spmm_probe(indptr, indices, n, F, ld=None, plan=None, label="", iters=50, blockdiag=None, plan_factory=None):
    """kernel-only SpMM throughput for one shape (row-padded operands as every op of the package produces).
    ``plan_factory(indptr, indices)`` -> plan: also time the launch MALL-cold, rotating over disjoint copies of ALL its
    operands (CSR arrays, plan tables, H, M) that add up to >= COLD_BYTES; launches whose operands exceed the 256 MB
    Infinity Cache anyway report their one figure as both"""
    from gae_dgl_amd import ops, workloads as W
    ld = ld or F
    H = torch.rand(n, ld, device=indptr.device)[:, :F]
    out = torch.empty(n, ld, device=indptr.device)[:, :F]
    # 30 warm-up launches: the probes start on a GPU that idled during the host-side graph generation, and
    # the first launches after an idle period run at ramping clocks
    t = time_launches(lambda: ops.spmm_raw(indptr, indices, H, n, out=out, plan=plan, blockdiag=blockdiag,
                                           out_padded=True), iters=iters, warmup=30)
    nnz = int(indices.numel())
    b = W.spmm_alg_bytes(n, n, nnz, F, 4)
    r = {"shape": label, "n": n, "nnz": nnz, "F": F, "ld": ld, "dtype": "float32", "us_per_launch": t * 1e6,
         "edges_per_s": nnz / t, "alg_bytes": b, "achieved_GBs": b / t / 1e9,
         "frac_hbm_peak": b / t / 1e9 / HBM_PEAK_GBS}
    if b >= COLD_BYTES // 2:
        r.update({"frac_cold": r["frac_hbm_peak"], "cold_note": "operands exceed the 256 MB Infinity Cache: every launch reads DRAM"})
    elif plan_factory is not None:
        K = cold_copies(b)
        fns = []
        for _ in range(K):
            ip_, ix_ = indptr.clone(), indices.clone()
            pl_ = plan_factory(ip_, ix_)
            H_ = torch.rand(n, ld, device=indptr.device)[:, :F]
            o_ = torch.empty(n, ld, device=indptr.device)[:, :F]
            fns.append(lambda ip_=ip_, ix_=ix_, pl_=pl_, H_=H_, o_=o_: ops.spmm_raw(ip_, ix_, H_, n, out=o_, plan=pl_,
                                                                                   out_padded=True))
        r.update(cold_fields(b, time_rotation(fns), indptr.device, K))
        del fns
        torch.cuda.empty_cache()
    return r


# ---------------------------------------------------------------------------------------------- workloads
# `dtype` of the line: what the arithmetic is carried out in.  Storage, the aggregations, the Linear forward and every
# accumulation are fp32; the products of the fused loss and of the weight gradients run on the bf16 matrix pipe as
# split-operand products (see config.loss_products / dW_products; `value_exact_fp32` is the same step with exact
# fp32 products everywhere)
DTYPE_SPLIT = ("f32 (storage, aggregations, accumulators; the dense products run on the 16-bit matrix pipe from split operands with fp32 accumulation: X W^T and dW from three bf16 pieces per operand / six pairs, the loss from two fp16 pieces (N >= 5120) or three bf16 pieces -- all fp32-grade, see ms_per_step_exact_fp32 for fp32 MFMAs everywhere)")


class CitationWorkload:
    def __init__(self, name, args, dev):
        import gae_dgl_amd as G
        from gae_dgl_amd import ops, workloads as W
        self.args, self.dev = args, dev
        n, src, dst, X = W.citation_graph(name, seed=0, degrees=getattr(args, "degrees", "uniform"))
        self.n, self.src, self.dst, self.X = n, src, dst, X
        self.F_in, self.hidden = X.shape[1], [32, 16]
        torch.manual_seed(0)
        self.model = G.GAE(self.F_in, self.hidden).to(dev)
        self.use_graph = (not args.no_hipgraph) and args.loss == "fused"
        self.opt, opt_name = make_adam(self.model.parameters(), 1e-2, args, self.use_graph)  # train_transductive.py:43
        self.g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
        self.Xd = ops.pad_rows(torch.from_numpy(X).to(dev))                  # rows padded to whole 128-B lines
        feats = getattr(args, "features", "auto")
        if feats == "sparse":
            self.Xd = G.SparseFeatures.from_dense(self.Xd)                   # once: X is constant across epochs
        elif feats == "auto" and args.layer1 == "transform-first":
            self.Xd = G.SparseFeatures.maybe_from_dense(self.Xd, self.hidden[0], graph=self.g)    # (train_transductive.py:37-38: loaded once)
        self.sparse = isinstance(self.Xd, G.SparseFeatures)
        self.g.csr(); self.g.csc(); self.g.spmm_plan(False); self.g.spmm_plan(True); self.g.scattered()   # static
        E = self.g.number_of_edges()
        self.edges_per_step = E * (2 * len(self.hidden) - 1)                   # L fwd + (L-1) bwd SpMM launches
        self.meta = {"workload": f"{name}-transductive-gae" + ("" if getattr(args, "degrees", "uniform") == "uniform" else "[planetoid degrees]"),
                     "longest_row": int(np.bincount(dst, minlength=n).max()), "n_nodes": n, "n_edges": E, "in_dim": self.F_in,
                     "hidden_dims": self.hidden, "norm": "none", "loss": args.loss + "-bce",
                     "optimizer": "adam lr=1e-2: " + opt_name, "parallelism": "1 GPU",
                     "launch": "hipGraph replay of the captured step" if self.use_graph else "eager",
                     "forward_products": "layer 1 X W^T: three bf16 pieces per operand, six pairs (knob xw_p3=1: 1.5e-7 of fp64, as the fp32 MFMAs it replaces); narrow layers: fp32 MFMA",
                     "loss_products": "split-operand products on K = 32 MFMAs, fp32 accumulate (knobs bce_s_bf16=3, bce_pv_bf16=1): N >= 5120 (symmetric kernel, 256-row panels; balanced schedule below 65 536 rows): two fp16 pieces per operand for S = Z Z^T and dZ = P Z (22 mantissa bits; embeddings beyond |z| = 32768 fall back to the bf16 form inside the same call); N < 5120: three bf16 pieces for S (24 bits), two for P Z",
                     "dW_products": "three bf16 pieces per operand, six piece pairs down to 2^-24, fp32 accumulate (knob atb_bf16=1; gae_xw_wgrad: exact fp32 MFMAs)",
                     "residency": f"operands of the dominant launch ({2 * n * self.F_in * 4 / 1e6:.0f} MB) stay in the "
                                  "256 MB Infinity Cache across the timed replays: its '% of 8 TB/s' is measured "
                                  "against the on-die fabric, not DRAM"}
        self.captured = None
        self.dtype = DTYPE_SPLIT if args.loss == "fused" else "f32"
        self.tf = args.layer1 == "transform-first"
        J = self.hidden[0]
        if self.sparse:
            nnz = self.Xd.nnz
            self.dominant = ("spx_fwd", n, self.F_in, J, nnz)
            self.dominant_desc = (f"spx_fwd P = X W^T from the {nnz} non-zeros of X ({100.0 * nnz / (n * self.F_in):.1f} % "
                                  f"of {n} x {self.F_in}; X is constant across steps and was compressed once at set-up)")
            self.alg_bytes = 8 * nnz + 4 * (n + 1) + 4 * (n * J + J * self.F_in)
            self.pmc_key = ""
            self.meta["layer1"] = ("act(A (X W^T) + b) with X W^T and dW = G^T X from the non-zeros of the constant input "
                                   "features (gae_dgl_amd.SparseFeatures, compressed once at set-up: --features "
                                   f"{feats}; gae_spx_fwd / gae_spx_wgrad; same values as the dense kernels)")
        elif self.tf:
            # the step's HBM-dominant launch is the dense pass over X (gae_xw_fwd); its compulsory bytes: X + P + W
            self.dominant = ("xw_fwd", n, self.F_in, J, "torch.float32")
            self.dominant_desc = (f"xw_fwd P = X W^T, {n} x {self.F_in} -> {J} (layer 1 in transform-first order: the "
                                  f"aggregation runs at F = {J}; X read once, W stationary in registers)")
            self.alg_bytes = 4 * (n * self.F_in + n * J + J * self.F_in)
            self.pmc_key = f"{name}-xw_fwd"
            self.meta["layer1"] = ("act(A (X W^T) + b): gae_xw_fwd, gae_spmm_csr_epilogue; backward gae_spmm_csr_epilogue "
                                   "(ReLU gate in the gather), gae_xw_wgrad -- value of gae.py:26-31 up to fp32 rounding, "
                                   "4 SpMM launches per step at F <= 32 (3 counted in `value`: the reference's L + (L - 1))")
        else:
            self.dominant = ("spmm", n, n, self.F_in, "torch.float32")
            self.dominant_desc = f"spmm F={self.F_in} (layer-1 aggregation A*X, {n} rows, {E} edges)"
            self.alg_bytes = W.spmm_alg_bytes(n, n, E, self.F_in, 4)
            self.pmc_key = f"{name}-F{self.F_in}"
            self.meta["layer1"] = "act((A X) W^T + b), the reference's order (gae.py:26-31)"
        self.scaling = "weak"

    def dominant_launch(self):
        """the step's dominant HBM launch on its real operands: X W^T (transform-first) or the aggregation A X"""
        from gae_dgl_amd import ops
        if self.sparse:
            W1 = self.model.layers[0].apply_mod.linear.weight.detach()
            return lambda: ops.spx_fwd_raw(self.Xd, W1)
        if self.tf:
            W1 = self.model.layers[0].apply_mod.linear.weight.detach()
            return lambda: ops.xw_fwd_raw(self.Xd, W1, None, 0, keep_splits=True)      # as the step launches it
        ip, ix = self.g.csr()
        out = ops.pad_rows(torch.empty(self.Xd.shape, device=self.dev))
        plan = self.g.spmm_plan(False)
        sc = self.Xd.shape[1] > ops.TILE_MIN_F and self.g.scattered(self.Xd.shape[1] * 4)
        return lambda: ops.spmm_raw(ip, ix, self.Xd, self.n, out=out, plan=plan, out_padded=True, scattered=sc)

    def aggregation_launch(self):
        """the DEFAULT step's layer-1 aggregation on its real operands: act(A P + b) at F = 32 with P = X W^T as
        gae_xw_fwd leaves it (gae_spmm_csr_epilogue) -- the SpMM the step actually runs (SURVEY 8(d)'s unit)"""
        from gae_dgl_amd import ops
        lin = self.model.layers[0].apply_mod.linear
        W1, b1 = lin.weight.detach(), lin.bias.detach()
        P, _ = ops.xw_fwd_raw(self.Xd, W1, None, 0, keep_splits=True)
        ip, ix = self.g.csr()
        plan = self.g.spmm_plan(False)
        return lambda: ops.spmm_epilogue_raw(ip, ix, P, self.n, plan, b1, 1)

    def layer1_launches(self, fresh=False):
        """the three layer-1 launches of the DEFAULT step on operands shaped like the step's own -- 'xw_fwd' (P = X W^T),
        'agg' (relu(A P + b), gae_spmm_csr_epilogue at F = 32) and 'xtg' (dW1 = G^T X, db1: gae_xw_wgrad) -- as closures.
        ``fresh``: on private copies of EVERY operand (graph arrays, plan tables, X, P, G ...): one member of a
        MALL-cold rotation"""
        import gae_dgl_amd as G
        from gae_dgl_amd import ops
        lin = self.model.layers[0].apply_mod.linear
        W1, b1 = lin.weight.detach(), lin.bias.detach()
        g, Xd = self.g, self.Xd
        if fresh:
            g = G.DGLGraph((self.src, self.dst), num_nodes=self.n).to(self.dev)
            Xd = ops.pad_rows(torch.from_numpy(self.X).to(self.dev))
            W1, b1 = W1.clone(), b1.clone()
        ip, ix = g.csr()
        plan = g.spmm_plan(False)
        P, _ = ops.xw_fwd_raw(Xd, W1, None, 0, keep_splits=True)
        J = self.hidden[0]
        Gm = torch.randn(self.n, J, device=self.dev)          # G = gate(A^T dM2), dY and the ReLU mask Y of the backward
        D = torch.randn(self.n, J, device=self.dev)
        Y = torch.relu(torch.randn(self.n, J, device=self.dev))

        def xtg():
            # as the captured step launches it: per-block partial sums only (gae_x_xw_wgrad_partials) -- their reduction
            # rides in the Adam launch (ops.StepContext) -- so this closure times the one-pass kernel alone
            with ops.StepContext(defer_grads=True) as c:
                out = ops.xw_wgrad_raw(Xd, Gm, None, D, Y, J)
                c.partials.clear()
            return out
        return {"xw_fwd": lambda: ops.xw_fwd_raw(Xd, W1, None, 0, keep_splits=True),
                "agg": lambda: ops.spmm_epilogue_raw(ip, ix, P, self.n, plan, b1, 1),
                "xtg": xtg}

    def capture(self):
        from gae_dgl_amd.capture import CapturedTrainStep
        self.captured = CapturedTrainStep(self.model, self.opt, self.g, self.Xd)

    def loss_launch(self):
        """the step's dominant launch sequence -- the fused decoder + BCE (loss and dZ: prepare, dense, edges,
        finalize) -- on an embedding of the trained model"""
        from gae_dgl_amd import ops
        self.g.ndata['h'] = self.Xd
        with torch.no_grad():
            Z = self.model.encode(self.g).clone()
        mask = ops.dropout_mask(tuple(Z.shape), 0.1, seed=1, device=self.dev)
        csr, csc = self.g.csr(), self.g.csc()
        E = self.g.number_of_edges()
        pw = (float(self.n) ** 2 - E) / E
        return lambda: ops.decoder_bce_raw(Z, mask, csr, csc, pw, want_grad=True)

    def step(self):
        if self.captured is not None:
            return self.captured()
        g, model = self.g, self.model
        g.ndata['h'] = self.Xd
        if self.args.loss == "fused":
            loss = model.reconstruction_loss(g)                               # same quantity, fused HIP kernel
        else:
            adj = g.dense_adjacency()                                         # train_transductive.py:59
            pw = (self.n * self.n - adj.sum()) / adj.sum()                    # :60
            loss = torch.nn.functional.binary_cross_entropy_with_logits(model(g), adj, pos_weight=pw)
        from gae_dgl_amd import ops
        self.opt.zero_grad(); ops.backward(loss); self.opt.step()
        return loss

    def cpu_baseline(self, seconds):
        from oracle import gae_oracle as O
        ref = O.CpuReferenceStep(self.src, self.dst, self.n, self.X, self.F_in, self.hidden, lr=1e-2, seed=0)
        ref.step()  # warm-up
        t0 = time.perf_counter(); k = 0
        while True:
            ref.step(); k += 1
            el = time.perf_counter() - t0
            if el >= seconds or k >= 50:
                break
        return {"value": self.edges_per_step * k / el, "unit": "edges/s", "cores": torch.get_num_threads(),
                "kind": "port", "ms_per_step": el / k * 1e3,
                "sample": f"{k} full training steps of the same workload by oracle/gae_oracle.py:CpuReferenceStep "
                          f"(dense N x N label/logits/BCE as train_inductive.py:44-52, torch CPU, "
                          f"{torch.get_num_threads()} threads, host cpu_count={os.cpu_count()})"}


class VgaeWorkload(CitationWorkload):
    """BASELINE config 5: VGAE (mu / log sigma heads, sampled Z Z^T decoder, BCE + KL) on Citeseer with bf16 feature
    storage (fp32 accumulation everywhere)"""

    def __init__(self, args, dev):
        import gae_dgl_amd as G
        from gae_dgl_amd import ops, workloads as W
        from gae_dgl_amd.vgae import VGAE
        self.args, self.dev = args, dev
        n, src, dst, X = W.citation_graph("citeseer", seed=0)
        self.n, self.src, self.dst, self.X = n, src, dst, X
        self.F_in, self.hidden = X.shape[1], [32, 16]
        torch.manual_seed(0)
        self.model = VGAE(self.F_in, self.hidden, seed=11).to(dev)
        self.use_graph = not args.no_hipgraph
        self.opt, opt_name = make_adam(self.model.parameters(), 1e-2, args, self.use_graph)
        self.g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
        self.Xd = ops.pad_rows(torch.from_numpy(X).to(dev).to(torch.bfloat16))     # rows of whole 128-byte lines
        self.g.csr(); self.g.csc(); self.g.spmm_plan(False); self.g.spmm_plan(True); self.g.scattered()
        E = self.g.number_of_edges()
        self.edges_per_step = E * 5        # A X, A H (mu), A H (log sigma) forward; A^T dM of the two heads backward
        self.meta = {"workload": "citeseer-vgae-bf16-features", "n_nodes": n, "n_edges": E, "in_dim": self.F_in,
                     "hidden_dims": self.hidden, "heads": "mu, log sigma (identity GCN heads on the shared ReLU layer)",
                     "loss": "fused BCE (sampled z) + KL", "optimizer": "adam lr=1e-2: " + opt_name,
                     "parallelism": "1 GPU",
                     "launch": "hipGraph replay of the captured step (the noise draw advances a device counter)"
                               if self.use_graph else "eager",
                     "feature_storage": "bf16 X (layer-1 aggregation: bf16 rows, fp32 accumulate, bf16 M); "
                                        "everything after the first Linear in fp32"}
        self.captured = None
        self.tf = args.layer1 == "transform-first"
        J = self.hidden[0]
        if self.tf:
            self.dominant = ("xw_fwd", n, self.F_in, J, "torch.bfloat16")
            self.dominant_desc = (f"xw_fwd P = X W^T, bf16-stored X {n} x {self.F_in} -> {J} (v_mfma_f32_16x16x32_bf16 on the "
                                  f"rows as stored, W = hi + lo bf16 fragments, fp32 accumulation)")
            self.alg_bytes = 2 * n * self.F_in + 4 * (n * J + J * self.F_in)
            self.pmc_key = "citeseer-bf16-xw_fwd"
            self.meta["feature_storage"] = ("bf16 X read once per direction by gae_xw_fwd / gae_xw_wgrad (shared layer in "
                                            "transform-first order: no bf16 aggregate, no fp32 copy of it); everything after "
                                            "in fp32")
        else:
            self.dominant = ("spmm", n, n, self.F_in, "torch.bfloat16")
            self.dominant_desc = f"spmm F={self.F_in}, bf16 storage (layer-1 aggregation A*X, {n} rows, {E} edges)"
            self.alg_bytes = W.spmm_alg_bytes(n, n, E, self.F_in, 2)
            self.pmc_key = "citeseer-bf16-F3703"
        self.scaling = "weak"
        self.dtype = "bf16 feature storage; " + DTYPE_SPLIT
        self._params = list(self.model.parameters())

    def dominant_launch(self):
        from gae_dgl_amd import ops
        if self.tf:
            W1 = self.model.shared.apply_mod.linear.weight.detach()
            return lambda: ops.xw_fwd_raw(self.Xd, W1, None, 0, keep_splits=True)      # as the step launches it
        ip, ix = self.g.csr()
        out = ops.pad_rows(torch.empty(self.Xd.shape, dtype=self.Xd.dtype, device=self.dev))
        plan = self.g.spmm_plan(False)
        sc = self.Xd.shape[1] > ops.TILE_MIN_F and self.g.scattered(self.Xd.shape[1] * 2)
        return lambda: ops.spmm_raw(ip, ix, self.Xd, self.n, out=out, plan=plan, out_padded=True, scattered=sc)

    def capture(self):
        from gae_dgl_amd.capture import CapturedTrainStep
        # tensors of the last eager step (model.last, the z left in ndata) keep that step's autograd graph alive,
        # and its AccumulateGrad nodes are bound to the eager stream: drop them before the capture
        self.model.last = {}
        self.g.ndata['h'] = self.Xd
        self.opt.zero_grad(set_to_none=True)
        self.captured = CapturedTrainStep(self.model, self.opt, self.g, self.Xd, loss_fn=lambda m, g: m.loss(g),
                                          defer_loss=True)

    loss_launch = None

    def step(self):
        if self.captured is not None:
            return self.captured()
        from gae_dgl_amd import ops
        self.g.ndata['h'] = self.Xd
        loss = self.model.loss(self.g)
        self.opt.zero_grad(); ops.backward(loss, self._params); self.opt.step()
        self.model.last = {}
        return loss.detach()

    def cpu_baseline(self, seconds):
        """the oracle's VGAE restatement (oracle/gae_oracle.py:vgae_forward + vgae_kl, dense N x N BCE) forward and
        backward on the host cores, same sizes"""
        from oracle import gae_oracle as O
        Xo = torch.from_numpy(self.X)
        P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in self.model.named_parameters()}
        ip, ix = O.csr_from_coo(self.src, self.dst, self.n)
        adj = O.dense_adjacency(self.src, self.dst, self.n)
        pw = O.pos_weight_of(adj)
        eps = torch.randn(self.n, self.hidden[-1])

        def one():
            mu, ls, z = O.vgae_forward(ip, ix, Xo, P["shared.apply_mod.linear.weight"], P["shared.apply_mod.linear.bias"],
                                       P["mu_head.apply_mod.linear.weight"], P["mu_head.apply_mod.linear.bias"],
                                       P["logstd_head.apply_mod.linear.weight"], P["logstd_head.apply_mod.linear.bias"],
                                       eps)
            loss = O.bce_with_logits_mean(z @ z.t(), adj, pw) + O.vgae_kl(mu, ls)
            for v in P.values():
                v.grad = None
            loss.backward()
        one()
        t0 = time.perf_counter(); k = 0
        while True:
            one(); k += 1
            el = time.perf_counter() - t0
            if el >= seconds or k >= 50:
                break
        return {"value": self.edges_per_step * k / el, "unit": "edges/s", "cores": torch.get_num_threads(),
                "kind": "port", "ms_per_step": el / k * 1e3,
                "sample": f"{k} forward + backward passes of the same VGAE step by oracle/gae_oracle.py (fp32, dense "
                          f"N x N BCE, no optimizer update; torch CPU, {torch.get_num_threads()} threads)"}


class ZincWorkload:
    """inductive step, batch of B molecules gathered on the device (train_inductive.py:31-53)"""

    def __init__(self, args, dev, n_graphs=None, rank=0, world=1, group=None):
        import gae_dgl_amd as G
        from gae_dgl_amd import workloads as W
        from gae_dgl_amd.dataset import DeviceGraphDataset
        self.args, self.dev = args, dev
        self.rank, self.world, self.group = rank, world, group
        B = args.batch_graphs
        n_graphs = n_graphs or max(4 * B, 32768)
        self.ds = DeviceGraphDataset.synthetic_zinc(n_graphs, seed=0, device=dev)
        self.B = B
        torch.manual_seed(0)
        self.model = G.GAE(39, [32, 16]).to(dev)
        self.use_graph = (not args.no_hipgraph) and args.loss == "fused"
        self.opt, opt_name = make_adam(self.model.parameters(), 1e-3, args, self.use_graph)   # train_inductive.py:25
        self.runner = None
        self.rng = np.random.default_rng(0)
        self.perm = self.rng.permutation(n_graphs)
        if world > 1:          # data-parallel replicas: every rank holds the dataset and trains on its share of the order
            from gae_dgl_amd.dataset import shard_order
            self.perm = shard_order(self.perm, rank, world)
            self.use_graph = self.use_graph and not args.oversubscribe     # (staged collectives cannot be captured)
        self.d_perm = torch.from_numpy(self.perm).to(dev)      # the epoch order lives on the device (ds.epoch())

        self.cursor = 0
        nb = int(self.ds.sizes_host[:B].sum()); eb = int(self.ds.edges_host[:B].sum())
        self.edges_per_step = 3 * eb * world                                  # nominal (batch 0; all replicas); varies < 1 % per batch
        self.meta = {"workload": "zinc250k-inductive-gae", "batch_graphs": B, "nodes_per_batch~": nb,
                     "edges_per_batch~": eb, "in_dim": 39, "hidden_dims": [32, 16], "loss": "fused-bce",
                     "optimizer": "adam lr=1e-3: " + opt_name, "dataset_graphs": n_graphs,
                     "parallelism": "1 GPU" if world == 1 else
                                    f"data-parallel x{world}: every replica trains on its share of the epoch order, ONE "
                                    f"all-reduce of the 1 808 parameter gradients per step (averaged; inside the "
                                    f"captured step) -- train_inductive.py:84-96 on {world} GPUs, global batch "
                                    f"{B * world} molecules",
                     "launch": "hipGraph replay per batch (collate + step on a fixed-capacity batch, "
                               "capture.CapturedInductiveStep)" if self.use_graph else "eager",
                     "batches_per_epoch_at_239455_graphs": int(np.ceil(239455 / B))}
        self.dominant = None
        self.dtype = DTYPE_SPLIT
        self.pmc_key = "zinc-batch4096-F39" if B == 4096 else ""
        self.W = W
        self.scaling = "weak"
        self._edges_done = 0

    def dominant_launch(self):
        from gae_dgl_amd import ops
        bg = self.ds.batch(self.perm[:self.B])
        ip, ix = bg.csr()
        H = bg.ndata['h']
        out = ops.pad_rows(torch.empty(H.shape, device=self.dev))
        nb, eb = bg.number_of_nodes(), bg.number_of_edges()
        self.alg_bytes = 4 * (nb + 1) + 4 * eb + 2 * 4 * 39 * nb
        self.dominant_desc = f"spmm F=39 (layer-1 aggregation of a {self.B}-molecule batch, {nb} rows, {eb} edges)"
        return lambda: ops.spmm_raw(ip, ix, H, bg.number_of_nodes(), out=out, out_padded=True,
                                    blockdiag=bg.block_diag, plan=bg.spmm_plan(False))

    def loss_launch(self):
        """the step's dominant launch sequence -- fused decoder + BCE of one batch (loss and dZ) -- on the embedding of
        batch 0"""
        from gae_dgl_amd import ops
        bg = self.ds.batch(self.perm[:self.B])
        with torch.no_grad():
            Z = self.model.encode(bg).clone()
        self.n = bg.number_of_nodes()
        mask = ops.dropout_mask(tuple(Z.shape), 0.1, seed=1, device=self.dev)
        csr, csc = bg.csr(), bg.csc()
        E = bg.number_of_edges()
        pw = (float(self.n) ** 2 - E) / E
        return lambda: ops.decoder_bce_raw(Z, mask, csr, csc, pw, want_grad=True)

    def capture(self):
        from gae_dgl_amd.capture import CapturedInductiveStep
        self.runner = CapturedInductiveStep(self.model, self.opt, self.ds, self.B, group=self.group,
                                            replicas=self.world > 1)
        self._left = 0
        self.meta["capacity"] = None

    def step(self):
        if self.runner is not None:
            if self._left == 0:                     # next epoch: upload the order again, reset the device cursor
                self._left = self.runner.begin_epoch(self.perm)
                self.meta["capacity"] = {"nodes": self.runner.cap_nodes, "edges": self.runner.cap_edges}
            self._left -= 1
            return self.runner.step()
        ids = self.perm[self.cursor:self.cursor + self.B]
        self._lo = self.cursor
        self.cursor = (self.cursor + self.B) % (len(self.perm) - self.B)
        bg = self.ds._assemble(self.d_perm[self._lo:self._lo + self.B], ids)   # dgl.batch on the device (K10), no H2D
        loss = self.model.reconstruction_loss(bg)
        from gae_dgl_amd import ops
        self.opt.zero_grad(); ops.backward(loss)
        if self.world > 1:
            from gae_dgl_amd.parallel import allreduce_grads
            allreduce_grads(list(self.model.parameters()), self.group, average=True)
        self.opt.step()
        return loss


class RmatShardedWorkload:
    def __init__(self, args, dev, rank, world, group):
        import gae_dgl_amd as G
        from gae_dgl_amd import workloads as W
        from gae_dgl_amd.parallel import ShardedGraph
        self.args, self.dev, self.rank, self.world, self.group = args, dev, rank, world, group
        scale = args.rmat_scale
        n = 1 << scale
        # every rank generates ITS slice of the edge list (1 / world of the chunks); the row blocks are assembled from
        # the slices (degree histogram all-reduce + one all-to-all-v of edges per direction): no rank holds the list
        src, dst = W.rmat_edges(scale, 16, seed=0, device=dev, part=(rank, world))
        E = 16 << scale
        self.overlap = not args.no_overlap
        # auto = the one-pass encoder (parallel.ShardedEncoder2Function): last layer as A (H W^T) + b, dense halves in two
        # one-pass kernels; "aggregate-first" keeps (A H) W^T for every layer (gae.py:26-31 literally)
        self.transform_first = args.layer_order in ("transform-first", "auto")
        self.sg = ShardedGraph.from_edge_slice(n, src, dst, group=group, mode=args.exchange, device=dev,
                                               balance=args.balance, overlap=self.overlap)
        del src, dst
        self.sg.cache_constant_inputs = not args.no_cache_input_exchange
        p = self.sg.part
        e_local = int(p.fwd_rows.numel())
        # structure + plans, built once: their cost and size are part of the line (one-off, outside `value`)
        torch.cuda.synchronize(); t_csr = time.perf_counter()
        for w in ("fwd", "bwd"):
            for part in (("own", "remote") if self.overlap else (None,)):
                self.sg.csr(w, part)
        torch.cuda.synchronize(); t_plan = time.perf_counter()
        plan_bytes = 0
        for w in ("fwd", "bwd"):
            for part in (("own", "remote") if self.overlap else (None,)):
                pl = self.sg.plan(w, part)
                if pl is not None:
                    plan_bytes += sum(t.numel() * t.element_size() for t in pl.tensors if t is not None)
        torch.cuda.synchronize()
        self.plan_build_ms = (time.perf_counter() - t_plan) * 1e3
        self.csr_build_ms = (t_plan - t_csr) * 1e3
        self.plan_bytes = plan_bytes
        # the edge lists of the plan are not needed once the device CSRs exist (2 x 2^28 int64 per direction)
        p.fwd_rows = p.fwd_cols = p.bwd_rows = p.bwd_cols = None
        p.cols_global = {}
        for k in p.split:
            p.split[k] = dict(n_remote_cols=p.split[k]["n_remote_cols"])
        torch.cuda.empty_cache()
        F, hidden = 32, [32, 16]
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.X = torch.rand(p.n_local, F, device=dev, generator=gen)
        self.dZ = torch.randn(p.n_local, hidden[-1], device=dev, generator=gen) / n
        torch.manual_seed(0)
        self.model = G.GAE(F, hidden).to(dev)
        self.opt, opt_name = make_adam(self.model.parameters(), 1e-2, args)
        self.params = list(self.model.parameters())
        self.n, self.E = n, E
        self.edges_per_step = 3 * E
        self.meta = {"workload": f"rmat-s{scale}-ef16-row-sharded-encoder", "n_nodes": n, "n_edges": E, "in_dim": F,
                     "hidden_dims": hidden, "parallelism": f"row-shard x{world}", "exchange": args.exchange,
                     "balance": args.balance,
                     "overlap": "own-column SpMM under the exchange, then M += remote-column SpMM" if self.overlap
                                else "exchange, then one SpMM (bit-identical to one GPU)",
                     "exchange_bytes_per_spmm_per_rank": self.sg.exchange_bytes(F),
                     "exchange_bytes_per_step_per_rank":
                         ((0 if self.sg.cache_constant_inputs else 1) * self.sg.exchange_bytes(F)
                          + 2 * self.sg.exchange_bytes(16 if self.transform_first else F)),
                     "input_exchange": "the remote rows of the constant input features X travel once, before the "
                                       "timed steps (inputs resident); A X itself is evaluated in every step"
                                       if self.sg.cache_constant_inputs else "every step",
                     "decoder": "excluded (O(N^2) = 2.8e14 logits at N = 2^24); synthetic dZ",
                     "layer_order": "(A H) W^T for every layer (gae.py:26-31)" if not self.transform_first else
                                    "32 -> 32 layer: (A X) W1^T; 32 -> 16 layer: A (H1 W2^T) + b2, its backward from "
                                    "G = A^T dZ: two of the three aggregations (and exchanges) run at width 16; dense "
                                    "halves: gae_linear2_fwd + gae_gcn2_bwd_dense, one pass each over the rows that have in-edges (the "
                                    "others' aggregate is zero: not written, not read; their share is a constant row / "
                                    "a rank-one term) (value of gae.py:26-31 up to fp32 rounding)",
                     "local_rows": p.n_local, "local_edges_fwd": e_local,
                     "csr_build_ms": self.csr_build_ms, "plan_build_ms": self.plan_build_ms,
                     "plan_bytes": self.plan_bytes,
                     "plan_note": "per-rank one-off costs outside `value`: device CSRs of A and A^T (csr_build_ms) and "
                                  "their gae_spmm_plan arrays (plan_build_ms, plan_bytes: heavy-row segments and "
                                  "descriptors, hot-column tags, XCD-pinned virtual CSR)"}
        self.pmc_key = f"rmat-s{scale}-F32" if world == 1 else ""
        self.dominant = None
        self.dominant_desc = (f"spmm F=32 on this rank's row block ({p.n_local} rows, {e_local} edges, skew plan"
                              + ("; own-column + remote-column launches" if self.overlap else "") + ")")
        # compulsory bytes of the local product: local indptr/indices + the referenced H + the local output
        self.alg_bytes = 4 * (p.n_local + 1) + 4 * e_local + 4 * F * min(n, p.n_cols["fwd"]) + 4 * F * p.n_local
        self.scaling = "strong"

    def dominant_launch(self):
        """this rank's forward product on already-exchanged rows (no collective inside the timed launches)"""
        from gae_dgl_amd import ops
        sg, n_local = self.sg, self.sg.part.n_local
        out = torch.empty(n_local, self.X.shape[1], device=self.dev)
        if not self.overlap:
            full = sg.exchange(self.X, "fwd")          # collective: every rank calls it
            ip, ix = sg.csr("fwd")
            plan = sg.plan("fwd")
            return lambda: ops.spmm_raw(ip, ix, full, n_local, out=out, plan=plan)
        recv, wait = sg.exchange_start(self.X, "fwd")
        wait()
        oip, oix = sg.csr("fwd", "own"); rip, rix = sg.csr("fwd", "remote")
        po, pr = sg.plan("fwd", "own"), sg.plan("fwd", "remote")

        def launch():
            ops.spmm_raw(oip, oix, self.X, n_local, out=out, plan=po)
            if rix.numel():
                ops.spmm_raw(rip, rix, recv, n_local, out=out, plan=pr, accumulate=True)
        return launch

    def comm_profile(self, steps=5):
        """HIP-event times of the exchange and SpMM parts over a few extra steps (outside the timed region)"""
        self.sg.timers = {}
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        t = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) * 1e-3 for k, v in self.sg.timers.items()}
        calls = {k: len(v) // steps for k, v in self.sg.timers.items()}
        self.sg.timers = None
        return t, calls

    def step(self):
        from gae_dgl_amd.parallel import allreduce_grads, sharded_encode
        z = sharded_encode(self.model, self.sg, self.X, transform_first=self.transform_first)
        self.opt.zero_grad()
        z.backward(self.dZ)
        if self.world > 1:
            allreduce_grads(self.params, self.group)
        self.opt.step()
        return z


def citation_spmm_probe(name, dev, cold=True):
    """the reference-order layer-1 aggregation A X of a citation shape at its input width (the north-star's SpMM;
    not part of the default step, which aggregates at the output width): kernel-only, operands as
    `--layer1 reference` launches them; warm (operands resident in the Infinity Cache across the back-to-back launches)
    and MALL-cold (rotation over disjoint copies of graph, plan, X and M adding up to >= COLD_BYTES)"""
    import gae_dgl_amd as G
    from gae_dgl_amd import ops, workloads as W
    n, src, dst, X = W.citation_graph(name, seed=0)
    F = X.shape[1]
    E = int(src.size)
    b = W.spmm_alg_bytes(n, n, E, F, 4)

    def make():
        g = G.DGLGraph((src, dst), num_nodes=n).to(dev)
        ip, ix = g.csr()
        Xd = ops.pad_rows(torch.from_numpy(X).to(dev))
        out = ops.pad_rows(torch.empty(Xd.shape, device=dev))
        plan = g.spmm_plan(False)
        sc = F > ops.TILE_MIN_F and g.scattered(F * 4)
        return (lambda: ops.spmm_raw(ip, ix, Xd, n, out=out, plan=plan, out_padded=True, scattered=sc)), Xd.stride(0)
    fn, ld = make()
    t = time_launches(fn, iters=50, warmup=30)
    r = {"shape": f"{name} layer-1 aggregation A*X in the reference's order (not in the default step)", "n": n, "nnz": E,
         "F": F, "ld": ld, "dtype": "float32", "us_per_launch": t * 1e6, "edges_per_s": E / t, "alg_bytes": b,
         "achieved_GBs": b / t / 1e9, "frac_hbm_peak": b / t / 1e9 / HBM_PEAK_GBS,
         "traffic": pmc_traffic(f"{name}-F{F}")}
    if cold:
        K = cold_copies(b)
        fns = [fn] + [make()[0] for _ in range(K - 1)]
        r.update(cold_fields(b, time_rotation(fns), dev, K))
        r["edges_per_s_cold"] = E / (r["avg_launch_us_cold"] * 1e-6)
        del fns
        torch.cuda.empty_cache()
    return r


def extras(dev):
    """kernel-only SpMM numbers on the other BASELINE shapes (F quoted with each)"""
    from gae_dgl_amd import ops, workloads as W
    out = [citation_spmm_probe(name, dev) for name in ("pubmed", "cora", "citeseer")]
    gptr, src, dst, X = W.zinc_like(249455, seed=0)
    N = int(gptr[-1])
    s = torch.from_numpy(src).to(dev); d = torch.from_numpy(dst).to(dev)
    ip, ix = ops.csr_from_coo(d, s, N, N)
    nb = int(gptr[4096]); eb = int(ip[nb])
    ipb, ixb = ip[:nb + 1].clone(), ix[:eb].clone()
    # as the product launches it: the packed neighbour table of the batch comes out of the batch gather
    pb = ops.spmm_plan(ipb, indices=ixb, ell=True, ell_width=ops.ell_width_for_degrees(ipb[1:] - ipb[:-1]))
    pf = lambda ip_, ix_: ops.spmm_plan(ip_, indices=ix_, ell=True, ell_width=ops.ell_width_for_degrees(ip_[1:] - ip_[:-1]))
    out.append(spmm_probe(ipb, ixb, nb, 39, ld=40, plan=pb, label="zinc-batch4096 layer1", plan_factory=pf))
    out.append(spmm_probe(ipb, ixb, nb, 32, plan=pb, label="zinc-batch4096 layer2", plan_factory=pf))
    bd = ops.BlockDiag(gptr, dev)     # whole molecules per thread block: LDS-staged block-diagonal kernel
    out.append(spmm_probe(ip, ix, N, 39, ld=40, label="zinc-250k whole set, one launch, layer1", iters=20,
                          blockdiag=bd))
    out.append(spmm_probe(ip, ix, N, 32, label="zinc-250k whole set, one launch, layer2", iters=20, blockdiag=bd))
    del s, d, ip, ix
    torch.cuda.empty_cache()
    src, dst = W.rmat_edges(24, 16, device=dev)
    n = 1 << 24
    ip, ix = ops.csr_from_coo(dst, src, n, n)
    del src, dst
    plan = ops.spmm_plan(ip, indices=ix, ell=False, n_cols=n)
    r = spmm_probe(ip, ix, n, 32, plan=plan, label="rmat-s24-ef16 (1 GPU, skew plan)", iters=10)
    r["gather_bytes_no_reuse"] = 4 * (n + 1) + 4 * r["nnz"] + 4 * 32 * r["nnz"] + 4 * 32 * n
    r["achieved_GBs_no_reuse_model"] = r["gather_bytes_no_reuse"] / (r["us_per_launch"] * 1e-6) / 1e9
    out.append(r)
    return out


def extra_steps(args, dev):
    """whole training steps of the other single-GPU BASELINE configurations, same method as the headline line (captured
    step, barrier + synchronize around exactly `steps` replays, regions repeated to >= 0.3 s, median region), each with
    the roofline fraction of its own HBM-dominant launch (back-to-back HIP-event timing on the step's operands)"""
    out = {}
    base = argparse.Namespace(**vars(args))
    base.steps, base.warmup = 20, 5
    def hubbed(name):
        def make(a):
            a.degrees = "planetoid"        # the real graph's hubs (100 - 170 neighbours): what a user's data looks like
            return CitationWorkload(name, a, dev)
        return make
    jobs = [("cora", lambda a: CitationWorkload("cora", a, dev), None),
            ("citeseer", lambda a: CitationWorkload("citeseer", a, dev), None),
            ("pubmed_planetoid_degrees", hubbed("pubmed"), None),
            ("cora_planetoid_degrees", hubbed("cora"), None),
            ("vgae", lambda a: VgaeWorkload(a, dev), None),
            ("zinc128", lambda a: ZincWorkload(a, dev), 128),
            ("zinc4096", lambda a: ZincWorkload(a, dev), 4096)]
    for name, make, B in jobs:
        t_job = time.perf_counter()
        a = argparse.Namespace(**vars(base))
        if B is not None:
            a.batch_graphs = B
        wl = make(a)
        for _ in range(a.warmup):
            wl.step()
        if getattr(wl, "use_graph", False):
            wl.capture()
            for _ in range(a.warmup):
                wl.step()
        regions = timed_regions(wl, a, torch.cuda.synchronize, 1, dev, min_total_s=0.3, max_regions=50)
        el = float(np.median(regions)) / a.steps
        dom_fn = wl.dominant_launch()
        t_dom = time_launches(dom_fn, iters=50, warmup=30)
        r = {"workload": wl.meta["workload"], **({"longest_row": wl.meta["longest_row"]} if "longest_row" in wl.meta else {}),
             "ms_per_step": el * 1e3, "value": wl.edges_per_step / el, "unit": "edges/s",
             "launch": wl.meta.get("launch"), "timing": {"regions": len(regions), "steps_per_region": a.steps},
             "dominant": {"kernel": wl.dominant_desc, "avg_launch_us": t_dom * 1e6, "alg_bytes_per_launch": wl.alg_bytes,
                          **({"note": "a launch on the non-zeros moves 1-3 MB: it is bound by its chain of dependent round trips "
                                      "(row pointers -> ids / values -> gathered rows), not by bytes; `frac` is small by "
                                      "construction"} if getattr(wl, "sparse", False) else {}),
                          "achieved_GBs": wl.alg_bytes / t_dom / 1e9, "frac": wl.alg_bytes / t_dom / 1e9 / HBM_PEAK_GBS,
                          "traffic": pmc_traffic(getattr(wl, "pmc_key", ""))}}
        if B is not None:
            r["epoch_time_s"] = el * wl.meta["batches_per_epoch_at_239455_graphs"]
            r["batch_graphs"] = B
        r["wall_s"] = time.perf_counter() - t_job
        out[name] = r
        del wl, dom_fn
        torch.cuda.empty_cache()
    return out


class MockWorkload:
    """host-logic stand-in (tests/test_bench_launch_cpu.py): runs on the CPU over gloo so that the launcher, the
    process group, the timed regions and the JSON line can be exercised without a GPU.  Never a bench result."""

    def __init__(self, args, rank, world, group):
        import torch.distributed as dist
        self.dist, self.group, self.world = dist, group, world
        self.t = torch.ones(16) * (rank + 1)
        self.edges_per_step = 1000 * world
        self.meta = {"workload": "mock (launcher self-test, CPU/gloo; not a measurement)", "parallelism": f"x{world}"}
        self.scaling = "weak"

    def step(self):
        t = self.t.clone()
        if self.world > 1:
            self.dist.all_reduce(t, group=self.group)
        return t


def collective_preflight(rank, world, dev, group=None):
    """Every collective primitive the N-rank paths use, once, on small tensors with KNOWN answers, before any workload
    is built (VERDICT r05 #7a): a primitive that fails or returns wrong values on this node is named in the line
    (`preflight`) and on stderr instead of surfacing as a hang or a wrong edge count an hour into the run.  Returns
    {primitive: "ok" | "FAILED: ..."}.  Also switches parallel.py's RCCL-only fast paths (receive views for the
    all-to-all, grouped uneven all-gather) back to their portable forms when those fail."""
    import torch.distributed as dist
    from gae_dgl_amd import parallel, transport
    res = {}

    def run(name, fn):
        try:
            fn()
            if dev.type == "cuda":
                torch.cuda.synchronize()
            res[name] = "ok"
        except Exception as ex:      # noqa: BLE001 -- reported, never swallowed silently
            res[name] = f"FAILED: {type(ex).__name__}: {ex}"[:300]

    def all_reduce_sum():
        t = torch.full((1808,), float(rank + 1), device=dev)              # the 39 -> 32 -> 16 model's gradient bucket
        transport.all_reduce(t, group=group)
        assert float(t[0]) == world * (world + 1) / 2 and bool((t == t[0]).all())

    def all_reduce_max():
        t = torch.tensor([1000 + 13 * ((rank * 5) % world), 5000 - 7 * rank], dtype=torch.int64, device=dev)
        transport.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        assert t.tolist() == [1000 + 13 * max((r * 5) % world for r in range(world)), 5000]

    def a2a_tables():
        # rank r sends (r + q) % 3 + (q != r) rows of 16 floats to rank q: uneven, with empty pairs on the diagonal
        send_counts = [((rank + q) % 3 + 1) * (q != rank) for q in range(world)]
        recv_counts = [((q + rank) % 3 + 1) * (q != rank) for q in range(world)]
        return send_counts, recv_counts

    def all_to_all_uneven():
        sc, rc = a2a_tables()
        send = torch.cat([torch.full((c, 16), float(100 * rank + q), device=dev) for q, c in enumerate(sc)])
        recv = torch.empty(sum(rc), 16, device=dev)
        transport.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=group)
        want = torch.cat([torch.full((c, 16), float(100 * q + rank), device=dev) for q, c in enumerate(rc)])
        assert torch.equal(recv, want)
        cnt = torch.tensor(sc, dtype=torch.int64, device=dev); got = torch.empty_like(cnt)
        transport.all_to_all_single(got, cnt, group=group)                  # the count exchange of the plan builders
        assert got.tolist() == rc

    def all_to_all_views():
        sc, rc = a2a_tables()
        send = torch.cat([torch.full((c, 16), float(100 * rank + q), device=dev) for q, c in enumerate(sc)])
        full = torch.full((sum(rc) + 5, 16), -1.0, device=dev)              # receive views into one assembled buffer
        outs, ins, so, ro = [], [], 0, 0
        for q in range(world):
            ins.append(send[so:so + sc[q]]); so += sc[q]
            at = ro if q < rank else ro + 5
            outs.append(full[at:at + rc[q]]); ro += rc[q]
        dist.all_to_all(outs, ins, group=group)
        for q in range(world):
            assert bool((outs[q] == float(100 * q + rank)).all())

    def all_gather_even():
        mine = torch.full((7, 16), float(rank), device=dev)
        full = torch.empty(7 * world, 16, device=dev)
        transport.all_gather_into_tensor(full, mine, group=group)
        assert torch.equal(full[::7, 0], torch.arange(world, device=dev, dtype=full.dtype))

    def all_gather_uneven():
        sizes = [3 + (q % 4) for q in range(world)]
        full = torch.empty(sum(sizes), 16, device=dev)
        offs = np.concatenate([[0], np.cumsum(sizes)])
        views = [full[int(offs[q]):int(offs[q + 1])] for q in range(world)]
        for w in transport.all_gather_uneven(views, torch.full((sizes[rank], 16), float(rank), device=dev), rank, group):
            w.wait()
        for q in range(world):
            assert bool((views[q] == float(q)).all())

    run("all_reduce_sum", all_reduce_sum)
    run("all_reduce_max", all_reduce_max)
    run("all_to_all_single_uneven", all_to_all_uneven)
    run("all_gather_into_tensor", all_gather_even)
    run("all_gather_uneven", all_gather_uneven)
    if dev.type == "cuda" and transport.backend(group) == "nccl":
        run("all_to_all_receive_views", all_to_all_views)
        if res["all_to_all_receive_views"] != "ok":
            parallel.A2A_RECEIVE_VIEWS = False
            res["all_to_all_receive_views"] += " -> parallel.A2A_RECEIVE_VIEWS = False (all_to_all_single + cat instead)"
    if res["all_gather_uneven"] != "ok" and transport.backend(group) == "nccl":
        transport.GROUPED_UNEVEN_ALLGATHER = False
        run("all_gather_uneven_by_broadcasts", all_gather_uneven)
    # every rank must agree on the outcome: a primitive that failed anywhere is failed everywhere
    bad = torch.tensor([sum(v != "ok" and not v.startswith("ok") for v in res.values())], dtype=torch.int64, device=dev)
    try:
        transport.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        res["ranks_agree_all_ok"] = bool(int(bad) == 0)
    except Exception as ex:      # noqa: BLE001
        res["ranks_agree_all_ok"] = f"FAILED: {ex}"[:200]
    return res


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks here, one
    process per GPU, by re-executing this script under torch.distributed.run (the way the driver launches N > 1).
    Refuses -- exit code 2, nothing measured -- when fewer than N GPUs are visible."""
    import subprocess
    n = args.gpus
    if args.workload != "mock" and not args.oversubscribe:
        have = torch.cuda.device_count()
        if have < n:
            print(f"[bench] --gpus {n}: only {have} GPU(s) visible on this box; refusing to report a {n}-GPU number "
                  f"from fewer devices", file=sys.stderr, flush=True)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.call(cmd, env=env)


def timed_regions(wl, args, barrier, world, dev, min_total_s=0.5, max_regions=200):
    """The contract's timed region -- EXACTLY --steps steps between barrier + synchronize on both sides, MAX over
    ranks -- repeated until the regions add up to >= min_total_s (a 20-step region of a 0.25 ms step is 5 ms: clock
    ramp and scheduling noise).  Returns the per-region seconds (the same list on every rank)."""
    import torch.distributed as dist
    out = []
    while True:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            from gae_dgl_amd import transport
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            transport.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt)
        out.append(el)
        if sum(out) >= min_total_s or len(out) >= max_regions:      # same decision on every rank (max-reduced times)
            return out


def region_stats(regions, steps):
    ms = np.asarray(regions) / steps * 1e3
    return {"regions": len(regions), "steps_per_region": steps, "ms_per_step_median": float(np.median(ms)),
            "ms_per_step_min": float(ms.min()), "ms_per_step_max": float(ms.max()),
            "ms_per_step_first_region": float(ms[0]),
            "note": "every region is the contract's timed region (barrier + synchronize, exactly --steps steps, max "
                    "over ranks); regions are repeated until they add up to >= 0.5 s; `ms_per_step` / `value` are "
                    "the median region"}


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing (the line "
              f"would carry the wrong n_gpus)", file=sys.stderr, flush=True)
        sys.exit(2)
    mock = args.workload == "mock"
    import torch.distributed as dist
    if mock:
        dev = torch.device("cpu")
    else:
        if args.oversubscribe and torch.cuda.is_available():
            local_rank = local_rank % torch.cuda.device_count()
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
            print(f"[bench] rank {rank}: needs GPU {local_rank}, {torch.cuda.device_count()} visible (bench.py "
                  f"measures MI355X GPUs only)", file=sys.stderr, flush=True)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    workload = args.workload or ("pubmed" if world == 1 else "rmat")
    group = None
    if world > 1 or workload == "rmat":
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29500")
        # long timeout: after the N-rank regions rank 0 times the same workload on ONE GPU while the others wait
        dist.init_process_group("gloo" if (mock or args.oversubscribe) else "nccl", rank=rank, world_size=world,
                                timeout=datetime.timedelta(minutes=60))
    n_gpus = dist.get_world_size() if dist.is_initialized() else 1       # from the LIVE process group
    preflight = None
    if dist.is_initialized():
        assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus, world)
        if args.row_cost is not None:
            from gae_dgl_amd import parallel as _par
            _par.ROW_COST = int(args.row_cost)
        name = torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu"
        print(f"[bench] rank {rank}/{world}: device {dev} ({name}), backend {dist.get_backend()}, "
              f"rccl_ranks {dist.get_world_size()}", file=sys.stderr, flush=True)
        t_pf = time.perf_counter()
        preflight = collective_preflight(rank, world, dev)
        preflight["seconds"] = time.perf_counter() - t_pf
        if rank == 0:
            print(f"[bench] collective preflight: {preflight}", file=sys.stderr, flush=True)

    def barrier():
        if world > 1:
            dist.barrier()
        if not mock:
            torch.cuda.synchronize()

    if mock:
        wl = MockWorkload(args, rank, world, group)
        for _ in range(args.warmup):
            wl.step()
        regions = timed_regions(wl, args, barrier, world, dev, min_total_s=0.05, max_regions=5)
        elapsed = float(np.median(regions))
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "edges aggregated/sec (SpMM fwd+bwd)",
                              "value": wl.edges_per_step * args.steps / elapsed, "unit": "edges/s", "n_gpus": n_gpus,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None, "dtype": "f32",
                              "data": "synthetic", "config": wl.meta, "timing": region_stats(regions, args.steps),
                              "rccl_ranks": n_gpus, "preflight": preflight}),
                  flush=True)
        return

    if args.knobs:
        from gae_dgl_amd import _lib
        for kv in args.knobs.split(","):
            k, v = kv.split("=")
            _lib.call("gae_tuning_set", k.encode(), int(v))
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if args.no_fused_layers:
        from gae_dgl_amd import gae as _gae
        _gae.FUSE_NARROW_LAYERS = False
    if args.layer1 == "reference":
        from gae_dgl_amd import gae as _gae
        _gae.TRANSFORM_FIRST_AUTO = False
    if world > 1:
        dist.barrier()
    from gae_dgl_amd import ops

    if workload == "rmat":
        t_build0 = time.perf_counter()
        wl = RmatShardedWorkload(args, dev, rank, world, group)
        torch.cuda.synchronize()
        wl.meta["workload_build_s"] = time.perf_counter() - t_build0
        wl.meta["transport"] = ("RCCL (backend nccl), one process per GPU" if not args.oversubscribe else
                                f"oversubscribed: {world} ranks share {torch.cuda.device_count()} GPU(s), collectives over gloo "
                                f"staged through host memory -- functional rehearsal, not a scaling measurement")
    elif workload == "zinc":
        wl = ZincWorkload(args, dev, rank=rank, world=world, group=group)
        wl.meta["transport"] = ("RCCL (backend nccl), one process per GPU" if not args.oversubscribe else
                                f"oversubscribed: {world} ranks share {torch.cuda.device_count()} GPU(s), collectives over gloo "
                                f"staged through host memory -- functional rehearsal, not a scaling measurement") \
            if world > 1 else "none (1 GPU)"
    elif workload == "vgae":
        wl = VgaeWorkload(args, dev)
    else:
        wl = CitationWorkload(workload, args, dev)

    graphed = getattr(wl, "use_graph", False)
    prof = ops.EventProfiler()
    if graphed:
        # Events cannot be recorded inside a graph replay: the per-kernel HIP-event timings come from an eager
        # pass of the SAME steps right before the capture; the timed regions then replay the captured graph.
        for _ in range(args.warmup):
            wl.step()
        ops.profiler = prof
        torch.cuda.synchronize()
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize()
        ops.profiler = None
        try:
            wl.capture()
        except Exception as ex:      # noqa: BLE001 -- a capture holding collectives is the first thing to fail on new hardware
            if world == 1:
                raise
            # every rank must take the same branch: the warm-up steps inside the capture hold collectives
            print(f"[bench] rank {rank}: capturing the step failed ({type(ex).__name__}: {ex}); eager steps instead",
                  file=sys.stderr, flush=True)
            wl.captured = None
            if hasattr(wl, "runner"):
                wl.runner = None
            wl.meta["launch"] = f"eager (capturing the step with its collectives failed: {type(ex).__name__}: {str(ex)[:200]})"
            graphed = False
    if world > 1 and getattr(wl, "use_graph", False):
        # all ranks or none: a rank that fell back to eager steps cannot pair its collectives with replaying peers
        flag = torch.tensor([0 if graphed else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag) and graphed:
            wl.captured = None
            if hasattr(wl, "runner"):
                wl.runner = None
            wl.meta["launch"] = "eager (another rank could not capture the step with its collectives)"
            graphed = False
    for _ in range(args.warmup):
        wl.step()
    if not graphed:
        # eager workloads: HIP-event pairs around the SpMM launches of one extra pass of --steps steps (recording
        # events costs host time; the timed regions run without them)
        ops.profiler = prof
        barrier()
        for _ in range(args.steps):
            wl.step()
        barrier()
        ops.profiler = None
    regions = timed_regions(wl, args, barrier, world, dev)
    elapsed = float(np.median(regions))
    dom_fn = wl.dominant_launch() if hasattr(wl, "dominant_launch") else None   # may contain a collective
    comm = wl.comm_profile() if hasattr(wl, "comm_profile") else None              # collective: every rank
    one_gpu = None
    if world > 1 and workload == "rmat":
        # the SAME workload on ONE GPU, timed in this invocation: rank 0 builds the whole graph on its GPU (in a
        # 1-rank group) while the other ranks wait at the closing barrier
        g1 = dist.new_group(ranks=[0])
        if rank == 0:
            torch.cuda.empty_cache()
            w1 = RmatShardedWorkload(args, dev, 0, 1, g1)
            for _ in range(max(2, min(args.warmup, 5))):
                w1.step()
            r1 = timed_regions(w1, args, torch.cuda.synchronize, 1, dev)
            e1 = float(np.median(r1))
            one_gpu = {"value": w1.edges_per_step * args.steps / e1, "ms_per_step": e1 / args.steps * 1e3,
                       "timing": region_stats(r1, args.steps),
                       "source": "same invocation: rank 0 alone on the whole graph (1-rank process group, same "
                                 "kernels and plans), timed after the N-rank regions"}
            del w1
            torch.cuda.empty_cache()
    if rank != 0:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    times = prof.summary()
    spmm_keys = [k for k in times if k[0] == "spmm"]
    if wl.dominant is None and workload == "rmat":
        dom = []
    elif wl.dominant is None and spmm_keys:   # zinc: batch shapes vary slightly; take the F=39 launches
        dom = [t for k in spmm_keys if k[3] == 39 for t in times[k]]
        k39 = [k for k in spmm_keys if k[3] == 39]
        nb = int(np.mean([k[1] for k in k39])); eb = wl.edges_per_step // 3 // max(world, 1)
        wl.alg_bytes = 4 * (nb + 1) + 4 * eb + 2 * 4 * 39 * nb
        wl.dominant_desc = f"spmm F=39 (layer-1 aggregation of a {wl.B}-molecule batch, ~{nb} rows, ~{eb} edges)"
    else:
        dom = times.get(wl.dominant, [])
    t_dom_instep = float(np.mean(dom)) if dom else float("nan")
    # Kernel duration for the roofline: HIP events (launch stream) around 50 back-to-back launches of the
    # dominant kernel on the step's own operands, right after the timed region.  The per-launch event pairs
    # recorded inside the steps (t_dom_instep) also contain the host-side launch gap of an eager step and
    # over-state the kernel time; the back-to-back figure is the one that agrees with rocprofv3's average
    # kernel duration (profiles/).
    t_dom = time_launches(dom_fn, iters=50 if wl.alg_bytes < 1e9 else 10) if dom_fn is not None else t_dom_instep
    spmm_t = sum(sum(times[k]) for k in spmm_keys)
    copy_gbs = copy_bandwidth(wl.alg_bytes, dev) if dom_fn is not None else None
    value = wl.edges_per_step * args.steps / elapsed
    line = {
        "metric": "edges aggregated/sec (SpMM fwd+bwd)", "value": value, "unit": "edges/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "rccl_ranks": n_gpus if dist.is_initialized() else 0, "preflight": preflight,
        "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None, "dtype": getattr(wl, "dtype", "f32"), "data": "synthetic",
        "config": wl.meta,
        "value_spmm_only": wl.edges_per_step / max(world, 1) * args.steps / spmm_t if spmm_t else None,
        "value_note": "value = SpMM edges of a step / WHOLE-step time (encoder + fused N^2 loss + backward + Adam: the loss is "
                      "two thirds of a Pubmed step); value_spmm_only = the same edges / the HIP-event time of the step's SpMM "
                      "launches alone (see spmm_only) -- the figure to set beside an SpMM-only CPU rate (cpu_baseline.spmm)",
        "epoch_time_s": elapsed / args.steps * (wl.meta.get("batches_per_epoch_at_239455_graphs", 1)),
        "spmm_only": {"edges_per_s": wl.edges_per_step / max(world, 1) * args.steps / spmm_t if spmm_t else None,
                      "sum_event_time_s": spmm_t, "launches": sum(len(times[k]) for k in spmm_keys),
                      "note": "rank-0 HIP-event time of its SpMM launches inside the timed steps"
                              + (" (per-rank share of the edges)" if world > 1 else "")},
        "roofline": {"bound": "hbm", "kernel": wl.dominant_desc,
                     "achieved": wl.alg_bytes / t_dom / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": wl.alg_bytes / t_dom / 1e9 / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(getattr(wl, "pmc_key", "")),
                     "traffic_note": "HBM bytes/launch from separate rocprofv3 --pmc passes of the same kernel and "
                                     "shape (profiles/pmc_traffic_r0N.json), not collected in this run",
                     "copy_GBs": copy_gbs, "frac_of_copy": (wl.alg_bytes / t_dom / 1e9 / copy_gbs) if copy_gbs else None,
                     "copy_note": "copy_GBs = measured rate of a device copy moving alg_bytes_per_launch in total "
                                  "(read + write), same timing method: the second denominator of SURVEY 8(d)",
                     "alg_bytes_per_launch": wl.alg_bytes, "avg_launch_us": t_dom * 1e6,
                     "avg_launch_us_event_pairs_inside_steps": t_dom_instep * 1e6, "launches_timed_inside_steps": len(dom)},
    }
    if comm is not None:
        t, calls = comm
        ex = t.get("exchange", 0.0) * calls.get("exchange", 0) + \
            t.get("exchange_start", 0.0) * calls.get("exchange_start", 0) + \
            t.get("exchange_wait", 0.0) * calls.get("exchange_wait", 0)
        sp = sum(t.get(k, 0.0) * calls.get(k, 0) for k in ("spmm", "spmm_own", "spmm_remote"))
        line["comm"] = {
            "exchange_s_per_step": ex, "spmm_s_per_step": sp, "step_s": elapsed / args.steps,
            "edges_per_s_spmm_only_no_comm": wl.edges_per_step / max(world, 1) / sp * world if sp else None,
            "edges_per_s_spmm_plus_exchange": wl.edges_per_step / (sp + ex) if sp + ex else None,
            "avg_us": {k: v * 1e6 for k, v in t.items()}, "calls_per_step": calls,
            "exchange_bytes_received_per_spmm": wl.sg.exchange_bytes(32),
            "note": "rank-0 HIP-event times over 5 extra steps outside the timed region; exchange_wait is what the "
                    "own-column SpMM did not hide; SpMM-only edges/s assumes every rank takes as long as rank 0 "
                    "(nnz-balanced blocks)"}
    line["timing"] = region_stats(regions, args.steps)
    if one_gpu is not None:
        one_gpu["speedup"] = value / one_gpu["value"]
        line["same_workload_1gpu"] = one_gpu
    if world == 1 and getattr(wl, "loss_launch", None) is not None:
        # ---- the step's DOMINANT launch: fused decoder + weighted BCE (loss + dZ), VALU / transcendental bound
        loss_fn = wl.loss_launch()                  # (sets wl.n for the molecule batches)
        n = wl.n
        t_loss = time_launches(loss_fn, iters=20 if n < 50000 else 5, warmup=5 if n < 50000 else 2)
        from gae_dgl_amd import _lib as _l
        kind = {1: "full", 2: "symmetric", 3: "symmetric256"}[_l.tuning_get("bce_last_kind")]     # what the launches above ran on
        per_logit, frac_eval = loss_slots_per_logit(kind)
        units = per_logit * frac_eval * float(n) * n
        isa = LOSS_ISA[kind][0]
        line["roofline_step_dominant"] = {
            "bound": "issue (VALU + MFMA)", "kernel": f"fused decoder + BCE, loss and dZ ({kind} dense kernel + edge / "
                                                      f"prepare / finalize launches), N = {n}, d = 16",
            "achieved": units / t_loss / 1e12, "peak": ISSUE_PEAK_LANE_SLOTS / 1e12, "unit": "T lane-slots/s",
            "frac": units / t_loss / ISSUE_PEAK_LANE_SLOTS, "avg_launch_us": t_loss * 1e6,
            "peak_spec": SPEC_LANE_OPS / 1e12, "frac_spec": units / t_loss / SPEC_LANE_OPS,
            "frac_alg": loss_alg_slots_per_logit(kind) * frac_eval * float(n) * n / t_loss / ISSUE_PEAK_LANE_SLOTS,
            "alg_note": f"frac_alg counts only the algorithmically required issue slots ({loss_alg_slots_per_logit(kind):.1f} of the "
                        f"{per_logit:.1f} per logit: {LOSS_ALG[kind]} per 32 logits -- S and the two P V products at the chosen "
                        "piece count, exp, one reciprocal per four logits, the log of the running product, the fp32 arithmetic "
                        "between them); the rest (splitting P into 16-bit pieces, copysign, re-laying P out for the mirror "
                        "product, LDS traffic) is the price of fp32-grade products on the 16-bit matrix pipe",
            "peak_note": "peak = the issue rate tools/probes/inst_cost.hip measures at 4 waves / SIMD (self-measured); "
                         "peak_spec = the data sheet's fp32 vector rate, 157.3 TFLOP/s / 2 = 78.6 T lane-operations/s",
            "logits_per_s": float(n) * n / t_loss,
            "model": f"{per_logit:.1f} issue slots (v_fma_f32 equivalents, measured costs) per evaluated logit and "
                     f"lane: {isa}, per 32 logits; {frac_eval:g} N^2 logits evaluated; MFMA and VALU time of a "
                     "SIMD add up on gfx950 (tools/probes/inst_cost.hip), so the matrix-core work is part of the "
                     "same budget; the launch sequence also holds the mirror reduction, the edge pass and two "
                     "bookkeeping launches, which this model does not count",
            "mfma_flops_per_s": isa["mfma"] * 16384.0 / 64 / 32 * frac_eval * float(n) * n / t_loss,
            "share_of_step": t_loss / (elapsed / args.steps),
            "sequence_note": "avg_launch_us = the loss called on its own: prepare + dense + edges + final reduction "
                             "(4 launches).  Inside the captured step the prepare work rides in the last encoder "
                             "launch's epilogue and the final reduction in the optimiser launch (2 launches, about "
                             "10 us less): share_of_step is an upper bound"}
        # ---- the same step with exact-fp32 products everywhere (no bf16 x 3 split)
        if graphed and isinstance(wl, CitationWorkload):
            from gae_dgl_amd import _lib
            knobs = {b"bce_s_bf16": 3, b"bce_pv_bf16": 1, b"atb_bf16": 1, b"xw_p3": 1}          # name -> the library's default
            for k in knobs:
                _lib.call("gae_tuning_set", k, 0)
            try:
                wl.captured = None
                wl.capture()
                for _ in range(3):
                    wl.step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    wl.step()
                torch.cuda.synchronize()
                line["ms_per_step_exact_fp32"] = (time.perf_counter() - t1) / args.steps * 1e3
                line["value_exact_fp32"] = wl.edges_per_step / (line["ms_per_step_exact_fp32"] * 1e-3)
            finally:
                for k, v in knobs.items():
                    _lib.call("gae_tuning_set", k, v)
    if world == 1 and type(wl) is CitationWorkload and getattr(wl, "tf", False) and not wl.sparse:
        # ---- The metric's kernel is the SpMM AGGREGATION: the top-level `roofline` is
        #      (1) the reference-order layer-1 aggregation A X at the input width -- the north-star's SpMM, the launch
        #      its 60 % target is quoted on; `--layer1 reference` runs it inside the step -- with (2) the aggregation the
        #      DEFAULT step runs, act(A P + b) at F = 32 on the step's own operands, beside it as `in_step`.  The dense
        #      layer-1 pair of the default step (xw_fwd, xtg) moves to `roofline_dense`.  Every entry: kernel-only,
        #      back-to-back launches between HIP events on the launch stream; WARM (operands stay in the 256 MB Infinity
        #      Cache between launches, as they do between the replays of the timed region) and MALL-COLD (rotation over
        #      disjoint operand copies adding up to >= 512 MB: every launch reads DRAM), each with the copy rate of the
        #      same bytes measured the same way as second denominator.
        from gae_dgl_amd import workloads as Wl
        name = wl.meta["workload"].split("-")[0]
        J, E_ = wl.hidden[0], wl.g.number_of_edges()
        dense_entry = dict(line["roofline"])          # the xw_fwd entry computed above
        warm = wl.layer1_launches()
        b_agg = Wl.spmm_alg_bytes(wl.n, wl.n, E_, J, 4)
        b_xtg = 4 * (wl.n * wl.F_in + 3 * wl.n * J + J * wl.F_in + J)
        Kc = {"xw_fwd": cold_copies(wl.alg_bytes), "agg": cold_copies(b_agg), "xtg": cold_copies(b_xtg)}
        rot = [wl.layer1_launches(fresh=True) for _ in range(max(Kc.values()))]
        t_warm = {k: time_launches(warm[k], iters=50, warmup=30) for k in ("agg", "xtg")}
        t_cold = {k: time_rotation([r[k] for r in rot[:Kc[k]]]) for k in ("xw_fwd", "agg", "xtg")}
        del rot
        torch.cuda.empty_cache()
        ref = citation_spmm_probe(name, dev)
        c_ref = copy_bandwidth(ref["alg_bytes"], dev)
        c_agg = copy_bandwidth(b_agg, dev)
        in_step = {
            "kernel": f"gae_spmm_csr_epilogue: act(A (X W^T) + b), F = {J}, {wl.n} rows, {E_} edges (the aggregation of the default step)",
            "alg_bytes_per_launch": b_agg, "avg_launch_us": t_warm["agg"] * 1e6, "achieved": b_agg / t_warm["agg"] / 1e9,
            "frac": b_agg / t_warm["agg"] / 1e9 / HBM_PEAK_GBS, "copy_GBs": c_agg,
            "frac_of_copy": b_agg / t_warm["agg"] / 1e9 / c_agg, "edges_per_s": E_ / t_warm["agg"],
            **cold_fields(b_agg, t_cold["agg"], dev, Kc["agg"]), "edges_per_s_cold": E_ / t_cold["agg"],
            "note": "5.5 MB of compulsory traffic: at 8 TB/s the launch would last 0.7 us -- it is bound by its chain of "
                    "dependent round trips (row pointers -> neighbour ids -> gathered rows), not by bytes"}
        line["roofline"] = {
            "bound": "hbm", "kernel": ref["shape"] + f": gae_spmm_csr, F = {ref['F']}, {wl.n} rows, {E_} edges",
            "achieved": ref["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ref["frac_hbm_peak"],
            "traffic": ref["traffic"],
            "traffic_note": "HBM bytes/launch from separate rocprofv3 --pmc passes of the same kernel and shape "
                            "(profiles/pmc_traffic_r0N.json; FETCH_SIZE / WRITE_SIZE, gfx950 corrections), not collected in this run",
            "alg_bytes_per_launch": ref["alg_bytes"], "avg_launch_us": ref["us_per_launch"], "edges_per_s": ref["edges_per_s"],
            "copy_GBs": c_ref, "frac_of_copy": ref["achieved_GBs"] / c_ref,
            **{k: ref[k] for k in ("avg_launch_us_cold", "achieved_GBs_cold", "frac_cold", "copy_GBs_cold",
                                   "frac_of_copy_cold", "cold_copies", "edges_per_s_cold") if k in ref},
            "residency_note": "frac: back-to-back launches on ONE operand set (79 MB: resident in the 256 MB Infinity Cache, "
                              "so '% of 8 TB/s' is against the on-die fabric); frac_cold: the same launch rotating over "
                              "disjoint operand sets adding up to >= 512 MB, so every launch reads DRAM",
            "in_step": in_step}
        dense_entry.pop("spmm", None)
        dense_entry.update(cold_fields(wl.alg_bytes, t_cold["xw_fwd"], dev, Kc["xw_fwd"]))
        c_xtg = copy_bandwidth(b_xtg, dev)
        line["roofline_dense"] = {
            "xw_fwd": dense_entry,
            "xtg": {"bound": "hbm", "kernel": f"xtg (gae_xw_wgrad) dW1 = G^T X, db1, {wl.n} x {wl.F_in} -> {J} x {wl.F_in} (X read once)",
                    "alg_bytes_per_launch": b_xtg, "avg_launch_us": t_warm["xtg"] * 1e6, "achieved": b_xtg / t_warm["xtg"] / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_xtg / t_warm["xtg"] / 1e9 / HBM_PEAK_GBS,
                    "copy_GBs": c_xtg, "frac_of_copy": b_xtg / t_warm["xtg"] / 1e9 / c_xtg,
                    **cold_fields(b_xtg, t_cold["xtg"], dev, Kc["xtg"])}}
    if "ms_per_step_exact_fp32" in line:        # (the driver keeps `config` whole: the same-arithmetic figure travels there too)
        line["config"]["ms_per_step_exact_fp32"] = line["ms_per_step_exact_fp32"]
        line["config"]["value_exact_fp32"] = line["value_exact_fp32"]
    if "decoder_bce" in {k[0] for k in times}:
        kb = [k for k in times if k[0] == "decoder_bce"]
        tb = float(np.mean([t for k in kb for t in times[k]]))
        n = kb[0][1]; d = kb[0][2]
        line["decoder_loss"] = {"kernel": "fused decoder+BCE fwd+bwd (prepare + dense + edges + finalize)",
                                "avg_us_event_pairs_in_eager_steps": tb * 1e6, "logits_per_s": n * n / tb,
                                "bound": "issue: 2 transcendentals + ~10 VALU ops + 0.9 (full kernel) / 1.75 "
                                         "(symmetric) bf16 MFMAs per 32 logits and lane; rocprofv3 kernel time in "
                                         "profiles/"}
    if not args.no_cpu_baseline and hasattr(wl, "cpu_baseline") and world == 1:      # (rank 0 at N = 1 only)
        line["cpu_baseline"] = wl.cpu_baseline(args.cpu_seconds)
        line["cpu_baseline"]["cpu_model"] = cpu_model_name()
        # the SpMM metric alone on the host cores: forward and backward, all cores and one thread
        line["cpu_baseline"]["spmm"] = cpu_spmm_baseline(wl.src, wl.dst, wl.n, [wl.F_in, 32], 0.3)
    if not args.no_extra and world == 1 and workload not in ("rmat",):
        del wl
        torch.cuda.empty_cache()
        line["extra"] = {"steps": extra_steps(args, dev), "spmm_kernel_only": extras(dev)}
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    # RCCL prints a version banner through C stdio; flush it first so the JSON line is the LAST line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    for k in ("ms_per_step_exact_fp32", "value_exact_fp32", "roofline_dense", "roofline"):       # last in the line: inside any tail of it
        if k in line:
            line[k] = line.pop(k)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
