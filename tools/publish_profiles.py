#!/usr/bin/env python3
"""Copy the measurements of tools/collect.sh <tag> (gpurun_out/<tag>/) into profiles/ and derive
profiles/pmc_traffic_<tag>.json + a markdown summary (stdout).  Runs in the build container, no GPU.
usage: python tools/publish_profiles.py r02"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
SRC = os.path.join(ROOT, "gpurun_out", TAG)
if TAG in ("r04c", "r05c", "r06c"):  # rounds 4-6 collect into gpurun_out/r0Nc; files are r0N_*
    TAG = TAG[:3]
DST = os.path.join(ROOT, "profiles")


def last_json(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def pmc_sums(d, needles=("spmm",)):
    """(kernel, counter) -> (sum over all recorded launches, number of launches)"""
    out = {}
    for f in sorted(glob.glob(os.path.join(d, "p*", "pmc_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if not any(t in r["Kernel_Name"] for t in needles):
                continue
            k = (r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0], r["Counter_Name"])
            a = out.setdefault(k, [0.0, 0])
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return out


def pmc_means(d, needles=("spmm",)):
    out = {}
    for f in sorted(glob.glob(os.path.join(d, "p*", "pmc_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if not any(t in r["Kernel_Name"] for t in needles):
                continue
            k = (r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0], r["Counter_Name"])
            out.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


benches = {}
for name in ("pubmed", "pubmed_reference_order", "cora", "citeseer", "cora_dense_features", "citeseer_dense_features", "vgae", "zinc", "zinc128", "zinc_eager", "zinc128_eager",
             "rmat_s24_1gpu", "rmat_s24_1gpu_aggregate_first"):
    p = os.path.join(SRC, f"bench_{name}.json")
    if os.path.exists(p):
        benches[name] = last_json(p)
        json.dump(benches[name], open(os.path.join(DST, f"{TAG}_bench_{name}.json"), "w"), indent=1)
for w in ("pubmed", "cora", "citeseer", "vgae", "zinc", "zinc128", "rmat"):
    st = os.path.join(SRC, f"prof_{w}", "bench_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(DST, f"{TAG}_{w}_step_kernel_stats.csv"))
        shutil.copy(os.path.join(SRC, f"{w}_step_kernel_stats_top.txt"), os.path.join(DST, f"{TAG}_{w}_step_kernel_stats_top.txt"))
for f in ("loss_sq.txt", "probe_wave_spec_overlap.txt", "bce_bench_sizes.txt", "spmm_tile_sweep_pubmed.txt", "spmm_tile_sweep_cora.txt",
          "rmat_windows.txt", "spx_bench.txt", "tall_bench.txt", "xtg_probe.txt", "zinc_l1.txt", "loss_condition.txt", "bce_bench_cora.txt", "plan_build_time.txt", "xw_bench.txt", "xw_sweep_pubmed.txt", "linear_bench.txt", "spmm_bench_pubmed.txt", "spmm_bench_diag.txt", "bce_bench_pubmed.txt", "bce_bench_zinc.txt",
          "probe_gather_l2.txt", "probe_gather_l2b.txt", "probe_valu_rate.txt", "probe_inst_cost.txt",
          "probe_mfma32_check.txt"):
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f"{TAG}_{f}"))

sys.path.insert(0, ROOT)
from gae_dgl_amd import workloads as W  # noqa: E402  (pure numpy helpers)
traffic = {"_how": "rocprofv3 --pmc, one counter set per pass (tools/pmc.sh: FETCH_SIZE | WRITE_SIZE | TCC_*), MI355X, "
                   "ROCm 7.2, per-launch means over the launches of tools/spmm_one.py (operands and flags as the "
                   "package launches them).  FETCH_SIZE is in KiB and doubled as MI355X_MICROARCH.md prescribes for "
                   "16-B/lane coalesced reads on gfx950 (calibrated on the zinc-250k F=32 launch, whose compulsory "
                   "read bytes are known); WRITE_SIZE is used as reported (matches the output bytes to 0.1 %)."}
shapes = {"pubmed500": ("pubmed-F500", "pubmed", 500), "pubmed500_tv8": ("pubmed-F500-tile_vecs8", "pubmed", 500),
          "pubmed500_tv24": ("pubmed-F500-tile_vecs24", "pubmed", 500), "pubmed500_tv-1": ("pubmed-F500-tile_vecs-1", "pubmed", 500),
          "pubmed500_plain": ("pubmed-F500-untiled", "pubmed", 500),
          "pubmed32": ("pubmed-F32", "pubmed", 32), "citeseer3703": ("citeseer-F3703", "citeseer", 3703),
          "cora1433": ("cora-F1433", "cora", 1433), "zincb39": ("zinc-batch4096-F39", "zinc", 39),
          "zinc32": ("zinc250k-F32", "zinc", 32), "zinc39": ("zinc250k-F39-ld40", "zinc", 39),
          "rmat32": ("rmat-s24-F32", "rmat", 32), "rmat16": ("rmat-s24-F16", "rmat", 16)}
for sh, (key, graph, F) in shapes.items():
    d = os.path.join(SRC, f"pmc_{sh}")
    if not os.path.isdir(d):
        continue
    m = pmc_means(d)
    kernels = sorted({k[0] for k in m})
    main = max(kernels, key=lambda k: m.get((k, "FETCH_SIZE"), 0))
    fetch, write = m.get((main, "FETCH_SIZE"), 0.0), m.get((main, "WRITE_SIZE"), 0.0)
    log = open(os.path.join(SRC, f"pmc_{sh}.txt")).read() if os.path.exists(os.path.join(SRC, f"pmc_{sh}.txt")) else ""
    done = [l for l in open(os.path.join(d, "p1.log")).read().splitlines() if l.startswith("done")]
    n, nnz = (int(done[0].split()[2]), int(done[0].split()[3])) if done else (0, 0)
    # a product on a skew plan is SEVERAL kernels (empty-row fill, light rows, two segment launches, two combines):
    # its traffic is the sum over all of them per call of the product (round 3 filed one mean segment launch: half)
    sums = pmc_sums(d)
    calls = min(v[1] for (k, c), v in sums.items() if c == "FETCH_SIZE") if sums else 1
    fetch_all = sum(v[0] for (k, c), v in sums.items() if c == "FETCH_SIZE") / calls
    write_all = sum(v[0] for (k, c), v in sums.items() if c == "WRITE_SIZE") / calls
    if len(kernels) > 1:
        fetch, write = fetch_all, write_all
    traffic[key] = {"kernel": main if len(kernels) == 1 else "all %d kernels of the product: " % len(kernels) + ", ".join(kernels),
                    "launch": done[0] if done else "", "fetch_kib_raw": fetch, "write_kib": write,
                    "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
                    "alg_bytes": W.spmm_alg_bytes(n, n, nnz, F, 4) if n else None,
                    "tcc_hit": m.get((main, "TCC_HIT_sum")), "tcc_miss": m.get((main, "TCC_MISS_sum"))}
    dd = os.path.join(DST, f"{TAG}_pmc_{sh}")
    os.makedirs(dd, exist_ok=True)
    for i, f in enumerate(sorted(glob.glob(os.path.join(d, "p*", "pmc_counter_collection.csv"))), 1):
        shutil.copy(f, os.path.join(dd, f"pass{i}.csv"))
# ---- the layer-1 dense passes (tools/r03/xw_one.py): kernel xw_fwd_kernel / xtg_kernel; compulsory bytes X + P (+ W) / X + G
xw = {"xwfwd_pubmed": ("pubmed-xw_fwd", "xw_fwd", 4), "xwfwd_cora": ("cora-xw_fwd", "xw_fwd", 4),
      "xwfwd_citeseer": ("citeseer-xw_fwd", "xw_fwd", 4), "xwfwd_citeseer_bf16": ("citeseer-bf16-xw_fwd", "xw_fwd", 2),
      "xwgrad_pubmed": ("pubmed-xw_wgrad", "xtg", 4), "xwgrad_cora": ("cora-xw_wgrad", "xtg", 4),
      "xwgrad_citeseer": ("citeseer-xw_wgrad", "xtg", 4), "xwgrad_citeseer_bf16": ("citeseer-bf16-xw_wgrad", "xtg", 2)}
for sh, (key, needle, elem) in xw.items():
    d = os.path.join(SRC, f"pmc_{sh}")
    if not os.path.isdir(d):
        continue
    m = pmc_means(d, (needle,))
    kernels = sorted({k[0] for k in m})
    if not kernels:
        continue
    main = max(kernels, key=lambda k: m.get((k, "FETCH_SIZE"), 0))
    fetch, write = m.get((main, "FETCH_SIZE"), 0.0), m.get((main, "WRITE_SIZE"), 0.0)
    done = [l for l in open(os.path.join(d, "p1.log")).read().splitlines() if l.startswith("done")]
    n, K, J = (int(done[0].split()[2]), int(done[0].split()[3]), int(done[0].split()[4])) if done else (0, 0, 0)
    alg = (elem * n * K + 4 * n * J + 4 * J * K) if needle == "xw_fwd" else (elem * n * K + 3 * 4 * n * J + 4 * J * K)
    traffic[key] = {"kernel": main, "launch": done[0] if done else "", "fetch_kib_raw": fetch, "write_kib": write,
                    "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024), "alg_bytes": alg,
                    "tcc_hit": m.get((main, "TCC_HIT_sum")), "tcc_miss": m.get((main, "TCC_MISS_sum"))}
    dd = os.path.join(DST, f"{TAG}_pmc_{sh}")
    os.makedirs(dd, exist_ok=True)
    for i, f in enumerate(sorted(glob.glob(os.path.join(d, "p*", "pmc_counter_collection.csv"))), 1):
        shutil.copy(f, os.path.join(dd, f"pass{i}.csv"))
json.dump(traffic, open(os.path.join(DST, f"pmc_traffic_{TAG}.json"), "w"), indent=1)

if "pubmed" in benches and "roofline_dense" in benches["pubmed"]:
    b = benches["pubmed"]
    rows = [("Pubmed A X, F = 500 (reference-order layer 1; the metric's SpMM)", b["roofline"]),
            ("Pubmed act(A P + b), F = 32 (the default step's aggregation)", b["roofline"]["in_step"]),
            ("Pubmed xw_fwd P = X W^T (default step, layer 1 forward)", b["roofline_dense"]["xw_fwd"]),
            ("Pubmed xtg dW1 = G^T X (default step, layer 1 backward)", b["roofline_dense"]["xtg"])]
    for e in b.get("extra", {}).get("spmm_kernel_only", []):
        if "frac_cold" in e:
            rows.append((e["shape"] + f", F = {e['F']}", {"alg_bytes_per_launch": e["alg_bytes"], "avg_launch_us": e["us_per_launch"], "frac": e["frac_hbm_peak"],
                                                      "avg_launch_us_cold": e.get("avg_launch_us_cold"), "frac_cold": e["frac_cold"],
                                                      "copy_GBs_cold": e.get("copy_GBs_cold"), "cold_copies": e.get("cold_copies")}))
    with open(os.path.join(DST, f"{TAG}_cold.txt"), "w") as f:
        f.write("MALL-warm vs MALL-cold roofline fractions (bench.py, round 6; VERDICT r05 #2b).  warm = back-to-back launches on ONE operand set\n"
                "(working sets below 256 MB stay in the Infinity Cache between launches, as they do between the replays of the timed region);\n"
                "cold = the same launch rotating over disjoint operand sets (CSR arrays, plan tables, H / X, outputs) that add up to >= 512 MB, so\n"
                "every launch reads DRAM.  Same timing method for both: one HIP event pair around a HIP-graph replay; frac = B_alg / t / 8 TB/s.\n"
                "copy cold = a device copy of the same byte count rotating the same way (GB/s, read + write).\n\n")
        f.write(f"{'launch':78s} {'B_alg MB':>9s} {'warm us':>8s} {'frac':>6s} {'cold us':>8s} {'frac_cold':>9s} {'copy cold GB/s':>14s} {'sets':>4s}\n")
        for name, r in rows:
            cu, fc = r.get("avg_launch_us_cold"), r.get("frac_cold")
            f.write(f"{name[:78]:78s} {r['alg_bytes_per_launch'] / 1e6:9.1f} {r['avg_launch_us']:8.2f} {r['frac']:6.3f} "
                    f"{(cu if cu is not None else float('nan')):8.2f} {fc:9.3f} {(r.get('copy_GBs_cold') or float('nan')):14.0f} {str(r.get('cold_copies') or '-'):>4s}\n")
print("| workload | ms/step | edges/s | dominant launch | us | % of 8 TB/s | CPU port ms/step |")
print("|---|---|---|---|---|---|---|")
for name, d in benches.items():
    r = d["roofline"]; c = d.get("cpu_baseline") or {}
    print(f"| {name} | {d['ms_per_step']:.4f} | {d['value']:.3e} | {r['kernel'][:60]} | {r['avg_launch_us']:.2f} | "
          f"{100 * r['frac']:.1f} | {c.get('ms_per_step', float('nan')):.1f} |")
if "pubmed" in benches and "extra" in benches["pubmed"]:
    print()
    for e in benches["pubmed"]["extra"]["spmm_kernel_only"]:
        print(f"| {e['shape']} | F={e['F']} ld={e['ld']} | {e['us_per_launch']:.1f} us | {e['edges_per_s'] / 1e9:.2f} Gedge/s | "
              f"{100 * e['frac_hbm_peak']:.1f} % |")
    print(json.dumps(benches["pubmed"].get("roofline_step_dominant")))
for k, v in traffic.items():
    if k != "_how":
        print(k, v["hbm_bytes_per_launch"], v["alg_bytes"], v["kernel"][:70])
