#!/usr/bin/env python3
"""profiles/r05_hubs.txt from what tools/r05/hubs.sh and tools/prof_bench.sh left under gpurun_out/: citation steps on
graphs with the real graphs' degree profile, by plan policy, and the per-kernel times next to the uniform graph's."""
import csv
import glob
import json
import os

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = ["citation training steps (bench.py --workload W --degrees D, captured step, 300 steps), by plan policy",
       "  GAE_TABLE_MAXDEG=64  : rule of rounds 2-4 (a row longer than 64 edges -> skew plan with heavy rows; the scripts then",
       "                         run the reference's layer order on dense features: bench.py --layer1 reference --features dense)",
       "  GAE_TABLE_MAXDEG=1024: round 5 default (table-only plan; rows beyond the 16 table slots gathered by the whole wave)",
       "", f"{'workload':10s} {'degrees':10s} {'policy':>6s} {'longest row':>11s} {'ms/step':>9s}  layer 1"]
for f in sorted(glob.glob(os.path.join(R, "gpurun_out/r05_hubs/*.json"))):
    w, deg, lim = os.path.basename(f)[:-5].split("_")
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        out.append(f"{w:10s} {deg:10s} {lim:>6s} {d['config'].get('longest_row', 0):11d} {d['ms_per_step']:9.4f}  "
                   f"{d['config'].get('layer1', '')[:64]}")
    except Exception as e:          # noqa: BLE001
        out.append(f"{w:10s} {deg:10s} {lim:>6s}  FAILED ({e})")
for w in ("pubmed", "cora"):
    tabs = {}
    for deg in ("uniform", "planetoid"):
        p = os.path.join(R, f"gpurun_out/r05_hubs_{w}_{deg}/bench_kernel_stats.csv")
        if os.path.exists(p):
            tabs[deg] = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(p))}
    if len(tabs) == 2:
        out += ["", f"{w}: kernels of the captured step (rocprofv3 --kernel-trace --stats), average us per launch, uniform -> planetoid degrees"]
        names = [k for k, (c, _) in tabs["planetoid"].items() if c > 1000]
        for k in sorted(names, key=lambda k: -tabs["planetoid"][k][1]):
            a = tabs["uniform"].get(k, (0, float("nan")))[1]
            b = tabs["planetoid"][k][1]
            nm = k.replace("(anonymous namespace)::", "").replace("void ", "")
            out.append(f"  {nm[:92]:92s} {a:7.1f} -> {b:7.1f}  ({b - a:+.1f})")
open(os.path.join(R, "profiles/r05_hubs.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
