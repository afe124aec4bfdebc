#!/usr/bin/env python3
"""Summary of tools/r06/loss_sq.sh: per kernel the mean of every SQ counter over its dispatches, and the derived busy
fractions.  Units as found on this box (checked on the calibration kernels of tools/probes/wave_spec_overlap.hip, whose
instruction counts are known): SQ_INSTS_* and SQ_VALU_MFMA_BUSY_CYCLES are whole-chip totals (BUSY_CYCLES = 16 per
v_mfma_*_16x16x32 = its 4 passes); GRBM_GUI_ACTIVE is the SUM over the 8 XCDs (a 2.22 ms kernel reads 4.3e7 = 8 x 2.4 GHz x
2.22 ms), so one SIMD had GUI_ACTIVE / 8 cycles; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over
waves.  The VALU has no busy-cycle counter of its own: its time is priced at the rate the pure-VALU calibration kernel
sustains (cycles per v_fma_f32 at 4 waves per SIMD), transcendentals x 2.5 and conversions x 1.7 (tools/probes/inst_cost.hip)."""
import collections
import csv
import glob
import os
import sys

O = sys.argv[1]
SIMD = 1024


def load(tag, want):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(O, f"{tag}_p*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if want and want not in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, d in agg.items():
        m = {}
        for c, v in d.items():
            if tag == "probe":          # 1 warm-up launch (16 iterations) + 3 timed ones: keep the long ones
                v = sorted(v)[-max(1, len(v) // 4):]
            m[c] = sum(v) / len(v)
        out[k] = m
    return out


probe = load("probe", None)
cal = next((m for k, m in probe.items() if k.startswith("void k<3, 4, false, 0, 1024>")), None)
valu_cyc = (cal["GRBM_GUI_ACTIVE"] / 8 * SIMD / cal["SQ_INSTS_VALU"]) if cal else 2.74
print(f"calibration: a saturated SIMD issues one v_fma_f32 per {valu_cyc:.2f} cycles (pure-VALU kernel, 4 waves / SIMD); one "
      f"v_mfma_f32_16x16x32 holds the matrix pipe 16 cycles")
print()


def report(k, m):
    g = m.get("GRBM_GUI_ACTIVE")
    if not g or "SQ_INSTS_VALU" not in m:
        return
    cyc = g / 8                                       # cycles one SIMD had
    mfma = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / SIMD
    n_mfma = m.get("SQ_INSTS_MFMA", 0)
    tr, cv = m.get("SQ_INSTS_VALU_TRANS_F32", 0), m.get("SQ_INSTS_VALU_CVT", 0)
    plain = m["SQ_INSTS_VALU"] - n_mfma - tr - cv
    valu = (plain + 2.5 * tr + 1.7 * cv) / SIMD * valu_cyc
    co = m.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / SIMD
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k[:110])
    print(f"    kernel: {cyc:12.0f} cycles per SIMD ({cyc / 2.4e3:8.1f} us at 2.4 GHz), {m.get('SQ_WAVES', 0):.0f} waves")
    print(f"    matrix pipe busy  {100 * mfma / cyc:5.1f} %   ({n_mfma / SIMD:10.0f} MFMAs per SIMD x 16 cycles)")
    print(f"    VALU issue time   {100 * valu / cyc:5.1f} %   ({plain / SIMD:10.0f} plain + {tr / SIMD:9.0f} transcendental + {cv / SIMD:9.0f} conversion "
          f"instructions per SIMD)")
    print(f"    SUM               {100 * (mfma + valu) / cyc:5.1f} %   <- the two ADD on a SIMD (tools/probes/wave_spec_overlap.hip): this is the issue budget used")
    print(f"    MFMA cycles with a VALU instruction co-executing (SQ_VALU_MFMA_COEXEC_CYCLES): {100 * co / max(mfma, 1):5.1f} % of the MFMA-busy cycles")
    print(f"    of the waves' resident time: issuing {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:4.1f} %, stalled at issue "
          f"(SQ_WAIT_INST_ANY) {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:4.1f} %, parked at s_waitcnt / barrier (SQ_WAIT_ANY) "
          f"{100 * m.get('SQ_WAIT_ANY', 0) / wc:4.1f} %")
    print(f"    LDS instructions {m.get('SQ_INSTS_LDS', 0) / SIMD:9.0f}, SALU {m.get('SQ_INSTS_SALU', 0) / SIMD:9.0f} per SIMD; VALU-class instructions per MFMA "
          f"{m['SQ_INSTS_VALU'] / max(n_mfma, 1):.2f}")


print("================ calibration kernels (tools/probes/wave_spec_overlap.hip: k<MODE, VALU per MFMA, transcendental, s_setprio, threads>,")
print("                 MODE 0 mixed / 1 wave-specialised / 2 MFMA only / 3 VALU only)")
for k, m in probe.items():
    if ", 1024>" in k:
        report(k, m)
for tag in ("pubmed", "zinc"):
    print()
    print(f"================ fused loss, dense kernel: {tag} " + ("(N = 19 717)" if tag == "pubmed" else "(N = 95 000, a ZINC batch of 4096 molecules)"))
    for k, m in load(tag, "bce_dense_sym").items():
        if m.get("SQ_INSTS_VALU", 0) > 0:
            report(k, m)
