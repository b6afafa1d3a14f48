#!/usr/bin/env python3
"""rocprofv3 --pmc passes of scripts/pmc_gemm_row.sh -> <tag>_row_sq.json: gemm256p_kernel (variant 5) beside gemm_row_kernel (variant 8) on the
attention output projection (K 768 -> N 768) and on fc2's shape (K 3072 -> N 768), per launch: duration, matrix pipe busy share, the clock
held, vector instructions, LDS bank conflicts, where the waves' cycles go.
    python scripts/parse_row_sq.py gpurun_out r06"""
import csv, glob, json, os, sys
from collections import defaultdict
root, tag = sys.argv[1], sys.argv[2]
FLOP = {"out": 2.0 * 126976 * 768 * 768, "fc2": 2.0 * 126976 * 768 * 3072}
out = {}
for v in (5, 8):
    kname = "gemm256p_kernel" if v == 5 else "gemm_row_kernel"
    for sh in ("out", "fc2"):
        m = defaultdict(list)
        for g in "AB":
            d = os.path.join(root, f"{tag}_row_sq_v{v}_{sh}_{g}")
            dur = {}
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kname in r["Kernel_Name"]:
                        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                per = defaultdict(float)
                for r in csv.DictReader(open(f)):
                    if kname in r["Kernel_Name"]:
                        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
                for (did, c), val in per.items():
                    m[g + "." + c].append(val)
                    if c == "GRBM_GUI_ACTIVE" and did in dur:
                        m[g + ".duration_us"].append(dur[did])
        if not m:
            continue
        a = {k: sum(x[2:]) / max(len(x[2:]), 1) for k, x in m.items()}      # (the first launches warm up)
        d_us = a.get("A.duration_us", float("nan")); cyc = a.get("A.GRBM_GUI_ACTIVE", float("nan")) / 8.0; wave = a.get("B.SQ_WAVE_CYCLES", float("nan"))
        out[f"v{v}.{sh}"] = {
            "kernel": kname, "duration_us_under_pmc": round(d_us, 1), "tflops_under_pmc": round(FLOP[sh] / d_us / 1e6, 1),
            "effective_clock_mhz": round(cyc / d_us), "mfma_busy_frac": round(a.get("A.SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (cyc * 1024.0), 4),
            "valu_insts_millions": round(a.get("A.SQ_INSTS_VALU", float("nan")) / 1e6, 2), "mfma_insts_millions": round(a.get("A.SQ_INSTS_MFMA", float("nan")) / 1e6, 2),
            "lds_insts_millions": round(a.get("B.SQ_INSTS_LDS", float("nan")) / 1e6, 2), "lds_bank_conflict_cycles_millions": round(a.get("A.SQ_LDS_BANK_CONFLICT", float("nan")) / 1e6, 2),
            "wave_cycles_share": {k: round(a.get("B." + c, float("nan")) / wave, 4) for k, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"),
                                                                                             ("wait_inst_lds", "SQ_WAIT_INST_LDS"), ("active_inst_any", "SQ_ACTIVE_INST_ANY"))}}
        o = out[f"v{v}.{sh}"]
        print(f"variant {v} {sh:4s} {o['duration_us_under_pmc']:7.1f} us {o['tflops_under_pmc']:7.1f} TF/s clock {o['effective_clock_mhz']} MHz MFMA busy {100 * o['mfma_busy_frac']:.1f} % "
              f"VALU {o['valu_insts_millions']} M LDS conflicts {o['lds_bank_conflict_cycles_millions']} Mcyc waits {o['wave_cycles_share']}")
json.dump(out, open(os.path.join(root, f"{tag}_row_sq.json"), "w"), indent=1)
