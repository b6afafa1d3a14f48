#!/usr/bin/env python3
"""rocprofv3 --pmc passes of scripts/pmc_gemm_sq.sh -> <tag>_gemm_sq.json: per GEMM shape, per launch of gemm256p_kernel: the matrix pipe's busy
share and the clock the chip held -- "x of the 2.4 GHz issue peak" = (effective clock / 2400) x (MFMA busy share) x (issue efficiency).

    python scripts/parse_gemm_sq.py gpurun_out r04

MFMA busy share  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)       (busy cycles are per SIMD, summed)
effective clock  = GRBM_GUI_ACTIVE / 8 / kernel duration  (the guide: reads a few percent high on dispatches well under 1 ms)
floor            = SQ_INSTS_MFMA x 16 cycles / (4 SIMDs x ...): v_mfma_f32_16x16x32 issues back to back every 16 cycles
"""
import csv, glob, json, os, sys
from collections import defaultdict

root, tag = sys.argv[1], sys.argv[2]
FLOP = {"qkv": 2 * 126976 * 2304 * 768, "out": 2 * 126976 * 768 * 768, "fc1": 2 * 126976 * 3072 * 768, "fc2": 2 * 126976 * 768 * 3072}
out = {}
for sh in ("qkv", "out", "fc1", "fc2"):
    per = defaultdict(lambda: defaultdict(float))      # dispatch id -> counter -> value
    dur = {}
    for g in "AB":
        for f in glob.glob(os.path.join(root, f"{tag}_sq_{sh}_{g}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm256p" not in r["Kernel_Name"]:
                    continue
                key = (g, r["Dispatch_Id"])
                per[key][r["Counter_Name"]] += float(r["Counter_Value"])
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3      # us
        if not dur:
            for f in glob.glob(os.path.join(root, f"{tag}_sq_{sh}_{g}", "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "gemm256p" in r["Kernel_Name"]:
                        dur[(g, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not per:
        continue
    keys = sorted(per)
    keys = [k for k in keys if k in dur][2:] or keys      # skip the first two (cold) launches of each pass
    avg = defaultdict(list)
    for k in keys:
        for c, v in per[k].items():
            avg[(k[0], c)].append(v)
        avg[(k[0], "duration_us")].append(dur.get(k, float("nan")))
    m = {f"{g}.{c}": sum(v) / len(v) for (g, c), v in avg.items()}
    gui = m.get("A.GRBM_GUI_ACTIVE", float("nan"))
    d_us = m.get("A.duration_us", float("nan"))
    cyc = gui / 8.0                                     # shader cycles of the dispatch (the counter is the sum over the 8 XCDs)
    busy = m.get("A.SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
    n_mfma = m.get("A.SQ_INSTS_MFMA", float("nan"))
    wave = m.get("B.SQ_WAVE_CYCLES", float("nan"))
    out[sh] = {
        "launches_averaged": len([k for k in keys if k[0] == "A"]),
        "duration_us_under_pmc": round(d_us, 1),
        "tflops_under_pmc": round(FLOP[sh] / d_us / 1e6, 1),
        "effective_clock_mhz": round(cyc / d_us, 0),
        "mfma_busy_frac": round(busy / (cyc * 1024.0), 4),
        "mfma_insts": round(n_mfma), "mfma_insts_expected": round(FLOP[sh] / (2 * 16 * 16 * 32) / 1.0),
        "valu_active_frac": round(m.get("A.SQ_ACTIVE_INST_VALU", float("nan")) * 4 / (cyc * 1024.0), 4),
        "mfma_valu_coexec_frac": round(m.get("A.SQ_VALU_MFMA_COEXEC_CYCLES", float("nan")) / (cyc * 1024.0), 4),
        "wave_cycles_share": {"wait_any": round(m.get("B.SQ_WAIT_ANY", float("nan")) / wave, 4), "wait_inst_any": round(m.get("B.SQ_WAIT_INST_ANY", float("nan")) / wave, 4),
                              "wait_inst_lds": round(m.get("B.SQ_WAIT_INST_LDS", float("nan")) / wave, 4), "active_inst_any": round(m.get("B.SQ_ACTIVE_INST_ANY", float("nan")) / wave, 4)},
        "raw_millions": {k: round(v / 1e6, 2) for k, v in sorted(m.items()) if not k.endswith("duration_us")},
    }
    o = out[sh]
    print(f"{sh:4s} {o['duration_us_under_pmc']:7.1f} us  {o['tflops_under_pmc']:7.1f} TFLOP/s  clock {o['effective_clock_mhz']:.0f} MHz  MFMA busy {100 * o['mfma_busy_frac']:.1f} %  "
          f"VALU active {100 * o['valu_active_frac']:.1f} %  waits {o['wave_cycles_share']}")
tot = sum(o["duration_us_under_pmc"] * n for o, n in ((out[s], 12) for s in out))
if out:
    w = {s: out[s]["duration_us_under_pmc"] * 12 / tot for s in out}
    out["layer_gemms_time_weighted"] = {"mfma_busy_frac": round(sum(out[s]["mfma_busy_frac"] * w[s] for s in w), 4),
                                        "effective_clock_mhz": round(sum(out[s]["effective_clock_mhz"] * w[s] for s in w)),
                                        "how": "the four shapes weighted by their share of a step's GEMM time (12 launches each)"}
    print("time-weighted:", out["layer_gemms_time_weighted"])
# which kernel source these counters describe: bench.py quotes the file only next to the same hash (or says that it is stale)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avex_amd.build import kernel_source_sha16  # noqa: E402
out["gemm_source_sha16"] = kernel_source_sha16(("gemm.hip", "gemm_epi.h", "common.h"))
json.dump(out, open(os.path.join(root, f"{tag}_gemm_sq.json"), "w"), indent=1)
