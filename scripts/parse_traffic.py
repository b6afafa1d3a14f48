#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) per kernel: average HBM bytes per launch.
gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly half the
bytes of wide (16 B/lane) coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores."""
import csv, glob, json, os, sys, collections
root, tag = sys.argv[1], sys.argv[2]
out = collections.defaultdict(lambda: {"launches": 0})
for name, sub, scale in (("fetch", f"{tag}_pmc_fetch", 2.0 * 1024), ("write", f"{tag}_pmc_write", 1.0 * 1024)):
    files = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
    per = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for short in ("gemm256p_kernel", "gemm256_kernel", "gemm_nt_kernel", "attention_kernel", "attention2_kernel", "layernorm_half_kernel", "layernorm_kernel", "posconv_kernel", "fbank_kernel", "mean_pool_kernel"):
                if short in k:
                    per[short].append(float(r["Counter_Value"]) * scale)
    for k, v in per.items():
        out[k][f"{name}_bytes_per_launch"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
res = {k: dict(v, hbm_bytes_per_launch=v.get("fetch_bytes_per_launch", 0) + v.get("write_bytes_per_launch", 0)) for k, v in out.items()}
path = os.path.join(root, f"{tag}_traffic.json")
json.dump(res, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
