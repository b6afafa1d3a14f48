#!/bin/bash
# HBM-side traffic of the EfficientNet-B0 forward (BASELINE config C5) from the PMC counters: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
# passes (kernel-trace only), program directly after `--`.  Writes gpurun_out/r05_effnet_traffic.json: bytes per clip, per kernel family and in
# total, beside bench.py's algorithmic 71.94 MB per clip.  gfx950 corrections as for the GEMM (MI355X_MICROARCH.md, HBM): both counters are in
# KiB; FETCH_SIZE reports half the bytes of wide (16 B / lane) coalesced reads and is doubled -- an UPPER bound for kernels that read narrower
# (the depthwise taps): the file gives the undoubled figure too.
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-256}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_effnet_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_effnet_$c -- python3 $R/scripts/effnet_bench.py $B > $R/gpurun_out/pmc_effnet_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json, re
B = $B
fam = lambda k: next((n for n in ("stft_fft_kernel", "melspec_norm_kernel", "melspec_kernel", "mbconv_kernel", "dwconv_lds_kernel", "dwconv_kernel", "se_pool_fc_kernel",
                                  "gemm_skinny_kernel", "gemm_nt_kernel", "gemm256p_kernel", "stem_conv", "nhwc", "pool") if n in k), re.sub(r"<.*", "", k.split("(")[0])[-40:])
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(float); disp = collections.defaultdict(int); fwd = 0
    for f in glob.glob("$R/gpurun_out/pmc_effnet_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != c: continue
            k = fam(row["Kernel_Name"])
            per[k] += float(row["Counter_Value"]) * 1024.0; disp[k] += 1
            if "stft_fft_kernel" in row["Kernel_Name"]: fwd += 1
    tot[c] = (per, disp, max(fwd, 1))
fwd = tot["FETCH_SIZE"][2]
out = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} -- python3 scripts/effnet_bench.py %d" % B, "clips_per_forward": B, "forwards_profiled": fwd,
       "corrections": "counters in KiB; fetch_bytes = 2 x FETCH_SIZE (gfx950: wide coalesced reads are reported at half), fetch_bytes_undoubled beside it; WRITE_SIZE as is",
       "per_kernel_family_mb_per_clip": {}}
F = W = 0.0
for k in sorted(set(tot["FETCH_SIZE"][0]) | set(tot["WRITE_SIZE"][0])):
    f = tot["FETCH_SIZE"][0].get(k, 0.0) / fwd / B; w = tot["WRITE_SIZE"][0].get(k, 0.0) / fwd / B
    out["per_kernel_family_mb_per_clip"][k] = {"fetch": round(2 * f / 1e6, 3), "fetch_undoubled": round(f / 1e6, 3), "write": round(w / 1e6, 3), "launches_per_forward": round(tot["FETCH_SIZE"][1].get(k, 0) / fwd, 1)}
    F += f; W += w
out["counted_mb_per_clip"] = round((2 * F + W) / 1e6, 2)
out["counted_mb_per_clip_fetch_undoubled"] = round((F + W) / 1e6, 2)
out["algorithmic_mb_per_clip"] = 71.94
out["counted_over_algorithmic"] = round((2 * F + W) / 1e6 / 71.94, 3)
json.dump(out, open("$R/gpurun_out/r05_effnet_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -2 $R/gpurun_out/pmc_effnet_FETCH_SIZE.log
