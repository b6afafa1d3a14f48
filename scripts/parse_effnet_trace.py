#!/usr/bin/env python3
"""The launches of one forward (the last complete one) in a rocprofv3 kernel trace of scripts/effnet_bench.py, in order: kernel, grid,
microseconds.  A forward starts at the STFT kernel of the mel frontend.
Usage: parse_effnet_trace.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "stft_fft_kernel" in r["Kernel_Name"] or "melspec_kernel" in r["Kernel_Name"]]
# the min/max initialisation kernel is launched just before the STFT kernel: include it
a, b = starts[-2] - 1, starts[-1] - 1
tot = 0.0
for r in rows[a:b]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    name = r["Kernel_Name"]
    m = re.search(r"N12_GLOBAL__N_1\d+(\w+?)(I.*)?$", name)
    if m:
        short = m.group(1) + re.sub(r"EEvN.*|EvPK.*|EEEvP.*|EvN.*", "", m.group(2) or "")
    else:
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", "")).replace(" ", "") or "?"
    print(f"{short[:58]:58s} grid {int(r['Grid_Size_X']):>8d} x {int(r['Grid_Size_Y']):>3d} x {int(r['Grid_Size_Z']):>2d}  wg {int(r['Workgroup_Size_X']):>4d}  {us:9.1f} us")
print(f"{b - a} launches, {tot/1e3:.2f} ms of kernel time")
