#!/usr/bin/env python3
"""The launches of the LAST of n forwards in a rocprofv3 kernel trace, in order: kernel, grid, microseconds.
Usage: parse_effnet_trace.py <kernel_trace.csv> <forwards in the trace>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // n
last = rows[-per:]
tot = 0.0
for r in last:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    name = r["Kernel_Name"]
    m = re.search(r"N12_GLOBAL__N_1\d+(\w+?)I", name)
    short = name if not m else m.group(1) + re.sub(r".*?kernel", "", name)[:28]
    if "(" in name: short = name.split("(")[0].replace("void (anonymous namespace)::", "")
    print(f"{short:60s} grid {int(r['Grid_Size_X']):>9d} x {int(r['Grid_Size_Y']):>3d}  wg {int(r['Workgroup_Size_X']):>4d}  lds {int(r.get('LDS_Block_Size', 0) or 0):>6d}  {us:9.1f} us")
print(f"{per} launches, {tot/1e3:.2f} ms of kernel time")
