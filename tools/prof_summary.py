#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats CSV dir: per-kernel us/step and one step's launches."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
rows = list(csv.DictReader(open(glob.glob(d + '/*/*kernel_stats.csv')[0])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms/step %.3f" % (tot / steps / 1e6))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls/step {int(r['Calls'])/steps:5.1f} us/step {int(r['TotalDurationNs'])/steps/1e3:8.1f} avg {float(r['AverageNs'])/1e3:7.1f}")
if len(sys.argv) > 3:
    tr = list(csv.DictReader(open(glob.glob(d + '/*/*kernel_trace.csv')[0])))
    idx = [i for i, r in enumerate(tr) if r['Kernel_Name'].startswith('adam_step')]
    for r in tr[idx[-2] + 1: idx[-1] + 1]:
        print(f"{r['Kernel_Name'][:36]:36s} grid {r['Grid_Size_X']:>7s} vgpr {r['VGPR_Count']:>4s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}us")
