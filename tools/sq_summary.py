#!/usr/bin/env python
"""Per-kernel sums of one rocprofv3 --pmc pass over the eight SQ counters below (one pass holds at most 8 SQ counters).

usage (GPU box):  cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
                      SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace \
                      -d $R/gpurun_out/TAG_sq -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 \
                      --no-graph --no-cpu-baseline --no-strong-ref --roofline-iters 1
       (here):    python tools/sq_summary.py gpurun_out/TAG_sq > profiles/TAG_sq_counters.csv"""
import collections
import csv
import sys

C = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
     "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_INST_LDS"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
with open(f"{sys.argv[1]}/pmc_counter_collection.csv") as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches"] + C + ["mfma_busy_frac = MFMA_BUSY / (32 SIMDs per SE * BUSY_CYCLES)",
                                         "wait_any/wave_cycles", "wait_inst_any/wave_cycles"])
for k in sorted(acc, key=lambda k: -acc[k]["SQ_BUSY_CYCLES"]):
    a = acc[k]
    busy, wave = max(a["SQ_BUSY_CYCLES"], 1.0), max(a["SQ_WAVE_CYCLES"], 1.0)
    w.writerow([k, n[k]["SQ_WAVE_CYCLES"]] + [f"{a[c]:.0f}" for c in C] +
               [f"{a['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * busy):.3f}", f"{a['SQ_WAIT_ANY'] / wave:.3f}",
                f"{a['SQ_WAIT_INST_ANY'] / wave:.3f}"])
