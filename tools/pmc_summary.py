#!/usr/bin/env python
"""Per-kernel averages of the rocprofv3 --pmc passes made by tools/pmc_dominant.sh.

usage: pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_summary.csv
FETCH_SIZE / WRITE_SIZE are in KiB.  hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE counts
128-byte read requests at 64 bytes (MI355X_MICROARCH.md, HBM); checked here against launches whose read
set is known exactly (conv1 forward reads its 12,845 KiB input: FETCH_SIZE 6,443)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    with open(f"{d}/pmc_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches", "avg_FETCH_SIZE_KiB", "avg_WRITE_SIZE_KiB", "avg_hbm_bytes_per_launch(2*fetch+write)"])
rows = []
for k, c in acc.items():
    f, wr = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
    af = sum(f) / len(f) if f else 0.0
    aw = sum(wr) / len(wr) if wr else 0.0
    rows.append((k, max(len(f), len(wr)), af, aw, (2 * af + aw) * 1024))
for r in sorted(rows, key=lambda r: -r[4] * r[1]):
    w.writerow([r[0], r[1], f"{r[2]:.1f}", f"{r[3]:.1f}", f"{r[4]:.0f}"])
