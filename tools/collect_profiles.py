#!/usr/bin/env python
"""Copies the summaries of tools/profile_round.sh from gpurun_out/ into profiles/ (tracked):
  profiles/TAG_<model>_bench.json, TAG_<model>_kernel_stats.csv, TAG[_<model>]_pmc_summary.csv
usage: collect_profiles.py TAG [models...]"""
import glob, os, shutil, subprocess, sys
tag = sys.argv[1]
models = sys.argv[2:] or ["mnist", "multiscale", "tcvae", "wide"]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for m in models:
    src = f"{root}/gpurun_out/{tag}_{m}_bench.json"
    if os.path.exists(src):
        shutil.copy(src, f"{root}/profiles/{tag}_{m}_bench.json")
    # (gpurun merges every call's files into gpurun_out/: an earlier call's trace of the same tag lies beside the new one)
    newest = lambda paths: sorted(paths, key=os.path.getmtime)[-1:]
    st = newest(glob.glob(f"{root}/gpurun_out/{tag}_{m}_prof/*/*kernel_stats.csv"))
    if st:
        shutil.copy(st[0], f"{root}/profiles/{tag}_{m}_kernel_stats.csv")
    f, w = (newest(glob.glob(f"{root}/gpurun_out/{tag}_{m}_pmc_{k}/**/pmc_counter_collection.csv", recursive=True)) for k in ("fetch", "write"))
    if f and w:
        name = f"{tag}_pmc_summary.csv" if m == "mnist" else f"{tag}_{m}_pmc_summary.csv"
        out = subprocess.run([sys.executable, f"{root}/tools/pmc_summary.py", os.path.dirname(f[0]), os.path.dirname(w[0])],
                             capture_output=True, text=True, check=True).stdout
        open(f"{root}/profiles/{name}", "w").write(out)
    sq = newest(glob.glob(f"{root}/gpurun_out/{tag}_{m}_sq/**/pmc_counter_collection.csv", recursive=True))
    if sq:
        name = f"{tag}_sq_counters.csv" if m == "mnist" else f"{tag}_{m}_sq_counters.csv"
        out = subprocess.run([sys.executable, f"{root}/tools/sq_summary.py", os.path.dirname(sq[0])],
                             capture_output=True, text=True, check=True).stdout
        open(f"{root}/profiles/{name}", "w").write(out)
    print(m, bool(st), bool(f and w), bool(sq))
