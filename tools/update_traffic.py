#!/usr/bin/env python3
"""usage: python tools/update_traffic.py <tag>   -- after tools/final_profile_r03.sh <tag> has run on the GPU box.
Copies gpurun_out/r03/<tag>_* (bench lines, kernel stats, PMC summary) into profiles/r03/ and rewrites profiles/r03/traffic.json
(HBM bytes per launch of the headline kernel) from the PMC summary, keyed by the workload and by the hash of the kernel sources
it was measured on (code only: comments and blank lines do not count); bench.py quotes the figure only while both still match."""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "r03")
dst = os.path.join(ROOT, "profiles", "r03")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, f"{tag}_*")):
    if f.endswith((".json", ".csv", ".txt")):
        shutil.copy(f, os.path.join(dst, os.path.basename(f).replace("_pmc.txt", "_pmc_summary.txt")))
txt = open(os.path.join(src, f"{tag}_pmc.txt")).read()
val = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)=([0-9.e+]+)", txt)}
kern = txt.split(" SQ_")[0].strip()
b = json.load(open(os.path.join(src, f"{tag}_bench.json")))
key = "config3:k20:full"
fetch, write, miss = val["FETCH_SIZE"] * 1024, val["WRITE_SIZE"] * 1024, val["TCC_MISS_sum"]
entry = {"source_hash": bench.source_hash(), "kernel": kern, "hbm_bytes_per_launch": int(fetch + write),
         "launches_per_pass": b["roofline"]["launches_per_pass"],
         "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per launch (a pass over config 3 = {b['roofline']['launches_per_pass']} launches), "
                 f"profiles/r03/{tag}_pmc_summary.txt: FETCH_SIZE {val['FETCH_SIZE']:.4e} KB (not doubled: this kernel's table reads are 64-byte "
                 f"requests -- one region by LDS-DMA -- which gfx950 tallies in full; TCC_MISS {miss:.4e} x 64 B = {miss * 64:.3e} B agrees) + WRITE_SIZE {val['WRITE_SIZE']:.3e} KB"}
json.dump({key: entry}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(key, entry)
