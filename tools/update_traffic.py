#!/usr/bin/env python3
"""usage: python tools/update_traffic.py <tag>   -- after tools/final_profile_r02.sh <tag> has run on the GPU box.
Copies gpurun_out/r02/<tag>_{bench.json,bench_under_rocprof.json,kernel_stats.csv,pmc.txt} into profiles/r02/ and rewrites
profiles/r02/traffic.json (HBM bytes per launch of the headline kernel) from the PMC summary, keyed by the workload and by the
hash of the kernel sources it was measured on (code only: comments and blank lines do not count) (bench.py quotes the figure only while both still match)."""
import json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "r02")
dst = os.path.join(ROOT, "profiles", "r02")
for suffix, name in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                     ("kernel_stats.csv", "kernel_stats.csv"), ("pmc.txt", "pmc_summary.txt")):
    shutil.copy(os.path.join(src, f"{tag}_{suffix}"), os.path.join(dst, f"{tag}_{name}"))
txt = open(os.path.join(src, f"{tag}_pmc.txt")).read()
val = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)=([0-9.e+]+)", txt)}
kern = txt.split(" SQ_")[0].strip()
b = json.load(open(os.path.join(src, f"{tag}_bench.json")))
wl = b["config"]
key = f"config3:3100Mb:{wl['reads_per_gpu']}reads:seed30"  # bench.py defaults (genome 3100 Mb, seed 30)
fetch, write, miss = val["FETCH_SIZE"] * 1024, val["WRITE_SIZE"] * 1024, val["TCC_MISS_sum"]
entry = {"source_hash": bench.source_hash(), "kernel": kern, "hbm_bytes_per_launch": int(fetch + write),
         "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/r02/{tag}_pmc_summary.txt: FETCH_SIZE {val['FETCH_SIZE']:.4e} KB "
                 f"(not doubled: this kernel's table reads are 64-byte requests -- one region, LDS-DMA -- which gfx950 tallies in full; "
                 f"TCC_MISS {miss:.4e} x 64 B = {miss * 64:.3e} B agrees) + WRITE_SIZE {val['WRITE_SIZE']:.3e} KB"}
json.dump({key: entry}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(key, entry)
