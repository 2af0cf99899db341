#!/usr/bin/env python3
"""usage: python tools/update_traffic.py [round dir, default r04]   -- after tools/final_profile_r04.sh a / b have run on the GPU box.
Copies gpurun_out/<round>/fin_* (bench lines, kernel stats, PMC summaries) into profiles/<round>/ and rewrites
profiles/<round>/traffic.json: HBM bytes per launch of the probe kernel for every workload that was profiled, keyed by the
workload and by the hash of the kernel sources it was measured on (code only: comments and blank lines do not count);
bench.py quotes a figure only while both still match."""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", rnd)
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
KEYS = {"config3": "config3:k20:full", "diploid": "diploid:k20:full", "config5_k21": "config5:k21:full", "config5_k25": "config5:k25:full"}
out = {}
for tag, key in KEYS.items():
    pmc = os.path.join(src, f"fin_{tag}_pmc.txt")
    bj = os.path.join(src, f"fin_{tag}_bench.json")
    if not (os.path.exists(pmc) and os.path.exists(bj)):
        continue
    for f in glob.glob(os.path.join(src, f"fin_{tag}_*")):
        if f.endswith((".json", ".csv", ".txt")):
            shutil.copy(f, os.path.join(dst, os.path.basename(f).replace("_pmc.txt", "_pmc_summary.txt")))
    txt = open(pmc).read()
    val = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)=([0-9.e+]+)", txt)}
    kern = txt.split(" SQ_")[0].strip()
    b = json.loads(open(bj).read().strip().splitlines()[0])
    fetch, write, miss = val["FETCH_SIZE"] * 1024, val["WRITE_SIZE"] * 1024, val["TCC_MISS_sum"]
    out[key] = {"source_hash": bench.source_hash(), "kernel": kern, "hbm_bytes_per_launch": int(fetch + write),
                "launches_per_pass": b["roofline"]["launches_per_pass"], "tcc_req_per_launch": val.get("TCC_REQ_sum"), "tcc_miss_per_launch": miss,
                "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per launch (a pass = {b['roofline']['launches_per_pass']} launches), "
                        f"profiles/{rnd}/fin_{tag}_pmc_summary.txt: FETCH_SIZE {val['FETCH_SIZE']:.4e} KB (not doubled: this kernel's table reads are 64-byte "
                        f"requests -- one region by LDS-DMA -- which gfx950 tallies in full; TCC_MISS {miss:.4e} x 64 B = {miss * 64:.3e} B agrees) + WRITE_SIZE {val['WRITE_SIZE']:.3e} KB"}
    print(key, out[key]["hbm_bytes_per_launch"], "alg", b["roofline"]["alg_bytes_per_launch"])
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
