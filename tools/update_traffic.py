#!/usr/bin/env python3
"""usage: python tools/update_traffic.py [round dir, default r05]   -- after tools/final_profile.sh <round> a / b / c / chain have run on the GPU box.
Copies gpurun_out/<round>/fin_* (bench lines, kernel stats, PMC summaries) into profiles/<round>/ and rewrites
profiles/<round>/traffic.json: HBM bytes per launch of the probe kernel for every workload that was profiled, keyed by the
workload and by the hash of the kernel sources it was measured on (code only: comments and blank lines do not count);
bench.py quotes a figure only while both still match.  Also writes profiles/<round>/README.md: one row per workload with the
numbers of the files beside it (so that the table cannot go stale against them)."""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", rnd)
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
KEYS = {"config3": "config3:k20:full", "diploid": "diploid:k20:full", "config5_k21": "config5:k21:full", "config5_k25": "config5:k25:full",
        "config3_group_len_200": "config3:k20:group_len_200"}
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

# ---- the second half's files, and the round's README ----
for f in glob.glob(os.path.join(src, "chain_*")) + glob.glob(os.path.join(src, "trace_overlap.txt")):
    if f.endswith((".json", ".csv", ".txt", ".log")):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
import csv

rows = []
for tag, key in KEYS.items():
    bj, ks = os.path.join(dst, f"fin_{tag}_bench.json"), os.path.join(dst, f"fin_{tag}_kernel_stats.csv")
    if not (os.path.exists(bj) and os.path.exists(ks) and key in out):
        continue
    b = json.loads(open(bj).read().strip().splitlines()[0])
    r = b["roofline"]
    probe = [x for x in csv.DictReader(open(ks)) if "k_probe" in x["Name"]]
    prof_ms = float(probe[0]["AverageNs"]) / 1e6 if probe else float("nan")
    rows.append(f"| `{tag}` | {b['value'] / 1e9:.1f} | {b['ms_per_step']:.1f} | {r['kernel_ms']:.3f} | {prof_ms:.3f} ({probe[0]['Calls'] if probe else 0} calls) | "
                f"{r['alg_bytes_per_launch'] / 1e9:.2f} | {r['frac']:.4f} | {r['alg_bytes_per_launch'] / (prof_ms * 1e-3) / 8e12:.4f} | "
                f"{out[key]['hbm_bytes_per_launch'] / 1e9:.2f} = {out[key]['hbm_bytes_per_launch'] / r['alg_bytes_per_launch']:.2f} x |")
with open(os.path.join(dst, "README.md"), "w") as f:
    f.write(f"# profiles/{rnd} -- written by tools/update_traffic.py from the files beside it\n\n"
            "Probe kernel (`k_probe_mz`), one launch = one resident device batch (8 per pass).  `fin_<w>_bench.json` = the workload's own bench line "
            "(HIP-event kernel time), `fin_<w>_kernel_stats.csv` = `rocprofv3 --kernel-trace --stats` of a short run of the same command, "
            "`fin_<w>_pmc_summary.txt` = SQ / TCC counters (separate `--pmc` passes), `traffic.json` = HBM bytes per launch keyed by workload and kernel-source hash.\n\n"
            "| workload | Gbase/s | ms per step | kernel ms per launch (HIP events) | kernel ms per launch (rocprofv3 average) | algorithmic GB per launch | frac (HIP events) | frac (rocprofv3) | HBM traffic per launch (PMC) |\n"
            "|---|---|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n\n"
            "Second half: `chain_config3_kernel_stats.csv` (a run with the `pair_stage` / `engine_both_halves` / `cli_config3_rows` legs), `chain_pmc.txt` "
            "(counters of its largest kernels), `trace_overlap*.txt` (which chain kernels ran beside probe kernels, and what each cost the other).\n")
    # ---- the reader side: BGZF inflate and the record parser (tools/bgzf_profile.sh) ----
    lines = []
    for lv in (1, 6):
        ks = os.path.join(dst, f"bgzf_level{lv}_kernel_stats.csv")
        if os.path.exists(ks):
            for x in csv.DictReader(open(ks)):
                if "k_bgzf" in x["Name"] or "k_count_newlines" in x["Name"]:
                    nm = (x["Name"][len("(anonymous namespace)::"):] if x["Name"].startswith("(anonymous namespace)::") else x["Name"]).split("(")[0]
                    lines.append(f"| level {lv} | `{nm}` | {x['Calls']} | {float(x['TotalDurationNs']) / 1e6:.1f} | {float(x['AverageNs']) / 1e6:.2f} | {float(x['MaxNs']) / 1e6:.2f} |")
    ks = os.path.join(dst, "ingest_kernel_stats.csv")
    if os.path.exists(ks):
        for x in csv.DictReader(open(ks)):
            nm = (x["Name"][len("(anonymous namespace)::"):] if x["Name"].startswith("(anonymous namespace)::") else x["Name"]).split("(")[0]
            if nm.startswith(("k_nl_", "k_line", "k_fastq", "k_records", "k_gather_lines", "k_pack_reads")):
                lines.append(f"| parser, 1.2 GB block | `{nm}` | {x['Calls']} | {float(x['TotalDurationNs']) / 1e6:.2f} | {float(x['AverageNs']) / 1e6:.3f} | {float(x['MaxNs']) / 1e6:.3f} |")
    if lines:
        f.write("\nReader side (`tools/bgzf_profile.sh`): `bgzf_level{1,6}_kernel_stats.csv` = `rocprofv3 --kernel-trace --stats` of `tools/bgzf_time.py 10 <level>` "
                "(3 passes over a 10 GB text file in bgzip's container, blocks of 0.4 ... 3.5 GB of text, one launch each; `bgzf_level*_time.txt` = the same run's "
                "per-block times without the profiler), `ingest_kernel_stats.csv` = the record parser on one 1.2 GB FASTQ block, three times.\n\n"
                "| run | kernel | calls | total ms | average ms | max ms |\n|---|---|---|---|---|---|\n" + "\n".join(lines) + "\n")
    extra = os.path.join(dst, "NOTES.md")
    if os.path.exists(extra):
        f.write("\n" + open(extra).read())
print("\n".join(rows))
