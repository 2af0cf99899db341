"""Reads a rocprofv3 --kernel-trace csv of a bench run: per probe-kernel launch its duration and how much chain-kernel time ran
inside it; per chain kernel its duration alone (no probe kernel running) and beside a probe kernel."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    short = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"wrapped_(\w+?)_config", n)
    short = "rocprim:" + m.group(1) if m else re.sub(r"\(.*", "", short).replace("void ", "")[:40]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "")))
ev.sort()
t0 = ev[0][0]
probes = [(s, e) for s, e, n, q in ev if n.startswith("k_probe_mz")]
tagger = ("k_probe", "k_emit", "k_scan", "k_tile", "k_chain_carry", "k_pack_reads", "k_mz", "k_reads", "k_window", "k_genome", "k_table", "k_mark_unique",
          "k_group_loc", "k_group_flags", "k_head_rows", "k_iota", "k_loc_stats", "k_pack_rec", "k_codes", "k_sel", "k_count_w", "k_emit_w")
chain = [(s, e, n) for s, e, n, q in ev if not n.startswith(tagger)]


def overlap(s, e):
    tot = 0
    for ps, pe in probes:
        if pe <= s or ps >= e:
            continue
        tot += min(e, pe) - max(s, ps)
    return tot


alone, beside = defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0])
for s, e, n in chain:
    ov = overlap(s, e)
    d = alone if ov < (e - s) / 2 else beside
    d[n][0] += 1
    d[n][1] += e - s
print("chain kernels: calls / mean us ALONE | calls / mean us BESIDE a probe kernel")
for n in sorted(set(alone) | set(beside), key=lambda n: -(alone[n][1] + beside[n][1])):
    a, b = alone[n], beside[n]
    if a[1] + b[1] < 2e5:
        continue
    print(f"{n:42s} {a[0]:5d} {a[1] / max(a[0], 1) / 1e3:10.1f} | {b[0]:5d} {b[1] / max(b[0], 1) / 1e3:10.1f}")
print("\nprobe launches: start ms, duration ms, chain-kernel ms inside | the chain kernels that ran inside it (ms of overlap)")
for ps, pe in probes[-40:]:
    ov = defaultdict(float)
    for s, e, n in chain:
        if e > ps and s < pe:
            ov[n] += (min(e, pe) - max(s, ps)) / 1e6
    inside = sum(ov.values())
    top = ", ".join(f"{n} {v:.1f}" for n, v in sorted(ov.items(), key=lambda kv: -kv[1])[:7] if v >= 0.2)
    print(f"{(ps - t0) / 1e6:10.1f} {(pe - ps) / 1e6:8.2f} {inside / 1e6 * 1e6:8.2f} | {top}")
