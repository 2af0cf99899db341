#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel name."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if not k.startswith(tuple(__import__("os").environ.get("PMC_KERNELS", "k_probe").split(","))):
                continue
            dur = None
            print(k, " ".join(f"{c}={sum(v)/len(v):.5g}" for c, v in sorted(cs.items())), f"(n={len(next(iter(cs.values())))})")
