#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: kernel, calls, average microseconds, share."""
import csv, sys
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        print(f"{n:40s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
