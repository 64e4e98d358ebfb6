#!/usr/bin/env python3
"""Per-kernel averages of every counter in rocprofv3 counter_collection.csv files.  usage: pmc_table.py a.csv [b.csv ...]"""
import csv
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = sorted({c for k in tot for c in tot[k]})
print("kernel".ljust(22), "us".rjust(8), *[n.replace("SQ_", "")[:14].rjust(15) for n in names])
for k in sorted(tot, key=lambda k: -sum(dur[k]) / len(dur[k])):
    print(k[:22].ljust(22), f"{sum(dur[k]) / len(dur[k]) / 1e3:8.1f}", *[f"{tot[k][n] / max(cnt[k][n], 1):15.4g}" for n in names])
