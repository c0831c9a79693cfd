#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


print(f"# rocprofv3 summary, tag={tag}")
for f in find(os.path.join(out, f"{tag}_trace"), "*kernel_stats.csv"):
    print(f"## kernel stats ({os.path.basename(f)})")
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i > 8:
                break
            print(", ".join(c[:70] for c in row))
for d in sorted(glob.glob(os.path.join(out, f"{tag}_pmc*"))):
    for f in find(d, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"## PMC {os.path.basename(d)}")
        for k, ctrs in acc.items():
            for c, v in ctrs.items():
                print(f"{k:60s} {c:32s} n={len(v):4d} mean={sum(v) / len(v):.6g}")
