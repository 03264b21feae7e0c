#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output: mean counter value per launch for every kernel whose name contains a filter.
usage: pmc_summary.py <dir> <kernel-substring>"""
import csv
import glob
import re
import sys
from collections import defaultdict

root, filt = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fp:
        for r in csv.DictReader(fp):
            name = r.get("Kernel_Name", "")
            if filt not in name:
                continue
            short = re.sub(r"\(.*", "", name)[:70]
            acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:70s} {c:32s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
