#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel family: mean counter value per launch.

    python tools/pmc_summary.py <dir with pass sub-directories> [more dirs] > summary.json
"""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_summary import family  # noqa: E402

out = {}
for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            a = acc[family(row["Kernel_Name"])][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    out[os.path.basename(d.rstrip("/"))] = {k: {c: {"launches": v[0], "mean_per_launch": v[1] / v[0]} for c, v in cs.items()}
                                            for k, cs in acc.items()}
json.dump(out, sys.stdout, indent=1)
