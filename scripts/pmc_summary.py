#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_<tag>/<pass>/**/_counter_collection.csv) per kernel."""
import csv, glob, sys, collections, json
tag = sys.argv[1]
root = f"gpurun_out/pmc_{tag}"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        if "k_trace<" not in k and "k_ft_" not in k:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in acc.items():
    out[k] = {c: sum(v) / len(v) for c, v in d.items()}
    out[k]["_dispatches"] = max(len(v) for v in d.values())
print(json.dumps(out, indent=1))
