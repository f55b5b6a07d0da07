#!/usr/bin/env python3
"""Turn gpurun_out/final/* into the committed artefacts under profiles/ (run on either side)."""
import csv, glob, json, os, shutil, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
tag = os.environ.get("ORT_ROUND", "r01")
traffic = {}
for pol in ("fast", "ieee"):
    vals = {}
    for ctr in ("fetch", "write"):
        fs = sorted(glob.glob(os.path.join(F, f"{ctr}_{pol}", "**", "*counter_collection.csv"), recursive=True),
                    key=os.path.getmtime)[-1:]
        if not fs:
            continue
        acc = []
        want = "k_trace<double, 1" if pol == "fast" else "k_trace<double, 0"
        for r in csv.DictReader(open(fs[0])):
            if want in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc.append(float(r["Counter_Value"]))
        if acc:
            vals[ctr] = sum(acc) / len(acc)
    if "fetch" in vals and "write" in vals:
        # rocprofv3 reports KiB.  gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads
        # -> x2 (guide, HBM section); WRITE_SIZE is exact for 16-B-per-lane streaming stores.
        fetch_b = vals["fetch"] * 1024.0 * 2.0
        write_b = vals["write"] * 1024.0
        traffic[f"history_k1024_{pol}"] = {"FETCH_SIZE_KiB_raw": vals["fetch"], "WRITE_SIZE_KiB": vals["write"],
                                          "hbm_bytes_per_launch": fetch_b + write_b,
                                          "note": "FETCH_SIZE x2 (gfx950 half-count correction) + WRITE_SIZE, per launch, "
                                                  "separate --pmc passes"}
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
for src, dst in (("bench.json", f"{tag}_bench.json"),):
    if os.path.exists(os.path.join(F, src)):
        shutil.copy(os.path.join(F, src), os.path.join(P, dst))
stats = sorted(glob.glob(os.path.join(F, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:                      # newest run only (gpurun merges every run's files into gpurun_out/)
    shutil.copy(stats[-1], os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
b = json.loads(open(os.path.join(F, "bench.json")).read().strip().splitlines()[-1])
print("bench:", b["value"], b["roofline"]["kernel_ms"], b["roofline"]["frac"], b.get("cpu_baseline", {}).get("value"))
for r in csv.DictReader(open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"))):
    if "k_trace" in r["Name"]:
        print("rocprof:", r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, "ms")
print(json.dumps(traffic))
