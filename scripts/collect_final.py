#!/usr/bin/env python3
"""Turn gpurun_out/final/* into the committed artefacts under profiles/ (run on either side)."""
import csv, glob, json, os, shutil, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
tag = os.environ.get("ORT_ROUND", "r04")


def counters(sub):
    """{kernel name: {counter: mean value}} of the newest counter_collection.csv under gpurun_out/final/<sub>/."""
    fs = sorted(glob.glob(os.path.join(F, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


traffic = {}
for pol in ("fast", "ieee"):
    want = "k_trace<double, 1" if pol == "fast" else "k_trace<double, 0"
    vals = {}
    for ctr, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for k, d in counters(f"{ctr}_{pol}").items():
            if want in k and name in d:
                vals[ctr] = d[name]
    if "fetch" in vals and "write" in vals:
        # rocprofv3 reports KiB.  gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads
        # -> x2 (guide, HBM section); WRITE_SIZE is exact for 16-B-per-lane streaming stores.
        traffic[f"history_k1024_{pol}"] = {"FETCH_SIZE_KiB_raw": vals["fetch"], "WRITE_SIZE_KiB": vals["write"],
                                          "hbm_bytes_per_launch": vals["fetch"] * 1024.0 * 2.0 + vals["write"] * 1024.0,
                                          "source": f"scripts/final_profile.sh, round {tag}",
                                          "note": "FETCH_SIZE x2 (gfx950 half-count correction) + WRITE_SIZE, per launch, "
                                                  "separate --pmc passes"}
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
# full_trace routes: bytes per kernel and per ray (config 2: 9,437,184 rays)
ft = {}
for route in ("place", "lookback", "fused"):
    per = {}
    for ctr, name, mul in (("ft_fetch", "FETCH_SIZE", 2.0), ("ft_write", "WRITE_SIZE", 1.0)):
        for k, d in counters(f"{ctr}_{route}").items():
            if name in d and ("k_trace<" in k or "k_ft_" in k):
                per.setdefault(k.split("(")[0][:60], {})[name + "_bytes"] = d[name] * 1024.0 * mul
    tot = sum(sum(v.values()) for v in per.values())
    ft[route] = {"kernels": per, "total_bytes": tot, "bytes_per_ray": tot / 9437184.0,
                 "note": "FETCH_SIZE x2 (half-count correction; uncalibrated for the 8-B-per-lane reads of the second pass) + WRITE_SIZE"}
ftp = os.path.join(P, f"{tag}_pmc_full_trace_routes.json")
if os.path.exists(ftp):                       # keep the record of the superseded (spilling) look-back build
    old = json.load(open(ftp))
    for key in old:
        if key not in ft:
            ft[key] = old[key]
json.dump(ft, open(ftp, "w"), indent=1)
sq = {k: d for k, d in counters("sq_both").items() if "k_trace<" in k}
json.dump(sq, open(os.path.join(P, f"{tag}_pmc_sq_history_both_policies.json"), "w"), indent=1)
if os.path.exists(os.path.join(F, "bench.json")):
    shutil.copy(os.path.join(F, "bench.json"), os.path.join(P, f"{tag}_bench.json"))
stats = sorted(glob.glob(os.path.join(F, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:                      # newest run only (gpurun merges every run's files into gpurun_out/)
    shutil.copy(stats[-1], os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
b = json.loads([l for l in open(os.path.join(F, "bench.json")).read().strip().splitlines() if l.startswith("{")][-1])
print("bench:", b["value"], b["roofline"]["kernel_ms"], b["roofline"]["frac"], b.get("cpu_baseline", {}).get("value"))
for r in csv.DictReader(open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"))):
    if "k_trace" in r["Name"] or "k_ft" in r["Name"]:
        print("rocprof:", r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6, "ms")
print(json.dumps(traffic))
print(json.dumps({k: v["bytes_per_ray"] for k, v in ft.items()}))
for k, d in sq.items():
    print(k[:50], {c: round(v) for c, v in d.items() if c in ("SQ_INSTS_VALU", "SQ_WAVES")})
