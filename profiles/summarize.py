#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/run_profiles.sh (gpurun_out/prof_<tag>/) into the
small tracked summaries under profiles/<tag>/ and refresh profiles/traffic_latest.json.

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KB and come from
separate --pmc passes; on gfx950 FETCH_SIZE counts exactly half of a wide coalesced streaming
read, so it is doubled.  Usage: python profiles/summarize.py r01b
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", f"prof_{tag}")
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copyfile(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
for sub, out in (("trace_d2", "kernel_stats_dist2.csv"), ("trace_ladder", "kernel_stats_ladder.csv")):
    f = os.path.join(src, sub, "trace_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copyfile(f, os.path.join(dst, out))
for name in ("bench_trace.json", "bench_trace_d2.json", "ladder.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))

pmc = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    path = os.path.join(src, d, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"vk_\w+(<[^>]*>)?", r["Kernel_Name"])
        if m:
            agg[(m.group(0), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kern, ctr), v in agg.items():
        pmc.setdefault(kern, {})[ctr] = {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)}
json.dump(pmc, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)

ck = next((k for k in pmc if k.startswith("vk_count_dense_kernel")), None) or \
    next((k for k in pmc if k.startswith("vk_count_kernel")), None)
if ck and "FETCH_SIZE" in pmc[ck] and "WRITE_SIZE" in pmc[ck]:
    fetch = pmc[ck]["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2   # KB -> B, gfx950 x2
    write = pmc[ck]["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    cfg = None  # the configuration the counter passes ran (bench.py quotes the traffic only for the same one)
    try:
        line = [l for l in open(os.path.join(src, "bench_fetch.json")) if l.startswith("{")][-1]
        c = json.loads(line)["config"]
        cfg = {"k": c["k"], "samples": c["samples_per_gpu"], "reads": c["reads_per_sample"], "readlen": c["read_len"],
               "pool": c["distinct_samples_in_hbm"], "dist": c["base_distribution"]}
    except Exception:
        pass
    t = {"kernel": ck, "tag": tag, "config": cfg, "fetch_bytes_corrected": fetch, "write_bytes": write,
         "hbm_bytes_per_launch": fetch + write,
         "note": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024, separate --pmc passes, 1000-sample launch"}
    json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    json.dump(t, open(os.path.join("profiles", "traffic_latest.json"), "w"), indent=1)
    print(json.dumps(t))
print(open(os.path.join(dst, "kernel_stats.csv")).read()[:1500])
