#!/usr/bin/env python3
"""Summaries of the k=9 (BASELINE config 4) profiles: python profiles/summarize_k9.py <tag_dist0> <tag_dist1> [<tag_dist2>]
reads gpurun_out/prof_<tag>/ (profiles/run_k9_pmc.sh), writes profiles/<tag>/ (kernel_stats.csv,
pmc_summary.json, traffic.json) and profiles/k9_latest.json, which bench.py's `config4` leg quotes when its
configuration matches.  HBM bytes: FETCH_SIZE (KB) x 2 (gfx950: a wide streaming read is tallied at half,
MI355X_MICROARCH.md "HBM") + WRITE_SIZE (KB), separate --pmc passes, summed over the spill path's kernels."""
import collections
import csv
import json
import os
import re
import shutil
import sys

legs = []
for tag in sys.argv[1:]:
    src = os.path.join("gpurun_out", f"prof_{tag}")
    dst = os.path.join("profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copyfile(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "bench_trace.json")):
        shutil.copyfile(os.path.join(src, "bench_trace.json"), os.path.join(dst, "bench_trace.json"))
    pmc = {}
    for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
        path = os.path.join(src, d, "pmc_counter_collection.csv")
        if not os.path.exists(path):
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            m = re.search(r"vk_\w+", r["Kernel_Name"])
            if m:
                agg[(m.group(0), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (kern, ctr), v in agg.items():
            # the warm-up launch and the timed one: per-dispatch means
            pmc.setdefault(kern, {})[ctr] = {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)}
    json.dump(pmc, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    kernel_ms = {}
    for r in csv.DictReader(open(os.path.join(dst, "kernel_stats.csv"))):
        m = re.search(r"vk_bucket\w*|vk_quad\w*|vk_check_kernel|vk_image\w*", r["Name"])
        if m:
            kernel_ms[m.group(0)] = float(r["AverageNs"]) / 1e6
    spill = [k for k in pmc if k.startswith(("vk_bucket", "vk_quad"))]
    fetch = sum(pmc[k].get("FETCH_SIZE", {}).get("mean_per_dispatch", 0.0) for k in spill) * 1024 * 2
    write = sum(pmc[k].get("WRITE_SIZE", {}).get("mean_per_dispatch", 0.0) for k in spill) * 1024
    cfg = None
    try:
        line = [l for l in open(os.path.join(src, "bench_fetch.json")) if l.startswith("{")][-1]
        c = json.loads(line)["config"]
        cfg = {"k": c["k"], "samples": c["samples_per_gpu"], "reads": c["reads_per_sample"], "readlen": c["read_len"],
               "pool": c["distinct_samples_in_hbm"], "dist": c["base_distribution"]}
    except Exception:
        pass
    t = {"tag": tag, "config": cfg, "kernel_ms": kernel_ms, "fetch_bytes_corrected": fetch, "write_bytes": write,
         "hbm_bytes_per_launch": fetch + write if fetch and write else None,
         "by_kernel": {k: {"fetch_bytes_corrected": pmc[k].get("FETCH_SIZE", {}).get("mean_per_dispatch", 0.0) * 2048,
                           "write_bytes": pmc[k].get("WRITE_SIZE", {}).get("mean_per_dispatch", 0.0) * 1024} for k in spill}}
    json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    legs.append(t)
    print(json.dumps(t))
json.dump({"legs": legs, "note": "per-launch HBM bytes and rocprofv3 kernel averages of the k=9 spill path; "
                                 "bench.py config4 quotes them for a matching configuration"},
          open(os.path.join("profiles", "k9_latest.json"), "w"), indent=1)
