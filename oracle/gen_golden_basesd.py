#!/usr/bin/env python3
"""Golden vectors for the base-frequency quality flag, made by running the reference's own
get_basefrequency_sd (varKoder/commands/image.py:45-88) unmodified on small fastp-style JSON
reports written here.  Build-container only (needs /root/reference).  Writes
tests/golden/basesd_cases.json: the report contents (inputs, data) and the value the reference
returned for each list of reports.

Usage:  python oracle/gen_golden_basesd.py
"""
import json
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ref_harness  # noqa: E402

ref_harness.install()
from varKoder.commands.image import get_basefrequency_sd  # noqa: E402


def curves(rng, n, wobble):
    """content_curves of one fastp section: per-cycle base fractions (plus N and GC, which the
    reference ignores)."""
    base = rng.dirichlet([8, 8, 8, 8], size=1)[0]
    cur = {b: (base[i] + wobble * rng.standard_normal(n)).round(6).tolist() for i, b in enumerate("ATCG")}
    cur["N"] = (0.001 * rng.random(n)).round(6).tolist()
    cur["GC"] = (np.array(cur["G"]) + np.array(cur["C"])).round(6).tolist()
    return cur


def main():
    rng = np.random.default_rng(20250824)
    reports = {
        "paired_merged.json": {"merged_and_filtered": {"content_curves": curves(rng, 150, 0.004)},
                               "read1_after_filtering": {"content_curves": curves(rng, 150, 0.02)}},
        "unpaired_only.json": {"read1_after_filtering": {"content_curves": curves(rng, 100, 0.03)}},
        "merged_only.json": {"merged_and_filtered": {"content_curves": curves(rng, 60, 0.001)}},
        "low_quality.json": {"read1_after_filtering": {"content_curves": curves(rng, 150, 0.08)}},
    }
    lists = [["paired_merged.json"], ["unpaired_only.json"], ["merged_only.json"], ["low_quality.json"],
             ["unpaired_only.json", "low_quality.json"],   # only the FIRST report counts (the return sits inside the loop)
             ["low_quality.json", "merged_only.json"]]
    cases = []
    with tempfile.TemporaryDirectory(prefix="golden_basesd_") as tmp:
        for name, js in reports.items():
            with open(Path(tmp) / name, "w") as f:
                json.dump(js, f)
        for names in lists:
            val = get_basefrequency_sd([Path(tmp) / n for n in names])
            cases.append({"files": names, "base_sd": float(val)})
            print(names, float(val))
        empty = get_basefrequency_sd([])
        print("empty list ->", empty)
    out = {"generator": "oracle/gen_golden_basesd.py", "reports": reports, "cases": cases,
           "empty_list_returns": None if empty is None else float(empty)}
    with open(ROOT / "tests" / "golden" / "basesd_cases.json", "w") as f:
        json.dump(out, f)
    print("wrote tests/golden/basesd_cases.json")


if __name__ == "__main__":
    main()
