#!/usr/bin/env python3
"""One-off (GPU box): `python -m varkoder_amd image` on a ladder-shaped folder with 1 rank and with N ranks on cuda:0
(torchrun, gloo control plane): same PNG bytes, per-sample times = the sum of the ranks' per-file rows, the tail of the
size order pulled from the shared cursor.  python tools/cli_ranks_check.py [ranks] [samples]"""
import json
import os
import socket
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, ".")
import pandas as pd  # noqa: E402

from varkoder_amd import synth  # noqa: E402

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 12
root = os.getcwd()
tmp = Path(tempfile.mkdtemp(prefix="vk_ranks_"))
split = tmp / "int" / "split_fastqs"
split.mkdir(parents=True)
rungs = [150, 300, 750, 1500, 3000]
for i in range(nsamp):
    for kbp in rungs:
        data = synth.sample_fastq(500 + 7 * i + kbp % 5, kbp * 1000 // 150, 150, dist=1).tobytes()
        (split / f"tax{i:02d}_S@{kbp:08d}K.fq").write_bytes(data)
env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
common = ["image", str(tmp / "int"), "-k", "7", "-p", "cgr", "-n", "2"]
one = subprocess.run([sys.executable, "-m", "varkoder_amd"] + common + ["-o", str(tmp / "img1"), "-f", str(tmp / "stats1.csv")],
                     capture_output=True, text=True, cwd=root, env=env)
assert one.returncode == 0, one.stderr[-2000:]
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                       "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "varkoder_amd"] + common +
                      ["-o", str(tmp / "imgN"), "-f", str(tmp / "statsN.csv")], capture_output=True, text=True, cwd=root,
                     env=dict(env, VARKODER_AMD_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", VARKODER_AMD_PER_FILE_STATS=str(tmp / "pf")))
assert many.returncode == 0, many.stderr[-3000:]
a = sorted(p.name for p in (tmp / "img1").rglob("*.png"))
b = sorted(p.name for p in (tmp / "imgN").rglob("*.png"))
assert a == b and len(a) == nsamp * len(rungs), (len(a), len(b))
for name in a:
    assert next((tmp / "img1").rglob(name)).read_bytes() == next((tmp / "imgN").rglob(name)).read_bytes(), name
parts = [json.load(open(str(tmp / "pf") + f".rank{r}.json")) for r in range(ranks)]
assert sum(len(p) for p in parts) == len(a) and len(set().union(*[set(p) for p in parts])) == len(a)
sN = pd.read_csv(tmp / "statsN.csv").set_index("sample")
for col in ("7mer_counting_time", "k7_img_time"):
    for s in sN.index:
        want = sum(v[col] for p in parts for k, v in p.items() if k.split("@")[0] == s)
        assert abs(float(sN.loc[s, col]) - want) <= 1e-9 * max(1.0, want), (s, col)
print("ok:", len(a), "PNGs identical at 1 and", ranks, "ranks; files per rank", [len(p) for p in parts],
      "; samples whose rungs were split over ranks:", sum(1 for s in sN.index if sum(any(k.split('@')[0] == s for k in p) for p in parts) > 1))
