#!/usr/bin/env python3
"""Probe (GPU box): can a file's page-cache pages be registered with HIP (mmap + hipHostRegister) and copied to the device
without the read() copy into a pinned buffer?  Prints registration and copy rates next to the staged route."""
import ctypes as C
import mmap
import os
import sys
import tempfile
import time

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = 179 << 20
tmp = tempfile.mkdtemp(prefix="vk_hostreg_")
blob = np.random.default_rng(0).integers(32, 127, size=size, dtype=np.uint8).tobytes()
files = []
for i in range(n):
    p = os.path.join(tmp, f"f{i}.fq")
    with open(p, "wb") as f:
        f.write(blob)
    files.append(p)
dev = torch.empty(n * size, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()

# staged route: readinto a pinned buffer, then one copy
pinned = torch.empty(n * size, dtype=torch.uint8).pin_memory()
pv = memoryview(pinned.numpy())
t0 = time.perf_counter()
for i, p in enumerate(files):
    with open(p, "rb", buffering=0) as f:
        f.readinto(pv[i * size:(i + 1) * size])
t1 = time.perf_counter()
dev.copy_(pinned, non_blocking=True)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"staged (1 thread): read {n * size / (t1 - t0) / 1e9:.1f} GB/s, copy {n * size / (t2 - t1) / 1e9:.1f} GB/s")

# registered route
maps = []
t0 = time.perf_counter()
rc_all = []
for p in files:
    fd = os.open(p, os.O_RDONLY)
    m = mmap.mmap(fd, size, flags=mmap.MAP_SHARED | mmap.MAP_POPULATE, prot=mmap.PROT_READ)
    os.close(fd)
    addr = C.addressof(C.c_char.from_buffer_copy(b"x"))  # placeholder
    buf = np.frombuffer(m, dtype=np.uint8)
    addr = buf.ctypes.data
    rc = hip.hipHostRegister(C.c_void_p(addr), size, 0)
    rc_all.append(rc)
    maps.append((m, buf, addr))
t1 = time.perf_counter()
print("hipHostRegister rc:", sorted(set(rc_all)), f"map+register {n * size / (t1 - t0) / 1e9:.1f} GB/s (1 thread)")
if all(r == 0 for r in rc_all):
    t1 = time.perf_counter()
    for i, (m, buf, addr) in enumerate(maps):
        hip.hipMemcpyAsync(C.c_void_p(dev.data_ptr() + i * size), C.c_void_p(addr), size, 1, None)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"registered copy {n * size / (t2 - t1) / 1e9:.1f} GB/s; equal to file: {bool((dev[:size].cpu().numpy() == np.frombuffer(blob, dtype=np.uint8)).all())}")
    t3 = time.perf_counter()
    for m, buf, addr in maps:
        hip.hipHostUnregister(C.c_void_p(addr))
    print(f"unregister {n * size / (time.perf_counter() - t3) / 1e9:.1f} GB/s")
import shutil
shutil.rmtree(tmp, ignore_errors=True)
