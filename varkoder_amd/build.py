"""Compile the HIP extension in-tree: varkoder_amd/libvkimg_hip.so (gfx950 only)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "vkimg.hip")
OUT = os.path.join(HERE, "libvkimg_hip.so")
INC = os.path.join(ROOT, "include")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIP extension cannot be built")


def needs_build():
    if not os.path.exists(OUT):
        return True
    csrc = os.path.join(HERE, "csrc")
    deps = [os.path.join(INC, "vkimg.h")] + [os.path.join(csrc, f) for f in os.listdir(csrc)
                                             if f.endswith((".hip", ".h"))]
    return os.path.getmtime(OUT) < max(os.path.getmtime(d) for d in deps)


LAST = {}  # what the last build_hip() call did: {"action": "compiled" | "reused", "so": path, "so_mtime": ..., "seconds": ...}


def compile_cmd(out):
    return [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-I", INC, SRC, "-o", out]


def verify_compiles():
    """Compile the tree as it stands to a temporary file (the library in place is left alone): seconds taken and
    the size of what came out.  Proof that the sources build, for a run that would otherwise reuse the binary."""
    import tempfile
    import time
    with tempfile.TemporaryDirectory(prefix="vkbuild_") as tmp:
        out = os.path.join(tmp, "libvkimg_hip.so")
        t0 = time.time()
        subprocess.check_call(compile_cmd(out))
        return {"seconds": time.time() - t0, "bytes": os.path.getsize(out)}


def build_hip(force=False, verbose=False, verify=False):
    """The library, compiled when it is missing or older than a source (or with force).  verify: when the
    binary in place is current, compile the tree to a temporary path anyway and record that it built."""
    import time
    if not force and not needs_build():
        LAST.update(action="reused", so=OUT, so_mtime=os.path.getmtime(OUT), seconds=0.0)
        if verify:
            v = verify_compiles()
            LAST.update(action="reused+verified", seconds=v["seconds"], verified_bytes=v["bytes"])
        return OUT
    t0 = time.time()
    cmd = compile_cmd(OUT)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    LAST.update(action="compiled", so=OUT, so_mtime=os.path.getmtime(OUT), seconds=time.time() - t0)
    return OUT


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
