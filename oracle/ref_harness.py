"""Import shim for the read-only reference at /root/reference -- build container only.

Used by oracle/gen_golden.py to run the reference's own make_image /
get_kmer_mapping / get_cgr UNMODIFIED and record golden vectors.  Nothing here
ships reference code: it only imports it in place.  /root/reference does not
exist on the GPU box, so no test, smoke() or bench imports this module.

Recipe: three in-memory stubs for packages that are not
installed and are not on the arithmetic path (humanfriendly, tenacity, the
installed-package version lookup), plus a fake `dsk2ascii` on PATH that cats
the `-file` argument, so make_image() (commands/image.py:808-936) parses the
`KMER count` text we wrote at the .h5 path.
"""
import importlib.metadata
import os
import stat
import sys
import tempfile
import types

REF_ROOT = "/root/reference"


def install():
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present; golden vectors can only be regenerated "
                           "in the build container")
    sys.dont_write_bytecode = True
    if "humanfriendly" not in sys.modules:
        hf = types.ModuleType("humanfriendly")

        def parse_size(s):
            s = str(s).strip().upper().rstrip("B")
            mult = {"K": 10 ** 3, "M": 10 ** 6, "G": 10 ** 9, "T": 10 ** 12}
            if s and s[-1] in mult:
                return int(float(s[:-1]) * mult[s[-1]])
            return int(float(s))
        hf.parse_size = parse_size
        sys.modules["humanfriendly"] = hf
    if "tenacity" not in sys.modules:
        tn = types.ModuleType("tenacity")

        class _Attempt:
            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

        def Retrying(**kw):
            yield _Attempt()
        tn.Retrying = Retrying
        tn.stop_after_attempt = lambda *a, **k: None
        tn.wait_random_exponential = lambda *a, **k: None
        sys.modules["tenacity"] = tn
    _orig = importlib.metadata.version

    def _version(name):
        if name == "varKoder":
            return "1.4.0"
        return _orig(name)
    importlib.metadata.version = _version
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


_FAKE_DIR = None


def fake_dsk2ascii_on_path():
    """Put a `dsk2ascii` that cats its -file argument first on PATH."""
    global _FAKE_DIR
    if _FAKE_DIR is None:
        _FAKE_DIR = tempfile.mkdtemp(prefix="fakedsk_")
        p = os.path.join(_FAKE_DIR, "dsk2ascii")
        with open(p, "w") as f:
            f.write('#!/bin/sh\nwhile [ $# -gt 0 ]; do\n  if [ "$1" = "-file" ]; then cat "$2"; exit 0; fi\n'
                    '  shift\ndone\nexit 1\n')
        os.chmod(p, os.stat(p).st_mode | stat.S_IEXEC)
        os.environ["PATH"] = _FAKE_DIR + os.pathsep + os.environ["PATH"]
    return _FAKE_DIR


def recording_tools_on_path(log_path):
    """Put `dsk` and `dsk2ascii` stand-ins first on PATH that append their argv (one JSON list per
    line) to log_path; `dsk` then creates its -out file, `dsk2ascii` cats its -file argument."""
    d = tempfile.mkdtemp(prefix="recdsk_")
    body = ('#!/usr/bin/env python3\nimport json, sys\n'
            'with open(%r, "a") as f:\n    f.write(json.dumps([%%r] + sys.argv[1:]) + "\\n")\n'
            'a = sys.argv[1:]\n' % log_path)
    tails = {"dsk": 'open(a[a.index("-out") + 1], "w").close()\n',
             "dsk2ascii": 'sys.stdout.write(open(a[a.index("-file") + 1]).read())\n'}
    for name, tail in tails.items():
        p = os.path.join(d, name)
        with open(p, "w") as f:
            f.write(body % name + tail)
        os.chmod(p, os.stat(p).st_mode | stat.S_IEXEC)
    os.environ["PATH"] = d + os.pathsep + os.environ["PATH"]
    return d


def drop_from_path(d):
    os.environ["PATH"] = os.pathsep.join(x for x in os.environ["PATH"].split(os.pathsep) if x != d)
