"""The C-ABI library builds for gfx950, loads, and exports every symbol include/vkimg.h
declares.  No compute calls: this runs without a GPU."""
import ctypes as C
import os
import re
import shutil

import pytest

from varkoder_amd import _capi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "vkimg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(_capi.SYMBOLS)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc not available")
def test_extension_builds_for_gfx950():
    so = build.build_hip()
    assert os.path.exists(so)


@pytest.mark.gpu
def test_extension_builds_for_gfx950_on_the_gpu_box():
    """The same under `-m gpu` (the suite the driver runs): the tree as shipped compiles from clean, to a temporary
    path, whatever binary travelled with it."""
    v = build.verify_compiles()
    assert v["bytes"] > 100000 and v["seconds"] > 0


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_capi.LIB_PATH):
        build.build_hip()
    L = C.CDLL(_capi.LIB_PATH)
    for name in header_functions():
        assert hasattr(L, name), name
    L.vk_abi_version.restype = C.c_int
    assert L.vk_abi_version() == 1
    L.vk_strerror.restype = C.c_char_p
    assert L.vk_strerror(0) == b"ok"
    assert b"mapping" in L.vk_strerror(3)
