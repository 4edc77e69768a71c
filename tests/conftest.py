import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_small():
    with np.load(os.path.join(GOLDEN, "image_small.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def engines():
    """Cache of ImageEngine objects keyed by (k, mapping); GPU tests only."""
    from varkoder_amd.engine import ImageEngine
    cache = {}

    def get(k, mapping="cgr"):
        key = (k, mapping)
        if key not in cache:
            cache[key] = ImageEngine(k=k, mapping=mapping, device=0)
        return cache[key]
    yield get
    for e in cache.values():
        e.close()
