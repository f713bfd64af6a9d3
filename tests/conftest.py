import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_data(spec):
    """Regenerates the bytes a golden record describes (see tests/golden/gen_golden.py)."""
    import oracle
    kind, n = spec["kind"], spec["len"]
    if kind == "splitmix64":
        return oracle.splitmix64(spec["seed"], n)
    if kind == "zeros":
        return np.zeros(n, dtype=np.uint8)
    if kind == "ones":
        return np.full(n, 0xFF, dtype=np.uint8)
    if kind == "incr":
        return (np.arange(n) & 0xFF).astype(np.uint8)
    if kind == "decr":
        return ((n - 1 - np.arange(n)) & 0xFF).astype(np.uint8)
    if kind == "ascii":
        return np.frombuffer(spec["text"].encode("ascii"), dtype=np.uint8).copy()
    raise ValueError(kind)


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(GOLDEN, "vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def reference_tables():
    with open(os.path.join(GOLDEN, "reference_tables.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def reference_held():
    """Digests the reference's own tests hold as literals for the JDK primitive this path uses
    (tests/golden/gen_reference_held.py)."""
    with open(os.path.join(GOLDEN, "reference_held_digests.json")) as f:
        return json.load(f)
