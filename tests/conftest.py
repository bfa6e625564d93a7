import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def golden_indexes(oracle, golden_dir, tmp_path_factory):
    """Index files for the committed golden genomes, written by the oracle's restatement of `sbwt build`
    (byte-identical to upstream's files for G1: md5 pinned in tests/golden/golden.json)."""
    d = tmp_path_factory.mktemp("idx")
    out = {}
    for g in ("g1", "g2"):
        prefix = str(d / g)
        oracle.build_index(os.path.join(golden_dir, g + ".fa"), prefix)
        out[g] = prefix
    return out
