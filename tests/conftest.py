import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(GOLDEN, "kats.json")) as f:
        data = json.load(f)

    def get(key):
        get.used.add(key)
        return data[key]["values"]
    get.keys = sorted(data)
    get.used = set()  # keys some test has asked for (tests/test_oracle_kats.py checks that none is left over)
    return get


@pytest.fixture(scope="session")
def s101_proof():
    import stark_symphony_amd as ss
    with open(os.path.join(GOLDEN, "stark101_proof.json")) as f:
        return ss.stark101_from_json(json.load(f))


@pytest.fixture(scope="session")
def stwo_small():
    import stark_symphony_amd as ss
    with open(os.path.join(GOLDEN, "stwo_proof_test.json")) as f:
        return ss.stwo_from_json(json.load(f))


@pytest.fixture(scope="session")
def stwo_prod():
    import stark_symphony_amd as ss
    with open(os.path.join(GOLDEN, "stwo_proof.json")) as f:
        return ss.stwo_from_json(json.load(f))
