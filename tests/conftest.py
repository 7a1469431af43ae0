import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope="session", params=["s128_k128", "s128_k256", "tiny_k8"])
def golden(request):
    return load_json("params_%s.json" % request.param), load_json("vectors_%s.json" % request.param)


@pytest.fixture(scope="session")
def params128():
    return load_json("params_s128_k128.json")
