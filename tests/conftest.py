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


def pytest_sessionstart(session):
    """COFHE_TEST_LIB=<path>: run the suite against another build of the SAME extension (kernel-tuning variants made by
    tools/build_variant.sh) instead of cofhe_amd/libcofhe_hip.so"""
    p = os.environ.get("COFHE_TEST_LIB")
    if p:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()          # PyTorch's HIP runtime first, as the tests' engine() helper does
        import cofhe_amd
        cofhe_amd.load_library(os.path.abspath(p))
