import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "green-marl_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np
    z = np.load(os.path.join(GOLD, "golden.npz"), allow_pickle=False)
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    cases, other = {}, {}
    for k in z.files:
        name, field = k.split("/")
        # "cases": graphs with the full set of reference-pinned results; "uniform": outputs of the reference's
        # uniform generator only
        (other if name.startswith("uniform_") else cases).setdefault(name, {})[field] = z[k]
    return {"cases": cases, "uniform": other, "manifest": man}
