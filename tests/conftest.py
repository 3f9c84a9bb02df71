import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "cloudmicrophysics.jl_amd", REPO / "oracle", REPO / "tests", REPO):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def golden():
    import json
    with open(REPO / "tests" / "golden" / "sb2006_kats.json") as f:
        return json.load(f)
