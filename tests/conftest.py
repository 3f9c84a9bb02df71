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


def pytest_sessionfinish(session, exitstatus):
    """Write the quantitative parity rows collected by parity.assert_parity (fraction of points inside the plain north-star
    relative bound, excluded near-branch counts, worst well-conditioned error) to gpurun_out/parity_report.json."""
    import json
    try:
        import parity
    except ImportError:
        return
    if not parity.REPORTS:
        return
    out = REPO / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        summary = {}
        for r in parity.REPORTS:
            s = summary.setdefault(r["ft"], {"rows": 0, "points": 0, "outside_plain_bound": 0, "excluded_near_branch": 0,
                                             "min_frac_within": 1.0, "worst_wellcond": 0.0})
            s["rows"] += 1
            s["points"] += r["n"]
            s["outside_plain_bound"] += r["n_outside"]
            s["excluded_near_branch"] += r["n_excluded"]
            s["min_frac_within"] = min(s["min_frac_within"], r["frac_within"])
            s["worst_wellcond"] = max(s["worst_wellcond"], r["worst_wellcond"])
        (out / "parity_report.json").write_text(json.dumps({"summary": summary, "rows": parity.REPORTS}, indent=1))
    except OSError:
        pass
