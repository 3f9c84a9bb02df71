"""Host-side pieces of bench.py that need no GPU: the amdgpu sysfs telemetry parser and the source digest."""
import importlib.util
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench", REPO / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_sysfs_telemetry_parser(tmp_path):
    """The three files rocm-smi itself reads: the starred level of pp_dpm_sclk / pp_dpm_mclk and hwmon power in microwatts (power1_average, or
    power1_input where the average is not exported — the MI355X boxes of this pool)."""
    b = _bench()
    d = tmp_path / "device"
    (d / "hwmon" / "hwmon3").mkdir(parents=True)
    (d / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 1931Mhz *\n2: 2400Mhz\n")
    (d / "pp_dpm_mclk").write_text("0: 900Mhz\n1: 2000Mhz *\n")
    (d / "hwmon" / "hwmon3" / "power1_input").write_text("1399000000\n")
    assert b._read_sysfs_telemetry(str(d)) == (1931, 2000, 1399.0)
    (d / "hwmon" / "hwmon3" / "power1_average").write_text("1250000000\n")
    assert b._read_sysfs_telemetry(str(d))[2] == 1250.0
    (d / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 2400Mhz\n")           # no starred level: unknown, not a guess
    assert b._read_sysfs_telemetry(str(d))[0] is None
    assert b._read_sysfs_telemetry(str(tmp_path / "missing")) == (None, None, None)


def test_source_digest_is_stable_and_covers_the_kernel_sources():
    b = _bench()
    a = b.source_digest()
    assert a == b.source_digest() and len(a) == 16
    import inspect
    src = inspect.getsource(b.source_digest)
    assert "*.hip" in src and "*.hpp" in src and "cmx.h" in src
