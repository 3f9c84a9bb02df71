"""The boundary as seen by a plain-C caller (tests/native/abi_caller.c): include/cmx.h compiled as C11 by gcc, parameter structs
filled from literals, device memory from the HIP C API, no Python host mirror on the calling side.

CPU (`-m "not gpu"`): the caller compiles and links against libcmx.so, its literal header is current, the size contract of
cmx.h (CMX_ASSERT_PARAM_STRUCT_SIZES) matches the ctypes mirror, and a struct with a missing field is REJECTED at compile
time.  GPU: the program runs and its numbers are compared with the reference's known-answer values (tests/golden), with
the oracle, and — bit for bit — with the ctypes path the other GPU tests use.
"""
import json
import math
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from cmx import _abi
from cmx import parameters as P

REPO = Path(__file__).resolve().parent.parent
NATIVE = REPO / "tests" / "native"
CSRC = REPO / "cloudmicrophysics.jl_amd" / "csrc"
ROCM = Path("/opt/rocm")


def _gcc(src: Path, out: Path, extra=()):
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror=implicit-function-declaration", "-D__HIP_PLATFORM_AMD__", "-I", str(REPO / "include"),
           "-I", str(ROCM / "include"), "-I", str(NATIVE), str(src), "-L", str(CSRC), "-lcmx", "-L", str(ROCM / "lib"), "-lamdhip64",
           f"-Wl,-rpath,{CSRC}", f"-Wl,-rpath,{ROCM / 'lib'}", "-o", str(out), *extra]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.fixture(scope="module")
def caller(tmp_path_factory):
    out = tmp_path_factory.mktemp("native") / "abi_caller"
    r = _gcc(NATIVE / "abi_caller.c", out)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


def test_literal_header_is_current():
    r = subprocess.run([sys.executable, str(NATIVE / "gen_params.py"), "--check"])
    assert r.returncode == 0, "tests/native/abi_caller_params.h is stale: run python tests/native/gen_params.py"


def test_c_caller_compiles_and_links(caller):
    assert caller.exists()
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([str(caller)], capture_output=True, text=True)
        assert r.returncode == 77 and "no HIP device" in r.stderr        # loads libcmx.so, finds no GPU, computes nothing


def test_size_contract_matches_the_ctypes_mirror():
    """cmx.h asserts sizeof(struct) == fields·sizeof(FT) (+ 8-byte count header); the ctypes structs must agree."""
    import ctypes as C
    import re
    text = (REPO / "include" / "cmx.h").read_text()
    rows = re.findall(r"CMX_STATIC_ASSERT\(sizeof\((cmx_\w+)_##SFX\) == (?:(\d+) \+ )?(\d+) \* sizeof\(FT\)", text)
    assert len(rows) == 54
    checked = 0
    for name, hdr, nft in rows:
        for fam, w in ((_abi.F32, 4), (_abi.F64, 8)):
            ct = getattr(fam, name[len("cmx_"):])                       # every struct of the header has a ctypes twin
            assert ct.__name__ == f"{name}_{fam.sfx}"
            assert C.sizeof(ct) == int(hdr or 0) + int(nft) * w, (name, fam.sfx)
            checked += 1
    assert checked == 108
    assert C.sizeof(_abi.F32.thermo) == 52 and C.sizeof(_abi.F64.thermo) == 104               # 13 fields: cv_l is the 13th


def test_a_binding_with_a_missing_field_does_not_compile(tmp_path):
    """The round-1 INTEGRATION.md shim declared CmxThermo with 12 fields; the same mistake made in C must be a compile error."""
    src = tmp_path / "bad.c"
    src.write_text('#include "cmx.h"\n'
                   "typedef struct { double R_v, R_d, cp_d, cp_v, cp_l, cp_i, LH_v0, LH_s0, T_0, T_triple, press_triple, T_freeze; } my_thermo;\n"
                   '_Static_assert(sizeof(my_thermo) == sizeof(cmx_thermo_f64), "binding struct does not match cmx_thermo_f64");\n'
                   "int main(void) { return 0; }\n")
    r = subprocess.run(["gcc", "-std=c11", "-I", str(REPO / "include"), "-c", str(src), "-o", str(tmp_path / "bad.o")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "does not match cmx_thermo_f64" in r.stderr


def test_header_compiles_as_cxx(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "cmx.h"\nint main() { return sizeof(cmx_thermo_f32) == 52 ? 0 : 1; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-I", str(REPO / "include"), str(src), "-o", str(tmp_path / "t")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def result(caller):
    r = subprocess.run([str(caller)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    return json.loads(r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("limited", [True, False])
def test_process_rates_match_the_reference_kats(result, golden, limited):
    """test/gpu_tests.jl:821-872 — the numbers the reference asserts for SB2006_2M_kernel on this state."""
    g = golden["process_rates_default_params"]
    got = result["process_rates_limited" if limited else "process_rates_notlimited"]
    n = 0
    for e in g["common"] + g["limited" if limited else "notlimited"]:
        x = got[e["col"]]
        if "rtol" in e:
            assert math.isclose(x, e["expected"], rel_tol=e["rtol"]), (e, x)
        else:
            assert abs(x - e["expected"]) <= e["atol"], (e, x)
        n += 1
    assert n >= 14


@pytest.mark.gpu
def test_sedimentation_velocities_match_the_reference_kats(result):
    """test/gpu_tests.jl:608-630 (test_chen2022_terminal_velocity_kernel!)."""
    g = json.loads((REPO / "tests" / "golden" / "mp1m_kats.json").read_text())["chen2022_sedimentation_velocities"]
    for k in ("w_lcl", "w_icl", "w_rai", "w_sno"):
        assert math.isclose(result["sedimentation_velocities"][k], g[k], rel_tol=1e-12), k


@pytest.mark.gpu
def test_fused_entries_match_oracle_and_ctypes_path(result, oracle):
    import torch

    import cmx
    import parity
    dev = torch.device("cuda:0")
    col = lambda v: np.full(8, v, dtype=np.float64)  # noqa: E731
    tps = P.ThermodynamicsParameters("f64")
    # (2) north-star entry on the KAT state
    cols = [col(1.2), col(290.0), col(7e-3), col(2e-3), col(1e8 / 1.2), col(5e-4), col(1e7 / 1.2)]
    ref = oracle.sb2006_warm_rain_tendencies(_abi.F64, P.WarmRainParams2M("f64").c, tps, P.rain_vel_params("f64"),
                                             _abi.CMX_SB2006_LIMITED | _abi.CMX_VEL_SB2006, *cols)
    got = {k: col(v) for k, v in result["warm_rain_tendencies"].items()}
    parity.assert_parity(got, ref, parity.RTOL["f64"], what="C caller, 2M fused")
    out = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), P.Microphysics2MParams("f64"), tps,
                                           *[torch.from_numpy(c).to(dev) for c in cols], vel=cmx.SB2006VelType)
    for k, v in out._asdict().items():
        assert v[0].item() == result["warm_rain_tendencies"][k], k            # the same library, the same bits
    # the fused tendencies are the sums of the KAT-pinned process rates (BMT:707-782) — ties (2) to the reference's numbers
    pr = result["process_rates_limited"]
    assert math.isclose(result["warm_rain_tendencies"]["vt_rai_n"], pr["rain_vel_n"], rel_tol=1e-12)
    assert math.isclose(result["warm_rain_tendencies"]["vt_rai_m"], pr["rain_vel_m"], rel_tol=1e-12)
    dq_rai = pr["acnv_dq_rai_dt"] + pr["accr_dq_rai_dt"] + pr["evap_dq_rai_dt"]
    assert math.isclose(result["warm_rain_tendencies"]["dq_rai_dt"], dq_rai, rel_tol=1e-9)
    # (3) 1-moment entry
    mp = P.Microphysics1MParams("f64")
    c1 = [col(1.2), col(268.0), col(6e-3), col(5e-4), col(5e-4), col(5e-4), col(5e-4)]
    ref1 = oracle.mp1m(_abi.F64, mp.c, tps, mp.flags, *c1, want_sources=False)
    names = ["dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"]
    parity.assert_parity({k: col(result["mp1m_tendencies"][k]) for k in names}, ref1, parity.RTOL["f64"], names=names, what="C caller, 1M")
    t = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *[torch.from_numpy(c).to(dev) for c in c1])
    for k in names:
        assert getattr(t, k)[0].item() == result["mp1m_tendencies"][k], k
        assert result["mp1m_tendencies"][k] != 0.0


@pytest.mark.gpu
def test_status_codes_seen_by_a_c_caller(result):
    assert result["status"] == {"null_params": _abi.CMX_ERR_BAD_ARG, "negative_n": _abi.CMX_ERR_BAD_ARG, "empty": 0}
    assert result["cmx_version"] == (0 << 16) | 5
