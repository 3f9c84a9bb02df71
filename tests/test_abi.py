"""CPU tests of the boundary: libcmx.so loads and exports every symbol include/cmx.h declares; the
ctypes struct mirror matches the C layout; argument validation returns the documented status codes
(no compute call is made without a GPU)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

from cmx import _abi, _lib
from cmx import parameters as P

REPO = Path(__file__).resolve().parent.parent
HEADER = (REPO / "include" / "cmx.h").read_text()


def declared_symbols():
    # function declarations: `int32_t cmx_xxx(` or `const char *cmx_xxx(` at the start of a line
    return sorted(set(re.findall(r"^(?:int32_t|const char \*)\s*(cmx_\w+)\s*\(", HEADER, flags=re.M)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for s in ("cmx_sb2006_warm_rain_tendencies_f32", "cmx_sb2006_warm_rain_tendencies_f64",
              "cmx_sb2006_process_rates_f32", "cmx_sb2006_process_rates_f64", "cmx_version"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    for s in declared_symbols():
        assert hasattr(lib, s), f"libcmx.so does not export {s}"
    assert lib.cmx_version() == (0 << 16) | 5          # include/cmx.h: the minor number moves with every layout change / new entry
    assert lib.cmx_last_hip_error() == b""


def test_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof from a C translation unit that includes the real header vs the ctypes mirror."""
    probes = []
    for fam in (_abi.F32, _abi.F64):
        for name in ("cloud_pdf_sb2006", "rain_pdf_sb2006", "acnv_sb2006", "accr_sb2006", "selfcol_sb2006",
                     "breakup_sb2006", "evap_sb2006", "numadj_horn2012", "sb2006", "air_properties", "warm_rain_2m",
                     "thermo", "kk2000", "b1994", "tc1980", "ld2004", "bulk_2m_schemes", "stokes_vel", "sb2006_vel", "chen2022_rain_vel", "rain_vel", "koop2000", "abifm_dust",
                     "particle_mass", "particle_area", "ventilation", "acnv_1m", "var_timescale_acnv", "cloud_liquid",
                     "cloud_ice", "rain", "snow", "blk1m_vel_rain", "blk1m_vel_snow", "process_params_1m",
                     "microphysics_1m", "aerosol_activation_params", "aerosol_mode", "aerosol_distribution",
                     "p3_params", "chen2022_small_ice_vel", "chen2022_large_ice_vel", "chen2022_ice_vel", "quadrature",
                     "parameters_0m", "local_rime_density", "rain_freezing", "frostenberg2023", "morrison_milbrandt2014", "p3_ice_params"):
            st = getattr(fam, name)
            probes.append((f"cmx_{name}_{fam.sfx}", st))
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{REPO}/include/cmx.h"', "int main(void){"]
    for cname, st in probes:
        src.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            src.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src.append(f'printf("NPROC %d\\n", (int)CMX_SB2006_NPROC);')
    src.append("return 0;}")
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-o", str(exe), str(c)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, st in probes:
        assert int(out[cname]) == C.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"
    assert int(out["NPROC"]) == _abi.CMX_SB2006_NPROC
    # Julia layout: 45 FT in SB2006 (limited), 50 in WarmRainParams2M
    assert C.sizeof(_abi.F32.sb2006) == 45 * 4 and C.sizeof(_abi.F64.warm_rain_2m) == 50 * 8


def test_argument_validation_without_gpu():
    lib = _lib.lib()
    wr, tps = P.WarmRainParams2M("f32"), P.ThermodynamicsParameters("f32")
    f = lib.cmx_sb2006_warm_rain_tendencies_f32
    null = [None] * 13
    assert f(None, C.byref(tps), None, 1, 10, *null, None) == _abi.CMX_ERR_BAD_ARG
    assert f(C.byref(wr.c), C.byref(tps), None, 1, -1, *null, None) == _abi.CMX_ERR_BAD_ARG
    assert f(C.byref(wr.c), C.byref(tps), None, 1, 10, *null, None) == _abi.CMX_ERR_BAD_ARG   # null columns
    assert f(C.byref(wr.c), C.byref(tps), None, 1, 0, *null, None) == _abi.CMX_OK            # empty input
    # layout adapters: exactly one output form, strides required for more than one run, runs must not overlap
    h = lib.cmx_sb2006_warm_rain_tendencies_fields_f32
    ins = (C.c_void_p * 7)(*[16 * (k + 1) for k in range(7)])
    outs = (C.c_void_p * 4)(*[1600 + 16 * k for k in range(4)])
    st7, st4 = (C.c_int64 * 7)(*[8] * 7), (C.c_int64 * 4)(*[8] * 4)
    assert h(C.byref(wr.c), C.byref(tps), 1, 0, 8, ins, st7, outs, st4, None, None) == _abi.CMX_OK                  # no runs
    assert h(C.byref(wr.c), C.byref(tps), 1, 2, 8, ins, st7, None, None, None, None) == _abi.CMX_ERR_BAD_ARG       # no output
    assert h(C.byref(wr.c), C.byref(tps), 1, 2, 8, ins, st7, outs, st4, C.c_void_p(4096), None) == _abi.CMX_ERR_BAD_ARG   # both outputs
    assert h(C.byref(wr.c), C.byref(tps), 1, 2, 8, ins, None, outs, st4, None, None) == _abi.CMX_ERR_BAD_ARG       # strides missing
    assert h(C.byref(wr.c), C.byref(tps), 1, 2, 16, ins, st7, outs, st4, None, None) == _abi.CMX_ERR_BAD_ARG      # stride < run length
    assert h(C.byref(wr.c), C.byref(tps), 1, 1, 8, ins, None, None, None, C.c_void_p(4100), None) == _abi.CMX_ERR_BAD_ARG   # AoS misaligned
    assert h(C.byref(wr.c), C.byref(tps), 4, 1, 8, ins, None, outs, None, None, None) == _abi.CMX_ERR_BAD_ARG      # unknown flag
    h1 = lib.cmx_mp1m_tendencies_fields_f32
    mp1 = P.Microphysics1MParams("f32")
    assert h1(C.byref(mp1.c), C.byref(tps), mp1.flags, 2, 8, ins, st7, None, None, None, None) == _abi.CMX_ERR_BAD_ARG       # no output
    assert h1(C.byref(mp1.c), C.byref(tps), mp1.flags, 0, 8, ins, st7, outs, st4, None, None) == _abi.CMX_OK
    z = lib.cmx_mp0m_tendencies_f64
    p0 = P.Parameters0M("f64")
    assert z(None, 4, None, None, None, None, None, None) == _abi.CMX_ERR_BAD_ARG
    assert z(C.byref(p0), 4, None, None, None, None, None, None) == _abi.CMX_ERR_BAD_ARG     # null columns
    assert z(C.byref(p0), 0, None, None, None, None, None, None) == _abi.CMX_OK
    # limited rain PSD with an unordered limiter pair: refused (the kernel clamps with v_med3, which needs lo <= hi)
    bad = P.WarmRainParams2M("f32")
    bad.c.seifert_beheng.pdf_r.lambda_min, bad.c.seifert_beheng.pdf_r.lambda_max = 1e4, 1e3
    assert f(C.byref(bad.c), C.byref(tps), None, 1, 0, *null, None) == _abi.CMX_OK             # n = 0 returns before the parameters matter
    assert f(C.byref(bad.c), C.byref(tps), None, 1, 10, *([C.c_void_p(4096)] * 11), None, None, None) == _abi.CMX_ERR_BAD_ARG
    assert f(C.byref(bad.c), C.byref(tps), None, 0, 10, *null, None) == _abi.CMX_ERR_BAD_ARG   # (not limited: falls through to the null-column check)
    # Chen-2022 rain fall speeds: every parameter set is accepted (round 3: Γ(b_i(ρ) + 1) fitted per parameter set, general instantiation
    # otherwise — tests/test_sb2006_gpu.py::test_chen2022_parameter_sets_outside_the_old_window); n = 0 returns before any launch
    velp = P.rain_vel_params("f32")
    cols11 = [C.c_void_p(4096)] * 11
    velp.chen2022.b[2] = 0.3
    assert f(C.byref(wr.c), C.byref(tps), C.byref(velp), 1 | _abi.CMX_VEL_CHEN2022, 0, *cols11, C.c_void_p(4096), C.c_void_p(4096), None) == _abi.CMX_OK
    # a point count no single launch can express (> 16·(2^31 − 1), cmx_launch.hpp kMaxPoints) is refused, not silently truncated
    huge = 16 * 0x7fffffff + 1
    assert z(C.byref(p0), huge, None, None, None, None, None, None) == _abi.CMX_ERR_UNSUPPORTED
    assert f(C.byref(wr.c), C.byref(tps), None, 1, huge, *null, None) == _abi.CMX_ERR_UNSUPPORTED
    assert h(C.byref(wr.c), C.byref(tps), 1, 1 << 31, 1 << 31, ins, st7, outs, st4, None, None) == _abi.CMX_ERR_UNSUPPORTED
    # SB2006 PSD accessors: the limited rain variant asked of a NOT-limited struct (its N0 / lambda limiters are zero) is refused
    # before anything is launched; the same struct without the flag, and the limited struct with it, pass the parameter check (and then
    # fail on the next one here: no output column)
    sd = lib.cmx_sb2006_size_distribution_f32
    fake = [C.c_void_p(4096)] * 3
    pr_nl, pr_l = P.RainParticlePDF_SB2006("f32", is_limited=False), P.RainParticlePDF_SB2006("f32", is_limited=True)
    out3 = [C.c_void_p(4096), None, None]
    assert pr_nl.N0_max == 0 and pr_nl.lambda_max == 0
    assert sd(None, C.byref(pr_nl), _abi.CMX_SB2006_LIMITED, 1e-6, 8, *fake, C.c_void_p(4096), *out3, None) == _abi.CMX_ERR_BAD_ARG
    swapped = P.RainParticlePDF_SB2006("f32", is_limited=True)
    swapped.N0_min, swapped.N0_max = swapped.N0_max, swapped.N0_min
    assert sd(None, C.byref(swapped), _abi.CMX_SB2006_LIMITED, 1e-6, 8, *fake, C.c_void_p(4096), *out3, None) == _abi.CMX_ERR_BAD_ARG
    assert sd(None, C.byref(pr_nl), _abi.CMX_SB2006_LIMITED, 1e-6, 0, *fake, None, None, None, None, None) == _abi.CMX_OK      # n = 0
    assert sd(None, C.byref(pr_l), _abi.CMX_SB2006_LIMITED, 1e-6, 8, *fake, None, None, None, None, None) == _abi.CMX_ERR_BAD_ARG   # no output
    g = lib.cmx_column_sums_f64
    assert g(-1, None, 0, None, None, None) == _abi.CMX_ERR_BAD_ARG
    two = (C.c_void_p * 2)(4096, None)
    assert g(2, two, 8, C.c_void_p(8192), C.c_void_p(16384), None) == _abi.CMX_ERR_BAD_ARG          # a NULL column is caught before anything is enqueued
    assert g(2, (C.c_void_p * 2)(4096, 4096), 8, C.c_void_p(8192), None, None) == _abi.CMX_ERR_BAD_ARG   # … and so is a missing workspace
    assert g(17, (C.c_void_p * 17)(*[4096] * 17), 8, C.c_void_p(8192), C.c_void_p(16384), None) == _abi.CMX_ERR_BAD_ARG     # more than CMX_COLUMN_SUMS_MAX_COLS
    assert g(0, None, 0, None, None, None) == _abi.CMX_OK


def test_product_never_imports_the_oracle():
    """The product package must not reference oracle/ (a CPU fallback would void parity claims)."""
    for p in (REPO / "cloudmicrophysics.jl_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".hpp", ".h", ".cpp") and p.is_file():
            txt = p.read_text()
            assert "oracle_binding" not in txt and "libcmx_oracle" not in txt and "cmxo_" not in txt, p


def test_contraction_is_bracketed_around_the_p3_code_only():
    """The library is built with -ffp-contract=off (Makefile) so that every kernel sharing a point function rounds alike; the P3 quadrature / incomplete-gamma code is
    the one bracketed exception (cmx_p3.hpp CMX_P3_CONTRACT_BEGIN / END).  The pointwise part of the 2M + P3 entry must stay OUTSIDE the bracket: it is bit-identical to
    the warm-rain entry (tests/test_mp2m_p3_gpu.py::test_reduces_to_warm_rain_without_ice_and_validates) only as long as it is compiled like that entry."""
    csrc = REPO / "cloudmicrophysics.jl_amd" / "csrc"
    assert "-ffp-contract=off" in (csrc / "Makefile").read_text()
    users = {p.name: p.read_text() for p in csrc.iterdir() if p.suffix in (".hip", ".hpp") and "CMX_P3_CONTRACT_BEGIN" in p.read_text()}
    assert sorted(users) == ["cmx_p3.hpp", "cmx_p3_collisions.hip", "cmx_p3_kernels.hip"]
    for name, txt in users.items():
        code = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith(("//", "#define", "#if", "#else", "#endif")))
        assert code.count("CMX_P3_CONTRACT_BEGIN") == code.count("CMX_P3_CONTRACT_END") == 1, name      # one bracket per file, closed
        assert code.index("CMX_P3_CONTRACT_BEGIN") < code.index("CMX_P3_CONTRACT_END"), name
    col = users["cmx_p3_collisions.hip"]
    begin, end = col.index("\nCMX_P3_CONTRACT_BEGIN"), col.index("\nCMX_P3_CONTRACT_END")
    for outside in ("void mp2m_p3_point(", "void mp2m_p3_pointwise_kernel(", "void liquid_freezing_kernel(", '#include "cmx_sb2006.hpp"'):
        at = col.index(outside)
        assert at < begin or at > end, outside
    assert begin < col.index("void p3_collision_kernel(") < end
    # no other file switches contraction on
    for p in csrc.iterdir():
        if p.suffix in (".hip", ".hpp", ".inc") and p.name != "cmx_p3.hpp":
            assert "fp contract" not in p.read_text() and "ffp-contract=fast" not in p.read_text(), p.name


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.CmxLibraryError):
        _lib.lib()


def test_parameter_defaults_and_overrides():
    sb = P.SB2006("f64")
    assert sb.acnv.x_star == sb.pdf_r.xr_min == sb.pdf_c.xc_max == 2.6e-10     # one ClimaParams key, three fields
    td = P.create_toml_dict("f64", P.SB2006_LIMITERS_OVERRIDE)
    sb2 = P.SB2006(td)
    assert sb2.acnv.x_star == sb2.pdf_r.xr_min == sb2.pdf_c.xc_max == 6.54e-11 and sb2.pdf_r.N0_max == 2e11
    with pytest.raises(KeyError):
        P.create_toml_dict("f64", {"not_a_parameter": 1.0})
    mp = P.Microphysics2MParams("f32", with_ice=True)            # P3IceParams defaults — Microphysics2MParams.jl:88-106
    assert mp.ice.c.quad.n == 16 and mp.ice.c.tau_act == 300.0 and mp.ice.is_limited
    assert P.build_quadrature("f64", 20).n == 20 and P.Microphysics2MParams("f64").ice is None
