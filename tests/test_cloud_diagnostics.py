"""CloudDiagnostics over columns (round 5 — the diagnostics a host model computes from the same state columns as the tendencies):
    CMD.radar_reflectivity_1M, radar_reflectivity_2M, effective_radius_2M, effective_radius_Liu_Hallet_97      src/CloudDiagnostics.jl:31-163
CPU: the oracle restatement (oracle/cmx_oracle_diag_impl.h) against the reference's own known answers (test/cloud_diagnostics.jl, committed as
tests/golden/cloud_diagnostics_kats.json) in Float64 and Float32 arithmetic; GPU: the device entry cmx_cloud_diagnostics_* against those KATs and,
on random states, against the oracle."""
import json
from pathlib import Path

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from cmx import _abi  # noqa: E402
from cmx import parameters as P  # noqa: E402

G = json.loads((Path(__file__).parent / "golden" / "cloud_diagnostics_kats.json").read_text())
NPF = {"f32": np.float32, "f64": np.float64}
DT = {"f32": torch.float32, "f64": torch.float64}


def _sb(ft, limited):
    return P.SB2006(P.create_toml_dict(ft, P.SB2006_LIMITERS_OVERRIDE), limited)


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_oracle_reproduces_the_reference_kats(oracle, ft):
    fam = _abi.family(ft)
    f32 = ft == "f32"
    g = G["radar_reflectivity_1M"]
    rain = P.Microphysics1MParams(ft).c.rain      # CMP.Rain(FT) = the `rain` member of Microphysics1MParams
    Z = oracle.cloud_diagnostics(fam, np.full(2, g["rho"]), np.zeros(2), g["q_rai"], rain=rain, float32_gates=f32, want=("Z_1m",))["Z_1m"]
    assert np.all(np.abs(Z - g["expected_dBZ"]) <= g["atol"])
    g = G["sb2006_2M"]
    for limited in (True, False):
        sb = _sb(ft, limited)
        out = oracle.cloud_diagnostics(fam, np.full(4, g["rho"]), g["q_lcl"], g["q_rai"], g["N_lcl"], g["N_rai"], pdf_c=sb.pdf_c, pdf_r=sb.pdf_r,
                                       limited=limited, float32_gates=f32, want=("Z_2m", "reff_2m"))
        assert np.all(np.abs(out["Z_2m"] - g["radar_reflectivity_dBZ"]) <= (g["atol_Z"] if ft == "f64" else 2e-3)), (limited, out["Z_2m"])
        assert np.all(np.abs(out["reff_2m"] - g["effective_radius_m"]) <= g["atol_reff"]), (limited, out["reff_2m"])
        s = G["sb2006_2M_small_numbers"]
        out = oracle.cloud_diagnostics(fam, [s["rho"]], [s["q_lcl"]], [s["q_rai"]], [s["N_lcl"]], [s["N_rai"]], pdf_c=sb.pdf_c, pdf_r=sb.pdf_r, limited=limited,
                                       float32_gates=f32, want=("Z_2m", "reff_2m"))
        assert abs(out["Z_2m"][0] - s["radar_reflectivity_dBZ"]) <= s["atol_Z"] and abs(out["reff_2m"][0] - s["effective_radius_m"]) <= s["atol_reff"]
    g = G["liu_hallett_97"]
    r = oracle.cloud_diagnostics(fam, [g["rho"]], [g["q_lcl"]], [g["q_rai"]], [g["N_lcl"]], [g["N_rai"]], rho_w=g["rho_w"], float32_gates=f32, want=("reff_lh97",))["reff_lh97"]
    assert abs(r[0] - g["expected_m"]) <= g["atol"]
    # the three-argument method = N_lcl 100, no rain (test/cloud_diagnostics.jl:118-126)
    a = oracle.cloud_diagnostics(fam, [g["rho"]], [g["q_lcl"]], rho_w=g["rho_w"], float32_gates=f32, want=("reff_lh97",))["reff_lh97"]
    b = oracle.cloud_diagnostics(fam, [g["rho"]], [g["q_lcl"]], [g["default_q_rai"]], [g["default_N_lcl"]], [g["default_N_rai"]], rho_w=g["rho_w"], float32_gates=f32,
                                 want=("reff_lh97",))["reff_lh97"]
    assert a[0] == b[0]
    c = G["effective_radius_const"]
    mp = P.Microphysics1MParams(ft)
    assert mp.c.cloud_liquid.r_eff == NPF[ft](c["cloud_liquid_m"]) and mp.c.cloud_ice.r_eff == NPF[ft](c["cloud_ice_m"])


# ---- device ------------------------------------------------------------------------------------------------------------------------------
import parity  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_device_reproduces_the_reference_kats(dev, ft):
    """The reference's own expected values (test/cloud_diagnostics.jl) through the C ABI, at its tolerances."""
    from cmx import cloud_diagnostics as CD
    to = lambda a: torch.tensor(np.atleast_1d(np.asarray(a, dtype=np.float64)), dtype=DT[ft], device=dev)  # noqa: E731
    back = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    g = G["radar_reflectivity_1M"]
    Z = back(CD.radar_reflectivity_1M(P.Microphysics1MParams(ft).c.rain, to(g["q_rai"]), to([g["rho"]] * 2)))
    assert np.all(np.abs(Z - g["expected_dBZ"]) <= g["atol"])
    g = G["sb2006_2M"]
    rho = to([g["rho"]] * 4)
    for limited in (True, False):
        sb = _sb(ft, limited)
        Z, r = CD.radar_reflectivity_and_effective_radius_2M(sb, to(g["q_lcl"]), to(g["q_rai"]), to(g["N_lcl"]), to(g["N_rai"]), rho)
        assert np.all(np.abs(back(Z) - g["radar_reflectivity_dBZ"]) <= (g["atol_Z"] if ft == "f64" else 2e-3)), (limited, back(Z))
        assert np.all(np.abs(back(r) - g["effective_radius_m"]) <= g["atol_reff"]), (limited, back(r))
        # each function on its own gives the same bits as the fused call
        assert torch.equal(CD.radar_reflectivity_2M(sb, to(g["q_lcl"]), to(g["q_rai"]), to(g["N_lcl"]), to(g["N_rai"]), rho), Z)
        assert torch.equal(CD.effective_radius_2M(sb, to(g["q_lcl"]), to(g["q_rai"]), to(g["N_lcl"]), to(g["N_rai"]), rho), r)
        s = G["sb2006_2M_small_numbers"]
        Z, r = CD.radar_reflectivity_and_effective_radius_2M(sb, to(s["q_lcl"]), to(s["q_rai"]), to(s["N_lcl"]), to(s["N_rai"]), to(s["rho"]))
        assert abs(back(Z)[0] - s["radar_reflectivity_dBZ"]) <= s["atol_Z"] and abs(back(r)[0] - s["effective_radius_m"]) <= s["atol_reff"]
    g = G["liu_hallett_97"]
    r = back(CD.effective_radius_Liu_Hallet_97(g["rho_w"], to(g["rho"]), to(g["q_lcl"]), to(g["N_lcl"]), to(g["q_rai"]), to(g["N_rai"])))
    assert abs(r[0] - g["expected_m"]) <= g["atol"]
    a = CD.effective_radius_Liu_Hallet_97(g["rho_w"], to(g["rho"]), to(g["q_lcl"]))
    b = CD.effective_radius_Liu_Hallet_97(g["rho_w"], to(g["rho"]), to(g["q_lcl"]), to(g["default_N_lcl"]), to(g["default_q_rai"]), to(g["default_N_rai"]))
    assert torch.equal(a, b)                                                     # test/cloud_diagnostics.jl:118-126
    c = G["effective_radius_const"]
    mp = P.Microphysics1MParams(ft)
    assert CD.effective_radius_const(mp.c.cloud_liquid) == float(NPF[ft](c["cloud_liquid_m"])) and CD.effective_radius_const(mp.c.cloud_ice) == float(NPF[ft](c["cloud_ice_m"]))


def _random_state(n, seed):
    rng = np.random.default_rng(seed)
    rho = rng.uniform(0.3, 1.3, n)
    present = lambda p: rng.random(n) < p  # noqa: E731
    q_lcl = np.where(present(0.8), 10 ** rng.uniform(-8, -2.5, n), 0.0)
    q_rai = np.where(present(0.8), 10 ** rng.uniform(-9, -2.5, n), 0.0)
    N_lcl = np.where(present(0.9), 10 ** rng.uniform(5, 9.5, n), 0.0)
    N_rai = np.where(present(0.9), 10 ** rng.uniform(0, 6.5, n), 0.0)
    # tiny values around the gates (eps(FT) and far below), as in the reference's "small numbers" case
    k = n // 20
    q_lcl[:k] = 10 ** rng.uniform(-30, -6, k); N_lcl[:k] = 10 ** rng.uniform(-15, 3, k)
    q_rai[k:2 * k] = 10 ** rng.uniform(-30, -6, k); N_rai[k:2 * k] = 10 ** rng.uniform(-20, 1, k)
    return rho, q_lcl, q_rai, N_lcl, N_rai


@pytest.mark.gpu
@pytest.mark.parametrize("ft", ["f64", "f32"])
@pytest.mark.parametrize("limited", [True, False])
def test_device_against_the_oracle_on_random_states(dev, oracle, ft, limited):
    from cmx import cloud_diagnostics as CD
    fam = _abi.family(ft)
    n = 200_003                                               # odd: vector body + scalar tail
    cols = [np.asarray(c).astype(NPF[ft]).astype(np.float64) for c in _random_state(n, 11 + limited)]
    rho, q_lcl, q_rai, N_lcl, N_rai = cols
    to = lambda a: torch.from_numpy(a).to(DT[ft]).to(dev)  # noqa: E731
    back = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    sb, sb64 = _sb(ft, limited), _sb("f64", limited)
    rain, rain64 = P.Microphysics1MParams(ft).c.rain, P.Microphysics1MParams("f64").c.rain
    ref = oracle.cloud_diagnostics(_abi.F64, rho, q_lcl, q_rai, N_lcl, N_rai, rain=rain64, pdf_c=sb64.pdf_c, pdf_r=sb64.pdf_r, rho_w=1000.0, limited=limited,
                                   float32_gates=(ft == "f32"))
    Z2, r2 = CD.radar_reflectivity_and_effective_radius_2M(sb, to(q_lcl), to(q_rai), to(N_lcl), to(N_rai), to(rho))
    got = {"Z_1m": back(CD.radar_reflectivity_1M(rain, to(q_rai), to(rho))), "Z_2m": back(Z2), "reff_2m": back(r2),
           "reff_lh97": back(CD.effective_radius_Liu_Hallet_97(1000.0, to(rho), to(q_lcl), to(N_lcl), to(q_rai), to(N_rai)))}
    # all four in ONE launch into caller-provided columns: the same bits as the per-function calls
    outc = CD.Diagnostics(*[torch.empty(n, dtype=DT[ft], device=dev) for _ in range(4)])
    fused = CD.cloud_diagnostics(rain, sb, 1000.0, to(rho), to(q_lcl), to(q_rai), to(N_lcl), to(N_rai), out=outc)
    assert fused.Z_2m.data_ptr() == outc.Z_2m.data_ptr()
    for k in ("Z_1m", "Z_2m", "reff_2m", "reff_lh97"):
        assert np.array_equal(back(getattr(fused, k)), got[k], equal_nan=True), k
    tol = parity.RTOL[ft]
    # reflectivities are 10·log10 of a power law: an ABSOLUTE tolerance in dB — the relative tolerance of the linear quantity, 10 log10(1 + tol) ≈ 4.34 tol, plus the
    # rounding of the (≈ 300 dB) offsets that cancel inside the logarithm's affine form in Float32
    dB = 4.35 * tol + (3e-4 if ft == "f32" else 1e-9)
    for k in ("Z_1m", "Z_2m"):
        assert np.array_equal(got[k] == -150.0, ref[k] == -150.0) or np.all(np.abs(got[k] - ref[k])[(got[k] == -150.0) != (ref[k] == -150.0)] <= dB), k
        assert np.all(np.abs(got[k] - ref[k]) <= dB), (k, np.max(np.abs(got[k] - ref[k])))
    # effective radii: relative, except where the M² gate (≤ ϵ) sits within rounding of the sum (a genuine discontinuity of the reference)
    for k in ("reff_2m", "reff_lh97"):
        x, r = got[k], ref[k]
        # "zero" = below 1e-12 m: where q ρ / N underflows Float32 (q ≈ 1e-30 in the tiny-value set) the hardware logarithm flushes the subnormal to 0 and the
        # radius comes out 0, the Float64 oracle's ∛ of the same quotient is ≈ 1e-15 m — neither is a radius
        same_gate = (x < 1e-12) == (r < 1e-12)
        assert np.mean(same_gate) >= 0.9999, (k, int(np.sum(~same_gate)))
        live = same_gate & (r >= 1e-12)
        assert np.all(np.abs(x[live] - r[live]) <= 4 * tol * np.abs(r[live])), (k, np.max(np.abs(x[live] - r[live]) / np.abs(r[live])))
    for k in ("Z_1m", "Z_2m", "reff_2m", "reff_lh97"):
        assert np.all(np.isfinite(got[k]))
    pin = "test/cloud_diagnostics.jl KATs (tests/golden/cloud_diagnostics_kats.json) + oracle restatement of src/CloudDiagnostics.jl"
    for k in ("reff_2m", "reff_lh97"):      # radii above 1e-12 m (below: the Float32 underflow of q ρ / N described above); the reflectivities are dB: absolute, asserted above
        parity.record(f"CMD cloud diagnostics {'limited' if limited else 'not limited'} {ft}", ft, {k: got[k]}, {k: ref[k]}, family="row g: CloudDiagnostics", pinned_by=pin,
                      keep=(ref[k] >= 1e-12) & (got[k] >= 1e-12), assert_wellcond=True,
                      note="effective radii at 4 x the plain bound (a ratio of two moments); reflectivities asserted in dB: 4.35 RTOL + the Float32 rounding of the 300-dB offsets")


@pytest.mark.gpu
def test_device_nan_and_argument_rules(dev):
    from cmx import _lib
    from cmx import cloud_diagnostics as CD
    import ctypes as C
    ft = "f32"
    sb = _sb(ft, True)
    t = lambda *v: torch.tensor(v, dtype=torch.float32, device=dev)  # noqa: E731
    nan = float("nan")
    Z, r = CD.radar_reflectivity_and_effective_radius_2M(sb, t(1e-4, nan, 1e-4), t(1e-4, 1e-4, 1e-4), t(1e8, 1e8, 1e8), t(1e4, 1e4, nan), t(1.0, 1.0, 1.0))
    assert torch.isfinite(Z[0]) and torch.isnan(Z[1]) and torch.isnan(Z[2]) and torch.isnan(r[1]) and torch.isnan(r[2])
    assert torch.isnan(CD.radar_reflectivity_1M(P.Microphysics1MParams(ft).c.rain, t(nan), t(1.0)))[0]
    # a negative rain number next to rain mass (left by advection; the limited PSD gates only on N < eps AND q < eps): the reference's N·B^(−n/μ) is a finite
    # negative moment, not a NaN — against the oracle; a negative q_rai with a positive number likewise
    import oracle_binding as ob
    cols = [np.array(v, dtype=np.float64) for v in ([1e-4, 1e-4, 0.0], [1e-4, -1e-6, 1e-4], [1e8, 1e8, 0.0], [-1e3, 1e4, -1e3], [1.0, 1.0, 1.0])]
    Z, r = CD.radar_reflectivity_and_effective_radius_2M(sb, *[torch.tensor(c, dtype=torch.float32, device=dev) for c in cols])
    sb64 = _sb("f64", True)
    ref = ob.cloud_diagnostics(_abi.F64, cols[4], cols[0], cols[1], cols[2], cols[3], pdf_c=sb64.pdf_c, pdf_r=sb64.pdf_r, limited=True, float32_gates=True,
                               want=("Z_2m", "reff_2m"))
    assert bool(torch.isfinite(Z).all()) and bool(torch.isfinite(r).all())
    np.testing.assert_allclose(Z.cpu().numpy(), ref["Z_2m"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(r.cpu().numpy(), ref["reff_2m"], rtol=1e-3, atol=1e-12)
    # limited flag with a not-limited struct, and a missing parameter struct: refused
    lib = _lib.lib()
    fn = lib.cmx_cloud_diagnostics_f32
    nl = _sb(ft, False)
    p = C.c_void_p(t(1.0, 1.0, 1.0, 1.0).data_ptr())
    assert fn(None, C.byref(nl.pdf_c), C.byref(nl.pdf_r), 0.0, _abi.CMX_SB2006_LIMITED, 4, p, p, p, p, p, None, p, None, None, None) == _abi.CMX_ERR_BAD_ARG
    assert fn(None, None, None, 0.0, 0, 4, p, p, p, p, p, p, None, None, None, None) == _abi.CMX_ERR_BAD_ARG      # Z_1m without `rain`
    assert fn(None, None, None, 0.0, 0, 4, p, p, p, p, p, None, None, None, None, None) == _abi.CMX_ERR_BAD_ARG   # no output
    with pytest.raises(ValueError):
        CD.effective_radius_Liu_Hallet_97(1000.0, t(1.0), t(1e-4), t(1e8))
