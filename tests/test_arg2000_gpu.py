"""GPU parity tests of the fused ARG2000 aerosol-activation kernel through the C ABI: the data the reference's
tests hold (digitised Fig. 1, κ-vs-B consistency), random-state parity against the oracle for the BASELINE config-3
distribution (5 modes) and other mode counts, the liquid / ice sink path, ragged / unaligned inputs and the 1e8-state
size-independent properties."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

import parity
from cmx import _abi
from cmx import parameters as P
from cmx import synthetic
from cmx.aerosol import AerosolDistribution, Mode_B, Mode_kappa

pytestmark = pytest.mark.gpu
DT = {"f32": torch.float32, "f64": torch.float64}
G = json.loads((Path(__file__).parent / "golden" / "arg2000_kats.json").read_text())


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _params(ft):
    return P.AerosolActivationParameters(ft), P.AirProperties(ft), P.ThermodynamicsParameters(ft)


def _erfc_form(oracle, a64, adc, i64, t64, cols64, ft):
    """N_i ½ erfc(u_i) per mode with the oracle's own u_i: the reference's activated number without the cancellation of its `1 − erf(u)`
    (src/AerosolActivation.jl:257 — its M_activated_per_mode :319 uses erfc for this very reason)."""
    from scipy.special import erfc
    u = oracle.arg2000_erf_argument(_abi.F64, a64, adc, i64, t64, *cols64, float32_gates=(ft == "f32"))
    return [0.5 * adc.modes[k].N * erfc(u[k]) for k in range(adc.n_modes)], u


# the reference's Float64 N ½ (1 − erf u) has three correct digits up to u = 5 (eps/2 ÷ erfc(5) = 7e-5): below it the activated NUMBER is held to the
# plain north-star bound at EVERY state, like any other output column
U_PLAIN = 5.0


def _compare(got, ref, adc, ft, what, erfc_form=None, amp=None):
    """S_max is a plain product: the plain bound.  M_act = M ½ erfc(·) against the oracle (which evaluates erfc itself, AA:319), on the operand scale M_i
    and, where more than 1e-3 of the mode activates, against |ref|.

    N_act: the reference forms N ½ (1 − erf u) (src/AerosolActivation.jl:257), which in Float64 equals N ½ erfc(u) to three digits up to u = 5 and is 0
    beyond u = 5.9.  With `erfc_form` = (N ½ erfc(u), u) from the oracle's own u (the cancellation-free statement of the same number):
      u < 5            |x − N ½ erfc(u)| ≤ rtol·N ½ erfc(u) at every state (assert_parity without an operand allowance: the rows are class A), after
                       checking that the oracle's literal ½(1 − erf u) agrees with the erfc form to 1e-4 there;
      5 ≤ u < u_max    rtol times the conditioning of erfc, max(1, 2u²) (the reference itself has fewer than three digits left);
      beyond           non-negative and below the last representable value.
    Without `erfc_form` the activated number is compared on the operand scale N_i only.

    `amp` (sink correction only, AA:187-197): the operand amplification ≥ 1 of the numerator αw − K_ice(ξ − 1) = αw − K_ice ξ + K_ice of S_max — a difference
    whose ξ − 1 = p_vs/p_vi − 1 itself cancels towards the triple point (test_liquid_and_ice_sinks forms it).  An error ε·amp of S_max is an error
    δu = 2/(3√2 ln σ)·ε·amp of u and |d ln erfc/du|·δu ≤ max(2u, 2/√π)·δu of the activated number: that product times (amp − 1) is the operand scale handed
    to assert_parity (its CTOL allowance, and its definition of a well-conditioned state; nothing without ice, where amp = 1 and the bound is the plain one)."""
    rtol, kap = parity.RTOL[ft], parity.CTOL[ft] / parity.RTOL[ft]
    rep = {}
    pin = "oracle restatement of src/AerosolActivation.jl:35-433 (pinned to 5-10 % by the reference's Fig.-1 test and an mpmath restatement)"
    fam = "ARG2000 (a3)"
    e = parity.scaled_err(got.S_max.cpu().numpy(), ref["S_max"], None, parity.FLOOR[ft], parity.CEIL[ft], kap)
    rep["S_max"] = float(np.nan_to_num(e, nan=np.inf).max())
    for name, grp, sc in (("N_act", got.N_act, lambda m: m.N), ("M_act", got.M_act, lambda m: m.molar_mass_mix)):
        if grp is None:
            continue
        for k in range(adc.n_modes):
            x = grp[k].cpu().numpy()
            e = parity.scaled_err(x, ref[name][k], np.full(x.shape, sc(adc.modes[k])), parity.FLOOR[ft], parity.CEIL[ft], kap)
            rep[f"{name}[{k}]"] = float(np.nan_to_num(e, nan=np.inf).max())
    worst = max(rep.values())
    assert worst <= rtol, (what, rep)
    parity.record("ARG2000 " + what, ft, {"S_max": got.S_max.cpu().numpy()}, {"S_max": ref["S_max"]}, family=fam, pinned_by=pin, assert_wellcond=True)
    if got.M_act is not None:
        for k in range(adc.n_modes):
            x = got.M_act[k].cpu().numpy()
            parity.record("ARG2000 " + what, ft, {f"M_act[{k}]": x}, {f"M_act[{k}]": ref["M_act"][k]}, family=fam, pinned_by=pin,
                          scale={f"M_act[{k}]": np.full(x.shape, adc.modes[k].molar_mass_mix)}, wellcond=1e-3, assert_wellcond=True,
                          note="well-conditioned = more than 1e-3 of the mode's total activates")
    if got.N_act is None:
        return rep
    for k in range(adc.n_modes):
        x = got.N_act[k].cpu().numpy()
        xx = x.astype(np.float64)
        key = f"N_act[{k}]"
        if erfc_form is None:
            parity.record("ARG2000 " + what, ft, {key: x}, {key: ref["N_act"][k]}, family=fam, pinned_by=pin,
                          scale={key: np.full(x.shape, adc.modes[k].N)}, wellcond=1e-3, assert_wellcond=True,
                          note="well-conditioned = more than 1e-3 of the mode's total activates")
            continue
        r_k, u = erfc_form[0][k], erfc_form[1][k]
        with np.errstate(invalid="ignore"):
            plain = u < U_PLAIN                                     # (a NaN u — w ≤ 0 — is compared by the callers that construct such states)
        # the literal reference value and its erfc form are the same number where the comparison is plain
        lit = ref["N_act"][k][plain]
        assert np.all(np.abs(lit - r_k[plain]) <= 1e-4 * r_k[plain] + 1e-300), (what, key, "oracle 1 − erf vs erfc")
        refd = {key: r_k[plain]}
        if amp is not None:
            du = 2.0 / (3.0 * math.sqrt(2.0) * math.log(adc.modes[k].stdev))
            refd["scale"] = {key: (r_k * np.maximum(2.0 * u, 2.0 / math.sqrt(math.pi)) * du * (amp - 1.0))[plain]}
        parity.assert_parity({key: xx[plain]}, refd, rtol, names=[key], what=f"ARG2000 {what} (u < {U_PLAIN:g})", family=fam, pinned_by=pin,
                             note=f"{int(plain.sum())} of {plain.size} states have u < {U_PLAIN:g}; reference = N ½ erfc(u) with the oracle's u "
                                  "(= the reference's ½(1 − erf u) to 1e-4 there, checked)")
        tiny = adc.modes[k].N * (1e-30 if ft == "f32" else 1e-280)            # below: not representable next to the Float32 / Float64 range
        # Float64: the device's erfc is a table of polynomials on [0, 6.5) (csrc/cmx_lean_f64.hpp: the reference's ½(1 − erf u) is EXACTLY 0 beyond
        # u = 5.9); beyond it the value is e^{−u²} times the last interval's factor — positive, monotone, within a factor ≈ u/6.5 of the true tail
        u_max = 6.5 if ft == "f64" else np.inf
        with np.errstate(invalid="ignore"):
            live = (r_k > tiny) & (u >= U_PLAIN) & (u < u_max)
            bound = rtol * np.maximum(1.0, 2.0 * u * u) * r_k
        assert np.all(np.abs(xx[live] - r_k[live]) <= bound[live]), (what, key, float(np.max(np.abs(xx[live] - r_k[live]) / bound[live])))
        far = u >= u_max
        if np.any(far):
            from scipy.special import erfc as _erfc
            assert np.all((xx[far] >= 0) & (xx[far] <= 0.5 * adc.modes[k].N * _erfc(6.5) * (1 + 1e-6))), (what, key)
        assert np.all(xx[(r_k <= tiny) & ~far & ~plain] <= 2 * tiny), (what, key)
    return rep


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_reference_test_data_through_the_abi(dev, ft):
    import cmx
    ap, aip, tps = _params(ft)
    t64 = P.ThermodynamicsParameters("f64")
    T, p, w = G["conditions"]["T"], G["conditions"]["p"], G["conditions"]["w"]
    dcl = t64.cp_v - t64.cp_l
    p_vs = t64.press_triple * (T / t64.T_triple) ** (dcl / t64.R_v) * math.exp((t64.LH_v0 - dcl * t64.T_0) / t64.R_v * (1 / t64.T_triple - 1 / T))
    q_vs = 1 / (1 - (t64.R_v / t64.R_d) * (p_vs - p) / p_vs)
    col = lambda v: torch.full((8,), v, dtype=DT[ft], device=dev)  # noqa: E731
    f = G["fig1"]
    s = P.Sulfate(ft)
    for chem, rtol in (("B", f["rtol_B"]), ("k", f["rtol_kappa"])):
        mk = (lambda N: Mode_B(0.05e-6, 2.0, N, (1.0,), (s.eps,), (s.phi,), (s.M,), (s.nu,), (s.rho,))) if chem == "B" else (
            lambda N: Mode_kappa(0.05e-6, 2.0, N, (1.0,), (1.0,), (s.M,), (s.kappa,)))
        frac = []
        for N2 in f["N_2_per_cm3"]:
            r = cmx.aerosol_activation(ap, AerosolDistribution([mk(100e6), mk(N2 * 1e6)]), aip, tps, col(T), col(p), col(w), col(q_vs))
            frac.append(r.N_act[0][0].item() / 100e6)
        obs = np.array(f["N_act_fraction_mode1"])
        assert np.linalg.norm(np.array(frac) - obs) / max(np.linalg.norm(frac), np.linalg.norm(obs)) <= rtol   # Julia vector isapprox
    g = G["gpu_consistency"]
    for m in g["modes"]:
        B = Mode_B(m["r_dry"], m["stdev"], m["N"], (1.0,), (m["eps"],), (m["phi"],), (m["M"],), (m["nu"],), (m["rho"],))
        K = Mode_kappa(m["r_dry"], m["stdev"], m["N"], (1.0,), (1.0,), (m["M"],), (m["kappa"],))
        rB = cmx.aerosol_activation(ap, AerosolDistribution([B]), aip, tps, col(T), col(p), col(w), col(q_vs), want=("N_act", "M_act"))
        rK = cmx.aerosol_activation(ap, AerosolDistribution([K]), aip, tps, col(T), col(p), col(w), col(q_vs), want=("N_act", "M_act"))
        tol = g["rtol_act"] if ft == "f64" else 2e-5
        assert math.isclose(rB.N_act[0][0].item(), rK.N_act[0][0].item(), rel_tol=tol)
        assert math.isclose(rB.M_act[0][0].item(), rK.M_act[0][0].item(), rel_tol=tol)
        assert rB.N_act[0][0].item() > 0


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_reference_property_tests_through_the_abi(dev, ft):
    """test/aerosol_activation_tests.jl:134-234 on the device, both parameter sets (default, PySDM-calibrated), B- and κ-type modes, the reference's
    conditions (T = 294 K, p = 1e5 Pa, w = 0.5 m/s, saturated, N_liq = N_ice = 1000): "callable and positive" (> 0 without sinks, ≥ 0 with), "same mean
    hygroscopicity for the same aerosol" (==), "B and kappa hygroscopicities are equivalent" (rtol 0.1) and "order of modes does not matter" — which the
    reference asserts with `==` on total_N_activated and total_M_activated: the kernel sums separately rounded terms for that reason (cmx_arg.hpp)."""
    import cmx
    t64 = P.ThermodynamicsParameters("f64")
    aip, tps = P.AirProperties(ft), P.ThermodynamicsParameters(ft)
    T, p, w = 294.0, 1e5, 0.5
    dcl = t64.cp_v - t64.cp_l
    p_vs = t64.press_triple * (T / t64.T_triple) ** (dcl / t64.R_v) * math.exp((t64.LH_v0 - dcl * t64.T_0) / t64.R_v * (1 / t64.T_triple - 1 / T))
    q_vs = 1 / (1 - (t64.R_v / t64.R_d) * (p_vs - p) / p_vs)
    col = lambda v: torch.full((5,), v, dtype=DT[ft], device=dev)  # noqa: E731
    ss = P.Seasalt(ft)
    mB = lambda r, sd, N: Mode_B(r, sd, N, (1.0,), (ss.eps,), (ss.phi,), (ss.M,), (ss.nu,), (ss.rho,))  # noqa: E731
    mK = lambda r, sd, N: Mode_kappa(r, sd, N, (1.0,), (1.0,), (ss.M,), (ss.kappa,))  # noqa: E731
    accum, coarse = (0.243e-6, 1.4, 100e6), (1.5e-6, 2.1, 1e6)
    state = (col(T), col(p), col(w), col(q_vs), col(0.0), col(0.0))
    sinks = (col(1000.0), col(1000.0))
    for override in (None, P.ARG2000_CALIBRATED_OVERRIDE):
        ap = P.AerosolActivationParameters(P.create_toml_dict(ft, override))
        for mk in (mB, mK):
            am1, am2, am3 = (AerosolDistribution(m) for m in ([mk(*accum)], [mk(*coarse), mk(*accum)], [mk(*accum), mk(*coarse)]))
            assert all(m.hygroscopicity(ap) > 0 for m in am3.modes)
            assert am3.modes[0].hygroscopicity(ap) == am1.modes[0].hygroscopicity(ap)                      # same aerosol → same mean hygroscopicity
            r = cmx.aerosol_activation(ap, am3, aip, tps, *state, want=("N_act", "M_act", "S_max"))
            rs = cmx.aerosol_activation(ap, am3, aip, tps, *state, *sinks, want=("N_act", "M_act", "S_max"))
            assert bool((r.S_max > 0).all()) and bool((rs.S_max >= 0).all())
            for k in range(2):
                assert bool((r.N_act[k] > 0).all()) and bool((r.M_act[k] > 0).all())
                assert bool((rs.N_act[k] >= 0).all()) and bool((rs.M_act[k] >= 0).all())
            # totals, and their independence of the order of the modes — bit for bit, like the reference's `==`
            t3 = cmx.total_activated(ap, am3, aip, tps, *state)
            t2 = cmx.total_activated(ap, am2, aip, tps, *state)
            assert bool((t3[0] > 0).all()) and bool((t3[1] > 0).all())
            assert torch.equal(t3[0], t2[0]) and torch.equal(t3[1], t2[1])
            ts = cmx.total_activated(ap, am3, aip, tps, *state, *sinks)
            assert bool((ts[0] >= 0).all()) and bool((ts[1] >= 0).all())
        kap, B = mK(*coarse).hygroscopicity(ap), mB(*coarse).hygroscopicity(ap)
        assert abs(kap - B) <= 0.1 * max(abs(kap), abs(B))


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("nmodes", [5, 1, 2, 8])
def test_random_state_parity(dev, oracle, ft, nmodes):
    import cmx
    from cmx import synthetic
    n = 1_000_003 if nmodes == 5 else 100_001
    st = synthetic.arg_state(n, dtype=DT[ft], seed=1234)
    base = synthetic.arg_config3_distribution().modes
    ad = AerosolDistribution([base[k % 5] for k in range(nmodes)])
    ap, aip, tps = _params(ft)
    r = cmx.aerosol_activation(ap, ad, aip, tps, *[c.to(dev) for c in st], want=("N_act", "M_act", "S_max"))
    torch.cuda.synchronize()
    a64, i64, t64 = _params("f64")
    adc = ad.c_struct(a64, _abi.F64)
    ref = oracle.arg2000_activation(_abi.F64, a64, adc, i64, t64, *[c.numpy().astype(np.float64) for c in st], nthreads=8,
                                    float32_gates=(ft == "f32"))
    cols64 = [c.numpy().astype(np.float64) for c in st]
    rep = _compare(r, ref, adc, ft, f"{ft} {nmodes} modes", erfc_form=_erfc_form(oracle, a64, adc, i64, t64, cols64, ft))
    print(f"\n[ARG parity] {ft} {nmodes} modes n={n}: worst {max(rep.values()):.2e} ({max(rep, key=rep.get)})")
    # activated fractions are fractions
    for k in range(nmodes):
        assert bool((r.N_act[k] >= 0).all()) and bool((r.N_act[k] <= ad.modes[k].N * (1 + 1e-6)).all())


@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_liquid_and_ice_sinks(dev, oracle, ft):
    """12-argument methods (AA:138-200 with N_liq, N_ice): pre-existing droplets / crystals lower S_max."""
    import cmx
    from cmx import synthetic
    n = 200_000
    st = synthetic.arg_state(n, dtype=DT[ft], seed=7)
    g = torch.Generator().manual_seed(3)
    u = lambda: torch.rand(n, dtype=torch.float64, generator=g)  # noqa: E731
    q_liq = (1e-4 * u()).to(DT[ft])
    q_ice = (1e-5 * u() * (st.T.double() < 273.15)).to(DT[ft])
    N_liq = torch.where(u() < 0.2, torch.zeros(n, dtype=torch.float64), torch.exp(math.log(1e6) + u() * math.log(1e3))).to(DT[ft])
    N_ice = torch.where(u() < 0.5, torch.zeros(n, dtype=torch.float64), torch.exp(math.log(1e2) + u() * math.log(1e3))).to(DT[ft])
    q_tot = (st.q_tot.double() + q_liq.double() + q_ice.double()).to(DT[ft])
    cols = (st.T, st.p, st.w, q_tot, q_liq, q_ice, N_liq, N_ice)
    ad = synthetic.arg_config3_distribution()
    ap, aip, tps = _params(ft)
    r = cmx.aerosol_activation(ap, ad, aip, tps, *[c.to(dev) for c in cols], want=("N_act", "S_max"))
    r0 = cmx.aerosol_activation(ap, ad, aip, tps, *[c.to(dev) for c in cols[:6]], want=("S_max",))
    torch.cuda.synchronize()
    a64, i64, t64 = _params("f64")
    adc = ad.c_struct(a64, _abi.F64)
    ref = oracle.arg2000_activation(_abi.F64, a64, adc, i64, t64, *[c.numpy().astype(np.float64) for c in cols], nthreads=8,
                                    float32_gates=(ft == "f32"))
    # S_max = S_ARG (αw − K_ice(ξ−1)) / (…): where the ice sink nearly cancels the updraught source the numerator is a
    # small difference of large terms (oracle: S_cond = amplification ≥ 1).  Strict parity on the well-conditioned
    # states (amplification < 8); on the rest the error must stay within amplification × tolerance.
    well = ref["S_cond"] < 8.0
    assert well.mean() > 0.5
    sel = lambda t: None if t is None else tuple(c[torch.from_numpy(well).to(c.device)] for c in t)  # noqa: E731
    sub = cmx.ActivationResult(sel(r.N_act), None, r.S_max[torch.from_numpy(well).to(dev)])
    ref_w = dict(N_act=[a[well] for a in ref["N_act"]], M_act=None, S_max=ref["S_max"][well])
    ef = _erfc_form(oracle, a64, adc, i64, t64, [c.numpy().astype(np.float64) for c in cols], ft)
    # operand amplification of the numerator for the activated number: S_cond counts the operands αw and K_ice(ξ − 1); ξ − 1 = p_vs/p_vi − 1 is itself a
    # difference (0.007 at 272.45 K), so in terms of the operands αw, K_ice ξ, K_ice the amplification is 1 + (S_cond − 1)·ξ/(ξ − 1)
    T64 = cols[0].numpy().astype(np.float64)
    xi = np.array([oracle.psat_liquid(_abi.F64, t64, float(t)) / oracle.psat_ice(_abi.F64, t64, float(t)) for t in np.unique(T64)])
    xi = xi[np.searchsorted(np.unique(T64), T64)]
    with np.errstate(divide="ignore", invalid="ignore"):
        amp_n = np.where(ref["S_cond"] > 1.0, 1.0 + (ref["S_cond"] - 1.0) * xi / np.abs(xi - 1.0), 1.0)
    _compare(sub, ref_w, adc, ft, f"{ft} sinks (well-conditioned)", erfc_form=([a[well] for a in ef[0]], [a[well] for a in ef[1]]), amp=amp_n[well])
    got_s = r.S_max.cpu().numpy().astype(np.float64)
    err = np.abs(got_s - ref["S_max"]) / np.maximum(ref["S_max"], 1e-12)
    pos = ref["S_max"] > 0
    assert np.all(err[pos & ~well] <= parity.RTOL[ft] * ref["S_cond"][pos & ~well])
    assert bool((r.S_max <= r0.S_max * (1 + 1e-5)).all())


@pytest.mark.parametrize("n", [0, 1, 3, 5, 257, 1023])
def test_ragged_and_unaligned(dev, oracle, n):
    import cmx
    from cmx import synthetic
    ad = synthetic.arg_config3_distribution()
    ap, aip, tps = _params("f32")
    st = [c.to(dev) for c in synthetic.arg_state(n + 1, seed=n)]
    a = cmx.aerosol_activation(ap, ad, aip, tps, *[c[1:] for c in st])             # misaligned by 4 bytes
    b = cmx.aerosol_activation(ap, ad, aip, tps, *[c[1:].clone() for c in st])
    assert a.N_act[0].shape == (n,) and a.M_act is None and a.S_max is None
    for x, y in zip(a.N_act, b.N_act):
        assert torch.equal(x, y)
    with pytest.raises(TypeError):
        cmx.aerosol_activation(P.AerosolActivationParameters("f64"), ad, aip, tps, *st)


def test_full_size_1e8_f32_properties(dev, oracle):
    """BASELINE config 3: 5 lognormal modes × 1e8 thermodynamic states, Float32."""
    import cmx
    from cmx import sharding, synthetic
    n = 100_000_000
    st = synthetic.arg_state(n, dtype=torch.float32, device=dev, seed=1234)
    ad = synthetic.arg_config3_distribution()
    ap, aip, tps = _params("f32")
    full = cmx.aerosol_activation(ap, ad, aip, tps, *st)
    torch.cuda.synchronize()
    for k, col in enumerate(full.N_act):
        assert bool(torch.isfinite(col).all()) and bool((col >= 0).all()) and bool((col <= ad.modes[k].N * (1 + 1e-6)).all())
    for lo, hi in ((0, 4096), (12_345_677, 12_400_001), (n - 1_000_003, n)):
        part = cmx.aerosol_activation(ap, ad, aip, tps, *[c[lo:hi] for c in st])
        for a, b in zip(full.N_act, part.N_act):
            assert torch.equal(a[lo:hi], b), (lo, hi)
    tot = cmx.column_sums(list(full.N_act))
    acc = torch.zeros_like(tot)
    for r in range(8):
        lo, hi = sharding.shard_bounds(n, r, 8)
        acc += cmx.column_sums([c[lo:hi] for c in full.N_act])
    assert torch.allclose(tot, acc, rtol=1e-9, atol=0)
    # monotonicity in updraught speed (stronger updraught → more activation), checked on sorted samples of one chunk
    stride = 101
    samp = [c[::stride].contiguous().cpu().numpy().astype(np.float64) for c in st]
    a64, i64, t64 = _params("f64")
    adc = ad.c_struct(a64, _abi.F64)
    ref = oracle.arg2000_activation(_abi.F64, a64, adc, i64, t64, *samp, nthreads=8, float32_gates=True)
    got = cmx.ActivationResult(tuple(c[::stride].contiguous() for c in full.N_act), None, torch.from_numpy(ref["S_max"]))
    rep = _compare(got, ref, adc, "f32", "1e8 sample", erfc_form=_erfc_form(oracle, a64, adc, i64, t64, samp, "f32"))
    print(f"\n[ARG parity 1e8 f32, {samp[0].size} sampled states] worst {max(rep.values()):.2e}")


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_spatially_varying_aerosol_columns(dev, oracle, ft):
    """Mode descriptors as per-state columns (the reference's aerosol_activation_kernel!, test/gpu_tests.jl:45-79,549-587):
    the reference's own κ-vs-B consistency check on its two test elements, equality with the shared-distribution entry
    when the columns are constant, and random-column parity with the oracle."""
    import cmx
    dt = {"f32": torch.float32, "f64": torch.float64}[ft]
    ap, aip, tps = P.AerosolActivationParameters(ft), P.AirProperties(ft), P.ThermodynamicsParameters(ft)
    col = lambda v: torch.tensor(v, dtype=dt, device=dev)  # noqa: E731
    # the two elements of the reference test: sulfate-like and seasalt-like single-mode aerosol
    r, sd, N = [0.243e-6, 1.5e-6], [1.4, 2.1], [100e6, 1e6]
    eps_, phi, Mm, nu, rho_a, kappa = [1.0, 1.0], [1.0, 0.9], [0.132, 0.058443], [3.0, 2.0], [1770.0, 2170.0], [0.53, 1.12]
    T = col([294.0, 294.0]); p = col([1e5, 1e5]); w = col([0.5, 0.5])
    p_vs = np.array([oracle.psat_liquid(_abi.F64, P.ThermodynamicsParameters("f64"), 294.0)] * 2)
    q_vs = 1 / (1 - 1 / (tps.R_d / tps.R_v) * (p_vs - 1e5) / p_vs)
    q_tot = col(q_vs.tolist())
    mB = cmx.Mode_B(col(r), col(sd), col(N), (1.0,), (col(eps_),), (col(phi),), (col(Mm),), (col(nu),), (col(rho_a),))
    mK = cmx.Mode_kappa(col(r), col(sd), col(N), (1.0,), (1.0,), (col(Mm),), (col(kappa),))
    res = {}
    for name, m in (("B", mB), ("kappa", mK)):
        mc = cmx.ModeColumns(m.r_dry, m.stdev, m.N, m.hygroscopicity(ap), m.molar_mass[0] * 1.0)
        res[name] = cmx.aerosol_activation_columns(ap, [mc], aip, tps, T, p, w, q_tot, want=("N_act", "M_act"))
    for k in range(2):   # gpu_tests.jl:580-587
        assert float(res["B"].N_act[0][k]) == pytest.approx(float(res["kappa"].N_act[0][k]), rel=0.3)
        assert float(res["B"].N_act[0][k]) > 0 and float(res["B"].M_act[0][k]) > 0
    # constant columns == the shared-distribution entry
    n = 50_001
    st = synthetic.arg_state(n, dtype=dt, device=dev, seed=9)
    ad = synthetic.arg_config3_distribution()
    shared = cmx.aerosol_activation(ap, ad, aip, tps, *st, want=("N_act", "S_max"))
    adc = ad.c_struct(ap, _abi.family(ft))
    full = lambda v: torch.full((n,), float(v), dtype=dt, device=dev)  # noqa: E731
    modes = [cmx.ModeColumns(full(adc.modes[k].r_dry), full(adc.modes[k].stdev), full(adc.modes[k].N), full(adc.modes[k].hygroscopicity))
             for k in range(adc.n_modes)]
    varying = cmx.aerosol_activation_columns(ap, modes, aip, tps, *st, want=("N_act", "S_max"))
    rt = 1e-9 if ft == "f64" else 2e-4
    assert torch.allclose(varying.S_max, shared.S_max, rtol=rt, atol=0)
    for a, b, m in zip(varying.N_act, shared.N_act, modes):
        assert float(((a - b).abs() / m.N).max()) <= rt
        big = b > 1e-3 * m.N
        assert float(((a - b).abs() / b)[big].max()) <= (1e-8 if ft == "f64" else 2e-4)
    # random columns vs the oracle
    g = torch.Generator(device="cpu").manual_seed(4)
    u = lambda lo, hi: (lo + (hi - lo) * torch.rand(n, generator=g, dtype=torch.float64)).to(dt)  # noqa: E731
    rm = [(10 ** u(-8.0, -6.0), u(1.3, 2.2), 10 ** u(6.0, 9.5), u(0.1, 1.3), u(0.05, 0.15)) for _ in range(3)]
    got = cmx.aerosol_activation_columns(ap, [cmx.ModeColumns(*[c.to(dev) for c in m]) for m in rm], aip, tps, *st, want=("N_act", "M_act", "S_max"))
    ref = oracle.arg2000_activation_columns(_abi.F64, P.AerosolActivationParameters("f64"), P.AirProperties("f64"),
                                            P.ThermodynamicsParameters("f64"), *[c.cpu().numpy().astype(np.float64) for c in st],
                                            [[c.numpy().astype(np.float64) for c in m] for m in rm], want_M=True,
                                            float32_gates=(ft == "f32"), nthreads=8)
    tol = 1e-6 if ft == "f64" else 1e-3
    sm = got.S_max.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(sm - ref["S_max"]) / ref["S_max"]) <= tol
    pin = "oracle restatement of src/AerosolActivation.jl:35-433 with per-state mode descriptors (test/gpu_tests.jl:45-79)"
    parity.record(f"ARG2000 per-element modes {ft}", ft, {"S_max": sm}, {"S_max": ref["S_max"]}, family="ARG2000 (a3)", pinned_by=pin, assert_wellcond=True)
    for k in range(3):
        Nk = rm[k][2].numpy().astype(np.float64)
        assert np.max(np.abs(got.N_act[k].cpu().numpy() - ref["N_act"][k]) / Nk) <= tol
        Mk = rm[k][4].numpy().astype(np.float64)
        assert np.max(np.abs(got.M_act[k].cpu().numpy() - ref["M_act"][k]) / Mk) <= tol
        # the plain bound on the activated number at EVERY state with u < 5, against N ½ erfc(u) with u restated from the oracle's S_max
        # (u = 2 ln(S_m/S_max)/(3√2 ln σ), S_m = 2/√κ (A/(3 r_dry))^1.5, A = 2 σ_w M_w/(ρ_w R T): AA:35-40,107-118,255) — class A rows like the
        # shared-distribution entry's; the oracle's literal ½(1 − erf u) agrees with it to 1e-4 there (checked)
        from scipy.special import erfc as _erfc
        a64 = P.AerosolActivationParameters("f64")
        T64 = st[0].cpu().numpy().astype(np.float64)
        A = 2 * a64.sigma * a64.M_w / (a64.rho_w * a64.R * T64)
        r_k, sd_k, hy_k = (rm[k][j].numpy().astype(np.float64) for j in (0, 1, 3))
        Sm = 2 / np.sqrt(hy_k) * (A / (3 * r_k)) ** 1.5
        u_k = 2 * np.log(Sm / ref["S_max"]) / (3 * np.sqrt(2.0) * np.log(sd_k))
        n_erfc = 0.5 * Nk * _erfc(u_k)
        plain = u_k < U_PLAIN
        assert plain.mean() > 0.3
        lit = ref["N_act"][k][plain]
        assert np.all(np.abs(lit - n_erfc[plain]) <= 1e-4 * n_erfc[plain] + 1e-300)
        gk = got.N_act[k].cpu().numpy().astype(np.float64)
        parity.assert_parity({f"N_act[{k}]": gk[plain]}, {f"N_act[{k}]": n_erfc[plain]}, tol, names=[f"N_act[{k}]"],
                             what=f"ARG2000 per-element modes {ft} (u < {U_PLAIN:g})", family="ARG2000 (a3)", pinned_by=pin,
                             note="reference = N ½ erfc(u), u from the oracle's S_max")
        # … and against |ref| itself wherever more than 1e-3 of the mode activates (round 4: a constant wrong in the 5th digit passed the
        # bound relative to the mode TOTAL in Float32)
        parity.record(f"ARG2000 per-element modes {ft}", ft, {f"M_act[{k}]": got.M_act[k].cpu().numpy()}, {f"M_act[{k}]": ref["M_act"][k]},
                      family="ARG2000 (a3)", pinned_by=pin, scale={f"M_act[{k}]": Mk}, wellcond=1e-3, assert_wellcond=True)


def test_small_activated_fractions_keep_relative_accuracy_f32(dev, oracle):
    """The reference forms M_act with erfc itself (AA:319), so a weakly activated mode keeps its leading digits; its N_act = N/2 (1 − erf u) (AA:257)
    keeps three digits in Float64 up to u = 5.  The Float32 kernel must too: a relative-accuracy erfc for BOTH (rounds 1-5 used the
    absolute-accuracy A&S 7.1.26 for the number).  Weak updraughts: activated fractions from 1e-1 down to 1e-12 of the mode, pure relative bound 1e-3."""
    import cmx
    from cmx import synthetic
    from scipy.special import erfc
    ft, n = "f32", 400_000
    st = synthetic.arg_state(n, dtype=DT[ft], seed=99)
    g = torch.Generator().manual_seed(5)
    w = torch.exp(torch.log(torch.tensor(1e-4)) + torch.rand(n, generator=g) * torch.log(torch.tensor(3e3))).to(DT[ft])     # 1e-4 … 0.3 m/s
    st = st._replace(w=w)
    ad = synthetic.arg_config3_distribution()
    ap, aip, tps = _params(ft)
    r = cmx.aerosol_activation(ap, ad, aip, tps, *[c.to(dev) for c in st], want=("N_act", "M_act"))
    torch.cuda.synchronize()
    a64, i64, t64 = _params("f64")
    adc = ad.c_struct(a64, _abi.F64)
    cols64 = [c.numpy().astype(np.float64) for c in st]
    ref = oracle.arg2000_activation(_abi.F64, a64, adc, i64, t64, *cols64, nthreads=8, float32_gates=True)
    n_erfc, u = _erfc_form(oracle, a64, adc, i64, t64, cols64, ft)
    checked = small = checked_n = small_n = 0
    for k in range(5):
        got = r.M_act[k].cpu().numpy().astype(np.float64)
        exp = ref["M_act"][k]
        total = exp.max()                                          # scale of the mode's activated mass in this sample
        sel = exp > 1e-30
        rel = np.abs(got[sel] - exp[sel]) / exp[sel]
        assert rel.max() <= 1e-3, (k, rel.max())
        checked += int(sel.sum())
        small += int((exp[sel] < 1e-4 * total).sum())
        # the activated number at every state with u < 5: fractions down to ½ erfc(5) = 7.7e-13 of the mode
        gn = r.N_act[k].cpu().numpy().astype(np.float64)
        sel = u[k] < U_PLAIN
        rel = np.abs(gn[sel] - n_erfc[k][sel]) / n_erfc[k][sel]
        assert rel.max() <= 1e-3, (k, rel.max())
        checked_n += int(sel.sum())
        small_n += int((n_erfc[k][sel] < 1e-4 * adc.modes[k].N).sum())
        assert 0.5 * erfc(U_PLAIN) < 1e-12
    assert checked > n and small > 1000          # the test did reach weakly activated states
    assert checked_n > n and small_n > 1000


@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_edge_states_updraft_and_cold(dev, oracle, ft):
    """Vanishing, zero and negative updrafts, very cold and very warm states, nearly saturated liquid: where the reference's formula is 0/0 or the
    root of a negative number (w <= 0, AA:168-183) the result is NaN in the oracle AND here (the kernel's exp2_fin / x^(-3/4) forms have no 0 / Inf
    cases to fall back on); everywhere else finite and in parity."""
    import cmx
    rows = []
    for w in (0.0, -0.5, 1e-6, 1e-3, 0.5, 10.0, 40.0):
        for T, p in ((294.0, 1e5), (273.15, 8e4), (235.0, 4e4), (200.0, 2e4), (310.0, 1.02e5)):
            for q_tot, q_liq in ((1e-2, 0.0), (1e-5, 0.0), (2e-2, 5e-3)):
                rows.append((T, p, w, q_tot, q_liq, 0.0))
    arr = np.array(rows, dtype=np.float64).T
    cols = [torch.tensor(a, dtype=DT[ft]) for a in arr]
    ad = synthetic_distribution()
    ap, aip, tps = _params(ft)
    r = cmx.aerosol_activation(ap, ad, aip, tps, *[c.to(dev) for c in cols], want=("N_act", "S_max"))
    torch.cuda.synchronize()
    a64, i64, t64 = _params("f64")
    adc = ad.c_struct(a64, _abi.F64)
    ref = oracle.arg2000_activation(_abi.F64, a64, adc, i64, t64, *[c.numpy().astype(np.float64) for c in cols], nthreads=4, float32_gates=(ft == "f32"))
    smax, rs = r.S_max.cpu().numpy().astype(np.float64), ref["S_max"]
    bad_w = arr[2] <= 0
    assert np.all(np.isnan(rs[bad_w])) and np.all(np.isnan(smax[bad_w]))
    ok = ~bad_w
    assert np.all(np.isfinite(smax[ok])) and np.all(np.isfinite(rs[ok]))
    assert np.all(np.abs(smax[ok] - rs[ok]) <= parity.RTOL[ft] * np.abs(rs[ok]) + 1e-300)
    for k in range(adc.n_modes):
        x = r.N_act[k].cpu().numpy().astype(np.float64)
        assert np.all(np.isnan(x[bad_w])) and np.all(np.isfinite(x[ok]))
        assert np.all(np.abs(x[ok] - ref["N_act"][k][ok]) <= parity.RTOL[ft] * np.abs(ref["N_act"][k][ok]) + parity.CTOL[ft] * adc.modes[k].N)


def synthetic_distribution():
    from cmx import synthetic
    return synthetic.arg_config3_distribution()
