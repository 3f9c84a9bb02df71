"""Synthetic atmospheric state columns for the benchmarks and the parity sweeps.

Mirrors the reference's generator `generate_atmospheric_states`
(test/gpu_performance.jl:80-136): a decaying temperature profile 300 K → 215 K over 0–15 km,
relative humidity sweeping 0.05 → 1.05, condensate = supersaturation excess + small noise —
randomised over the profile instead of a linear ramp, and extended with the number columns and
the threshold / clamp edge cases the 2-moment scheme gates on.
Pure torch, runs on any device; Julia's MersenneTwister stream is not reproducible outside Julia,
so parity never depends on reproducing the reference's random numbers: the oracle is fed the very
same arrays.
"""
from __future__ import annotations

from collections import namedtuple

import torch

from . import parameters as P

State2M = namedtuple("State2M", ["rho", "T", "q_tot", "q_lcl", "n_lcl", "q_rai", "n_rai"])


def _psat_liquid(T, td):
    """Rankine–Kirchhoff p_sat over liquid (same closed form as the kernels; float64 torch)."""
    R_v = td["gas_constant_vapor"]
    dcp = td["isobaric_specific_heat_vapor"] - td["isobaric_specific_heat_liquid"]
    T_tr, p_tr = td["temperature_triple_point"], td["pressure_triple_point"]
    LH, T0 = td["latent_heat_vaporization_at_reference"], td["thermodynamics_temperature_reference"]
    return p_tr * (T / T_tr) ** (dcp / R_v) * torch.exp((LH - dcp * T0) / R_v * (1.0 / T_tr - 1.0 / T))


State1M = namedtuple("State1M", ["rho", "T", "q_tot", "q_lcl", "q_icl", "q_rai", "q_sno"])


def _psat_ice(T, td):
    R_v = td["gas_constant_vapor"]
    dcp = td["isobaric_specific_heat_vapor"] - td["isobaric_specific_heat_ice"]
    T_tr, p_tr = td["temperature_triple_point"], td["pressure_triple_point"]
    LH, T0 = td["latent_heat_sublimation_at_reference"], td["thermodynamics_temperature_reference"]
    return p_tr * (T / T_tr) ** (dcp / R_v) * torch.exp((LH - dcp * T0) / R_v * (1.0 / T_tr - 1.0 / T))


def mp1m_state(n: int, dtype=torch.float32, device="cpu", seed: int = 1234, chunk: int = 1 << 24) -> State1M:
    """n random 1-moment states (ρ, T, q_tot, q_lcl, q_icl, q_rai, q_sno): the reference's
    `generate_atmospheric_states` (test/gpu_performance.jl:80-136) randomised — liquid / ice condensate =
    supersaturation excess over liquid / ice + U[0,1e-4] noise, rain / snow = U[0,1e-4] noise, plus heavier
    precipitation tails, exact zeros and slightly negative values (clamp path), both sides of T_freeze."""
    td = P.DEFAULT_PARAMETERS
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cols = [torch.empty(n, dtype=dtype, device=device) for _ in range(7)]
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        u = lambda: torch.rand(m, dtype=torch.float64, device=device, generator=g)  # noqa: E731
        logu = lambda a, b: torch.exp(torch.log(torch.tensor(a, dtype=torch.float64, device=device)) +  # noqa: E731
                                      u() * torch.log(torch.tensor(b / a, dtype=torch.float64, device=device)))
        z = 15000.0 * u()
        T = torch.clamp(300.0 - 6.5e-3 * z, min=215.0) + (4.0 * u() - 2.0)
        p = 1.0e5 * torch.exp(-z / 8000.0)
        RH = 0.05 + u()
        p_sat = _psat_liquid(T, td)
        eps_m = td["gas_constant_dry_air"] / td["gas_constant_vapor"]
        q_vap = (RH * eps_m * p_sat / (p - (1 - eps_m) * RH * p_sat).clamp(min=1.0)).clamp(max=0.04)
        rho = p / (td["gas_constant_dry_air"] * T * (1.0 + 0.61 * q_vap))
        q_sat_l = p_sat / (rho * td["gas_constant_vapor"] * T)
        q_sat_i = _psat_ice(T, td) / (rho * td["gas_constant_vapor"] * T)
        q_lcl = (q_vap - q_sat_l).clamp(min=0.0) + 1e-4 * u() * (u() < 0.7)
        q_icl = ((q_vap - q_sat_i).clamp(min=0.0) + 1e-4 * u()) * (u() < 0.6)
        q_rai = torch.where(u() < 0.1, logu(1e-7, 5e-3), 1e-4 * u() * (u() < 0.5))
        q_sno = torch.where(u() < 0.1, logu(1e-7, 5e-3), 1e-4 * u() * (u() < 0.5))
        q_tot = q_vap + q_lcl + q_icl + q_rai + q_sno
        for c in (q_lcl, q_icl, q_rai, q_sno):
            r = u()
            c.masked_fill_(r < 0.010, 0.0)
            c.copy_(torch.where((r >= 0.010) & (r < 0.015), -1e-9 * (1 + c.abs()), c))
        for dst, src in zip(cols, (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)):
            dst[lo:lo + m] = src.to(dtype)
    return State1M(*cols)


ArgState = namedtuple("ArgState", ["T", "p", "w", "q_tot"])


def arg_state(n: int, dtype=torch.float32, device="cpu", seed: int = 1234, chunk: int = 1 << 24) -> ArgState:
    """Thermodynamic states for BASELINE config 3: T ~ U[253, 303] K, p ~ U[5e4, 1.02e5] Pa,
    w ~ log-U[0.01, 10] m/s, q_tot = q_vs(T, p)·U[0.98, 1] (q_vs as in test/aerosol_activation_tests.jl:33-34:
    saturated, no condensate)."""
    td = P.DEFAULT_PARAMETERS
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cols = [torch.empty(n, dtype=dtype, device=device) for _ in range(4)]
    rv_rd = td["gas_constant_vapor"] / td["gas_constant_dry_air"]
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        u = lambda: torch.rand(m, dtype=torch.float64, device=device, generator=g)  # noqa: E731
        T = 253.0 + 50.0 * u()
        p = 5e4 + 5.2e4 * u()
        w = torch.exp(-4.605170185988091 + u() * 6.907755278982137)
        p_vs = _psat_liquid(T, td)
        q_vs = 1.0 / (1.0 - rv_rd * (p_vs - p) / p_vs)
        q_tot = q_vs * (0.98 + 0.02 * u())
        for dst, src in zip(cols, (T, p, w, q_tot)):
            dst[lo:lo + m] = src.to(dtype)
    return ArgState(*cols)


def arg_config3_distribution():
    """The 5 lognormal κ-modes of BASELINE config 3: r_dry = (0.01, 0.05, 0.1, 0.25, 1.5) µm,
    σ = (1.6, 2.0, 1.8, 1.4, 2.1), N = (1e9, 1e8, 5e7, 1e8, 1e6) m⁻³, κ = (0.53, 0.53, 1.12, 1.12, 1.12)
    (sulfate, sulfate, sea salt ×3), shared by all states."""
    from .aerosol import AerosolDistribution, Mode_kappa
    spec = [(0.01e-6, 1.6, 1e9, 0.53, 0.132), (0.05e-6, 2.0, 1e8, 0.53, 0.132), (0.1e-6, 1.8, 5e7, 1.12, 0.058443),
            (0.25e-6, 1.4, 1e8, 1.12, 0.058443), (1.5e-6, 2.1, 1e6, 1.12, 0.058443)]
    return AerosolDistribution([Mode_kappa(r, s, N, (1.0,), (1.0,), (M,), (k,)) for r, s, N, k, M in spec])


P3State4 = namedtuple("P3State4", ["rho_q_ice", "rho_n_ice", "rho_q_rim", "rho_b_rim"])


def p3_state(n: int, dtype=torch.float64, device="cpu", seed: int = 1234, chunk: int = 1 << 22) -> P3State4:
    """Prognostic P3 ice columns for BASELINE config 5 (the sweep of test/p3_tests.jl:247-250 randomised):
    L_ice ~ log-U[1e-6, 1e-3] kg/m³, N_ice ~ log-U[1e2, 1e6] m⁻³, F_rim ∈ {0 (30 %), U[0, 0.95]}, ρ_rim ~ U[200, 800];
    ρq_rim = F_rim·L_ice, ρb_rim = ρq_rim/ρ_rim; 1 % of the points have no ice at all (logλ = −Inf path)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cols = [torch.empty(n, dtype=dtype, device=device) for _ in range(4)]
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        u = lambda: torch.rand(m, dtype=torch.float64, device=device, generator=g)  # noqa: E731
        L = torch.exp(-13.815510557964274 + u() * 6.907755278982137)
        N = torch.exp(4.605170185988092 + u() * 9.210340371976184)
        F = torch.where(u() < 0.3, torch.zeros_like(L), 0.95 * u())
        rho_rim = 200.0 + 600.0 * u()
        none = u() < 0.01
        L = torch.where(none, torch.zeros_like(L), L)
        q_rim = F * L
        b_rim = q_rim / rho_rim
        for dst, src in zip(cols, (L, N, q_rim, b_rim)):
            dst[lo:lo + m] = src.to(dtype)
    return P3State4(*cols)


def p3_air_density(n: int, dtype=torch.float64, device="cpu", seed: int = 4321):
    """Air density column for the P3 fall-speed integrals: ρₐ ~ U[0.4, 1.3] kg/m³."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return (0.4 + 0.9 * torch.rand(n, dtype=torch.float64, device=device, generator=g)).to(dtype)


IceNucState = namedtuple("IceNucState", ["T", "a_w", "r"])


def ice_nucleation_state(n: int, dtype=torch.float32, device="cpu", seed: int = 1234, chunk: int = 1 << 24,
                         out_of_range: float = 0.05) -> IceNucState:
    """(T, a_w, r) columns for BASELINE config 4: T ~ U[190, 240] K; Δa_w ~ U[0.26, 0.34] (the Koop
    cubic's validity window) for 95 % of the points and U[0, 0.26) ∪ (0.34, 0.40] for `out_of_range` of them
    (exercises the DomainError → NaN + count path); a_w = a_w_ice(T) + Δa_w; r ~ log-U[1e-8, 1e-5] m."""
    td = P.DEFAULT_PARAMETERS
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cols = [torch.empty(n, dtype=dtype, device=device) for _ in range(3)]
    R_v, T_tr, T0 = td["gas_constant_vapor"], td["temperature_triple_point"], td["thermodynamics_temperature_reference"]
    dcl = td["isobaric_specific_heat_vapor"] - td["isobaric_specific_heat_liquid"]
    dci = td["isobaric_specific_heat_vapor"] - td["isobaric_specific_heat_ice"]
    LHv, LHs = td["latent_heat_vaporization_at_reference"], td["latent_heat_sublimation_at_reference"]
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        u = lambda: torch.rand(m, dtype=torch.float64, device=device, generator=g)  # noqa: E731
        T = 190.0 + 50.0 * u()
        ln_a_ice = ((dci - dcl) / R_v) * torch.log(T / T_tr) + ((LHs - dci * T0) - (LHv - dcl * T0)) / R_v * (1 / T_tr - 1 / T)
        d_in = 0.26 + 0.08 * u()
        w = u()
        d_out = torch.where(w < 0.8125, 0.26 * u(), 0.34 + 0.06 * u())      # |[0,0.26)| : |(0.34,0.40]| = 13 : 3
        delta = torch.where(u() < out_of_range, d_out, d_in)
        a_w = torch.exp(ln_a_ice) + delta
        r = torch.exp(torch.log(torch.tensor(1e-8, dtype=torch.float64, device=device)) + u() * 6.907755278982137)
        for dst, src in zip(cols, (T, a_w, r)):
            dst[lo:lo + m] = src.to(dtype)
    return IceNucState(*cols)


def sb2006_state(n: int, dtype=torch.float32, device="cpu", seed: int = 1234, chunk: int = 1 << 24) -> State2M:
    """n random warm-rain states as 7 contiguous columns (ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)."""
    td = P.DEFAULT_PARAMETERS
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cols = [torch.empty(n, dtype=dtype, device=device) for _ in range(7)]
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        u = lambda: torch.rand(m, dtype=torch.float64, device=device, generator=g)  # noqa: E731
        logu = lambda a, b: torch.exp(torch.log(torch.tensor(a, dtype=torch.float64, device=device)) +  # noqa: E731
                                      u() * (torch.log(torch.tensor(b / a, dtype=torch.float64, device=device))))
        z = 15000.0 * u()
        T = torch.clamp(300.0 - 6.5e-3 * z, min=215.0) + (4.0 * u() - 2.0)
        p = 1.0e5 * torch.exp(-z / 8000.0)
        RH = 0.05 + u()
        p_sat = _psat_liquid(T, td)
        eps_m = td["gas_constant_dry_air"] / td["gas_constant_vapor"]
        q_vap = RH * eps_m * p_sat / (p - (1 - eps_m) * RH * p_sat).clamp(min=1.0)
        q_vap = q_vap.clamp(max=0.04)
        rho = p / (td["gas_constant_dry_air"] * T * (1.0 + 0.61 * q_vap))
        q_sat = p_sat / (rho * td["gas_constant_vapor"] * T)
        q_lcl = (q_vap - q_sat).clamp(min=0.0) + 1e-4 * u() * (u() < 0.7)
        q_rai = 1e-4 * u() * (u() < 0.5)
        heavy = u() < 0.1
        q_rai = torch.where(heavy, logu(1e-7, 5e-3), q_rai)
        n_lcl = torch.where(u() < 0.2, torch.full_like(z, 1e8), logu(1e6, 1e9))
        n_rai = logu(1e1, 1e7)
        q_tot = q_vap + q_lcl + q_rai
        # threshold / clamp edge cases: 1 % exact zeros and 0.5 % slightly negative per q/n column
        for c in (q_lcl, q_rai, n_lcl, n_rai):
            r = u()
            c.masked_fill_(r < 0.010, 0.0)
            c.copy_(torch.where((r >= 0.010) & (r < 0.015), -1e-9 * (1 + c.abs()), c))
        for dst, src in zip(cols, (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)):
            dst[lo:lo + m] = src.to(dtype)
    return State2M(*cols)
