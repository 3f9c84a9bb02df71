"""P3 ice scheme over columns — host-side mirror of `CloudMicrophysics.P3Scheme` for the shape-solver path
(include/cmx.h §7): `state_from_prognostic` / `P3State`, `get_distribution_logλ`, `D_m`, `get_logN₀`.

Reference broadcast being replaced (KA wrappers test/gpu_tests.jl:436-451, test/gpu_performance.jl:59-67):

    state = P3.state_from_prognostic.(Ref(params), ρq_ice, ρn_ice, ρq_rim, ρb_rim)
    logλ  = P3.get_distribution_logλ.(state);   Dₘ = P3.D_m.(state, logλ)
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

import torch

from . import _abi, _lib
from .bulk_tendencies import _check_cols, _fam_of, _ptr
from .parameters import ParametersP3

P3Shape = namedtuple("P3Shape", ["F_rim", "rho_rim", "log_lambda", "D_m", "log_N0"])


def p3_shape(params: ParametersP3, rho_q_ice, rho_n_ice, x3, x4, *, from_state=False,
             want=("log_lambda", "D_m"), brent_iters=0, log_lambda_guess=None, stream=None) -> P3Shape:
    """Solve the P3 size distribution for every point.

    Inputs: ρq_ice [kg/m³], ρn_ice [1/m³] and either the prognostic rime variables (ρq_rim [kg/m³], ρb_rim [m³/m³];
    `state_from_prognostic`, src/P3_particle_properties.jl:101-106) or, with `from_state=True`, (F_rim, ρ_rim) as in
    `P3State(params, L, N, F_rim, ρ_rim)`.  Outputs (`want`): F_rim, rho_rim (the regularised state), log_lambda
    (`get_distribution_logλ`, src/P3_size_distribution.jl:284-320; −inf for absent ice), D_m (mass-weighted mean
    diameter, src/P3_integral_properties.jl:56-61) and log_N0 (:233-237).
    `brent_iters` = 0 keeps the reference's fixed Brent budget (8 Float32 / 10 Float64 iterations, :311);
    `log_lambda_guess` is the reference's optional warm start (`_narrow_bracket`, :336-353)."""
    if not isinstance(params, ParametersP3):
        raise TypeError("params must be ParametersP3")
    cols = (rho_q_ice, rho_n_ice, x3, x4)
    names = ("rho_q_ice", "rho_n_ice", "x3", "x4")
    if log_lambda_guess is not None:
        cols, names = cols + (log_lambda_guess,), names + ("log_lambda_guess",)
    ref = _check_cols(cols, names)
    fam = _fam_of(ref)
    if fam is not params.fam:
        raise TypeError("parameter float type does not match the state columns")
    unknown = set(want) - set(P3Shape._fields)
    if unknown:
        raise ValueError(f"unknown output(s) {sorted(unknown)}")
    outs = {k: (torch.empty_like(ref) if k in want else None) for k in P3Shape._fields}
    flags = params.flags | (_abi.CMX_P3_INPUT_IS_STATE if from_state else 0)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_p3_shape_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(params.c), flags, int(brent_iters), ref.numel(), *[_ptr(t) for t in cols[:4]],
                _ptr(log_lambda_guess) if log_lambda_guess is not None else None,
                *[_ptr(outs[k]) for k in P3Shape._fields], C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return P3Shape(*[outs[k] for k in P3Shape._fields])


P3Velocities = namedtuple("P3Velocities", ["v_n", "v_m"])


def p3_terminal_velocities(params: ParametersP3, velocity_params, rho_air, rho_q_ice, rho_n_ice, x3, x4, log_lambda, *,
                           from_state=False, aspect_ratio=True, p=1e-6, quad=None, want=("v_n", "v_m"),
                           stream=None) -> P3Velocities:
    """Number- and mass-weighted ice fall speeds for every point — `P3.ice_terminal_velocity_number_weighted` /
    `_mass_weighted(velocity_params, ρₐ, state, logλ; p, quad)` (src/P3_terminal_velocity.jl:72-137) and their
    `*_from_prognostic` wrappers (:152-178).  `velocity_params` = `parameters.Chen2022VelTypeIce(FT)`; `quad` =
    `parameters.ChebyshevGauss(FT, n)` (default n = 100, as in the reference) or `parameters.GaussLegendre(FT, n)`;
    `aspect_ratio=False` ↔ `ParametersP3(FT; aspect_ratio = NoAspectRatio())`."""
    if not isinstance(params, ParametersP3):
        raise TypeError("params must be ParametersP3")
    cols = (rho_q_ice, rho_n_ice, x3, x4, rho_air, log_lambda)
    ref = _check_cols(cols, ("rho_q_ice", "rho_n_ice", "x3", "x4", "rho_air", "log_lambda"))
    fam = _fam_of(ref)
    if fam is not params.fam or not isinstance(velocity_params, fam.chen2022_ice_vel):
        raise TypeError("parameter float type does not match the state columns")
    if quad is None:
        from .parameters import ChebyshevGauss
        quad = ChebyshevGauss(fam.sfx, 100)
    if not isinstance(quad, fam.quadrature):
        raise TypeError("quadrature float type does not match the state columns")
    unknown = set(want) - set(P3Velocities._fields)
    if unknown:
        raise ValueError(f"unknown output(s) {sorted(unknown)}")
    outs = {k: (torch.empty_like(ref) if k in want else None) for k in P3Velocities._fields}
    flags = params.flags | (_abi.CMX_P3_INPUT_IS_STATE if from_state else 0) | (0 if aspect_ratio else _abi.CMX_P3_NO_ASPECT_RATIO)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_p3_terminal_velocities_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(params.c), C.byref(velocity_params), C.byref(quad), flags, p, ref.numel(), *[_ptr(t) for t in cols],
                _ptr(outs["v_n"]), _ptr(outs["v_m"]), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return P3Velocities(outs["v_n"], outs["v_m"])


P3ShapeVelocities = namedtuple("P3ShapeVelocities", ["log_lambda", "D_m", "v_n", "v_m"])


def p3_shape_and_terminal_velocities(params: ParametersP3, velocity_params, rho_air, rho_q_ice, rho_n_ice, x3, x4, *, log_lambda_guess=None,
                                     from_state=False, aspect_ratio=True, brent_iters=0, p=1e-6, quad=None, stream=None) -> P3ShapeVelocities:
    """`p3_shape` (log λ, D_m) followed by `p3_terminal_velocities` as ONE launch (`cmx_p3_shape_terminal_velocities_*`) — BASELINE
    config 5: `P3.get_distribution_logλ(state)`, `P3.D_m`, `P3.ice_terminal_velocity_number_weighted / _mass_weighted` on the same
    columns (src/P3_size_distribution.jl:284-320, :56-61, src/P3_terminal_velocity.jl:72-137).  Bit-identical to the two calls."""
    if not isinstance(params, ParametersP3):
        raise TypeError("params must be ParametersP3")
    cols = (rho_q_ice, rho_n_ice, x3, x4, rho_air)
    ref = _check_cols(cols, ("rho_q_ice", "rho_n_ice", "x3", "x4", "rho_air"))
    fam = _fam_of(ref)
    if fam is not params.fam or not isinstance(velocity_params, fam.chen2022_ice_vel):
        raise TypeError("parameter float type does not match the state columns")
    guess = log_lambda_guess
    if guess is not None:
        _check_cols([ref, guess], ["rho_q_ice", "log_lambda_guess"])
    if quad is None:
        from .parameters import ChebyshevGauss
        quad = ChebyshevGauss(fam.sfx, 100)
    if not isinstance(quad, fam.quadrature):
        raise TypeError("quadrature float type does not match the state columns")
    out = P3ShapeVelocities(*[torch.empty_like(ref) for _ in range(4)])
    flags = params.flags | (_abi.CMX_P3_INPUT_IS_STATE if from_state else 0) | (0 if aspect_ratio else _abi.CMX_P3_NO_ASPECT_RATIO)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_p3_shape_terminal_velocities_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(params.c), C.byref(velocity_params), C.byref(quad), flags, int(brent_iters), p, ref.numel(), *[_ptr(t) for t in cols],
                _ptr(guess), *[_ptr(t) for t in out], C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


P3Melt = namedtuple("P3Melt", ["dNdt", "dLdt"])


def p3_ice_melt(params: ParametersP3, velocity_params, aps, tps, vent, T, rho_air, rho_q_ice, rho_n_ice, x3, x4, log_lambda, *,
                from_state=False, aspect_ratio=True, p=1e-6, quad=None, stream=None) -> P3Melt:
    """`P3.ice_melt(vel, aps, tps, T, ρₐ, state, logλ; quad)` for every point (src/P3_processes.jl:64-94): melting rates
    (dNdt [1/m³/s], dLdt [kg/m³/s]); zero at and below T_freeze.  `vent` = parameters.VentilationFactorP3(FT)."""
    if not isinstance(params, ParametersP3):
        raise TypeError("params must be ParametersP3")
    cols = (rho_q_ice, rho_n_ice, x3, x4, rho_air, T, log_lambda)
    ref = _check_cols(cols, ("rho_q_ice", "rho_n_ice", "x3", "x4", "rho_air", "T", "log_lambda"))
    fam = _fam_of(ref)
    if fam is not params.fam or not (isinstance(velocity_params, fam.chen2022_ice_vel) and isinstance(aps, fam.air_properties)
                                     and isinstance(tps, fam.thermo) and isinstance(vent, fam.ventilation)):
        raise TypeError("parameter float type does not match the state columns")
    if quad is None:
        from .parameters import ChebyshevGauss
        quad = ChebyshevGauss(fam.sfx, 100)
    if not isinstance(quad, fam.quadrature):
        raise TypeError("quadrature float type does not match the state columns")
    out = P3Melt(torch.empty_like(ref), torch.empty_like(ref))
    flags = params.flags | (_abi.CMX_P3_INPUT_IS_STATE if from_state else 0) | (0 if aspect_ratio else _abi.CMX_P3_NO_ASPECT_RATIO)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_p3_ice_melt_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(params.c), C.byref(velocity_params), C.byref(aps), C.byref(tps), C.byref(vent), C.byref(quad), flags, p,
                ref.numel(), *[_ptr(t) for t in cols], _ptr(out.dNdt), _ptr(out.dLdt), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


def p3_ice_self_collection(params: ParametersP3, velocity_params, rho_air, rho_q_ice, rho_n_ice, x3, x4, log_lambda, *, from_state=False,
                           aspect_ratio=True, quad=None, stream=None):
    """`P3.ice_self_collection(state, logλ, vel, ρₐ; quad).dNdt` for every point (src/P3_processes.jl:676-712): the ice number
    loss rate by aggregation [1/m³/s].  8·n² integrand evaluations per point for an n-point rule (default ChebyshevGauss(100),
    as in the reference; GaussLegendre(40) is 6× cheaper for the same accuracy)."""
    if not isinstance(params, ParametersP3):
        raise TypeError("params must be ParametersP3")
    cols = (rho_q_ice, rho_n_ice, x3, x4, rho_air, log_lambda)
    ref = _check_cols(cols, ("rho_q_ice", "rho_n_ice", "x3", "x4", "rho_air", "log_lambda"))
    fam = _fam_of(ref)
    if fam is not params.fam or not isinstance(velocity_params, fam.chen2022_ice_vel):
        raise TypeError("parameter float type does not match the state columns")
    if quad is None:
        from .parameters import ChebyshevGauss
        quad = ChebyshevGauss(fam.sfx, 100)
    if not isinstance(quad, fam.quadrature):
        raise TypeError("quadrature float type does not match the state columns")
    out = torch.empty_like(ref)
    flags = params.flags | (_abi.CMX_P3_INPUT_IS_STATE if from_state else 0) | (0 if aspect_ratio else _abi.CMX_P3_NO_ASPECT_RATIO)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_p3_ice_self_collection_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(params.c), C.byref(velocity_params), C.byref(quad), flags, ref.numel(), *[_ptr(t) for t in cols], _ptr(out),
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


COLLISION_SOURCES = ("dq_c", "dq_r", "dN_c", "dN_r", "dL_rim", "dL_ice", "dB_rim")
COLLISION_RATES = ("QCFRZ", "QCSHD", "NCCOL", "QRFRZ", "QRSHD", "NRCOL", "int_M_col", "BCCOL", "BRCOL", "int_wet_M_col")


def p3_liquid_ice_collisions(ice_params, aps, tps, rho_air, T, rho_q_ice, rho_n_ice, x3, x4, log_lambda, L_c, N_c, L_r, N_r, *,
                             from_state=False, aspect_ratio=True, quad=None, want_rates=False, stream=None):
    """`P3.bulk_liquid_ice_collision_sources(state, logλ, psd_c, psd_r, L_c, N_c, L_r, N_r, aps, tps, vel, ρₐ, T; quad)` for every
    point (src/P3_processes.jl:600-655): dict of the seven bulk tendencies (∂ₜq_c, ∂ₜq_r [kg/kg/s], ∂ₜN_c, ∂ₜN_r [1/m³/s], ∂ₜL_rim,
    ∂ₜL_ice [kg/m³/s], ∂ₜB_rim [1/s]); with `want_rates` also the ten ∫liquid_ice_collisions integrals (:527-562).  `ice_params` is a
    `P3IceParams` (scheme, fall-speed tables, cloud and rain PSDs); `quad` defaults to its quadrature rule."""
    from .parameters import P3IceParams
    if not isinstance(ice_params, P3IceParams):
        raise TypeError("ice_params must be P3IceParams")
    cols = (rho_q_ice, rho_n_ice, x3, x4, L_c, N_c, L_r, N_r, rho_air, T, log_lambda)
    ref = _check_cols(cols, ("rho_q_ice", "rho_n_ice", "x3", "x4", "L_c", "N_c", "L_r", "N_r", "rho_air", "T", "log_lambda"))
    fam = _fam_of(ref)
    if fam is not ice_params.fam or not isinstance(aps, fam.air_properties) or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    quad = ice_params.c.quad if quad is None else quad
    if not isinstance(quad, fam.quadrature):
        raise TypeError("quadrature float type does not match the state columns")
    src = {k: torch.empty_like(ref) for k in COLLISION_SOURCES}
    rates = {k: torch.empty_like(ref) for k in COLLISION_RATES} if want_rates else None
    src_p = (C.c_void_p * 7)(*[t.data_ptr() for t in src.values()])
    rates_p = (C.c_void_p * 10)(*[t.data_ptr() for t in rates.values()]) if want_rates else None
    flags = ice_params.flags | (_abi.CMX_P3_INPUT_IS_STATE if from_state else 0) | (0 if aspect_ratio else _abi.CMX_P3_NO_ASPECT_RATIO)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_p3_liquid_ice_collisions_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(ice_params.c), C.byref(aps), C.byref(tps), C.byref(quad), flags, ref.numel(), *[_ptr(t) for t in cols], src_p, rates_p,
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return (src, rates) if want_rates else src


P3HetNucleation = namedtuple("P3HetNucleation", ["dNdt", "dLdt"])


def p3_het_ice_nucleation(aerosol, tps, q_lcl, N_lcl, RH, T, rho_air, *, stream=None):
    """`P3.het_ice_nucleation(aerosol, tps, q_lcl, N_lcl, RH, T, ρₐ)` for every point (src/P3_processes.jl:20-46): ABIFM immersion
    freezing rates (dNdt [1/m³/s], dLdt [kg/m³/s]) for a dust type with ABIFM coefficients (`parameters.Illite / Kaolinite / …`)."""
    cols = (q_lcl, N_lcl, RH, T, rho_air)
    ref = _check_cols(cols, ("q_lcl", "N_lcl", "RH", "T", "rho_air"))
    fam = _fam_of(ref)
    if not isinstance(aerosol, fam.abifm_dust) or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    out = P3HetNucleation(torch.empty_like(ref), torch.empty_like(ref))
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_p3_het_ice_nucleation_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(aerosol), C.byref(tps), ref.numel(), *[_ptr(t) for t in cols], _ptr(out.dNdt), _ptr(out.dLdt), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out
