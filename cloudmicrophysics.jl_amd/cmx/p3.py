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
             want=("log_lambda", "D_m"), brent_iters=0, stream=None) -> P3Shape:
    """Solve the P3 size distribution for every point.

    Inputs: ρq_ice [kg/m³], ρn_ice [1/m³] and either the prognostic rime variables (ρq_rim [kg/m³], ρb_rim [m³/m³];
    `state_from_prognostic`, src/P3_particle_properties.jl:101-106) or, with `from_state=True`, (F_rim, ρ_rim) as in
    `P3State(params, L, N, F_rim, ρ_rim)`.  Outputs (`want`): F_rim, rho_rim (the regularised state), log_lambda
    (`get_distribution_logλ`, src/P3_size_distribution.jl:284-320; −inf for absent ice), D_m (mass-weighted mean
    diameter, src/P3_integral_properties.jl:56-61) and log_N0 (:233-237).
    `brent_iters` = 0 keeps the reference's fixed Brent budget (8 Float32 / 10 Float64 iterations, :311)."""
    if not isinstance(params, ParametersP3):
        raise TypeError("params must be ParametersP3")
    cols = (rho_q_ice, rho_n_ice, x3, x4)
    ref = _check_cols(cols, ("rho_q_ice", "rho_n_ice", "x3", "x4"))
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
        st = fn(C.byref(params.c), flags, int(brent_iters), ref.numel(), *[_ptr(t) for t in cols], *[_ptr(outs[k]) for k in P3Shape._fields],
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return P3Shape(*[outs[k] for k in P3Shape._fields])
