"""Cloud diagnostics over columns — host-side mirror of `CloudMicrophysics.CloudDiagnostics` (include/cmx.h §10).

Reference broadcasts being replaced (src/CloudDiagnostics.jl):

    CMD.radar_reflectivity_1M.(Ref(rain), q_rai, ρ)                                         # :31-46
    CMD.radar_reflectivity_2M.(Ref(SB2006), q_lcl, q_rai, N_lcl, N_rai, ρ)                  # :64-84
    CMD.effective_radius_2M.(Ref(SB2006), q_lcl, q_rai, N_lcl, N_rai, ρ)                    # :100-125
    CMD.effective_radius_Liu_Hallet_97.(Ref(wtr), ρ, q_lcl[, N_lcl, q_rai, N_rai])          # :143-180
    CMD.effective_radius_const(cloud_params)                                                # :188-193

N per m³, as in the reference.  The rain PSD variant of the two-moment functions is the one the `SB2006` struct was built with."""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

import torch

from . import _abi, _lib
from .bulk_tendencies import _check_cols, _fam_of, _ptr


def _call(ref, *, rain=None, sb=None, rho_w=0.0, rho, q_lcl=None, q_rai=None, N_lcl=None, N_rai=None, want, stream=None, out=None):
    fam = _fam_of(ref)
    outs = {k: (torch.empty_like(ref) if k in want else None) for k in ("Z_1m", "Z_2m", "reff_2m", "reff_lh97")}
    if out is not None:      # caller-provided output columns (no allocation in a time loop)
        missing = [k for k in want if getattr(out, k, None) is None]
        if missing:
            raise ValueError(f"`out` holds no column for the requested output(s) {missing}: pass want=(…) for the columns it does hold")
        for k in want:
            _check_cols([ref, getattr(out, k)], ["rho", k])
            outs[k] = getattr(out, k)
    flags = 0
    pdf_c = pdf_r = None
    if sb is not None:
        if not isinstance(sb, fam.sb2006):
            raise TypeError("SB2006 parameter float type does not match the state columns")
        pdf_c, pdf_r = sb.pdf_c, sb.pdf_r
        limited = getattr(sb, "is_limited", None)
        if limited is None:     # a bare C struct: the not-limited constructor leaves the N0 / lambda limiters at zero
            # (all three limiter pairs, like the entry's own check sb_limiters_ok: a struct with N0 / lambda limiters but a zero x_r pair is not a limited PSD)
            limited = ((0 < pdf_r.N0_min <= pdf_r.N0_max) and (0 < pdf_r.lambda_min <= pdf_r.lambda_max) and (0 < pdf_r.xr_min <= pdf_r.xr_max))
        flags = _abi.CMX_SB2006_LIMITED if limited else 0
    if rain is not None and not isinstance(rain, fam.rain):
        raise TypeError("rain parameter float type does not match the state columns")
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_cloud_diagnostics_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(rain) if rain is not None else None, C.byref(pdf_c) if pdf_c is not None else None, C.byref(pdf_r) if pdf_r is not None else None,
                rho_w, flags, ref.numel(), _ptr(rho), _ptr(q_lcl), _ptr(q_rai), _ptr(N_lcl), _ptr(N_rai), _ptr(outs["Z_1m"]), _ptr(outs["Z_2m"]),
                _ptr(outs["reff_2m"]), _ptr(outs["reff_lh97"]), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return outs


Diagnostics = namedtuple("Diagnostics", ["Z_1m", "Z_2m", "reff_2m", "reff_lh97"])


def cloud_diagnostics(rain, sb, wtr, rho, q_lcl, q_rai, N_lcl, N_rai, *, want=("Z_1m", "Z_2m", "reff_2m", "reff_lh97"), out=None, stream=None) -> Diagnostics:
    """All (or some) of the four diagnostics in ONE pass over the five state columns — what a host model's diagnostics step needs per cell.
    `rain` (Z_1m), `sb` (Z_2m, reff_2m) and `wtr` (reff_lh97) may be None when their outputs are not wanted; `out` = a Diagnostics of output columns."""
    ref = _check_cols([rho, q_lcl, q_rai, N_lcl, N_rai], ["rho", "q_lcl", "q_rai", "N_lcl", "N_rai"])
    rho_w = float(getattr(wtr, "rho_w", wtr)) if wtr is not None else 0.0
    o = _call(ref, rain=rain, sb=sb, rho_w=rho_w, rho=rho, q_lcl=q_lcl, q_rai=q_rai, N_lcl=N_lcl, N_rai=N_rai, want=tuple(want), stream=stream, out=out)
    return Diagnostics(o["Z_1m"], o["Z_2m"], o["reff_2m"], o["reff_lh97"])


def radar_reflectivity_1M(rain, q_rai: torch.Tensor, rho: torch.Tensor, *, stream=None) -> torch.Tensor:
    """CMD.radar_reflectivity_1M over columns [dBZ]; `rain` = the `rain` member of Microphysics1MParams (CMP.Rain)."""
    ref = _check_cols([q_rai, rho], ["q_rai", "rho"])
    return _call(ref, rain=rain, rho=rho, q_rai=q_rai, want=("Z_1m",), stream=stream)["Z_1m"]


def radar_reflectivity_2M(sb, q_lcl, q_rai, N_lcl, N_rai, rho, *, stream=None) -> torch.Tensor:
    """CMD.radar_reflectivity_2M over columns [dBZ]; `sb` = parameters.SB2006(FT, is_limited)."""
    ref = _check_cols([q_lcl, q_rai, N_lcl, N_rai, rho], ["q_lcl", "q_rai", "N_lcl", "N_rai", "rho"])
    return _call(ref, sb=sb, rho=rho, q_lcl=q_lcl, q_rai=q_rai, N_lcl=N_lcl, N_rai=N_rai, want=("Z_2m",), stream=stream)["Z_2m"]


def effective_radius_2M(sb, q_lcl, q_rai, N_lcl, N_rai, rho, *, stream=None) -> torch.Tensor:
    """CMD.effective_radius_2M over columns [m]."""
    ref = _check_cols([q_lcl, q_rai, N_lcl, N_rai, rho], ["q_lcl", "q_rai", "N_lcl", "N_rai", "rho"])
    return _call(ref, sb=sb, rho=rho, q_lcl=q_lcl, q_rai=q_rai, N_lcl=N_lcl, N_rai=N_rai, want=("reff_2m",), stream=stream)["reff_2m"]


def radar_reflectivity_and_effective_radius_2M(sb, q_lcl, q_rai, N_lcl, N_rai, rho, *, stream=None):
    """Both two-moment diagnostics in ONE pass over the five columns (they share the PSD parameters): (Z_2m [dBZ], reff_2m [m])."""
    ref = _check_cols([q_lcl, q_rai, N_lcl, N_rai, rho], ["q_lcl", "q_rai", "N_lcl", "N_rai", "rho"])
    o = _call(ref, sb=sb, rho=rho, q_lcl=q_lcl, q_rai=q_rai, N_lcl=N_lcl, N_rai=N_rai, want=("Z_2m", "reff_2m"), stream=stream)
    return o["Z_2m"], o["reff_2m"]


def effective_radius_Liu_Hallet_97(wtr, rho, q_lcl, N_lcl=None, q_rai=None, N_rai=None, *, stream=None) -> torch.Tensor:
    """CMD.effective_radius_Liu_Hallet_97 over columns [m]; `wtr` = anything with a `rho_w` field (WaterProperties / CloudLiquid) or the density
    itself.  The three-argument method (N_lcl = 100 m⁻³, no rain) when N_lcl, q_rai and N_rai are all omitted."""
    rho_w = float(getattr(wtr, "rho_w", wtr))
    given = [c for c in (N_lcl, q_rai, N_rai) if c is not None]
    if len(given) not in (0, 3):
        raise ValueError("pass all of N_lcl, q_rai, N_rai or none of them")
    ref = _check_cols([rho, q_lcl] + given, ["rho", "q_lcl", "N_lcl", "q_rai", "N_rai"][:2 + len(given)])
    return _call(ref, rho_w=rho_w, rho=rho, q_lcl=q_lcl, q_rai=q_rai, N_lcl=N_lcl, N_rai=N_rai, want=("reff_lh97",), stream=stream)["reff_lh97"]


def effective_radius_const(cloud_params) -> float:
    """CMD.effective_radius_const: the r_eff field of CloudLiquid / CloudIce."""
    return float(cloud_params.r_eff)
