"""Zero-moment scheme over columns — host-side mirror of `CloudMicrophysics.Microphysics0M` and of the 0M methods of
`BulkMicrophysicsTendencies.bulk_microphysics_tendencies` (include/cmx.h §0).

Reference broadcasts being replaced:

    BMT.bulk_microphysics_tendencies.(Ref(BMT.Microphysics0Moment()), Ref(mp), Ref(tps), T, q_lcl, q_icl[, q_vap_sat])   # BMT:658-680
    CM0.remove_precipitation.(Ref(p0m), q_lcl, q_icl[, q_vap_sat])                                                       # CM0:35-46
    CM0.∂remove_precipitation_∂q_tot.(Ref(p0m), q_lcl, q_icl[, q_vap_sat])                                               # CM0:64-75
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .bulk_tendencies import _check_cols, _fam_of, _ptr
from .parameters import Microphysics0MParams


class Microphysics0Moment:
    """BMT.Microphysics0Moment — scheme tag (src/BulkMicrophysicsTendencies.jl:45-49)."""


def _call(p0m, q_lcl, q_icl, q_vap_sat, want, want_derivative, out, stream):
    cols = [q_lcl, q_icl] + ([q_vap_sat] if q_vap_sat is not None else [])
    ref = _check_cols(cols, ["q_lcl", "q_icl", "q_vap_sat"])
    fam = _fam_of(ref)
    if not isinstance(p0m, fam.parameters_0m):
        raise TypeError("parameter float type does not match the state columns")
    if out is None:
        out = torch.empty_like(ref)
    else:
        _check_cols([ref, out], ["q_lcl", "out"])
    der = torch.empty_like(ref) if want_derivative else None
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_mp0m_tendencies_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(p0m), ref.numel(), _ptr(q_lcl), _ptr(q_icl), _ptr(q_vap_sat), _ptr(out), _ptr(der), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out, der


def bulk_microphysics_tendencies_0m(scheme, mp, tps, T, q_lcl, q_icl, q_vap_sat=None, *, out=None, stream=None) -> torch.Tensor:
    """dq_tot_dt [kg/kg/s] from precipitation removal — BMT:658-680.  `tps` and `T` are accepted and unused, as in the reference."""
    if not isinstance(scheme, Microphysics0Moment) or not isinstance(mp, Microphysics0MParams):
        raise TypeError("scheme must be Microphysics0Moment() and mp Microphysics0MParams")
    return _call(mp.precip, q_lcl, q_icl, q_vap_sat, True, False, out, stream)[0]


def remove_precipitation(p0m, q_lcl, q_icl, q_vap_sat=None, *, stream=None) -> torch.Tensor:
    """CM0.remove_precipitation over columns (src/Microphysics0M.jl:35-46); inputs are taken as given apart from the clamp to ≥ 0."""
    return _call(p0m, q_lcl, q_icl, q_vap_sat, True, False, None, stream)[0]


def d_remove_precipitation_d_q_tot(p0m, q_lcl, q_icl, q_vap_sat=None, *, stream=None) -> torch.Tensor:
    """CM0.∂remove_precipitation_∂q_tot over columns (src/Microphysics0M.jl:64-75)."""
    return _call(p0m, q_lcl, q_icl, q_vap_sat, True, True, None, stream)[1]
