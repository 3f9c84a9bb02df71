"""One-moment (Marshall–Palmer) scheme over columns — host-side mirror of the 1M entry of
`CloudMicrophysics.BulkMicrophysicsTendencies` and of `CloudMicrophysics.Microphysics1M.terminal_velocity`
(include/cmx.h §5).

Reference broadcasts being replaced:

    BMT.bulk_microphysics_tendencies.(Ref(BMT.Instantaneous()), Ref(BMT.Microphysics1Moment()), Ref(mp), Ref(tps),
                                      ρ, T, q_tot, q_lcl, q_icl, q_rai, q_sno)        # BMT:505-514
    @. w = CM1.terminal_velocity(rain, vel, ρ, q)                                      # test/gpu_clima_core_test.jl:39-42
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

import torch

from . import _abi, _lib
from .bulk_tendencies import _check_cols, _fam_of, _fields_call, _ptr
from .parameters import Chen2022VelTypeRain, Microphysics1MParams


class Microphysics1Moment:
    """BMT.Microphysics1Moment — scheme tag (src/BulkMicrophysicsTendencies.jl:52-56)."""


class Instantaneous:
    """BMT.Instantaneous — tendency mode tag: raw point-wise tendencies (BMT:84-90)."""


class LinearizedAverage:
    """BMT.LinearizedAverage — tendency mode tag: average tendencies over Δt from `nsub` linearized implicit substeps
    (BMT:96-115, 572-632).  `bulk_microphysics_tendencies_1m(LinearizedAverage(), scheme, mp, tps, …, dt, nsub)`.
    `q_min` = TD.Parameters.q_min(tps), the donor floor of the linearization (parameters.DEFAULT_PARAMETERS
    "specific_humidity_minimum"; not pinned by any reference test)."""


Tendencies1M = namedtuple("Tendencies1M", ["dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"])
SourceTerms1M = namedtuple("SourceTerms1M", _abi.MP1M_SOURCE_COLUMNS)
TerminalVelocities1M = namedtuple("TerminalVelocities1M", ["vt_rai_blk1m", "vt_sno_blk1m", "vt_rai_chen"])

_NAMES = ("rho", "T", "q_tot", "q_lcl", "q_icl", "q_rai", "q_sno")


def _prep(mp, tps, cols):
    if not isinstance(mp, Microphysics1MParams):
        raise TypeError("mp must be Microphysics1MParams")
    ref = _check_cols(cols, _NAMES)
    fam = _fam_of(ref)
    if fam is not mp.fam or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    return ref, fam


def bulk_microphysics_tendencies_1m(mode, scheme, mp, tps, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dt=None, nsub=1, *,
                                    q_min=None, out=None, stream=None) -> Tendencies1M:
    """1-moment tendencies over columns — Instantaneous mode (BMT:505-514 → :141-252) or, with `dt` [, `nsub`],
    LinearizedAverage mode (BMT:572-632)."""
    if not isinstance(scheme, Microphysics1Moment) or not isinstance(mode, (Instantaneous, LinearizedAverage)):
        raise TypeError("mode must be Instantaneous() or LinearizedAverage() and scheme Microphysics1Moment()")
    cols = (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)
    ref, fam = _prep(mp, tps, cols)
    if out is None:
        out = Tendencies1M(*[torch.empty_like(ref) for _ in range(4)])
    else:
        _check_cols([ref] + list(out), ["rho"] + ["out"] * 4)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    if isinstance(mode, LinearizedAverage):
        if dt is None or not dt > 0 or int(nsub) < 1:
            raise ValueError("LinearizedAverage needs dt > 0 and nsub >= 1")
        if q_min is None:
            from .parameters import DEFAULT_PARAMETERS
            q_min = DEFAULT_PARAMETERS["specific_humidity_minimum"]
        fn = getattr(_lib.lib(), f"cmx_mp1m_linearized_average_{fam.sfx}")
        with torch.cuda.device(ref.device):
            st = fn(C.byref(mp.c), C.byref(tps), mp.flags, q_min, dt, int(nsub), ref.numel(), *[_ptr(t) for t in cols],
                    *[_ptr(o) for o in out], C.c_void_p(s.cuda_stream))
    else:
        if dt is not None:
            raise TypeError("Instantaneous() takes no dt")
        fn = getattr(_lib.lib(), f"cmx_mp1m_tendencies_{fam.sfx}")
        with torch.cuda.device(ref.device):
            st = fn(C.byref(mp.c), C.byref(tps), mp.flags, ref.numel(), *[_ptr(t) for t in cols], *[_ptr(o) for o in out],
                    C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


def bulk_microphysics_tendencies_1m_fields(mode, scheme, mp, tps, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dt=None, nsub=1, *, q_min=None, out=None,
                                           aos=False, stream=None):
    """The 1-moment tendencies — Instantaneous, or LinearizedAverage with `dt` [, `nsub`] — on the host model's own storage (SURVEY §8f-3,
    `cmx_mp1m_tendencies_fields_*` / `cmx_mp1m_linearized_average_fields_*`): every column
    a contiguous 1-D tensor or a (n_seg, seg_len) view with contiguous rows (a ClimaCore `VIJFH` field component in place); the result
    goes into `out` (4 tensors of the same shape → `Tendencies1M`) or, with `aos=True`, into the reference's own result layout — an
    (n, 4) tensor of NamedTuple rows (dq_lcl_dt, dq_icl_dt, dq_rai_dt, dq_sno_dt).  Bit-identical to `bulk_microphysics_tendencies_1m`."""
    if not isinstance(scheme, Microphysics1Moment) or not isinstance(mode, (Instantaneous, LinearizedAverage)):
        raise TypeError("mode must be Instantaneous() or LinearizedAverage() and scheme Microphysics1Moment()")
    if not isinstance(mp, Microphysics1MParams):
        raise TypeError("mp must be Microphysics1MParams")
    fam = _fam_of(rho)
    if fam is not mp.fam or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    name, extra = "cmx_mp1m_tendencies_fields", ()
    if isinstance(mode, LinearizedAverage):     # cmx_mp1m_linearized_average_fields_*: (q_min, Δt, nsub) after the flags
        if dt is None or not dt > 0 or int(nsub) < 1:
            raise ValueError("LinearizedAverage needs dt > 0 and nsub >= 1")
        if q_min is None:
            from .parameters import DEFAULT_PARAMETERS
            q_min = DEFAULT_PARAMETERS["specific_humidity_minimum"]
        name, extra = "cmx_mp1m_linearized_average_fields", (fam.ft(q_min), fam.ft(dt), C.c_int32(int(nsub)))
    elif dt is not None:
        raise TypeError("Instantaneous() takes no dt")
    r = _fields_call(name, mp.c, tps, mp.flags, (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno),
                     ("rho", "T", "q_tot", "q_lcl", "q_icl", "q_rai", "q_sno"), 4, 4, out, aos, stream, extra)
    return r if aos else Tendencies1M(*r)


def microphysics_source_terms_1m(mp, tps, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, *, stream=None) -> SourceTerms1M:
    """The 18 individual source terms of `_microphysics_source_terms` (BMT:141-217) over columns."""
    cols = (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)
    ref, fam = _prep(mp, tps, cols)
    outs = [torch.empty_like(ref) for _ in range(_abi.CMX_MP1M_NSRC)]
    arr = (C.c_void_p * _abi.CMX_MP1M_NSRC)(*[o.data_ptr() for o in outs])
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_mp1m_source_terms_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(mp.c), C.byref(tps), mp.flags, ref.numel(), *[_ptr(t) for t in cols], arr, C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return SourceTerms1M(*outs)


def terminal_velocity_1m(mp, rho, q_rai=None, q_sno=None, *, chen=False, stream=None) -> TerminalVelocities1M:
    """Mass-weighted 1M fall speeds — CM1.terminal_velocity (CM1:223-270): Blk1M rain (needs q_rai), Blk1M snow
    (needs q_sno) and, with `chen=True`, Chen-2022 rain."""
    if not isinstance(mp, Microphysics1MParams):
        raise TypeError("mp must be Microphysics1MParams")
    cols = [c for c in (rho, q_rai, q_sno) if c is not None]
    ref = _check_cols(cols, ("rho", "q", "q"))
    fam = _fam_of(ref)
    if fam is not mp.fam:
        raise TypeError("parameter float type does not match the state columns")
    if chen and q_rai is None:
        raise ValueError("the Chen-2022 rain velocity needs q_rai")
    mk = lambda on: torch.empty_like(ref) if on else None  # noqa: E731
    out = TerminalVelocities1M(mk(q_rai is not None), mk(q_sno is not None), mk(chen))
    ch = Chen2022VelTypeRain(fam.sfx) if chen else None
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_mp1m_terminal_velocity_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(mp.c), C.byref(ch) if ch is not None else None, ref.numel(), _ptr(rho), _ptr(q_rai), _ptr(q_sno),
                *[_ptr(o) for o in out], C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


SedimentationVelocities = namedtuple("SedimentationVelocities", ["w_lcl", "w_icl", "w_rai", "w_sno"])


def sedimentation_velocities(mp, stokes, chen_rain, chen_ice, rho, q_lcl=None, q_icl=None, q_rai=None, q_sno=None, *,
                             stream=None) -> SedimentationVelocities:
    """The bulk fall speeds a host model precomputes for sedimentation (ClimaAtmos `set_sedimentation_precomputed_quantities`,
    test/gpu_clima_core_test.jl:36-45): cloud liquid (Stokes), cloud ice (Chen-2022 small ice), rain (Chen-2022 rain) and
    snow (Chen-2022 large ice) — `CMNonEq.terminal_velocity` (NonEq:250-281) and `CM1.terminal_velocity` (CM1:251-297).
    Species whose q column is None are skipped (None in the result)."""
    if not isinstance(mp, Microphysics1MParams):
        raise TypeError("mp must be Microphysics1MParams")
    qs = (q_lcl, q_icl, q_rai, q_sno)
    cols = [rho] + [q for q in qs if q is not None]
    ref = _check_cols(cols, ["rho"] + ["q"] * (len(cols) - 1))
    fam = _fam_of(ref)
    if fam is not mp.fam:
        raise TypeError("parameter float type does not match the state columns")
    outs = [torch.empty_like(ref) if q is not None else None for q in qs]
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    ref_or_null = lambda x: C.byref(x) if x is not None else None  # noqa: E731
    fn = getattr(_lib.lib(), f"cmx_sedimentation_velocities_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(mp.c), ref_or_null(stokes), ref_or_null(chen_rain), ref_or_null(chen_ice), ref.numel(), _ptr(rho),
                *[_ptr(q) for q in qs], *[_ptr(o) for o in outs], C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return SedimentationVelocities(*outs)


ColumnStep1M = namedtuple("ColumnStep1M", ["dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt", "precip_rai", "precip_sno"])


def column_tendencies_sedimentation_1m(mode, scheme, mp, tps, stokes, chen_rain, chen_ice, inv_dz, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno,
                                       dt=None, nsub=1, *, q_min=None, precip=True, stream=None) -> ColumnStep1M:
    """The operational 1-moment column step in one pass (`cmx_mp1m_column_tendencies_sedimentation_*`): the 1-moment tendencies
    (Instantaneous, or LinearizedAverage with `dt` [, `nsub`]) + the four sedimentation velocities (`sedimentation_velocities`) + the host
    model's first-order upwind flux divergence of q_lcl, q_icl, q_rai, q_sno.  State tensors of shape (n_col, n_lev), level 0 lowest,
    contiguous; `inv_dz` = n_lev values 1/Δz.  Returns the four total tendencies and (precip=True) the surface rain / snow fluxes."""
    if not isinstance(scheme, Microphysics1Moment) or not isinstance(mode, (Instantaneous, LinearizedAverage)):
        raise TypeError("mode must be Instantaneous() or LinearizedAverage() and scheme Microphysics1Moment()")
    cols = (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)
    names = ("rho", "T", "q_tot", "q_lcl", "q_icl", "q_rai", "q_sno")
    if rho.dim() != 2:
        raise ValueError("state tensors must have shape (n_col, n_lev)")
    # validate the 2-D tensors THEMSELVES: the kernel reads and writes flat row-major (n_col, n_lev) storage, so a transposed
    # view, a column of another shape with the same number of elements or any non-contiguous tensor must be refused, not silently copied
    for c, nm in zip(cols, names):
        if c.shape != rho.shape:
            raise ValueError(f"column {nm}: shape {tuple(c.shape)} differs from rho's {tuple(rho.shape)}")
        if not c.is_contiguous():
            raise ValueError(f"column {nm} must be a contiguous (n_col, n_lev) tensor (levels contiguous within a column)")
    flat = [c.view(-1) for c in cols]
    ref, fam = _prep(mp, tps, flat)
    for obj, cls, nm in ((stokes, fam.stokes_vel, "stokes"), (chen_rain, fam.chen2022_rain_vel, "chen_rain"), (chen_ice, fam.chen2022_ice_vel, "chen_ice")):
        if not isinstance(obj, cls):
            raise TypeError(f"{nm}: expected the {cls.__name__} struct of the state's float type, got {type(obj).__name__}")
    n_col, n_lev = rho.shape
    if inv_dz.numel() != n_lev or inv_dz.dtype != ref.dtype or inv_dz.device != ref.device or not inv_dz.is_contiguous():
        raise ValueError("inv_dz: n_lev contiguous values of the state's dtype on the state's device")
    lin = isinstance(mode, LinearizedAverage)
    if lin:
        if dt is None or not dt > 0 or int(nsub) < 1:
            raise ValueError("LinearizedAverage needs dt > 0 and nsub >= 1")
        if q_min is None:
            from .parameters import DEFAULT_PARAMETERS
            q_min = DEFAULT_PARAMETERS["specific_humidity_minimum"]
    elif dt is not None:
        raise TypeError("Instantaneous() takes no dt")
    out = [torch.empty(rho.shape, dtype=ref.dtype, device=ref.device) for _ in range(4)]      # contiguous row-major, whatever rho's strides
    pr = [torch.empty(n_col, dtype=ref.dtype, device=ref.device) if precip else None for _ in range(2)]
    in_p = (C.c_void_p * 7)(*[c.data_ptr() for c in flat])
    out_p = (C.c_void_p * 4)(*[o.data_ptr() for o in out])
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_mp1m_column_tendencies_sedimentation_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(mp.c), C.byref(tps), C.byref(stokes), C.byref(chen_rain), C.byref(chen_ice), mp.flags, q_min if lin else 0.0,
                dt if lin else 0.0, int(nsub) if lin else 0, n_col, n_lev, _ptr(inv_dz), in_p, out_p, _ptr(pr[0]), _ptr(pr[1]),
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return ColumnStep1M(*out, *pr)
