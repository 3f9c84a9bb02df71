"""Array entry points — host-side mirror of `CloudMicrophysics.BulkMicrophysicsTendencies` (BMT)
and of the per-process `CloudMicrophysics.Microphysics2M` (CM2) functions, evaluated over device
columns by the fused gfx950 kernels behind the C ABI (include/cmx.h).

Reference call being replaced (test/type_stability_tests.jl:131-137):

    BMT.bulk_microphysics_tendencies.(Ref(BMT.Microphysics2Moment()), Ref(mp), Ref(tps),
                                      ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)

Here:

    bulk_microphysics_tendencies(Microphysics2Moment(), mp, tps, ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)

with ρ… torch tensors resident in HBM; the result is a NamedTuple of columns (SoA) instead of the
reference's array of NamedTuples, with the same field names.
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple
from typing import Optional

import torch

from . import _abi, _lib
from .parameters import Microphysics2MParams, WarmRainParams2M, rain_vel_params


class Microphysics2Moment:
    """BMT.Microphysics2Moment — scheme tag (src/BulkMicrophysicsTendencies.jl:59-63)."""


class SB2006VelType:
    """velocity-scheme tag: CM2.rain_terminal_velocity(sb, ::SB2006VelType, …)  CM2:685-702"""
    flag = _abi.CMX_VEL_SB2006


class Chen2022VelTypeRain:
    """velocity-scheme tag: CM2.rain_terminal_velocity(sb, ::Chen2022VelTypeRain, …)  CM2:703-719"""
    flag = _abi.CMX_VEL_CHEN2022


WarmRainTendencies2M = namedtuple(
    "WarmRainTendencies2M",
    ["dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt", "vt_rai_n", "vt_rai_m"])

SB2006ProcessRates = namedtuple("SB2006ProcessRates", _abi.SB2006_PROCESS_COLUMNS)

# return of the 2M + P3 method (BMT:1079-1082)
Tendencies2MP3 = namedtuple(
    "Tendencies2MP3",
    ["dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt", "dq_ice_dt", "dn_ice_dt", "dq_rim_dt", "db_rim_dt", "dn_lcl_activation_dt"])


def _fam_of(t: torch.Tensor):
    if t.dtype == torch.float32:
        return _abi.F32
    if t.dtype == torch.float64:
        return _abi.F64
    raise TypeError(f"state columns must be float32 or float64, got {t.dtype}")


def _check_cols(cols, names):
    ref = cols[0]
    if not ref.is_cuda:
        raise ValueError("state columns must be resident on the GPU (torch device 'cuda'); cmx has no CPU path")
    for t, nm in zip(cols, names):
        if t.dtype != ref.dtype or t.device != ref.device or t.shape != ref.shape:
            raise ValueError(f"column {nm}: dtype/device/shape differ from {names[0]}")
        if not t.is_contiguous():
            raise ValueError(f"column {nm} must be contiguous (structure-of-arrays)")
    return ref


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _warm_rain(mp):
    if isinstance(mp, Microphysics2MParams):
        return mp.warm_rain
    if isinstance(mp, WarmRainParams2M):
        return mp
    raise TypeError("mp must be Microphysics2MParams or WarmRainParams2M")


def _vel_flag(vel):
    if vel is None:
        return 0
    flag = getattr(vel, "flag", None)
    if flag is None:
        raise TypeError("vel must be SB2006VelType, Chen2022VelTypeRain (class or instance) or None")
    return flag


def _bulk_tendencies_2m_p3(mp, tps, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log_lambda, inpc_log_shift,
                           aspect_ratio, stream):
    """bulk_microphysics_tendencies(::Microphysics2Moment, mp{WR, P3IceParams}, …) — BMT:898-1083 behind
    cmx_microphysics_2m_p3_tendencies_*."""
    cols = (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log_lambda)
    ref = _check_cols(cols, ("rho", "T", "q_tot", "q_lcl", "n_lcl", "q_rai", "n_rai", "q_ice", "n_ice", "q_rim", "b_rim", "log_lambda"))
    fam = _fam_of(ref)
    if fam is not mp.fam or fam is not mp.ice.fam or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    if mp.ice.is_limited != mp.warm_rain.is_limited:
        raise ValueError("warm_rain and ice must use the same rain PSD variant (is_limited)")
    if inpc_log_shift is not None:
        _check_cols([ref, inpc_log_shift], ["rho", "inpc_log_shift"])
    outs = [torch.empty_like(ref) for _ in range(8)]
    out_p = (C.c_void_p * 8)(*[t.data_ptr() for t in outs])
    flags = mp.ice.flags | (0 if aspect_ratio else _abi.CMX_P3_NO_ASPECT_RATIO)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_microphysics_2m_p3_tendencies_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(mp.warm_rain.c), C.byref(mp.ice.c), C.byref(tps), flags, ref.numel(), *[_ptr(t) for t in cols], _ptr(inpc_log_shift),
                out_p, C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return Tendencies2MP3(*outs, torch.zeros((), dtype=ref.dtype, device=ref.device).expand_as(ref))


def bulk_microphysics_tendencies(scheme, mp, tps, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice=None, n_ice=None, q_rim=None,
                                 b_rim=None, log_lambda=None, inpc_log_shift=None, *, vel=None, vel_params=None, aspect_ratio=True,
                                 out: Optional[WarmRainTendencies2M] = None, stream=None):
    """2-moment tendencies over columns.

    With `mp = Microphysics2MParams(FT; with_ice = true)` and the P3 ice columns (q_ice, n_ice, q_rim, b_rim, logλ[, inpc_log_shift])
    this is the warm rain + P3 ice method, BMT:898-1083 (returns `Tendencies2MP3`).  Otherwise:

    2-moment warm-rain tendencies over columns — BMT:820-854 → warm_rain_tendencies_2m BMT:707-782.

    `vel` (None | SB2006VelType | Chen2022VelTypeRain) additionally fuses
    CM2.rain_terminal_velocity (CM2:685-719) into the same pass (two more output columns); `vel_params` = a `cmx_rain_vel` struct
    with non-default tables (default: `parameters.rain_vel_params(FT)`).
    `out` lets the caller provide the output columns (KA-kernel style, test/gpu_tests.jl:407-415).
    Asynchronous on `stream` (default: torch's current stream)."""
    if not isinstance(scheme, Microphysics2Moment):
        raise TypeError("only Microphysics2Moment() is on this path")
    if isinstance(mp, Microphysics2MParams) and mp.ice is not None:
        if any(x is None for x in (q_ice, n_ice, q_rim, b_rim, log_lambda)):
            raise TypeError("the 2M + P3 method needs q_ice, n_ice, q_rim, b_rim and log_lambda columns")
        if vel is not None or out is not None:
            raise TypeError("vel / out are options of the warm-rain method")
        return _bulk_tendencies_2m_p3(mp, tps, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log_lambda,
                                      inpc_log_shift, aspect_ratio, stream)
    if q_ice is not None:
        raise TypeError("ice columns given but mp has no ice parameters (Microphysics2MParams(FT, with_ice=True))")
    wr = _warm_rain(mp)
    cols = (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)
    ref = _check_cols(cols, ("rho", "T", "q_tot", "q_lcl", "n_lcl", "q_rai", "n_rai"))
    fam = _fam_of(ref)
    if fam is not wr.fam or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    vflag = _vel_flag(vel)
    if out is None:
        mk = lambda: torch.empty_like(ref)  # noqa: E731
        out = WarmRainTendencies2M(mk(), mk(), mk(), mk(), mk() if vflag else None, mk() if vflag else None)
    else:
        _check_cols([ref] + [o for o in out if o is not None], ["rho"] + ["out"] * 6)
    flags = (_abi.CMX_SB2006_LIMITED if wr.is_limited else 0) | vflag
    velp = (vel_params if vel_params is not None else rain_vel_params(fam.sfx)) if vflag else None
    if velp is not None and not isinstance(velp, fam.rain_vel):
        raise TypeError("vel_params must be the cmx_rain_vel struct of the state's float type")
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_sb2006_warm_rain_tendencies_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(wr.c), C.byref(tps), C.byref(velp) if velp is not None else None, flags, ref.numel(),
                *[_ptr(t) for t in cols], *[_ptr(o) for o in out], C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


AOS_FIELDS = ("dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt", "dq_ice_dt", "dq_rim_dt", "db_rim_dt", "dn_lcl_activation_dt")


def _segments(t: torch.Tensor, name: str):
    """(n_seg, seg_len, stride) of a column given as a 1-D contiguous tensor or a 2-D (n_seg, seg_len) view whose rows are
    contiguous — e.g. `parent(field)[h, f, :]` of a ClimaCore VIJFH array seen from C order as (Nh, Nf, Nv·Ni·Nj)."""
    if t.dim() == 1 and t.stride(0) == 1:
        return 1, t.numel(), 0
    if t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1) and t.stride(0) >= t.shape[1]:
        return t.shape[0], t.shape[1], t.stride(0)
    raise TypeError(f"{name}: expected a contiguous 1-D column or a (n_seg, seg_len) view with contiguous rows")


def _fields_call(fn_name, params_c, tps, flags, cols, names, n_out, n_aos, out, aos, stream, extra=()):
    """Shared driver of the `_fields` entry points (segmented columns in; segmented columns or an (n, n_aos) array of rows out)."""
    ref = cols[0]
    fam = _fam_of(ref)
    n_seg, seg_len, _ = _segments(ref, names[0])
    strides = []
    for c, nm in zip(cols, names):
        if not c.is_cuda or c.device != ref.device or c.dtype != ref.dtype:
            raise TypeError(f"{nm}: all columns must live on the same GPU with the same dtype")
        ns, sl, st = _segments(c, nm)
        if (ns, sl) != (n_seg, seg_len):
            raise ValueError(f"{nm}: shape differs from {names[0]}")
        strides.append(st)
    n = n_seg * seg_len
    in_p = (C.c_void_p * len(cols))(*[c.data_ptr() for c in cols])
    in_s = (C.c_int64 * len(cols))(*strides)
    out_p = out_s = aos_t = None
    if aos:
        if out is not None:
            raise TypeError("out is the SoA form; aos=True allocates the array-of-rows result")
        aos_t = torch.empty((n, n_aos), dtype=ref.dtype, device=ref.device)
    else:
        if out is None:
            out = [torch.empty(ref.shape, dtype=ref.dtype, device=ref.device) for _ in range(n_out)]
        ostr = []
        for o in out:
            ns, sl, st = _segments(o, "out")
            if (ns, sl) != (n_seg, seg_len) or o.dtype != ref.dtype or o.device != ref.device:
                raise ValueError("out: shape / dtype / device differs from the inputs")
            ostr.append(st)
        out_p = (C.c_void_p * n_out)(*[o.data_ptr() for o in out])
        out_s = (C.c_int64 * n_out)(*ostr)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"{fn_name}_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(params_c), C.byref(tps), flags, *extra, n_seg, seg_len, in_p, in_s, out_p, out_s, C.c_void_p(aos_t.data_ptr()) if aos else None,
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return aos_t if aos else out


def bulk_microphysics_tendencies_2m_p3_fields(scheme, mp, tps, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log_lambda,
                                              inpc_log_shift=None, *, aspect_ratio=True, out=None, stream=None):
    """The 2M + P3 method of `bulk_microphysics_tendencies` (BMT:898-1083) on the host model's own storage — the layout adapter of
    `cmx_microphysics_2m_p3_tendencies_fields_*`: every column is a contiguous 1-D tensor or a 2-D strided view (n_seg, seg_len) with
    contiguous rows (a component of a ClimaCore `VIJFH` field in place; each column may have its own row stride, all share the shape); the
    eight tendencies go into `out` (8 tensors of that shape, e.g. components of the tendency field; allocated contiguous if None).
    Bit-identical to the SoA call on the same states.  Returns `Tendencies2MP3`."""
    if not isinstance(scheme, Microphysics2Moment):
        raise TypeError("only Microphysics2Moment() is on this path")
    if not (isinstance(mp, Microphysics2MParams) and mp.ice is not None):
        raise TypeError("mp must be Microphysics2MParams(FT, with_ice=True)")
    names = ["rho", "T", "q_tot", "q_lcl", "n_lcl", "q_rai", "n_rai", "q_ice", "n_ice", "q_rim", "b_rim", "log_lambda"]
    cols = [rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log_lambda]
    if inpc_log_shift is not None:
        names.append("inpc_log_shift"); cols.append(inpc_log_shift)
    ref = cols[0]
    fam = _fam_of(ref)
    if fam is not mp.fam or fam is not mp.ice.fam or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    if mp.ice.is_limited != mp.warm_rain.is_limited:
        raise ValueError("warm_rain and ice must use the same rain PSD variant (is_limited)")
    n_seg, seg_len, _ = _segments(ref, names[0])
    strides = []
    for c, nm in zip(cols, names):
        if not c.is_cuda or c.device != ref.device or c.dtype != ref.dtype:
            raise TypeError(f"{nm}: all columns must live on the same GPU with the same dtype")
        ns, sl, st = _segments(c, nm)
        if (ns, sl) != (n_seg, seg_len):
            raise ValueError(f"{nm}: shape differs from rho")
        strides.append(st)
    ptrs = [c.data_ptr() for c in cols]
    if inpc_log_shift is None:
        ptrs.append(None); strides.append(seg_len)
    if out is None:
        out = [torch.empty(ref.shape, dtype=ref.dtype, device=ref.device) for _ in range(8)]
    if len(out) != 8:
        raise ValueError("out: eight tendency columns")
    ostr = []
    for o in out:
        ns, sl, st = _segments(o, "out")
        if (ns, sl) != (n_seg, seg_len) or o.dtype != ref.dtype or o.device != ref.device:
            raise ValueError("out: shape / dtype / device differs from the inputs")
        ostr.append(st)
    in_p, in_s = (C.c_void_p * 13)(*ptrs), (C.c_int64 * 13)(*strides)
    out_p, out_s = (C.c_void_p * 8)(*[o.data_ptr() for o in out]), (C.c_int64 * 8)(*ostr)
    flags = mp.ice.flags | (0 if aspect_ratio else _abi.CMX_P3_NO_ASPECT_RATIO)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_microphysics_2m_p3_tendencies_fields_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(mp.warm_rain.c), C.byref(mp.ice.c), C.byref(tps), flags, n_seg, seg_len, in_p, in_s, out_p, out_s, C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return Tendencies2MP3(*out, torch.zeros((), dtype=ref.dtype, device=ref.device).expand(ref.shape))


def bulk_microphysics_tendencies_fields(scheme, mp, tps, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, *, out=None, aos=False, stream=None):
    """The 2-moment warm-rain tendencies (BMT:820-854) on the host model's own storage — zero-copy layout adapters of
    `cmx_sb2006_warm_rain_tendencies_fields_*`:

    * inputs: each column is a contiguous 1-D tensor or a 2-D strided view (n_seg, seg_len) with contiguous rows — a component of a
      ClimaCore `VIJFH` field in place (every column may have its own row stride, all share the shape);
    * `aos=False`: the four tendencies go into `out` (4 tensors of the same shape, e.g. components of the tendency field; allocated
      contiguous if None) → `WarmRainTendencies2M` without velocities;
    * `aos=True`: returns the reference's result layout, an (n, 8) tensor whose rows are the NamedTuple
      (dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt, dq_ice_dt, dq_rim_dt, db_rim_dt, dn_lcl_activation_dt) — `AOS_FIELDS`.

    The values are bit-identical to `bulk_microphysics_tendencies` on the same points."""
    if not isinstance(scheme, Microphysics2Moment):
        raise TypeError("only Microphysics2Moment() is on this path")
    wr = _warm_rain(mp)
    cols = (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)
    fam = _fam_of(rho)
    if fam is not wr.fam or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    flags = _abi.CMX_SB2006_LIMITED if wr.is_limited else 0
    r = _fields_call("cmx_sb2006_warm_rain_tendencies_fields", wr.c, tps, flags, cols, ("rho", "T", "q_tot", "q_lcl", "n_lcl", "q_rai", "n_rai"),
                     4, 8, out, aos, stream)
    return r if aos else WarmRainTendencies2M(*r, None, None)


def sb2006_process_rates(mp, tps, q_tot, q_lcl, q_rai, N_lcl, N_rai, rho, T, *, vel=SB2006VelType, stream=None):
    """The individual SB2006 process rates over columns — the reference's SB2006_2M_kernel
    (test/gpu_tests.jl:220-235) + cond/evap (NonEq:117-140).  N_* are per m³ (CM2 convention)."""
    wr = _warm_rain(mp)
    cols = (q_tot, q_lcl, q_rai, N_lcl, N_rai, rho, T)
    ref = _check_cols(cols, ("q_tot", "q_lcl", "q_rai", "N_lcl", "N_rai", "rho", "T"))
    fam = _fam_of(ref)
    if fam is not wr.fam or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    vflag = _vel_flag(vel)
    outs = [torch.empty_like(ref) for _ in range(_abi.CMX_SB2006_NPROC)]
    if not vflag:
        outs[_abi.SB2006_PROCESS_COLUMNS.index("rain_vel_n")] = None
        outs[_abi.SB2006_PROCESS_COLUMNS.index("rain_vel_m")] = None
    arr = (C.c_void_p * _abi.CMX_SB2006_NPROC)(*[o.data_ptr() if o is not None else None for o in outs])
    flags = (_abi.CMX_SB2006_LIMITED if wr.is_limited else 0) | vflag
    velp = rain_vel_params(fam.sfx) if vflag else None
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_sb2006_process_rates_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(wr.c), C.byref(tps), C.byref(velp) if velp is not None else None, flags, ref.numel(),
                *[_ptr(t) for t in cols], arr, C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return SB2006ProcessRates(*outs)


def column_sums(cols, stream=None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Σ of each device column (double accumulation) → float64 tensor [len(cols)] on the device (≤ 16 columns).
    Building block of the optional diagnostic reduction; no communication here.  Deterministic: one launch over all
    columns into `workspace` (len(cols) × 1024 doubles, allocated here unless given), then a fixed-tree finish — bit-identical from run
    to run (include/cmx.h §3)."""
    cols = list(cols)
    if len(cols) > _abi.CMX_COLUMN_SUMS_MAX_COLS:
        raise ValueError(f"at most {_abi.CMX_COLUMN_SUMS_MAX_COLS} columns per call")
    ref = _check_cols(cols, [f"col{i}" for i in range(len(cols))])
    fam = _fam_of(ref)
    sums = torch.empty(len(cols), dtype=torch.float64, device=ref.device)
    need = len(cols) * _abi.CMX_COLUMN_SUMS_PARTIALS
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.float64, device=ref.device)
    elif workspace.dtype != torch.float64 or workspace.device != ref.device or workspace.numel() < need or not workspace.is_contiguous():
        raise ValueError(f"workspace: a contiguous float64 device tensor of at least {need} elements")
    arr = (C.c_void_p * len(cols))(*[t.data_ptr() for t in cols])
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_column_sums_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(len(cols), arr, ref.numel(), _ptr(sums), _ptr(workspace), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return sums


CloudTerminalVelocity = namedtuple("CloudTerminalVelocity", ["vt_n", "vt_m"])


def cloud_terminal_velocity(pdf_c, vel, q_liq, rho, N_liq, *, stream=None) -> CloudTerminalVelocity:
    """`CM2.cloud_terminal_velocity.(Ref(pdf_c), Ref(vel), q_liq, ρₐ, N_liq)` (src/Microphysics2M.jl:647-664): number- and
    mass-weighted mean fall speeds of the cloud droplets; `pdf_c` = SB2006(FT).pdf_c, `vel` = StokesRegimeVelType(FT);
    N_liq per m³."""
    cols = (q_liq, rho, N_liq)
    ref = _check_cols(cols, ("q_liq", "rho", "N_liq"))
    fam = _fam_of(ref)
    if not isinstance(pdf_c, fam.cloud_pdf_sb2006) or not isinstance(vel, fam.stokes_vel):
        raise TypeError("parameter float type does not match the state columns")
    out = CloudTerminalVelocity(torch.empty_like(ref), torch.empty_like(ref))
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_sb2006_cloud_terminal_velocity_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(pdf_c), C.byref(vel), ref.numel(), *[_ptr(t) for t in cols], _ptr(out.vt_n), _ptr(out.vt_m),
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


ColumnTendencies2M = namedtuple("ColumnTendencies2M", ["dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt", "precip_flux"])


def column_tendencies_sedimentation(mp, tps, inv_dz, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, *, vel=SB2006VelType, cloud_vel=None,
                                    want_precip_flux: bool = False, out: Optional[ColumnTendencies2M] = None,
                                    stream=None) -> ColumnTendencies2M:
    """SURVEY §8f-4 — the fused column step behind `cmx_sb2006_column_tendencies_sedimentation_*`: per column of `n_lev`
    contiguous levels (state tensors of shape (n_col, n_lev), level 0 = lowest)

        BMT.bulk_microphysics_tendencies(Microphysics2Moment(), mp, tps, ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)   BMT:820-854
        CM2.rain_terminal_velocity(sb, vel, q_rai, ρ, ρ n_rai)  [+ CM2.cloud_terminal_velocity iff `cloud_vel`]     CM2:685-719, 647-664

    plus the host model's first-order upwind sedimentation flux divergence ∂χ/∂t += (F_{k+1} − F_k)/(ρ_k Δz_k), F = ρ χ w
    (the flux scheme is the host model's, not the reference's: include/cmx.h).  `inv_dz`: the n_lev values 1/Δz_k.
    `cloud_vel` = parameters.StokesRegimeVelType(FT) adds cloud-droplet sedimentation of q_lcl / n_lcl."""
    wr = _warm_rain(mp)
    cols = (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)
    ref = _check_cols(cols, ("rho", "T", "q_tot", "q_lcl", "n_lcl", "q_rai", "n_rai"))
    if ref.dim() != 2:
        raise ValueError("state columns must have shape (n_col, n_lev)")
    n_col, n_lev = ref.shape
    fam = _fam_of(ref)
    if fam is not wr.fam or not isinstance(tps, fam.thermo) or (cloud_vel is not None and not isinstance(cloud_vel, fam.stokes_vel)):
        raise TypeError("parameter float type does not match the state columns")
    if inv_dz.dtype != ref.dtype or inv_dz.device != ref.device or inv_dz.numel() != n_lev or not inv_dz.is_contiguous():
        raise ValueError("inv_dz must be a contiguous device vector of n_lev values of the state dtype")
    flag = _vel_flag(vel)
    if not flag:
        raise ValueError("the column step needs a rain fall-speed scheme (SB2006VelType or Chen2022VelTypeRain)")
    if out is None:
        out = ColumnTendencies2M(*[torch.empty_like(ref) for _ in range(4)],
                                 torch.empty(n_col, dtype=ref.dtype, device=ref.device) if want_precip_flux else None)
    else:
        _check_cols([ref, *out[:4]], ["rho", *ColumnTendencies2M._fields[:4]])
    flags = (_abi.CMX_SB2006_LIMITED if wr.is_limited else 0) | flag
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_sb2006_column_tendencies_sedimentation_{fam.sfx}")
    velp = rain_vel_params(fam.sfx)
    with torch.cuda.device(ref.device):
        st = fn(C.byref(wr.c), C.byref(tps), C.byref(velp), C.byref(cloud_vel) if cloud_vel is not None else None, flags, n_col, n_lev,
                _ptr(inv_dz), *[_ptr(t) for t in cols], *[_ptr(t) for t in out[:4]], _ptr(out.precip_flux), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


_SCHEMES_2M = {"KK2000": _abi.CMX_2M_KK2000, "B1994": _abi.CMX_2M_B1994, "TC1980": _abi.CMX_2M_TC1980, "LD2004": _abi.CMX_2M_LD2004}
CloudToRain2M = namedtuple("CloudToRain2M", ["acnv", "accr"])


def bulk_2m_cloud_to_rain(schemes, scheme: str, q_lcl, rho, N_d=None, q_rai=None, *, smooth_transition=False,
                          stream=None) -> CloudToRain2M:
    """`CM2.conv_q_lcl_to_q_rai.(Ref(scheme), q_lcl, ρ, N_d[, smooth_transition])` (→ `acnv`, needs N_d [1/m³]) and
    `CM2.accretion.(Ref(scheme), q_lcl, q_rai, ρ)` (→ `accr`, needs q_rai) for scheme ∈ KK2000 | B1994 | TC1980 | LD2004
    (src/Microphysics2M.jl:920-1003); `schemes` = parameters.Bulk2MSchemes(FT)."""
    if scheme not in _SCHEMES_2M:
        raise ValueError(f"scheme must be one of {sorted(_SCHEMES_2M)}")
    if scheme == "LD2004" and q_rai is not None:
        raise ValueError("LD2004 has no accretion parameterization")
    cols = [c for c in (q_lcl, rho, N_d, q_rai) if c is not None]
    ref = _check_cols(cols, ["q_lcl", "rho", "N_d", "q_rai"][:len(cols)])
    fam = _fam_of(ref)
    if not isinstance(schemes, fam.bulk_2m_schemes):
        raise TypeError("parameter float type does not match the state columns")
    if N_d is None and q_rai is None:
        raise ValueError("pass N_d (autoconversion) and / or q_rai (accretion)")
    acnv = torch.empty_like(ref) if N_d is not None else None
    accr = torch.empty_like(ref) if q_rai is not None else None
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_bulk_2m_cloud_to_rain_{fam.sfx}")
    flags = _SCHEMES_2M[scheme] | (_abi.CMX_2M_SMOOTH_TRANSITION if smooth_transition else 0)
    with torch.cuda.device(ref.device):
        st = fn(C.byref(schemes), flags, ref.numel(), _ptr(q_lcl), _ptr(q_rai), _ptr(rho), _ptr(N_d), _ptr(acnv), _ptr(accr),
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return CloudToRain2M(acnv, accr)
