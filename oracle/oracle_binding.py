"""ctypes binding of the CPU oracle (oracle/libcmx_oracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It reuses the ABI struct definitions of the product package
(cloudmicrophysics.jl_amd/cmx/_abi.py ↔ include/cmx.h) — the oracle sees exactly
the parameter bytes the device library sees — but nothing in the product imports
anything from oracle/.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ORACLE_DIR = Path(__file__).resolve().parent
REPO = ORACLE_DIR.parent
sys.path.insert(0, str(REPO / "cloudmicrophysics.jl_amd"))
from cmx import _abi  # noqa: E402

import os  # noqa: E402

# CMX_ORACLE_LIB: bench.py's cpu_baseline leg points this at a copy built on the box it runs on (`make -C oracle native`:
# -O3 -march=native); the tests always use the portable default build.
LIB_PATH = Path(os.environ.get("CMX_ORACLE_LIB", ORACLE_DIR / "libcmx_oracle.so"))
_lib = None


def build(force: bool = False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force and LIB_PATH.exists():
        LIB_PATH.unlink()
    subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, capture_output=True)


def _thresholds_type(fam):
    return type(f"cmxo_thresholds_{fam.sfx}", (C.Structure,),
                {"_fields_": [(n, fam.ft) for n in ("eps_m", "eps_n", "eps_1m", "eps_ft")]})


TH = {"f32": _thresholds_type(_abi.F32), "f64": _thresholds_type(_abi.F64)}
NP = {"f32": np.float32, "f64": np.float64}


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(LIB_PATH))
        for fam in (_abi.F32, _abi.F64):
            s = fam.sfx
            getattr(_lib, f"cmxo_psat_liquid_{s}").restype = fam.ft
            getattr(_lib, f"cmxo_psat_liquid_{s}").argtypes = [C.POINTER(fam.thermo), fam.ft]
            getattr(_lib, f"cmxo_psat_ice_{s}").restype = fam.ft
            getattr(_lib, f"cmxo_psat_ice_{s}").argtypes = [C.POINTER(fam.thermo), fam.ft]
            getattr(_lib, f"cmxo_gamma_incl_{s}").restype = fam.ft
            getattr(_lib, f"cmxo_gamma_incl_{s}").argtypes = [fam.ft, fam.ft]
    return _lib


def thresholds(fam, float32_gates: bool):
    """eps(FT)/cbrt(floatmin(FT)) gates (src/Utilities.jl:318-340) of Float32 or Float64, stored as fam.ft."""
    th = TH[fam.sfx]()
    getattr(lib(), f"cmxo_default_thresholds_{fam.sfx}")(C.byref(th), int(float32_gates))
    return th


def _col(fam, a):
    a = np.ascontiguousarray(a, dtype=NP[fam.sfx])
    return a, a.ctypes.data_as(C.c_void_p)


def sb2006_warm_rain_tendencies(fam, wr, tps, vel, flags, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, *,
                                float32_gates=None, nthreads=1, want_scale=True, branch_margin=1e-5):
    """Oracle twin of cmx_sb2006_warm_rain_tendencies_*: numpy in → dict of numpy columns.

    `fam` selects the ARITHMETIC (F64 = the reference's Float64 path); `float32_gates`
    the THRESHOLDS (default: those of `fam`).  `scale[k]` = Σ|cancelling terms| of output k;
    `near_branch` marks points within `branch_margin` (relative) of the Φ_br jump (CM2:596)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)]
    n = ins[0][0].size
    names = ["dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt", "vt_rai_n", "vt_rai_m"]
    outs = {k: np.empty(n, dtype=NP[fam.sfx]) for k in names}
    scale = [np.empty(n, dtype=NP[fam.sfx]) for _ in range(6)] if want_scale else None
    scale_arr = (C.c_void_p * 6)(*[s.ctypes.data for s in scale]) if want_scale else None
    near = np.zeros(n, dtype=np.uint8) if want_scale else None
    fn = getattr(lib(), f"cmxo_sb2006_warm_rain_tendencies_{fam.sfx}")
    fn.restype = None
    fn(C.byref(wr), C.byref(tps), C.byref(vel) if vel is not None else None, C.c_uint32(flags), C.byref(th),
       C.c_int64(n), *[p for _, p in ins], *[outs[k].ctypes.data_as(C.c_void_p) for k in names],
       scale_arr, near.ctypes.data_as(C.c_void_p) if want_scale else None, fam.ft(branch_margin),
       C.c_int32(nthreads))
    if want_scale:
        outs["scale"] = dict(zip(names, scale))
        outs["near_branch"] = near.astype(bool)
    return outs


def sb2006_column_tendencies_sedimentation(fam, wr, tps, vel, cloud_vel, flags, inv_dz, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, *,
                                           float32_gates=None, nthreads=1, branch_margin=1e-5):
    """Oracle twin of cmx_sb2006_column_tendencies_sedimentation_* (flux step: parity unpinned, see cmx_oracle_column_impl.h).
    State arrays of shape (n_col, n_lev); returns the 4 tendencies (flat), `precip_flux` (n_col), `scale`, `near_branch`."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    n_col, n_lev = np.shape(rho)
    ins = [_col(fam, np.reshape(a, -1)) for a in (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)]
    dz = _col(fam, inv_dz)
    n = n_col * n_lev
    names = ["dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt"]
    outs = [np.empty(n, dtype=NP[fam.sfx]) for _ in names]
    scale = [np.empty(n, dtype=NP[fam.sfx]) for _ in names]
    precip = np.empty(n_col, dtype=NP[fam.sfx])
    near = np.zeros(n, dtype=np.uint8)
    fn = getattr(lib(), f"cmxo_sb2006_column_tendencies_sedimentation_{fam.sfx}")
    fn.restype = None
    fn(C.byref(wr), C.byref(tps), C.byref(vel), C.byref(cloud_vel) if cloud_vel is not None else None, C.c_uint32(flags), C.byref(th),
       C.c_int64(n_col), C.c_int32(n_lev), dz[1], *[p for _, p in ins], *[o.ctypes.data_as(C.c_void_p) for o in outs],
       precip.ctypes.data_as(C.c_void_p), (C.c_void_p * 4)(*[c.ctypes.data for c in scale]), near.ctypes.data_as(C.c_void_p),
       fam.ft(branch_margin), C.c_int32(nthreads))
    res = dict(zip(names, outs))
    res["precip_flux"] = precip
    res["scale"] = dict(zip(names, scale))
    res["near_branch"] = near.astype(bool)
    return res


def sb2006_process_rates(fam, wr, tps, vel, flags, q_tot, q_lcl, q_rai, N_lcl, N_rai, rho, T, *, float32_gates=None):
    """Oracle twin of cmx_sb2006_process_rates_* (SB2006_2M_kernel, test/gpu_tests.jl:220-235)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (q_tot, q_lcl, q_rai, N_lcl, N_rai, rho, T)]
    n = ins[0][0].size
    outs = [np.empty(n, dtype=NP[fam.sfx]) for _ in range(_abi.CMX_SB2006_NPROC)]
    arr = (C.c_void_p * _abi.CMX_SB2006_NPROC)(*[o.ctypes.data for o in outs])
    fn = getattr(lib(), f"cmxo_sb2006_process_rates_{fam.sfx}")
    fn.restype = None
    fn(C.byref(wr), C.byref(tps), C.byref(vel) if vel is not None else None, C.c_uint32(flags), C.byref(th),
       C.c_int64(n), *[p for _, p in ins], arr)
    return dict(zip(_abi.SB2006_PROCESS_COLUMNS, outs))


def ice_nucleation_rates(fam, tps, dust, koop, flags, T, a_w, r):
    """Oracle twin of cmx_ice_nucleation_rates_*: dict of the five columns + 'n_domain_errors'."""
    ins = [_col(fam, a) for a in (T, a_w, r)]
    n = ins[0][0].size
    names = ["delta_a_w", "J_het", "J_hom", "rate_het", "rate_hom"]
    outs = {k: np.empty(n, dtype=NP[fam.sfx]) for k in names}
    fn = getattr(lib(), f"cmxo_ice_nucleation_rates_{fam.sfx}")
    fn.restype = C.c_int64
    nerr = fn(C.byref(tps), C.byref(dust), C.byref(koop), C.c_uint32(flags), C.c_int64(n), *[p for _, p in ins],
              *[outs[k].ctypes.data_as(C.c_void_p) for k in names])
    outs["n_domain_errors"] = int(nerr)
    return outs


def water_activity(fam, tps, T, e=None):
    Tn, Tp = _col(fam, T)
    n = Tn.size
    ice = np.empty(n, dtype=NP[fam.sfx])
    eT = np.empty(n, dtype=NP[fam.sfx]) if e is not None else None
    en, ep = _col(fam, e) if e is not None else (None, None)
    fn = getattr(lib(), f"cmxo_water_activity_{fam.sfx}")
    fn.restype = None
    fn(C.byref(tps), C.c_int64(n), Tp, ep, ice.ctypes.data_as(C.c_void_p),
       eT.ctypes.data_as(C.c_void_p) if eT is not None else None)
    return ice, eT


def mp1m(fam, mp, tps, flags, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, *, float32_gates=None, nthreads=1,
         want_sources=True):
    """Oracle twin of cmx_mp1m_tendencies_* + cmx_mp1m_source_terms_*: the 4 tendencies, their Σ|terms| scales and
    (optionally) the 18 source terms of BMT:141-217."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)]
    n = ins[0][0].size
    tn = ["dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"]
    tend = [np.empty(n, dtype=NP[fam.sfx]) for _ in tn]
    scale = [np.empty(n, dtype=NP[fam.sfx]) for _ in tn]
    src = [np.empty(n, dtype=NP[fam.sfx]) for _ in range(_abi.CMX_MP1M_NSRC)] if want_sources else None
    arr = lambda cols: (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])  # noqa: E731
    fn = getattr(lib(), f"cmxo_mp1m_{fam.sfx}")
    fn.restype = None
    fn(C.byref(mp), C.byref(tps), C.c_uint32(flags), C.byref(th), C.c_int64(n), *[p for _, p in ins], arr(tend),
       arr(scale), arr(src) if want_sources else None, C.c_int32(nthreads))
    out = dict(zip(tn, tend))
    out["scale"] = dict(zip(tn, scale))
    if want_sources:
        out["sources"] = dict(zip(_abi.MP1M_SOURCE_COLUMNS, src))
    return out


def mp1m_linearized_average(fam, mp, tps, flags, q_min, dt, nsub, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, *, float32_gates=None,
                            nthreads=1):
    """Oracle twin of cmx_mp1m_linearized_average_*: dict of the 4 average tendencies."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)]
    n = ins[0][0].size
    tn = ["dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"]
    tend = [np.empty(n, dtype=NP[fam.sfx]) for _ in tn]
    fn = getattr(lib(), f"cmxo_mp1m_linearized_average_{fam.sfx}")
    fn.restype = None
    fn(C.byref(mp), C.byref(tps), C.c_uint32(flags), C.byref(th), fam.ft(q_min), fam.ft(dt), C.c_int32(nsub), C.c_int64(n),
       *[p for _, p in ins], (C.c_void_p * 4)(*[c.ctypes.data for c in tend]), C.c_int32(nthreads))
    return dict(zip(tn, tend))


def mp1m_linearize(fam, mp, tps, flags, q_min, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, float32_gates=None):
    """(M11, M12, M22, M31, M33, M34, M41, M42, M43, M44, e1, e2, e4) of BMT._linearize at one state."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    out = (fam.ft * 13)()
    fn = getattr(lib(), f"cmxo_mp1m_linearize_{fam.sfx}")
    fn.restype = None
    fn(C.byref(mp), C.byref(tps), C.c_uint32(flags), C.byref(th), *[fam.ft(v) for v in (q_min, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)], out)
    return dict(zip(("M11", "M12", "M22", "M31", "M33", "M34", "M41", "M42", "M43", "M44", "e1", "e2", "e4"), list(out)))


def sedimentation_velocities(fam, mp, stokes, chen_rain, chen_ice, rho, q_lcl, q_icl, q_rai, q_sno, float32_gates=None):
    """Oracle twin of cmx_sedimentation_velocities_*: dict(w_lcl, w_icl, w_rai, w_sno)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho, q_lcl, q_icl, q_rai, q_sno)]
    n = ins[0][0].size
    names = ("w_lcl", "w_icl", "w_rai", "w_sno")
    outs = [np.empty(n, dtype=NP[fam.sfx]) for _ in names]
    fn = getattr(lib(), f"cmxo_sedimentation_velocities_{fam.sfx}")
    fn.restype = None
    fn(C.byref(mp), C.byref(stokes), C.byref(chen_rain), C.byref(chen_ice), C.byref(th), C.c_int64(n), *[p for _, p in ins],
       *[o.ctypes.data_as(C.c_void_p) for o in outs])
    return dict(zip(names, outs))


def mp1m_column_tendencies_sedimentation(fam, mp, tps, stokes, chen_rain, chen_ice, flags, inv_dz, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, *,
                                         q_min=0.0, dt=0.0, nsub=0, chen_ice_scale=None, float32_gates=None, nthreads=1):
    """Oracle twin of cmx_mp1m_column_tendencies_sedimentation_* (flux step: parity unpinned, see cmx_oracle_column_impl.h).  State
    arrays of shape (n_col, n_lev); nsub = 0: Instantaneous tendencies.  Returns the 4 tendencies (flat), the two surface fluxes, `scale`."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    n_col, n_lev = np.shape(rho)
    ins = [_col(fam, np.reshape(a, -1)) for a in (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)]
    dz = _col(fam, inv_dz)
    n = n_col * n_lev
    names = ["dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"]
    outs = [np.empty(n, dtype=NP[fam.sfx]) for _ in names]
    scale = [np.empty(n, dtype=NP[fam.sfx]) for _ in names]
    pr, ps = np.empty(n_col, dtype=NP[fam.sfx]), np.empty(n_col, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_mp1m_column_tendencies_sedimentation_{fam.sfx}")
    fn.restype = None
    fn(C.byref(mp), C.byref(tps), C.byref(stokes), C.byref(chen_rain), C.byref(chen_ice), C.byref(chen_ice_scale) if chen_ice_scale is not None else None,
       C.c_uint32(flags), C.byref(th), fam.ft(q_min), fam.ft(dt), C.c_int32(nsub), C.c_int64(n_col), C.c_int32(n_lev), dz[1], *[p for _, p in ins],
       (C.c_void_p * 4)(*[c.ctypes.data for c in outs]), pr.ctypes.data_as(C.c_void_p), ps.ctypes.data_as(C.c_void_p),
       (C.c_void_p * 4)(*[c.ctypes.data for c in scale]), C.c_int32(nthreads))
    res = dict(zip(names, outs))
    res["precip_rai"], res["precip_sno"] = pr, ps
    res["scale"] = dict(zip(names, scale))
    return res


def mp1m_terminal_velocity(fam, mp, chen, rho, q_rai, q_sno, float32_gates=None):
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho, q_rai, q_sno)]
    n = ins[0][0].size
    outs = [np.empty(n, dtype=NP[fam.sfx]) for _ in range(3)]
    fn = getattr(lib(), f"cmxo_mp1m_terminal_velocity_{fam.sfx}")
    fn.restype = None
    fn(C.byref(mp), C.byref(chen), C.byref(th), C.c_int64(n), *[p for _, p in ins],
       *[o.ctypes.data_as(C.c_void_p) for o in outs])
    return dict(zip(("vt_rai_blk1m", "vt_sno_blk1m", "vt_rai_chen"), outs))


def logistic_function_integral(fam, x, x_0, k, eps=None):
    if eps is None:
        eps = thresholds(fam, fam.sfx == "f32").eps_1m
    fn = getattr(lib(), f"cmxo_logistic_function_integral_{fam.sfx}")
    fn.restype = fam.ft
    fn.argtypes = [fam.ft] * 4
    return fn(x, x_0, k, eps)


def arg2000_activation(fam, ap, ad, aip, tps, T, p, w, q_tot, q_liq=None, q_ice=None, N_liq=None, N_ice=None, *,
                       float32_gates=None, nthreads=1):
    """Oracle twin of cmx_arg2000_activation_*: dict(N_act=[…per mode], M_act=[…], S_max=array)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    req = [_col(fam, a) for a in (T, p, w, q_tot)]
    opt = [(_col(fam, a) if a is not None else (None, None)) for a in (q_liq, q_ice, N_liq, N_ice)]
    n = req[0][0].size
    nm = ad.n_modes
    n_act = [np.empty(n, dtype=NP[fam.sfx]) for _ in range(nm)]
    m_act = [np.empty(n, dtype=NP[fam.sfx]) for _ in range(nm)]
    s_max = np.empty(n, dtype=NP[fam.sfx])
    s_cond = np.empty(n, dtype=NP[fam.sfx])
    arr = lambda cols: (C.c_void_p * nm)(*[c.ctypes.data for c in cols])  # noqa: E731
    fn = getattr(lib(), f"cmxo_arg2000_activation_{fam.sfx}")
    fn.restype = None
    fn(C.byref(ap), C.byref(ad), C.byref(aip), C.byref(tps), C.byref(th), C.c_int64(n), *[pp for _, pp in req],
       *[pp for _, pp in opt], arr(n_act), arr(m_act), s_max.ctypes.data_as(C.c_void_p),
       s_cond.ctypes.data_as(C.c_void_p), C.c_int32(nthreads))
    # S_cond ≥ 1: amplification of operand rounding in the ice-sink numerator αw − K_ice(ξ−1) of S_max (AA:197)
    return dict(N_act=n_act, M_act=m_act, S_max=s_max, S_cond=s_cond)


def arg2000_erf_argument(fam, ap, ad, aip, tps, T, p, w, q_tot, q_liq=None, q_ice=None, N_liq=None, N_ice=None, *, float32_gates=None):
    """u_i of N_activated_per_mode (src/AerosolActivation.jl:254) per mode: list of arrays."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    req = [_col(fam, a) for a in (T, p, w, q_tot)]
    opt = [(_col(fam, a) if a is not None else (None, None)) for a in (q_liq, q_ice, N_liq, N_ice)]
    n, nm = req[0][0].size, ad.n_modes
    u = [np.empty(n, dtype=NP[fam.sfx]) for _ in range(nm)]
    fn = getattr(lib(), f"cmxo_arg2000_erf_argument_{fam.sfx}")
    fn.restype = None
    fn(C.byref(ap), C.byref(ad), C.byref(aip), C.byref(tps), C.byref(th), C.c_int64(n), *[pp for _, pp in req], *[pp for _, pp in opt],
       (C.c_void_p * nm)(*[c.ctypes.data for c in u]))
    return u


def arg2000_activation_columns(fam, ap, aip, tps, T, p, w, q_tot, modes, *, want_M=False, float32_gates=None, nthreads=1):
    """Oracle twin of cmx_arg2000_activation_columns_*; `modes` = sequence of (r_dry, stdev, N, hygroscopicity, molar_mass_mix)
    numpy columns.  Returns dict(N_act=[…], M_act=[…] or None, S_max)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    req = [_col(fam, a) for a in (T, p, w, q_tot)]
    n, nm = req[0][0].size, len(modes)
    have_mm = all(len(m) > 4 and m[4] is not None for m in modes)
    if want_M and not have_mm:
        raise ValueError("M_act needs molar_mass_mix for every mode")
    mc = [[_col(fam, m[j]) for m in modes] for j in range(5 if have_mm else 4)] + ([] if have_mm else [None])
    arr = lambda cols: (C.c_void_p * nm)(*[c[0].ctypes.data for c in cols]) if cols is not None else None  # noqa: E731
    n_act = [np.empty(n, dtype=NP[fam.sfx]) for _ in range(nm)]
    m_act = [np.empty(n, dtype=NP[fam.sfx]) for _ in range(nm)] if want_M else None
    s_max = np.empty(n, dtype=NP[fam.sfx])
    out = lambda cols: (C.c_void_p * nm)(*[c.ctypes.data for c in cols]) if cols is not None else None  # noqa: E731
    fn = getattr(lib(), f"cmxo_arg2000_activation_columns_{fam.sfx}")
    fn.restype = None
    fn(C.byref(ap), C.byref(aip), C.byref(tps), C.byref(th), C.c_int32(nm), C.c_int64(n), *[q for _, q in req], None, None, None, None,
       arr(mc[0]), arr(mc[1]), arr(mc[2]), arr(mc[3]), arr(mc[4]), out(n_act), out(m_act), s_max.ctypes.data_as(C.c_void_p),
       C.c_int32(nthreads))
    return dict(N_act=n_act, M_act=m_act, S_max=s_max)


def p3_shape(fam, params, flags, rho_q_ice, rho_n_ice, x3, x4, *, guess=None, float32_gates=None, maxiters=0, gi_iters=0,
             nthreads=1):
    """Oracle twin of cmx_p3_shape_*: dict of F_rim, rho_rim, rho_g, D_gr, D_cr, log_lambda, D_m, log_N0.
    maxiters / gi_iters ≤ 0 → the reference's fixed budgets (Brent 8/10, gamma_inc 20/30 for Float32/Float64)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho_q_ice, rho_n_ice, x3, x4)]
    n = ins[0][0].size
    names = ["F_rim", "rho_rim", "rho_g", "D_gr", "D_cr", "log_lambda", "D_m", "log_N0"]
    outs = {k: np.empty(n, dtype=NP[fam.sfx]) for k in names}
    fn = getattr(lib(), f"cmxo_p3_shape_{fam.sfx}")
    fn.restype = None
    g = _col(fam, guess) if guess is not None else (None, None)
    fn(C.byref(params), C.c_uint32(flags), C.byref(th), C.c_int(maxiters), C.c_int(gi_iters), C.c_int64(n),
       *[p for _, p in ins], g[1], *[outs[k].ctypes.data_as(C.c_void_p) for k in names], C.c_int32(nthreads))
    return outs


def p3_terminal_velocities(fam, params, vel, quad, flags, rho_q_ice, rho_n_ice, x3, x4, rho_a, log_lambda, *, p=1e-6,
                           float32_gates=None, gi_iters=0, nthreads=1):
    """Oracle twin of cmx_p3_terminal_velocities_*: (v_n, v_m)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho_q_ice, rho_n_ice, x3, x4, rho_a, log_lambda)]
    n = ins[0][0].size
    v_n, v_m = np.empty(n, dtype=NP[fam.sfx]), np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_p3_terminal_velocities_{fam.sfx}")
    fn.restype = None
    fn(C.byref(params), C.byref(vel), C.byref(quad), C.c_uint32(flags), C.byref(th), C.c_int(gi_iters), fam.ft(p), C.c_int64(n),
       *[q for _, q in ins], v_n.ctypes.data_as(C.c_void_p), v_m.ctypes.data_as(C.c_void_p), C.c_int32(nthreads))
    return v_n, v_m


def p3_ice_melt(fam, params, vel, aps, tps, vent, quad, flags, rho_q_ice, rho_n_ice, x3, x4, rho_a, T, log_lambda, *, p=1e-6,
                float32_gates=None, nthreads=1):
    """Oracle twin of cmx_p3_ice_melt_*: (dNdt, dLdt)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho_q_ice, rho_n_ice, x3, x4, rho_a, T, log_lambda)]
    n = ins[0][0].size
    dN, dL = np.empty(n, dtype=NP[fam.sfx]), np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_p3_ice_melt_{fam.sfx}")
    fn.restype = None
    fn(C.byref(params), C.byref(vel), C.byref(aps), C.byref(tps), C.byref(vent), C.byref(quad), C.c_uint32(flags), C.byref(th), fam.ft(p),
       C.c_int64(n), *[q for _, q in ins], dN.ctypes.data_as(C.c_void_p), dL.ctypes.data_as(C.c_void_p), C.c_int32(nthreads))
    return dN, dL


def p3_ice_self_collection(fam, params, vel, quad, flags, rho_q_ice, rho_n_ice, x3, x4, rho_a, log_lambda, *, float32_gates=None,
                           nthreads=1):
    """Oracle twin of cmx_p3_ice_self_collection_*: dNdt column."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho_q_ice, rho_n_ice, x3, x4, rho_a, log_lambda)]
    n = ins[0][0].size
    out = np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_p3_ice_self_collection_{fam.sfx}")
    fn.restype = None
    fn(C.byref(params), C.byref(vel), C.byref(quad), C.c_uint32(flags), C.byref(th), C.c_int64(n), *[q for _, q in ins],
       out.ctypes.data_as(C.c_void_p), C.c_int32(nthreads))
    return out


def p3_particle_velocity(fam, params, vel, flags, F_rim, rho_rim, rho_a, D):
    fn = getattr(lib(), f"cmxo_p3_particle_velocity_{fam.sfx}")
    fn.restype = fam.ft
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, fam.ft, fam.ft, fam.ft, fam.ft]
    return fn(C.addressof(params), C.addressof(vel), flags, F_rim, rho_rim, rho_a, D)


def gamma_inc_inv(fam, a, p, q):
    fn = getattr(lib(), f"cmxo_gamma_inc_inv_{fam.sfx}")
    fn.restype = fam.ft
    fn.argtypes = [fam.ft] * 3
    return fn(a, p, q)


def p3_rho_d(fam, params, F_rim, rho_rim):
    fn = getattr(lib(), f"cmxo_p3_rho_d_{fam.sfx}")
    fn.restype = fam.ft
    fn.argtypes = [C.c_void_p, fam.ft, fam.ft]
    return fn(C.addressof(params), F_rim, rho_rim)


def set_quadrature_override(fam, nodes=None, weights=None):
    """A quadrature rule of any order (≤ 1024) for every P3 integral of this float type, replacing the `quad` argument until cleared (no arguments).
    The ABI's cmx_quadrature carries ≤ 128 nodes; the reference's order sweep compares against order 200."""
    fn = getattr(lib(), f"cmxo_set_quadrature_override_{fam.sfx}")
    fn.restype = C.c_int32
    fn.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
    if nodes is None:
        assert fn(0, None, None) == 0
        return
    x = np.ascontiguousarray(nodes, dtype=NP[fam.sfx])
    w = np.ascontiguousarray(weights, dtype=NP[fam.sfx])
    assert x.size == w.size and fn(x.size, x.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p)) == 0


def set_crossover_iters(n: int):
    """Iteration count of the collision integrals' crossover solve (0 = the reference's 8 / 10).  Both float types."""
    for sfx in ("f32", "f64"):
        fn = getattr(lib(), f"cmxo_set_crossover_iters_{sfx}")
        fn.restype = None
        fn.argtypes = [C.c_int32]
        fn(n)


def set_brent_variant(v: int):
    """0: Brent's zeroin (default); 1: the Wikipedia pseudo-code variant of rounds 1-5 (cmx_oracle_p3_impl.h o_brent_fixed).  Both float types."""
    for sfx in ("f32", "f64"):
        fn = getattr(lib(), f"cmxo_set_brent_variant_{sfx}")
        fn.restype = None
        fn.argtypes = [C.c_int32]
        fn(v)


def p3_rho_g(fam, params, F_rim, rho_rim):
    fn = getattr(lib(), f"cmxo_p3_rho_g_{fam.sfx}")
    fn.restype = fam.ft
    fn.argtypes = [C.c_void_p, fam.ft, fam.ft]
    return fn(C.addressof(params), F_rim, rho_rim)


def p3_particle_properties(fam, params, F_rim, rho_rim, D):
    """dict(mass, area, phi, D_th, D_gr, D_cr, rho_g) of an ice particle of diameter D in the state (F_rim, ρ_rim) — P3.ice_mass / ice_area / ϕᵢ and the thresholds."""
    out = (fam.ft * 7)()
    fn = getattr(lib(), f"cmxo_p3_particle_properties_{fam.sfx}")
    fn.restype = None
    fn.argtypes = [C.c_void_p, fam.ft, fam.ft, fam.ft, C.c_void_p]
    fn(C.addressof(params), F_rim, rho_rim, D, out)
    return dict(zip(("mass", "area", "phi", "D_th", "D_gr", "D_cr", "rho_g"), list(out)))


def p3_ventilation_factor(fam, params, vel, aps, vent, flags, F_rim, rho_rim, rho_a, D):
    """CO.ventilation_factor(vent, aps, P3.ice_particle_terminal_velocity(vel, ρₐ, state))(D) — src/Common.jl:506-514."""
    fn = getattr(lib(), f"cmxo_p3_ventilation_factor_{fam.sfx}")
    fn.restype = fam.ft
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, fam.ft, fam.ft, fam.ft, fam.ft]
    return fn(C.addressof(params), C.addressof(vel), C.addressof(aps), C.addressof(vent), flags, F_rim, rho_rim, rho_a, D)


def unrolled_logsumexp(fam, x):
    """UT.unrolled_logsumexp — src/Utilities.jl:399-412."""
    a = np.ascontiguousarray(x, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_unrolled_logsumexp_{fam.sfx}")
    fn.restype = fam.ft
    fn.argtypes = [C.c_int32, C.c_void_p]
    return fn(a.size, a.ctypes.data_as(C.c_void_p))


def p3_logLdivN(fam, params, flags, F_rim, rho_rim, loglam):
    fn = getattr(lib(), f"cmxo_p3_logLdivN_{fam.sfx}")
    fn.restype = fam.ft
    fn.argtypes = [C.c_void_p, C.c_uint32, fam.ft, fam.ft, fam.ft]
    return fn(C.addressof(params), flags, F_rim, rho_rim, loglam)


def gamma_inc(fam, a, x):
    out = (fam.ft * 2)()
    fn = getattr(lib(), f"cmxo_gamma_inc_{fam.sfx}")
    fn.restype = None
    fn.argtypes = [fam.ft, fam.ft, C.c_void_p]
    fn(a, x, out)
    return out[0], out[1]


def psat_liquid(fam, tps, T):
    return getattr(lib(), f"cmxo_psat_liquid_{fam.sfx}")(C.byref(tps), T)


def psat_ice(fam, tps, T):
    return getattr(lib(), f"cmxo_psat_ice_{fam.sfx}")(C.byref(tps), T)


def gamma_incl(fam, a, x):
    return getattr(lib(), f"cmxo_gamma_incl_{fam.sfx}")(a, x)


def pdf_rain_parameters(fam, pdf_r, limited, q, rho, N, float32_gates=None):
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    out = (fam.ft * 3)()
    fn = getattr(lib(), f"cmxo_pdf_rain_parameters_{fam.sfx}")
    fn.restype = None
    fn(C.byref(pdf_r), C.c_int(int(limited)), fam.ft(q), fam.ft(rho), fam.ft(N), C.byref(th), out)
    return dict(N0r=out[0], Dr_mean=out[1], xr_mean=out[2])


def cloud_terminal_velocity(fam, pdf_c, rho_w, grav, nu_air, q_liq, rho, N_liq, float32_gates=None):
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    out = (fam.ft * 2)()
    fn = getattr(lib(), f"cmxo_cloud_terminal_velocity_{fam.sfx}")
    fn.restype = None
    fn(C.byref(pdf_c), fam.ft(rho_w), fam.ft(grav), fam.ft(nu_air), fam.ft(q_liq), fam.ft(rho), fam.ft(N_liq),
       C.byref(th), out)
    return out[0], out[1]


def bulk_2m_cloud_to_rain(fam, schemes, scheme, q_lcl, q_rai, rho, N_d, float32_gates=None):
    """Oracle twin of cmx_bulk_2m_cloud_to_rain_*: (acnv, accr) columns (accr None for LD2004 / q_rai None)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    want_accr = q_rai is not None and (scheme & 0xff) != _abi.CMX_2M_LD2004
    ins = [_col(fam, a) if a is not None else (None, None) for a in (q_lcl, q_rai, rho, N_d)]
    n = ins[0][0].size
    acnv = np.empty(n, dtype=NP[fam.sfx])
    accr = np.empty(n, dtype=NP[fam.sfx]) if want_accr else None
    fn = getattr(lib(), f"cmxo_bulk_2m_cloud_to_rain_{fam.sfx}")
    fn.restype = None
    fn(C.byref(schemes), C.c_uint32(scheme), C.byref(th), C.c_int64(n), *[p for _, p in ins], acnv.ctypes.data_as(C.c_void_p),
       accr.ctypes.data_as(C.c_void_p) if want_accr else None)
    return acnv, accr


def sb2006_cloud_terminal_velocity(fam, pdf_c, vel, q_liq, rho, N_liq, float32_gates=None):
    """Oracle twin of cmx_sb2006_cloud_terminal_velocity_*: (vt_n, vt_m) columns."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (q_liq, rho, N_liq)]
    n = ins[0][0].size
    v0, v1 = np.empty(n, dtype=NP[fam.sfx]), np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_sb2006_cloud_terminal_velocity_{fam.sfx}")
    fn.restype = None
    fn(C.byref(pdf_c), C.byref(vel), C.byref(th), C.c_int64(n), *[p for _, p in ins], v0.ctypes.data_as(C.c_void_p),
       v1.ctypes.data_as(C.c_void_p))
    return v0, v1


def chen2022_rain_coeffs(fam, chen, rho):
    out = (fam.ft * 9)()
    fn = getattr(lib(), f"cmxo_chen2022_rain_coeffs_{fam.sfx}")
    fn.restype = None
    fn(C.byref(chen), fam.ft(rho), out)
    return list(out[0:3]), list(out[3:6]), list(out[6:9])


def p3_liquid_ice_collisions(fam, ice_params, aps, tps, quad, flags, rho_q_ice, rho_n_ice, x3, x4, L_c, N_c, L_r, N_r, rho_a, T,
                             log_lambda, *, float32_gates=None, nthreads=1):
    """Oracle twin of cmx_p3_liquid_ice_collisions_*: (sources[7, n], rates[10, n]) — bulk_liquid_ice_collision_sources and the ten
    ∫liquid_ice_collisions integrals."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho_q_ice, rho_n_ice, x3, x4, L_c, N_c, L_r, N_r, rho_a, T, log_lambda)]
    n = ins[0][0].size
    src = np.empty((7, n), dtype=NP[fam.sfx])
    rates = np.empty((10, n), dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_p3_liquid_ice_collisions_{fam.sfx}")
    fn.restype = None
    fn(C.byref(ice_params), C.byref(aps), C.byref(tps), C.byref(quad), C.c_uint32(flags), C.byref(th), C.c_int64(n),
       *[q for _, q in ins], src.ctypes.data_as(C.c_void_p), rates.ctypes.data_as(C.c_void_p), C.c_int32(nthreads))
    return src, rates


def p3_collision_probes(fam, ice_params, aps, tps, flags, L, N, F_rim, rho_rim, rho_a, T, log_lambda, D_ice, D_liq):
    """(compute_max_freeze_rate(…)(D_ice), compute_local_rime_density(…)(D_ice, D_liq)) for a P3State(L, N, F_rim, ρ_rim)."""
    fn = getattr(lib(), f"cmxo_p3_collision_probes_{fam.sfx}")
    fn.restype = None
    out = (fam.ft * 2)()
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32] + [fam.ft] * 9 + [C.c_void_p]
    fn(C.addressof(ice_params), C.addressof(aps), C.addressof(tps), flags, L, N, F_rim, rho_rim, rho_a, T, log_lambda, D_ice, D_liq,
       C.addressof(out))
    return out[0], out[1]


def microphysics_2m_p3_tendencies(fam, warm_rain, ice_params, tps, flags, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim,
                                  b_rim, log_lambda, inpc_log_shift=None, *, float32_gates=None, nthreads=1):
    """Oracle twin of cmx_microphysics_2m_p3_tendencies_* (BMT:898-1083): (out[8, n], scale[8, n]) with rows
    (dq_lcl, dn_lcl, dq_rai, dn_rai, dq_ice, dn_ice, dq_rim, db_rim)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log_lambda)]
    n = ins[0][0].size
    sh = _col(fam, inpc_log_shift) if inpc_log_shift is not None else (None, None)
    out = np.empty((8, n), dtype=NP[fam.sfx])
    scale = np.empty((8, n), dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_microphysics_2m_p3_tendencies_{fam.sfx}")
    fn.restype = None
    fn(C.byref(warm_rain), C.byref(ice_params), C.byref(tps), C.c_uint32(flags), C.byref(th), C.c_int64(n), *[q for _, q in ins], sh[1],
       out.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p), C.c_int32(nthreads))
    return out, scale


def liquid_freezing_rate(fam, rf, pdf, tps, q, rho, N, T, *, cloud, limited=True, float32_gates=None):
    """CMI_het.liquid_freezing_rate (Bigg) for the cloud (generalized gamma) or rain (exponential) PSD: (∂ₜn_frz, ∂ₜq_frz)."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    ins = [_col(fam, a) for a in (q, rho, N, T)]
    n = ins[0][0].size
    dn, dq = np.empty(n, dtype=NP[fam.sfx]), np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_liquid_freezing_rate_{fam.sfx}")
    fn.restype = None
    fn(C.byref(rf), C.byref(pdf) if cloud else None, None if cloud else C.byref(pdf), C.c_int32(int(limited)), C.byref(tps), C.byref(th),
       C.c_int64(n), *[p for _, p in ins], dn.ctypes.data_as(C.c_void_p), dq.ctypes.data_as(C.c_void_p))
    return dn, dq


def f23_deposition_rate(fam, ip, tps, m_nuc, tau_act, T, rho, q_tot, q_liq, q_ice, n_ice, shift=None):
    ins = [_col(fam, a) for a in (T, rho, q_tot, q_liq, q_ice, n_ice)]
    n = ins[0][0].size
    sh = _col(fam, shift) if shift is not None else (None, None)
    dn, dq = np.empty(n, dtype=NP[fam.sfx]), np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_f23_deposition_rate_{fam.sfx}")
    fn.restype = None
    fn(C.byref(ip), C.byref(tps), fam.ft(m_nuc), fam.ft(tau_act), C.c_int64(n), *[p for _, p in ins], sh[1],
       dn.ctypes.data_as(C.c_void_p), dq.ctypes.data_as(C.c_void_p))
    return dn, dq


def f23_immersion_limit_rate(fam, ip, tau, T, rho, n_active=None, shift=None):
    ins = [_col(fam, a) for a in (T, rho)]
    n = ins[0][0].size
    na = _col(fam, n_active) if n_active is not None else (None, None)
    sh = _col(fam, shift) if shift is not None else (None, None)
    dn = np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_f23_immersion_limit_rate_{fam.sfx}")
    fn.restype = None
    fn(C.byref(ip), fam.ft(tau), C.c_int64(n), *[p for _, p in ins], na[1], sh[1], dn.ctypes.data_as(C.c_void_p))
    return dn


def _scalar(fam, name, argtypes, *args):
    fn = getattr(lib(), f"{name}_{fam.sfx}")
    fn.restype = fam.ft
    fn.argtypes = argtypes
    return fn(*args)


def INP_concentration_mean(fam, ip, T):
    return _scalar(fam, "cmxo_INP_concentration_mean", [C.c_void_p, fam.ft], C.addressof(ip), T)


def INP_concentration_frequency(fam, ip, INPC, T):
    return _scalar(fam, "cmxo_INP_concentration_frequency", [C.c_void_p, fam.ft, fam.ft], C.addressof(ip), INPC, T)


def P3_deposition_N_i(fam, ip, T):
    return _scalar(fam, "cmxo_P3_deposition_N_i", [C.c_void_p, fam.ft], C.addressof(ip), T)


def P3_het_N_i(fam, ip, T, N_l, V_l, dt):
    return _scalar(fam, "cmxo_P3_het_N_i", [C.c_void_p] + [fam.ft] * 4, C.addressof(ip), T, N_l, V_l, dt)


def p3_het_ice_nucleation(fam, dust, tps, q_lcl, N_lcl, RH, T, rho):
    """Oracle twin of cmx_p3_het_ice_nucleation_*: (dNdt, dLdt)."""
    ins = [_col(fam, a) for a in (q_lcl, N_lcl, RH, T, rho)]
    n = ins[0][0].size
    dN, dL = np.empty(n, dtype=NP[fam.sfx]), np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_p3_het_ice_nucleation_{fam.sfx}")
    fn.restype = None
    fn(C.byref(dust), C.byref(tps), C.c_int64(n), *[p for _, p in ins], dN.ctypes.data_as(C.c_void_p), dL.ctypes.data_as(C.c_void_p))
    return dN, dL


def p3_closed_rain_probe(fam, ice_params, aps, tps, flags, L, N, F_rim, rho_rim, rho_a, log_lambda, L_r, N_r, D_ice):
    """closed_rain_inner_NM at one outer diameter: dict(N, M, Dstar, v_i, r_i, D_min, D_max, N0r, Dr_mean)."""
    fn = getattr(lib(), f"cmxo_p3_closed_rain_probe_{fam.sfx}")
    fn.restype = None
    out = (fam.ft * 9)()
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32] + [fam.ft] * 9 + [C.c_void_p]
    fn(C.addressof(ice_params), C.addressof(aps), C.addressof(tps), flags, L, N, F_rim, rho_rim, rho_a, log_lambda, L_r, N_r, D_ice, C.addressof(out))
    return dict(zip(("N", "M", "Dstar", "v_i", "r_i", "D_min", "D_max", "N0r", "Dr_mean"), list(out)))


def p3_crossover_probe(fam, chen, rho_a, v_target, D_min, D_max, maxiters=10):
    """P3.crossover_diameter: (D*, v_r(D*), v_r(D_min), v_r(D_max))."""
    fn = getattr(lib(), f"cmxo_p3_crossover_probe_{fam.sfx}")
    fn.restype = None
    fn.argtypes = [C.c_void_p] + [fam.ft] * 4 + [C.c_int, C.c_void_p]
    out = (fam.ft * 4)()
    fn(C.addressof(chen), rho_a, v_target, D_min, D_max, maxiters, C.addressof(out))
    return tuple(out)


def h2so4_solution(fam, prs, tps, x, T):
    """(H2SO4_soln_saturation_vapor_pressure, a_w_xT) over columns — src/Common.jl:188-246."""
    xs, Ts = _col(fam, x), _col(fam, T)
    n = xs[0].size
    p, a = np.empty(n, dtype=NP[fam.sfx]), np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_h2so4_solution_{fam.sfx}")
    fn.restype = None
    fn(C.byref(prs), C.byref(tps), C.c_int64(n), xs[1], Ts[1], p.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p))
    return p, a


def mohler2006_deposition(fam, dust, ip, S_i, T, dSi_dt, N_aer):
    """(dust_activated_number_fraction, MohlerDepositionRate) over columns — src/IceNucleation.jl:44-79 (NaN where the reference asserts)."""
    cols = [_col(fam, a) for a in (S_i, T, dSi_dt, N_aer)]
    n = cols[0][0].size
    f, r = np.empty(n, dtype=NP[fam.sfx]), np.empty(n, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_mohler2006_deposition_{fam.sfx}")
    fn.restype = None
    fn(C.byref(dust), C.byref(ip), C.c_int64(n), *[p for _, p in cols], f.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
    return f, r


def deposition_J(fam, dust, delta_a_w):
    d = _col(fam, delta_a_w)
    J = np.empty(d[0].size, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_deposition_J_{fam.sfx}")
    fn.restype = None
    fn(C.byref(dust), C.c_int64(d[0].size), d[1], J.ctypes.data_as(C.c_void_p))
    return J


def mp0m_tendencies(fam, p0m, q_lcl, q_icl, q_vap_sat=None):
    """Oracle twin of cmx_mp0m_tendencies_*: (dq_tot_dt, ∂dq_tot_dt/∂q_tot)."""
    a, ap = _col(fam, q_lcl)
    b, bp = _col(fam, q_icl)
    s, sp = _col(fam, q_vap_sat) if q_vap_sat is not None else (None, None)
    out = np.empty(a.size, dtype=NP[fam.sfx])
    der = np.empty(a.size, dtype=NP[fam.sfx])
    fn = getattr(lib(), f"cmxo_mp0m_tendencies_{fam.sfx}")
    fn.restype = None
    fn(C.byref(p0m), C.c_int64(a.size), ap, bp, sp, out.ctypes.data_as(C.c_void_p), der.ctypes.data_as(C.c_void_p))
    return out, der


# ---- size-distribution helpers (oracle/cmx_oracle_dist_impl.h) ---------------------------------------------------------------------------
def generalized_gamma(fam, nu, mu, B, Y=None, x=None):
    """Oracle twin of cmx_generalized_gamma_*: (quantile or None, cdf or None) — DT.generalized_gamma_quantile / _cdf."""
    Bc = _col(fam, B)
    n = Bc[0].size
    Yc = _col(fam, Y) if Y is not None else (None, None)
    xc = _col(fam, x) if x is not None else (None, None)
    qt = np.empty(n, dtype=NP[fam.sfx]) if Y is not None else None
    cdf = np.empty(n, dtype=NP[fam.sfx]) if x is not None else None
    fn = getattr(lib(), f"cmxo_generalized_gamma_{fam.sfx}")
    fn.restype = None
    fn.argtypes = [fam.ft, fam.ft, C.c_int64] + [C.c_void_p] * 5
    fn(nu, mu, n, Bc[1], Yc[1], xc[1], qt.ctypes.data if qt is not None else None, cdf.ctypes.data if cdf is not None else None)
    return qt, cdf


def exponential_distribution(fam, D_mean, Y=None, D=None):
    """Oracle twin of cmx_exponential_distribution_*: (quantile or None, cdf or None) — DT.exponential_quantile / _cdf."""
    Mc = _col(fam, D_mean)
    n = Mc[0].size
    Yc = _col(fam, Y) if Y is not None else (None, None)
    Dc = _col(fam, D) if D is not None else (None, None)
    qt = np.empty(n, dtype=NP[fam.sfx]) if Y is not None else None
    cdf = np.empty(n, dtype=NP[fam.sfx]) if D is not None else None
    fn = getattr(lib(), f"cmxo_exponential_distribution_{fam.sfx}")
    fn.restype = None
    fn.argtypes = [C.c_int64] + [C.c_void_p] * 5
    fn(n, Mc[1], Yc[1], Dc[1], qt.ctypes.data if qt is not None else None, cdf.ctypes.data if cdf is not None else None)
    return qt, cdf


def sb2006_size_distribution(fam, pdf_c, pdf_r, q, rho, N, D=None, *, cloud=False, limited=True, p=None, bounds=True, float32_gates=None):
    """Oracle twin of cmx_sb2006_size_distribution_*: dict(n_D, D_min, D_max) — CM2.size_distribution_value / get_size_distribution_bounds."""
    if float32_gates is None:
        float32_gates = fam.sfx == "f32"
    th = thresholds(fam, float32_gates)
    if p is None:
        p = float(np.finfo(np.float32 if float32_gates else np.float64).eps)
    qc, rc, Nc = _col(fam, q), _col(fam, rho), _col(fam, N)
    n = qc[0].size
    Dc = _col(fam, D) if D is not None else (None, None)
    mk = lambda on: np.empty(n, dtype=NP[fam.sfx]) if on else None  # noqa: E731
    n_D, D_min, D_max = mk(D is not None), mk(bounds), mk(bounds)
    ptr = lambda a: a.ctypes.data if a is not None else None  # noqa: E731
    fn = getattr(lib(), f"cmxo_sb2006_size_distribution_{fam.sfx}")
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, fam.ft, C.c_void_p, C.c_int64] + [C.c_void_p] * 7
    fn(C.addressof(pdf_c) if pdf_c is not None else None, C.addressof(pdf_r) if pdf_r is not None else None, int(cloud), int(limited), p, C.addressof(th), n,
       qc[1], rc[1], Nc[1], Dc[1], ptr(n_D), ptr(D_min), ptr(D_max))
    return dict(n_D=n_D, D_min=D_min, D_max=D_max)


# ---- cloud diagnostics (oracle/cmx_oracle_diag_impl.h) -----------------------------------------------------------------------------------
def cloud_diagnostics(fam, rho, q_lcl, q_rai=None, N_lcl=None, N_rai=None, *, rain=None, pdf_c=None, pdf_r=None, rho_w=1000.0, limited=True,
                      float32_gates=False, want=("Z_1m", "Z_2m", "reff_2m", "reff_lh97")):
    """Oracle twin of cmx_cloud_diagnostics_*: dict of the requested columns — CMD.radar_reflectivity_1M / _2M, effective_radius_2M,
    effective_radius_Liu_Hallet_97 (src/CloudDiagnostics.jl)."""
    r, rp = _col(fam, rho)
    cols = {k: (_col(fam, v) if v is not None else (None, None)) for k, v in (("q_lcl", q_lcl), ("q_rai", q_rai), ("N_lcl", N_lcl), ("N_rai", N_rai))}
    out = {k: (np.empty(r.size, dtype=NP[fam.sfx]) if k in want else None) for k in ("Z_1m", "Z_2m", "reff_2m", "reff_lh97")}
    th = thresholds(fam, float32_gates)
    fn = getattr(lib(), f"cmxo_cloud_diagnostics_{fam.sfx}")
    fn.restype = None
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
    fn(C.byref(rain) if rain is not None else None, C.byref(pdf_c) if pdf_c is not None else None, C.byref(pdf_r) if pdf_r is not None else None,
       fam.ft(rho_w), C.c_int(int(limited)), C.byref(th), C.c_int64(r.size), rp, cols["q_lcl"][1], cols["q_rai"][1], cols["N_lcl"][1], cols["N_rai"][1],
       ptr(out["Z_1m"]), ptr(out["Z_2m"]), ptr(out["reff_2m"]), ptr(out["reff_lh97"]))
    return {k: v for k, v in out.items() if v is not None}
