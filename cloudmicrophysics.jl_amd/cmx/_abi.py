"""ctypes mirror of include/cmx.h (structs, flags, column ids).

One factory stamps the Float32 and the Float64 struct family, exactly like the
CMX_DECLARE_PARAM_STRUCTS macro of the header.  Field order is the declaration
order of the reference's Julia structs (cited in include/cmx.h).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

# ---- flags (include/cmx.h) ---------------------------------------------------
CMX_SB2006_LIMITED = 1 << 0
CMX_VEL_SB2006 = 1 << 1
CMX_VEL_CHEN2022 = 1 << 2

CMX_ICENUC_HOM_LINEAR = 1 << 0
CMX_ICENUC_ERR_SLOTS = 64
CMX_ICENUC_ERR_WORDS = 1024

CMX_OK = 0
CMX_ERR_BAD_ARG = -1
CMX_ERR_HIP = -2
CMX_ERR_UNSUPPORTED = -3

# cmx_sb2006_process_column
SB2006_PROCESS_COLUMNS = (
    "acnv_dq_lcl_dt", "acnv_dN_lcl_dt", "acnv_dq_rai_dt", "acnv_dN_rai_dt",
    "lcl_self_collection",
    "accr_dq_lcl_dt", "accr_dN_lcl_dt", "accr_dq_rai_dt",
    "rain_self_collection", "rain_breakup",
    "rain_vel_n", "rain_vel_m",
    "evap_dN_rai_dt", "evap_dq_rai_dt",
    "numadj_rai", "numadj_lcl",
    "condevap",
)
CMX_SB2006_NPROC = len(SB2006_PROCESS_COLUMNS)


def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


def _family(ft, sfx):
    s = lambda *names: [(n, ft) for n in names]  # noqa: E731
    ns = SimpleNamespace(ft=ft, sfx=sfx)
    ns.cloud_pdf_sb2006 = _struct(f"cmx_cloud_pdf_sb2006_{sfx}",
                                  s("nu_c", "mu_c", "xc_min", "xc_max", "rho_w", "loggamma_z1", "loggamma_z2"))
    ns.rain_pdf_sb2006 = _struct(f"cmx_rain_pdf_sb2006_{sfx}",
                                 s("nu_r", "mu_r", "xr_min", "xr_max", "N0_min", "N0_max",
                                   "lambda_min", "lambda_max", "rho_w", "rho_0"))
    ns.acnv_sb2006 = _struct(f"cmx_acnv_sb2006_{sfx}", s("kcc", "x_star", "rho_0", "A", "a", "b"))
    ns.accr_sb2006 = _struct(f"cmx_accr_sb2006_{sfx}", s("kcr", "tau_0", "rho_0", "c"))
    ns.selfcol_sb2006 = _struct(f"cmx_selfcol_sb2006_{sfx}", s("krr", "kappa_rr", "d"))
    ns.breakup_sb2006 = _struct(f"cmx_breakup_sb2006_{sfx}", s("Deq", "Dr_th", "kbr", "kappa_br"))
    ns.evap_sb2006 = _struct(f"cmx_evap_sb2006_{sfx}",
                             s("av", "bv", "alpha", "beta", "rho_0", "a_vent_1", "b_vent_1",
                               "a_vent_0_coeff", "b_vent_0_coeff", "beta_vent_0"))
    ns.numadj_horn2012 = _struct(f"cmx_numadj_horn2012_{sfx}", s("tau"))
    ns.sb2006 = _struct(f"cmx_sb2006_{sfx}", [
        ("pdf_c", ns.cloud_pdf_sb2006), ("pdf_r", ns.rain_pdf_sb2006), ("acnv", ns.acnv_sb2006),
        ("accr", ns.accr_sb2006), ("self", ns.selfcol_sb2006), ("brek", ns.breakup_sb2006),
        ("evap", ns.evap_sb2006), ("numadj", ns.numadj_horn2012)])
    ns.air_properties = _struct(f"cmx_air_properties_{sfx}", s("K_therm", "D_vapor", "nu_air"))
    ns.warm_rain_2m = _struct(f"cmx_warm_rain_2m_{sfx}", [
        ("seifert_beheng", ns.sb2006), ("air_properties", ns.air_properties),
        ("condevap_tau_relax", ft), ("subdep_tau_relax", ft)])
    ns.thermo = _struct(f"cmx_thermo_{sfx}",
                        s("R_v", "R_d", "cp_d", "cp_v", "cp_l", "cp_i", "LH_v0", "LH_s0", "T_0",
                          "T_triple", "press_triple", "T_freeze"))
    ns.sb2006_vel = _struct(f"cmx_sb2006_vel_{sfx}", s("rho_0", "aR", "bR", "cR", "rho_w", "nu_air", "grav"))
    ns.chen2022_rain_vel = _struct(f"cmx_chen2022_rain_vel_{sfx}", [
        ("rho_0", ft), ("a", ft * 3), ("a3_pow", ft), ("b", ft * 3), ("b_rho", ft), ("c", ft * 3)])
    ns.rain_vel = _struct(f"cmx_rain_vel_{sfx}", [("sb2006", ns.sb2006_vel), ("chen2022", ns.chen2022_rain_vel)])
    ns.koop2000 = _struct(f"cmx_koop2000_{sfx}",
                          s("delta_a_w_min", "delta_a_w_max", "c1", "c2", "c3", "c4", "linear_c1", "linear_c2"))
    ns.abifm_dust = _struct(f"cmx_abifm_dust_{sfx}", s("ABIFM_m", "ABIFM_c"))
    return ns


F32 = _family(C.c_float, "f32")
F64 = _family(C.c_double, "f64")


def family(FT):
    """Struct family for a float type given as 'f32'/'f64', numpy dtype or torch dtype."""
    name = str(FT)
    if name in ("f32", "float32", "torch.float32", "<class 'numpy.float32'>") or FT is np.float32:
        return F32
    if name in ("f64", "float64", "torch.float64", "<class 'numpy.float64'>") or FT is np.float64 or FT is float:
        return F64
    raise TypeError(f"unsupported float type {FT!r}")
