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

CMX_ARG_MAX_MODES = 8
CMX_2M_KK2000, CMX_2M_B1994, CMX_2M_TC1980, CMX_2M_LD2004 = 0, 1, 2, 3
CMX_2M_SMOOTH_TRANSITION = 1 << 8
CMX_P3_INPUT_IS_STATE = 1 << 0
CMX_P3_SLOPE_CONSTANT = 1 << 1
CMX_P3_NO_ASPECT_RATIO = 1 << 2
CMX_P3_RAIN_PDF_LIMITED = 1 << 3
CMX_FREEZE_CLOUD_PSD = 1 << 4
CMX_QUAD_MAX = 128
CMX_PSD_CLOUD = 1 << 5
CMX_COLUMN_SUMS_MAX_COLS = 16
CMX_COLUMN_SUMS_PARTIALS = 1024

CMX_ICENUC_HOM_LINEAR = 1 << 0
CMX_ICENUC_ERR_SLOTS = 64
CMX_ICENUC_ERR_WORDS = 1024

# Microphysics1MOptions bits (include/cmx.h §5)
CMX_1M_CLOUD_LIQUID_FORMATION = 1 << 0
CMX_1M_CLOUD_ICE_FORMATION_CONST = 1 << 1
CMX_1M_CLOUD_ICE_FORMATION_TDEP = 1 << 2
CMX_1M_CLOUD_ICE_MELT = 1 << 3
CMX_1M_RAIN_ACNV_KESSLER = 1 << 4
CMX_1M_RAIN_ACNV_PRESCRIBED_ND = 1 << 5
CMX_1M_SNOW_ACNV_NO_SUPERSAT = 1 << 6
CMX_1M_SNOW_ACNV_WITH_SUPERSAT = 1 << 7
CMX_1M_RAIN_EVAPORATION = 1 << 8
CMX_1M_SNOW_SUBLIMATION_ONLY = 1 << 9
CMX_1M_SNOW_DEP_AND_SUBL = 1 << 10
CMX_1M_SNOW_MELT = 1 << 11
CMX_1M_ACCR_LCL_RAI = 1 << 12
CMX_1M_ACCR_LCL_SNO = 1 << 13
CMX_1M_ACCR_ICL_RAI = 1 << 14
CMX_1M_ACCR_ICL_SNO = 1 << 15
CMX_1M_ACCR_RAI_SNO = 1 << 16
CMX_1M_DEFAULT_OPTIONS = (CMX_1M_CLOUD_LIQUID_FORMATION | CMX_1M_CLOUD_ICE_FORMATION_CONST | CMX_1M_CLOUD_ICE_MELT |
                          CMX_1M_RAIN_ACNV_KESSLER | CMX_1M_SNOW_ACNV_NO_SUPERSAT | CMX_1M_RAIN_EVAPORATION |
                          CMX_1M_SNOW_DEP_AND_SUBL | CMX_1M_SNOW_MELT | CMX_1M_ACCR_LCL_RAI | CMX_1M_ACCR_LCL_SNO |
                          CMX_1M_ACCR_ICL_RAI | CMX_1M_ACCR_ICL_SNO | CMX_1M_ACCR_RAI_SNO)

# cmx_mp1m_source_column
MP1M_SOURCE_COLUMNS = (
    "S_phase_change_vap_lcl", "S_phase_change_vap_icl", "S_acnv_lcl_rai", "S_acnv_icl_sno",
    "S_accr_lcl_rai", "S_accr_lcl_sno_cold", "S_accr_lcl_sno_warm", "S_accr_melt_lcl_sno",
    "S_accr_icl_rai", "S_accr_freeze_icl_rai", "S_accr_icl_sno",
    "S_accr_rai_sno_cold", "S_accr_rai_sno_warm", "S_accr_melt_rai_sno",
    "S_phase_change_vap_rai", "S_phase_change_vap_sno", "S_melt_icl_lcl", "S_melt_sno_rai",
)
CMX_MP1M_NSRC = len(MP1M_SOURCE_COLUMNS)

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
    "devap_dN_rai", "devap_dq_rai",
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
    ns.kk2000 = _struct(f"cmx_kk2000_{sfx}", s("acnv_A", "acnv_a", "acnv_b", "acnv_c", "accr_A", "accr_a", "accr_b"))
    ns.b1994 = _struct(f"cmx_b1994_{sfx}", s("acnv_C", "acnv_a", "acnv_b", "acnv_c", "acnv_N_0", "acnv_d_low", "acnv_d_high", "acnv_k", "accr_A"))
    ns.tc1980 = _struct(f"cmx_tc1980_{sfx}", s("acnv_a", "acnv_b", "acnv_D", "acnv_r_0", "acnv_me_liq", "acnv_m0_liq_coeff", "acnv_k", "accr_A"))
    ns.ld2004 = _struct(f"cmx_ld2004_{sfx}", s("R_6C_0", "E_0", "rho_w", "k"))
    ns.bulk_2m_schemes = _struct(f"cmx_bulk_2m_schemes_{sfx}", [("kk2000", ns.kk2000), ("b1994", ns.b1994), ("tc1980", ns.tc1980),
                                                                  ("ld2004", ns.ld2004)])
    ns.stokes_vel = _struct(f"cmx_stokes_vel_{sfx}", s("rho_w", "nu_air", "grav"))
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
                          "T_triple", "press_triple", "T_freeze", "cv_l"))
    ns.sb2006_vel = _struct(f"cmx_sb2006_vel_{sfx}", s("rho_0", "aR", "bR", "cR", "rho_w", "nu_air", "grav"))
    ns.chen2022_rain_vel = _struct(f"cmx_chen2022_rain_vel_{sfx}", [
        ("rho_0", ft), ("a", ft * 3), ("a3_pow", ft), ("b", ft * 3), ("b_rho", ft), ("c", ft * 3)])
    ns.rain_vel = _struct(f"cmx_rain_vel_{sfx}", [("sb2006", ns.sb2006_vel), ("chen2022", ns.chen2022_rain_vel)])
    ns.koop2000 = _struct(f"cmx_koop2000_{sfx}",
                          s("delta_a_w_min", "delta_a_w_max", "c1", "c2", "c3", "c4", "linear_c1", "linear_c2"))
    ns.abifm_dust = _struct(f"cmx_abifm_dust_{sfx}", s("ABIFM_m", "ABIFM_c"))
    # ---- 1-moment scheme
    ns.particle_mass = _struct(f"cmx_particle_mass_{sfx}", s("r0", "m0", "me", "delta_m", "chi_m", "gamma_coeff"))
    ns.particle_area = _struct(f"cmx_particle_area_{sfx}", s("a0", "ae", "delta_a", "chi_a"))
    ns.ventilation = _struct(f"cmx_ventilation_{sfx}", s("a", "b"))
    ns.acnv_1m = _struct(f"cmx_acnv_1m_{sfx}", s("tau", "q_threshold", "k"))
    ns.var_timescale_acnv = _struct(f"cmx_var_timescale_acnv_{sfx}", s("tau", "alpha", "Nc"))
    ns.cloud_liquid = _struct(f"cmx_cloud_liquid_{sfx}", s("rho_w", "r_eff", "N_0"))
    ns.cloud_ice = _struct(f"cmx_cloud_ice_{sfx}", [("n0", ft), ("mass", ns.particle_mass)] + s("rho_i", "r_eff", "N_0"))
    ns.rain = _struct(f"cmx_rain_{sfx}", [("n0", ft), ("mass", ns.particle_mass), ("area", ns.particle_area),
                                          ("vent", ns.ventilation)])
    ns.snow = _struct(f"cmx_snow_{sfx}", s("mu", "nu") + [("mass", ns.particle_mass), ("area", ns.particle_area),
                                                         ("vent", ns.ventilation)] +
                      s("phi", "kappa", "rho_i", "gamma_aspect_oblate", "gamma_aspect_prolate"))
    ns.blk1m_vel_rain = _struct(f"cmx_blk1m_vel_rain_{sfx}",
                                s("r0", "ve", "delta_v", "chi_v", "rho_w", "C_drag", "grav", "gamma_vent", "gamma_term",
                                  "gamma_accr", "gamma_accr_rain_sink"))
    ns.blk1m_vel_snow = _struct(f"cmx_blk1m_vel_snow_{sfx}",
                                s("r0", "ve", "delta_v", "chi_v", "v0", "gamma_vent", "gamma_term", "gamma_accr"))
    ns.frostenberg2023 = _struct(f"cmx_frostenberg2023_{sfx}", s("sigma", "a", "b", "T_freeze", "log_a"))
    ns.process_params_1m = _struct(f"cmx_process_params_1m_{sfx}", s(
        "cloud_liquid_formation_tau_relax", "cloud_ice_formation_tau_relax") + [
        ("cloud_ice_formation_frostenberg", ns.frostenberg2023), ("rain_autoconversion", ns.acnv_1m), ("rain_autoconversion_nd", ns.var_timescale_acnv),
        ("snow_autoconversion", ns.acnv_1m)] + s(
        "r_ice_snow", "e_lcl_rai", "e_lcl_sno", "e_icl_rai", "e_icl_sno", "e_rai_sno", "coeff_disp"))
    ns.microphysics_1m = _struct(f"cmx_microphysics_1m_{sfx}", [
        ("process_params", ns.process_params_1m), ("cloud_liquid", ns.cloud_liquid), ("cloud_ice", ns.cloud_ice),
        ("rain", ns.rain), ("snow", ns.snow), ("air_properties", ns.air_properties),
        ("vel_rain", ns.blk1m_vel_rain), ("vel_snow", ns.blk1m_vel_snow)])
    # ---- ARG2000 aerosol activation
    ns.aerosol_activation_params = _struct(f"cmx_aerosol_activation_params_{sfx}",
                                           s("M_w", "R", "rho_w", "rho_i", "sigma", "g", "f1", "f2", "g1", "g2", "p1", "p2"))
    ns.aerosol_mode = _struct(f"cmx_aerosol_mode_{sfx}", s("r_dry", "stdev", "N", "hygroscopicity", "molar_mass_mix"))
    ns.aerosol_distribution = _struct(f"cmx_aerosol_distribution_{sfx}", [
        ("n_modes", C.c_int32), ("pad_", C.c_int32), ("modes", ns.aerosol_mode * CMX_ARG_MAX_MODES)])
    # ---- P3
    ns.p3_params = _struct(f"cmx_p3_params_{sfx}",
                           s("alpha_va", "beta_va", "gamma", "sigma", "slope_a", "slope_b", "slope_c", "mu_max", "mu_const",
                             "rho_i", "rho_l", "tau_wet", "T_freeze"))
    ns.chen2022_small_ice_vel = _struct(f"cmx_chen2022_small_ice_vel_{sfx}", [
        ("A", ft * 3), ("B", ft * 3), ("C", ft * 4), ("E", ft * 3), ("F", ft * 3), ("G", ft * 3), ("cutoff", ft)])
    ns.chen2022_large_ice_vel = _struct(f"cmx_chen2022_large_ice_vel_{sfx}", [
        ("A", ft * 3), ("B", ft * 3), ("C", ft * 3), ("E", ft * 3), ("F", ft * 3), ("G", ft * 3), ("H", ft * 3), ("cutoff", ft)])
    ns.chen2022_ice_vel = _struct(f"cmx_chen2022_ice_vel_{sfx}", [
        ("small_ice", ns.chen2022_small_ice_vel), ("large_ice", ns.chen2022_large_ice_vel)])
    ns.quadrature = _struct(f"cmx_quadrature_{sfx}", [
        ("n", C.c_int32), ("reserved", C.c_int32), ("node", ft * CMX_QUAD_MAX), ("weight", ft * CMX_QUAD_MAX)])
    # ---- P3 liquid–ice collisions, immersion / deposition nucleation, P3IceParams
    ns.parameters_0m = _struct(f"cmx_parameters_0m_{sfx}", s("tau_precip", "qc_0", "S_0"))
    ns.local_rime_density = _struct(f"cmx_local_rime_density_{sfx}", s("a", "b", "c", "rho_ice"))
    ns.rain_freezing = _struct(f"cmx_rain_freezing_{sfx}", s("het_a", "het_B"))
    ns.mohler2006 = _struct(f"cmx_mohler2006_{sfx}", s("S_i_max", "T_thr"))
    ns.mohler_dust = _struct(f"cmx_mohler_dust_{sfx}", s("S0_warm", "S0_cold", "a_warm", "a_cold"))
    ns.deposition_dust = _struct(f"cmx_deposition_dust_{sfx}", s("deposition_m", "deposition_c"))
    ns.h2so4_solution_params = _struct(f"cmx_h2so4_solution_params_{sfx}", s("T_max", "T_min", "w_2", "c1", "c2", "c3", "c4", "c5", "c6", "c7"))
    ns.morrison_milbrandt2014 = _struct(f"cmx_morrison_milbrandt2014_{sfx}", s("T_dep_thres", "c1", "c2", "T0", "het_a", "het_B"))
    ns.p3_ice_params = _struct(f"cmx_p3_ice_params_{sfx}", [
        ("scheme", ns.p3_params), ("vent", ns.ventilation), ("rho_rim_local", ns.local_rime_density),
        ("vel_rain", ns.chen2022_rain_vel), ("vel_ice", ns.chen2022_ice_vel), ("cloud_pdf", ns.cloud_pdf_sb2006),
        ("rain_pdf", ns.rain_pdf_sb2006), ("ice_nucleation", ns.frostenberg2023), ("rain_freezing", ns.rain_freezing),
        ("tau_act", ft), ("quad", ns.quadrature)])
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
