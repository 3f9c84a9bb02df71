"""Parameter structs — host-side mirror of `CloudMicrophysics.Parameters` (CMP).

The reference builds every parameter struct from a ClimaParams TOML dictionary
through a name map (e.g. src/parameters/Microphysics2M.jl:337-348).  ClimaParams
is not vendored in the reference tree, so the default VALUES below were recovered
from the reference's docs tables and pinned by its known-answer tests (SURVEY.md
§8c lists value → source → KAT for every entry).  Keys are the ClimaParams names
the reference's name maps use, so an override file of the reference
(src/parameters/toml/SB2006_limiters.toml) is just a dict update here.

Constructors keep the reference's names and keyword (`is_limited`) and return
ctypes structs laid out like the Julia structs (include/cmx.h), ready to be
passed by pointer through the C ABI.
"""
from __future__ import annotations

import math
from typing import Mapping, Optional

from . import _abi

# ---------------------------------------------------------------------------
# ClimaParams defaults (name → value), CloudMicrophysics + Thermodynamics subset
# ---------------------------------------------------------------------------
DEFAULT_PARAMETERS = {
    # Thermodynamics.jl parameters (accessors: src/ThermodynamicsInterface.jl:9-25)
    "gas_constant_vapor": 461.5,               # R_v  (= R/M_w, value pinned by test/gpu_tests.jl:909)
    "gas_constant_dry_air": 287.0,             # R_d
    "isobaric_specific_heat_dry_air": 1004.5,  # cp_d = R_d / kappa_d, kappa_d = 2/7
    "isobaric_specific_heat_vapor": 1859.0,
    "isobaric_specific_heat_liquid": 4181.0,
    "isobaric_specific_heat_ice": 2070.0,      # pinned by test/gpu_tests.jl:924
    "latent_heat_vaporization_at_reference": 2.5008e6,
    "latent_heat_sublimation_at_reference": 2.8344e6,
    "thermodynamics_temperature_reference": 273.16,
    "temperature_triple_point": 273.16,
    "pressure_triple_point": 611.657,
    "temperature_water_freeze": 273.15,
    # air properties (src/parameters/AirProperties.jl:24-28)
    "thermal_conductivity_of_air": 0.024,
    "diffusivity_of_water_vapor": 2.26e-5,
    "kinematic_viscosity_of_air": 1.6e-5,
    "density_liquid_water": 1000.0,
    "gravitational_acceleration": 9.81,
    # relaxation timescales (src/parameters/Microphysics2M.jl:676-689)
    "condensation_evaporation_timescale": 10.0,
    "sublimation_deposition_timescale": 10.0,
    # SB2006 (src/parameters/Microphysics2M.jl name maps; docs/src/Microphysics2M.md tables)
    "SB2006_cloud_gamma_distribution_coeff_nu": 1.0,
    "SB2006_cloud_gamma_distribution_coeff_mu": 1.0,
    "SB2006_cloud_droplets_min_mass": 4.2e-15,
    "SB2006_raindrops_min_mass": 2.6e-10,      # x* = xr_min = xc_max (one key, three fields)
    "SB2006_raindrops_max_mass": 5e-6,
    "SB2006_rain_distribution_coeff_nu": -2.0 / 3.0,
    "SB2006_rain_distribution_coeff_mu": 1.0 / 3.0,
    "SB2006_raindrops_size_distribution_coeff_N0_min": 2.5e5,
    "SB2006_raindrops_size_distribution_coeff_N0_max": 2e7,
    "SB2006_raindrops_size_distribution_coeff_lambda_min": 1e3,
    "SB2006_raindrops_size_distribution_coeff_lambda_max": 1e4,
    "SB2006_reference_air_density": 1.225,
    "SB2006_collection_kernel_coeff_kcc": 4.44e9,
    "SB2006_autoconversion_correcting_function_coeff_A": 400.0,
    "SB2006_autoconversion_correcting_function_coeff_a": 0.7,
    "SB2006_autoconversion_correcting_function_coeff_b": 3.0,
    "SB2006_collection_kernel_coeff_kcr": 5.25,
    "SB2006_accretion_correcting_function_coeff_tau0": 5e-5,
    "SB2006_accretion_correcting_function_coeff_c": 4.0,
    "SB2006_collection_kernel_coeff_krr": 7.12,
    "SB2006_collection_kernel_coeff_kapparr": 60.7,
    "SB2006_raindrops_self-collection_coeff_d": -5.0,
    "SB2006_raindrops_equilibrium_mean_diameter": 0.9e-3,
    "SB2006_raindrops_breakup_mean_diameter_threshold": 0.35e-3,
    "SB2006_raindrops_breakup_coeff_kbr": 1000.0,
    "SB2006_raindrops_breakup_coeff_kappabr": 2300.0,
    "SB2006_ventilation_factor_coeff_av": 0.78,
    "SB2006_ventilation_factor_coeff_bv": 0.308,
    "SB2006_rain_evaporation_coeff_alpha": 159.0,
    "SB2006_rain_evaporation_coeff_beta": 0.266,
    "Horn2012_number_concentration_adjustment_timescale": 100.0,
    "SB2006_raindrops_terminal_velocity_coeff_aR": 9.65,
    "SB2006_raindrops_terminal_velocity_coeff_bR": 10.3,
    "SB2006_raindrops_terminal_velocity_coeff_cR": 600.0,
    # Chen et al. 2022, Table B1 (rain)
    "Chen2022_table_B1_q_coeff": 0.115231,
    "Chen2022_table_B1_ai": (0.044612, -0.263166, 4.7178),
    "Chen2022_table_B1_a3_pow_coeff": -0.47335,
    "Chen2022_table_B1_bi": (2.2955, 2.2955, 1.1451),
    "Chen2022_table_B1_b_rho_coeff": 0.038465,
    "Chen2022_table_B1_ci": (0.0, 0.184325, 0.184325),
    # Koop et al. 2000 homogeneous freezing (src/parameters/IceNucleation.jl:57-68; docs/src/IceNucleation.md:214-217;
    # cubic pinned by test/gpu_tests.jl:1067; window: 0.25 and 0.35 throw, test/homogeneous_ice_nucleation_tests.jl:22-23)
    "Koop2000_min_delta_aw": 0.26,
    "Koop2000_max_delta_aw": 0.34,
    "Koop2000_J_hom_coeff1": -906.7,
    "Koop2000_J_hom_coeff2": 8502.0,
    "Koop2000_J_hom_coeff3": 26924.0,
    "Koop2000_J_hom_coeff4": 29180.0,
    # linear fit: in-tree values papers/ice_nucleation_2024/calibration_setup.jl:149, consistent with the KAT
    # test/gpu_tests.jl:1069 (one KAT, two coefficients)
    "Linear_J_hom_coeff1": -68.553283,
    "Linear_J_hom_coeff2": 255.927125,
    # ABIFM, Knopf & Alpert 2013 (src/parameters/AerosolKaolinite.jl:28-29, AerosolIllite.jl:27-28);
    # pinned by test/gpu_tests.jl:987-997
    "KnopfAlpert2013_J_ABIFM_m_Kaolinite": 54.58834,
    "KnopfAlpert2013_J_ABIFM_c_Kaolinite": -10.54758,
    "KnopfAlpert2013_J_ABIFM_m_Illite": 54.48075,
    "KnopfAlpert2013_J_ABIFM_c_Illite": -10.66873,
}

# the reference's override file src/parameters/toml/SB2006_limiters.toml (used by its CPU tests,
# test/microphysics2M_tests.jl:26-31)
SB2006_LIMITERS_OVERRIDE = {
    "SB2006_raindrops_min_mass": 6.54e-11,
    "SB2006_raindrops_size_distribution_coeff_N0_min": 3.5e5,
    "SB2006_raindrops_size_distribution_coeff_N0_max": 2e11,
    "SB2006_raindrops_size_distribution_coeff_lambda_max": 4e4,
}


class ParamDict(dict):
    """`CP.create_toml_dict(FT; override_file)` analogue: defaults + overrides, typed by FT."""

    def __init__(self, FT, override: Optional[Mapping] = None):
        super().__init__(DEFAULT_PARAMETERS)
        if override:
            unknown = set(override) - set(DEFAULT_PARAMETERS)
            if unknown:
                raise KeyError(f"unknown parameter name(s): {sorted(unknown)}")
            self.update(override)
        self.fam = _abi.family(FT)
        self.FT = self.fam.sfx


def create_toml_dict(FT, override: Optional[Mapping] = None) -> ParamDict:
    return ParamDict(FT, override)


def _td(FT_or_td) -> ParamDict:
    return FT_or_td if isinstance(FT_or_td, ParamDict) else ParamDict(FT_or_td)


def _rounded(fam, x):
    """Round a Python float to FT the way `FT(x)` does (so host-derived constants follow FT arithmetic)."""
    return fam.ft(x).value


# ---------------------------------------------------------------------------
# constructors (names = the reference's)
# ---------------------------------------------------------------------------
def ThermodynamicsParameters(FT):
    """TD.Parameters.ThermodynamicsParameters(FT), flattened (include/cmx.h: cmx_thermo)."""
    td = _td(FT)
    return td.fam.thermo(
        R_v=td["gas_constant_vapor"], R_d=td["gas_constant_dry_air"],
        cp_d=td["isobaric_specific_heat_dry_air"], cp_v=td["isobaric_specific_heat_vapor"],
        cp_l=td["isobaric_specific_heat_liquid"], cp_i=td["isobaric_specific_heat_ice"],
        LH_v0=td["latent_heat_vaporization_at_reference"], LH_s0=td["latent_heat_sublimation_at_reference"],
        T_0=td["thermodynamics_temperature_reference"], T_triple=td["temperature_triple_point"],
        press_triple=td["pressure_triple_point"], T_freeze=td["temperature_water_freeze"])


def AirProperties(FT):
    """CMP.AirProperties — src/parameters/AirProperties.jl:11-30."""
    td = _td(FT)
    return td.fam.air_properties(K_therm=td["thermal_conductivity_of_air"],
                                 D_vapor=td["diffusivity_of_water_vapor"],
                                 nu_air=td["kinematic_viscosity_of_air"])


def CloudParticlePDF_SB2006(FT):
    """src/parameters/Microphysics2M.jl:401-433 (loggamma_z1/z2 derived host-side, :428-432)."""
    td = _td(FT)
    nu, mu = td["SB2006_cloud_gamma_distribution_coeff_nu"], td["SB2006_cloud_gamma_distribution_coeff_mu"]
    return td.fam.cloud_pdf_sb2006(
        nu_c=nu, mu_c=mu, xc_min=td["SB2006_cloud_droplets_min_mass"], xc_max=td["SB2006_raindrops_min_mass"],
        rho_w=td["density_liquid_water"], loggamma_z1=math.lgamma((nu + 1) / mu),
        loggamma_z2=math.lgamma((nu + 2) / mu))


def RainParticlePDF_SB2006(FT, is_limited: bool = True):
    """src/parameters/Microphysics2M.jl:314-375.  One C layout for both variants; the not-limited
    one leaves the N0 / λ limiter fields at zero (never read: flag CMX_SB2006_LIMITED is clear)."""
    td = _td(FT)
    kw = dict(nu_r=td["SB2006_rain_distribution_coeff_nu"], mu_r=td["SB2006_rain_distribution_coeff_mu"],
              xr_min=td["SB2006_raindrops_min_mass"], xr_max=td["SB2006_raindrops_max_mass"],
              rho_w=td["density_liquid_water"], rho_0=td["SB2006_reference_air_density"])
    if is_limited:
        kw.update(N0_min=td["SB2006_raindrops_size_distribution_coeff_N0_min"],
                  N0_max=td["SB2006_raindrops_size_distribution_coeff_N0_max"],
                  lambda_min=td["SB2006_raindrops_size_distribution_coeff_lambda_min"],
                  lambda_max=td["SB2006_raindrops_size_distribution_coeff_lambda_max"])
    return td.fam.rain_pdf_sb2006(**kw)


def EvaporationSB2006(FT):
    """src/parameters/Microphysics2M.jl:567-607 — the five derived fields follow :599-606 in FT arithmetic."""
    td = _td(FT)
    fam = td.fam
    r = lambda x: _rounded(fam, x)  # noqa: E731
    av, bv = r(td["SB2006_ventilation_factor_coeff_av"]), r(td["SB2006_ventilation_factor_coeff_bv"])
    beta = r(td["SB2006_rain_evaporation_coeff_beta"])
    return fam.evap_sb2006(
        av=av, bv=bv, alpha=td["SB2006_rain_evaporation_coeff_alpha"], beta=beta,
        rho_0=td["SB2006_reference_air_density"],
        a_vent_1=av / 6.0 ** (1.0 / 3.0),
        b_vent_1=bv * math.gamma(2.5 + 1.5 * beta) / 6.0 ** (beta / 2 + 0.5),
        a_vent_0_coeff=av * 36.0 ** (1.0 / 3.0),
        b_vent_0_coeff=bv / 6.0 ** (beta / 2 - 0.5),
        beta_vent_0=-0.5 + 1.5 * beta)


def SB2006(FT, is_limited: bool = True):
    """CMP.SB2006(toml_dict; is_limited) — src/parameters/Microphysics2M.jl:642-671."""
    td = _td(FT)
    fam = td.fam
    sb = fam.sb2006()
    sb.pdf_c = CloudParticlePDF_SB2006(td)
    sb.pdf_r = RainParticlePDF_SB2006(td, is_limited)
    sb.acnv = fam.acnv_sb2006(
        kcc=td["SB2006_collection_kernel_coeff_kcc"], x_star=td["SB2006_raindrops_min_mass"],
        rho_0=td["SB2006_reference_air_density"], A=td["SB2006_autoconversion_correcting_function_coeff_A"],
        a=td["SB2006_autoconversion_correcting_function_coeff_a"],
        b=td["SB2006_autoconversion_correcting_function_coeff_b"])
    sb.accr = fam.accr_sb2006(
        kcr=td["SB2006_collection_kernel_coeff_kcr"], tau_0=td["SB2006_accretion_correcting_function_coeff_tau0"],
        rho_0=td["SB2006_reference_air_density"], c=td["SB2006_accretion_correcting_function_coeff_c"])
    sb.self = fam.selfcol_sb2006(krr=td["SB2006_collection_kernel_coeff_krr"],
                                 kappa_rr=td["SB2006_collection_kernel_coeff_kapparr"],
                                 d=td["SB2006_raindrops_self-collection_coeff_d"])
    sb.brek = fam.breakup_sb2006(Deq=td["SB2006_raindrops_equilibrium_mean_diameter"],
                                 Dr_th=td["SB2006_raindrops_breakup_mean_diameter_threshold"],
                                 kbr=td["SB2006_raindrops_breakup_coeff_kbr"],
                                 kappa_br=td["SB2006_raindrops_breakup_coeff_kappabr"])
    sb.evap = EvaporationSB2006(td)
    sb.numadj = fam.numadj_horn2012(tau=td["Horn2012_number_concentration_adjustment_timescale"])
    sb.is_limited = bool(is_limited)  # python-side tag (islimited(pdf_r), Microphysics2M.jl:393-394)
    return sb


def SB2006VelType(FT):
    """src/parameters/TerminalVelocity.jl:174-196."""
    td = _td(FT)
    return td.fam.sb2006_vel(
        rho_0=td["SB2006_reference_air_density"], aR=td["SB2006_raindrops_terminal_velocity_coeff_aR"],
        bR=td["SB2006_raindrops_terminal_velocity_coeff_bR"], cR=td["SB2006_raindrops_terminal_velocity_coeff_cR"],
        rho_w=td["density_liquid_water"], nu_air=td["kinematic_viscosity_of_air"],
        grav=td["gravitational_acceleration"])


def Chen2022VelTypeRain(FT):
    """src/parameters/TerminalVelocity.jl:288-311 (Chen et al. 2022 Table B1)."""
    td = _td(FT)
    fam = td.fam
    arr = fam.ft * 3
    return fam.chen2022_rain_vel(
        rho_0=td["Chen2022_table_B1_q_coeff"], a=arr(*td["Chen2022_table_B1_ai"]),
        a3_pow=td["Chen2022_table_B1_a3_pow_coeff"], b=arr(*td["Chen2022_table_B1_bi"]),
        b_rho=td["Chen2022_table_B1_b_rho_coeff"], c=arr(*td["Chen2022_table_B1_ci"]))


class WarmRainParams2M:
    """CMP.WarmRainParams2M — src/parameters/Microphysics2MParams.jl:14-28."""

    def __init__(self, FT, is_limited: bool = True):
        td = _td(FT)
        self.fam = td.fam
        self.is_limited = bool(is_limited)
        self.c = td.fam.warm_rain_2m()
        self.c.seifert_beheng = SB2006(td, is_limited)
        self.c.air_properties = AirProperties(td)
        self.c.condevap_tau_relax = td["condensation_evaporation_timescale"]
        self.c.subdep_tau_relax = td["sublimation_deposition_timescale"]

    seifert_beheng = property(lambda self: self.c.seifert_beheng)
    air_properties = property(lambda self: self.c.air_properties)


class Microphysics2MParams:
    """CMP.Microphysics2MParams(FT; with_ice = false, is_limited = true) —
    src/parameters/Microphysics2MParams.jl:134-162.  Only the warm-rain (`ice == nothing`) form is
    on this path."""

    def __init__(self, FT, with_ice: bool = False, is_limited: bool = True):
        if with_ice:
            raise NotImplementedError("2M + P3 ice entry (BMT:898-1083) is outside the current hot path (DESIGN.md §7)")
        self.warm_rain = WarmRainParams2M(FT, is_limited)
        self.ice = None
        self.fam = self.warm_rain.fam


def Koop2000(FT):
    """CMP.Koop2000 — src/parameters/IceNucleation.jl:38-69."""
    td = _td(FT)
    return td.fam.koop2000(
        delta_a_w_min=td["Koop2000_min_delta_aw"], delta_a_w_max=td["Koop2000_max_delta_aw"],
        c1=td["Koop2000_J_hom_coeff1"], c2=td["Koop2000_J_hom_coeff2"], c3=td["Koop2000_J_hom_coeff3"],
        c4=td["Koop2000_J_hom_coeff4"], linear_c1=td["Linear_J_hom_coeff1"], linear_c2=td["Linear_J_hom_coeff2"])


def Kaolinite(FT):
    """ABIFM fields of CMP.Kaolinite — src/parameters/AerosolKaolinite.jl:12-34."""
    td = _td(FT)
    return td.fam.abifm_dust(ABIFM_m=td["KnopfAlpert2013_J_ABIFM_m_Kaolinite"],
                             ABIFM_c=td["KnopfAlpert2013_J_ABIFM_c_Kaolinite"])


def Illite(FT):
    """ABIFM fields of CMP.Illite — src/parameters/AerosolIllite.jl:12-32."""
    td = _td(FT)
    return td.fam.abifm_dust(ABIFM_m=td["KnopfAlpert2013_J_ABIFM_m_Illite"],
                             ABIFM_c=td["KnopfAlpert2013_J_ABIFM_c_Illite"])


def ABIFMDust(FT, ABIFM_m: float, ABIFM_c: float):
    """Any other dust type (DesertDust, ArizonaTestDust, …): the caller supplies its ABIFM m, c (their ClimaParams
    defaults are not in the reference tree and are pinned by no reference test)."""
    return _abi.family(FT).abifm_dust(ABIFM_m=ABIFM_m, ABIFM_c=ABIFM_c)


def rain_vel_params(FT):
    """Both rain terminal-velocity parameter sets in the C layout `cmx_rain_vel`."""
    td = _td(FT)
    v = td.fam.rain_vel()
    v.sb2006 = SB2006VelType(td)
    v.chen2022 = Chen2022VelTypeRain(td)
    return v
