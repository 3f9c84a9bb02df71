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
    # ---- 1-moment scheme (docs/src/Microphysics1M.md:71-87,130-135,191-207, TerminalVelocity.md:63-70; pinned by the
    # accretion / velocity / melt KATs test/gpu_tests.jl:627-630,737-743,778 — SURVEY §8c)
    "isochoric_specific_heat_liquid": 4181.0,   # cv_l (= cp_l in Thermodynamics.jl)
    "rain_drop_length_scale": 1e-3, "rain_mass_size_relation_coefficient_me": 3.0,
    "rain_mass_size_relation_coefficient_delm": 0.0, "rain_mass_size_relation_coefficient_chim": 1.0,
    "rain_cross_section_size_relation_coefficient_ae": 2.0, "rain_cross_section_size_relation_coefficient_dela": 0.0,
    "rain_cross_section_size_relation_coefficient_chia": 1.0,
    "rain_terminal_velocity_size_relation_coefficient_ve": 0.5,
    "rain_terminal_velocity_size_relation_coefficient_delv": 0.0,
    "rain_terminal_velocity_size_relation_coefficient_chiv": 1.0,
    "rain_drop_size_distribution_coefficient_n0": 16e6, "rain_drop_drag_coefficient": 0.55,
    "rain_ventilation_coefficient_a": 1.5, "rain_ventilation_coefficient_b": 0.53,
    "snow_flake_length_scale": 1e-3, "snow_mass_size_relation_coefficient_me": 2.0,
    "snow_mass_size_relation_coefficient_delm": 0.0, "snow_mass_size_relation_coefficient_chim": 1.0,
    "snow_cross_section_size_relation_coefficient": 2.0, "snow_cross_section_size_relation_coefficient_dela": 0.0,
    "snow_cross_section_size_relation_coefficient_chia": 1.0,
    "snow_terminal_velocity_size_relation_coefficient": 0.25,
    "snow_terminal_velocity_size_relation_coefficient_delv": 0.0,
    "snow_terminal_velocity_size_relation_coefficient_chiv": 1.0,
    "snow_flake_size_distribution_coefficient_mu": 4.36e9, "snow_flake_size_distribution_coefficient_nu": 0.63,
    "snow_ventilation_coefficient_a": 0.65, "snow_ventilation_coefficient_b": 0.44,
    "snow_apparent_density": 100.0,            # Chen-2022 snow velocity; pinned with the two below by gpu_tests.jl:627
    "snow_aspect_ratio": 0.15, "snow_aspect_ratio_coefficient": 1.0 / 3.0,   # ϕ^κ = 0.15^(1/3)
    "cloud_ice_crystals_length_scale": 1e-5, "cloud_ice_mass_size_relation_coefficient_me": 3.0,
    "cloud_ice_mass_size_relation_coefficient_delm": 0.0, "cloud_ice_mass_size_relation_coefficient_chim": 1.0,
    "cloud_ice_size_distribution_coefficient_n0": 2e7,
    "cloud_ice_apparent_density": 500.0,       # back-solved from the ice accretion KATs
    "liquid_cloud_effective_radius": 14e-6, "ice_cloud_effective_radius": 25e-6,            # not used by the rates
    "cloud_liquid_sedimentation_number_concentration": 5e8, "cloud_ice_sedimentation_number_concentration": 5e8,  # pinned: gpu_tests.jl:624-625
    "rain_autoconversion_timescale": 1e3, "snow_autoconversion_timescale": 1e2,
    "cloud_liquid_water_specific_humidity_autoconversion_threshold": 5e-4,
    "cloud_ice_specific_humidity_autoconversion_threshold": 1e-6,
    "threshold_smooth_transition_steepness": 2.0,   # ClimaParams default; the reference tests only bound it loosely
    "Variable_time_scale_autoconversion_coeff_alpha": 0.73, "prescribed_cloud_droplet_number_concentration": 1e8,
    "ice_snow_threshold_radius": 62.5e-6,
    "cloud_liquid_rain_collision_efficiency": 0.8, "cloud_liquid_snow_collision_efficiency": 0.1,
    "cloud_ice_rain_collision_efficiency": 1.0, "cloud_ice_snow_collision_efficiency": 0.1,
    "rain_snow_collision_efficiency": 1.0, "rain_snow_velocity_dispersion_coefficient": 0.2,
    # ---- ARG2000 aerosol activation (src/parameters/AerosolActivation.jl:38-53).  f/g/p: docs/src/AerosolActivation.md:
    # 190-204; the physical constants are the ClimaParams defaults (not in the reference tree; pinned only to ≈5 % by the
    # digitised ARG2000 Fig. 1 test, test/aerosol_activation_tests.jl:236-299 — "parity unpinned" beyond that)
    "molar_mass_water": 0.01801528, "universal_gas_constant": 8.3144598, "density_ice_water": 916.7,
    "surface_tension_water": 0.072,
    "ARG2000_f_coeff_1": 0.5, "ARG2000_f_coeff_2": 2.5, "ARG2000_g_coeff_1": 1.0, "ARG2000_g_coeff_2": 0.25,
    "ARG2000_pow_1": 1.5, "ARG2000_pow_2": 0.75,
    # aerosol species used by the reference tests (test/gpu_tests.jl:563-571; src/parameters/AerosolSeasalt.jl, AerosolSulfate.jl)
    "seasalt_aerosol_molar_mass": 0.058443, "seasalt_aerosol_density": 2170.0, "seasalt_aerosol_osmotic_coefficient": 0.9,
    "seasalt_aerosol_ion_number": 2.0, "seasalt_aerosol_water_soluble_mass_fraction": 1.0, "seasalt_aerosol_kappa": 1.12,
    "sulfate_aerosol_molar_mass": 0.132, "sulfate_aerosol_density": 1770.0, "sulfate_aerosol_osmotic_coefficient": 1.0,
    "sulfate_aerosol_ion_number": 3.0, "sulfate_aerosol_water_soluble_mass_fraction": 1.0, "sulfate_aerosol_kappa": 0.53,
    # ---- P3 (src/parameters/MicrophysicsP3.jl name maps :33-36,67,115-120,144,305-308; docs/src/P3Scheme.md:56-59,327)
    # α_va = BF1995_mass_coeff_alpha · 10^(6β−3) (MicrophysicsP3.jl:37-41)
    "BF1995_mass_coeff_alpha": 7.38e-11, "BF1995_mass_exponent_beta": 1.9,
    "M1996_area_coeff_gamma": 0.2285, "M1996_area_exponent_sigma": 1.88,
    "Heymsfield_mu_coeff1": 0.00191, "Heymsfield_mu_coeff2": 0.8, "Heymsfield_mu_coeff3": 2.0, "Heymsfield_mu_cutoff": 6.0,
    "P3_constant_slope_parameterization_value": 0.0,   # SlopeConstant default (unpinned; only used with slope_law="constant")
    # bulk 2M autoconversion / accretion variants — docs/src/Microphysics2M.md:690-880; pinned by test/gpu_tests.jl:782-818
    "KK2000_autoconversion_coeff_A": 7.42e13, "KK2000_autoconversion_coeff_a": 2.47, "KK2000_autoconversion_coeff_b": -1.79,
    "KK2000_autoconversion_coeff_c": -1.47, "KK2000_accretion_coeff_A": 67.0, "KK2000_accretion_coeff_a": 1.15,
    "KK2000_accretion_coeff_b": -1.3,
    "B1994_autoconversion_coeff_C": 3e34, "B1994_autoconversion_coeff_a": -1.7, "B1994_autoconversion_coeff_b": 4.7,
    "B1994_autoconversion_coeff_c": -3.3, "B1994_autoconversion_coeff_N_0": 2e8, "B1994_autoconversion_coeff_d_low": 3.9,
    "B1994_autoconversion_coeff_d_high": 9.9, "B1994_accretion_coeff_A": 6.0,
    "TC1980_autoconversion_coeff_a": 7.0 / 3.0, "TC1980_autoconversion_coeff_b": -1.0 / 3.0, "TC1980_autoconversion_coeff_D": 3268.0,
    "TC1980_autoconversion_coeff_r_0": 7e-6, "TC1980_autoconversion_coeff_me_liq": 3.0, "TC1980_accretion_coeff_A": 4.7,
    "LD2004_R_6C_coeff": 7.5, "LD2004_E_0_coeff": 1.08e10,
    "P3_wet_growth_timescale": 100.0,                  # not read by the shape solver
    # TD.Parameters.q_min(tps): donor floor of the 1M LinearizedAverage linearization (BMT:395); ClimaParams value not
    # in the tree and not pinned by any reference test (only states with a hydrometeor below it are affected)
    "specific_humidity_minimum": 1e-10,
    # Chen et al. (2022) ice tables B3 (small ice) / B5 (large ice) and the small/large cutoff; pinned through the P3
    # particle-velocity KATs (test/p3_tests.jl:283-307) and, to 14 digits, the bulk fall-speed KATs (:376-379)
    "Chen2022_table_B3_As": (-0.263503, 0.00174079, 0.0378769), "Chen2022_table_B3_Bs": (0.575231, 0.0909307, 0.515579),
    "Chen2022_table_B3_Cs": (-0.345387, 0.177362, -0.000427794, 0.00419647),
    "Chen2022_table_B3_Es": (-0.156593, 0.0189334, 0.1377817), "Chen2022_table_B3_Fs": (-3.35641, 0.0156199, 0.765337),
    "Chen2022_table_B3_Gs": (-0.0309715, 1.55054, 0.518349),
    "Chen2022_table_B5_Al": (-0.475897, -0.00231270, 1.12293), "Chen2022_table_B5_Bl": (-2.56289, -0.00513504, 0.608459),
    "Chen2022_table_B5_Cl": (-0.756064, 0.935922, -1.70952), "Chen2022_table_B5_El": (0.00639847, 0.00906454, -0.108232),
    "Chen2022_table_B5_Fl": (0.515453, -0.0725042, -1.86810e19), "Chen2022_table_B5_Gl": (2.65236, 0.00158269, 259.935),
    "Chen2022_table_B5_Hl": (-0.346044, -7.17829e-11, -1.24394e20),
    "Chen2022_ice_cutoff": 0.000625,
    # 0-moment scheme (src/parameters/Microphysics0M.jl:12-28).  ClimaParams' values, not pinned by a reference test: its 0M tests
    # (test/microphysics0M_tests.jl, test/gpu_tests.jl:105-141) compare with the formula evaluated on the struct's own fields
    "precipitation_timescale": 1000.0, "specific_humidity_precipitation_threshold": 5e-6,
    "supersaturation_precipitation_threshold": 0.02,
    # Cober & List (1993) local rime density, Eq. 17 in kg/m³ — pinned by test/p3_tests.jl:719-726 (a + b + c = 159.5,
    # ρ′(8) = 51 + 114·8 − 5.5·64, ρ′(12) = ρ_ice)
    "CL1993_local_rime_density_constant_coeff": 51.0, "CL1993_local_rime_density_linear_coeff": 114.0,
    "CL1993_local_rime_density_quadratic_coeff": -5.5,
    # Bigg (1953) / Barklie & Gokhale (1959) immersion freezing: B·exp(a·33.15) pinned by the P3_het_N_i KAT
    # (test/gpu_tests.jl:1014-1037: 0.0002736160475969029 at T = 240, N = 2000, V = 3e-18, Δt = 0.1) with a = 0.65
    "BarklieGokhale1959_a_parameter": 0.65, "BarklieGokhale1959_B_parameter": 200.0,
    # Thompson et al. (2004) / Cooper deposition number: pinned by P3_deposition_N_i(240 K) = 119018.93920746
    # (test/gpu_tests.jl:1001-1012); T_dep_thres ∈ (232, 234] by test/heterogeneous_ice_nucleation_tests.jl:155-168
    "Thompson2004_c1_Cooper": 0.005, "Thompson2004_c2_Cooper": 0.304, "temperature_homogenous_nucleation": 233.0,
    # Frostenberg et al. (2023) INPC(T) climatology: a = b = 1 pinned by INP_concentration_mean = 9 log 2
    # (test/heterogeneous_ice_nucleation_tests.jl:250-253); σ = 1.37 from docs/src/IceNucleation.md:299
    "Frostenberg2023_standard_deviation": 1.37, "Frostenberg2023_a_coefficient": 1.0, "Frostenberg2023_b_coefficient": 1.0,
    # ---- round 3 (VERDICT r02 row g) -------------------------------------------------------------------------------------------
    # Luo et al. (1995) H2SO4 / H2O solution vapour pressure: every coefficient printed in docs/src/WaterActivity.md:28-36; pinned by
    # H2SO4_soln_saturation_vapor_pressure(0.1, 230) = 12.685507586924 and a_w_xT = 0.928418590276476 (test/gpu_tests.jl:876-893)
    "p_over_sulphuric_acid_solution_T_max": 235.0, "p_over_sulphuric_acid_solution_T_min": 185.0,
    "p_over_sulphuric_acid_solution_w_2": 1.4408, "p_over_sulphuric_acid_solution_c1": 23.306, "p_over_sulphuric_acid_solution_c2": 5.3465,
    "p_over_sulphuric_acid_solution_c3": 12.0, "p_over_sulphuric_acid_solution_c4": 8.19, "p_over_sulphuric_acid_solution_c5": -5814.0,
    "p_over_sulphuric_acid_solution_c6": 928.9, "p_over_sulphuric_acid_solution_c7": 1876.7,
    # Mohler et al. (2006) deposition on dust.  WARM branch (T > T_thr) PINNED by the two KATs at T = 240 K, S_i = 1.2
    # (test/gpu_tests.jl:930-966): MohlerDepositionRate = N_aer·a·dSi_dt = 38.7 / 423 with N_aer = 3000, dSi_dt = 0.03 → a = 0.43 / 4.7;
    # dust_activated_number_fraction = exp(a (S_i − S₀)) − 1 = 0.0129835639 / 1.2233164999 → S₀ = 1.17 / 1.03.
    # COLD branch, T_thr and Sᵢ_max: PARITY UNPINNED — no reference number constrains them beyond the orderings of
    # test/heterogeneous_ice_nucleation_tests.jl:39-90 (cold activates more than warm at S_i = 1.2; 1.34 < Sᵢ_max ≤ 1.5;
    # T_warm = 250 > T_thr > T_cold = 210); the values below are Table 2 of the paper as recalled (223 K / 210 K series).
    "Mohler2006_maximum_allowed_Si": 1.35, "Mohler2006_threshold_T": 220.0,
    "Mohler2006_S0_warm_DesertDust": 1.17, "Mohler2006_a_warm_DesertDust": 0.43,
    "Mohler2006_S0_cold_DesertDust": 1.05, "Mohler2006_a_cold_DesertDust": 2.35,
    "Mohler2006_S0_warm_ArizonaTestDust": 1.03, "Mohler2006_a_warm_ArizonaTestDust": 4.7,
    "Mohler2006_S0_cold_ArizonaTestDust": 1.07, "Mohler2006_a_cold_ArizonaTestDust": 9.2,
    # water-activity based deposition nucleation, log10 J[cm⁻² s⁻¹] = m Δa_w + c.  Kaolinite (China et al. 2017): the "true" coefficients
    # of the reference's own perfect-model calibration, papers/ice_nucleation_2024/calibration_setup.jl:145, reproduce the KAT
    # deposition_J(kaolinite, 0.16) = 1.5390757663075784e6 (test/gpu_tests.jl:968-984) to all printed digits: PINNED.
    # Feldspar / Ferrihydrite (Alpert et al. 2022): ONE KAT each (Δa_w = 0.15 → 5.693312205851678e6 / 802555.3607426438) for two
    # unknowns — slope from the two digitised points of the paper's Fig. 6 in docs/src/plots/activity_based_deposition.jl:31-35,
    # intercept from the KAT: the KAT is reproduced exactly, any other Δa_w is PARITY UNPINNED to the accuracy of that slope.
    "China2017_J_deposition_m_Kaolinite": 27.551, "China2017_J_deposition_c_Kaolinite": -2.2209,
    "Alpert2022_J_deposition_m_Feldspar": (4.165563 - 1.039735) / (0.256216 - 0.019459),
    "Alpert2022_J_deposition_c_Feldspar": 0.7749623086329027,
    "Alpert2022_J_deposition_m_Ferrihydrite": (4.21854 - 1.2781457) / (0.336486 - 0.0989189),
    "Alpert2022_J_deposition_c_Ferrihydrite": 0.04790839208164743,
}

# the reference's calibrated override file src/parameters/toml/ARG2000.toml (PySDM-based calibration)
ARG2000_CALIBRATED_OVERRIDE = {
    "ARG2000_f_coeff_1": 0.26583888195264627, "ARG2000_f_coeff_2": 2.3851515425961853,
    "ARG2000_g_coeff_1": 0.779519468021862, "ARG2000_g_coeff_2": 0.10571967167118024,
    "ARG2000_pow_1": 1.6523365679298359, "ARG2000_pow_2": 0.7578626397779737,
}

# the reference's override file src/parameters/toml/SB2006_limiters.toml (used by its CPU tests,
# test/microphysics2M_tests.jl:26-31)
SB2006_LIMITERS_OVERRIDE = {
    "SB2006_raindrops_min_mass": 6.54e-11,
    "SB2006_raindrops_size_distribution_coeff_N0_min": 3.5e5,
    "SB2006_raindrops_size_distribution_coeff_N0_max": 2e11,
    "SB2006_raindrops_size_distribution_coeff_lambda_max": 4e4,
}


class UnpinnedParameterWarning(UserWarning):
    """A default value that NO number held by the reference pins (its ClimaParams source is not in the reference tree): results computed
    with it are "parity unpinned" (DESIGN.md §6).  Pass the value explicitly (`create_toml_dict(FT, override={…})`) to silence it."""


# names → what is (not) known about them.  Every constructor that reads one of these from the DEFAULTS warns and tags the struct it returns
# with `.unpinned` (unpinned values must not pass silently as reference defaults).
UNPINNED_DEFAULTS = {
    "Mohler2006_maximum_allowed_Si": "only 1.34 < S_i,max <= 1.5 follows from test/heterogeneous_ice_nucleation_tests.jl:39-90",
    "Mohler2006_threshold_T": "only 210 K < T_thr < 250 K follows from the reference's tests",
    "Mohler2006_S0_cold_DesertDust": "cold branch: no reference number", "Mohler2006_a_cold_DesertDust": "cold branch: no reference number",
    "Mohler2006_S0_cold_ArizonaTestDust": "cold branch: no reference number", "Mohler2006_a_cold_ArizonaTestDust": "cold branch: no reference number",
    "Alpert2022_J_deposition_m_Feldspar": "slope digitised from a figure script; one KAT pins the intercept given the slope",
    "Alpert2022_J_deposition_c_Feldspar": "back-solved from one KAT with the digitised slope",
    "Alpert2022_J_deposition_m_Ferrihydrite": "slope digitised from a figure script; one KAT pins the intercept given the slope",
    "Alpert2022_J_deposition_c_Ferrihydrite": "back-solved from one KAT with the digitised slope",
}


def _tag_unpinned(struct, td, names):
    """Warn about — and record on the returned struct — the unpinned defaults among `names` that the caller did not override."""
    import warnings
    used = tuple(n for n in names if n in UNPINNED_DEFAULTS and n not in td.overridden)
    struct.unpinned = used
    if used:
        warnings.warn("parity-unpinned default(s) in use: " + "; ".join(f"{n} ({UNPINNED_DEFAULTS[n]})" for n in used), UnpinnedParameterWarning, stacklevel=3)
    return struct


class ParamDict(dict):
    """`CP.create_toml_dict(FT; override_file)` analogue: defaults + overrides, typed by FT."""

    def __init__(self, FT, override: Optional[Mapping] = None):
        super().__init__(DEFAULT_PARAMETERS)
        if override:
            unknown = set(override) - set(DEFAULT_PARAMETERS)
            if unknown:
                raise KeyError(f"unknown parameter name(s): {sorted(unknown)}")
            self.update(override)
        self.overridden = frozenset(override or ())
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
        press_triple=td["pressure_triple_point"], T_freeze=td["temperature_water_freeze"],
        cv_l=td["isochoric_specific_heat_liquid"])


def Bulk2MSchemes(FT):
    """CMP.KK2000(FT), CMP.B1994(FT), CMP.TC1980(FT), CMP.LD2004(FT) bundled (include/cmx.h: cmx_bulk_2m_schemes) —
    src/parameters/Microphysics2M.jl:11-279."""
    td = _td(FT)
    f = td.fam
    k = td["threshold_smooth_transition_steepness"]
    return f.bulk_2m_schemes(
        kk2000=f.kk2000(acnv_A=td["KK2000_autoconversion_coeff_A"], acnv_a=td["KK2000_autoconversion_coeff_a"],
                        acnv_b=td["KK2000_autoconversion_coeff_b"], acnv_c=td["KK2000_autoconversion_coeff_c"],
                        accr_A=td["KK2000_accretion_coeff_A"], accr_a=td["KK2000_accretion_coeff_a"], accr_b=td["KK2000_accretion_coeff_b"]),
        b1994=f.b1994(acnv_C=td["B1994_autoconversion_coeff_C"], acnv_a=td["B1994_autoconversion_coeff_a"],
                      acnv_b=td["B1994_autoconversion_coeff_b"], acnv_c=td["B1994_autoconversion_coeff_c"],
                      acnv_N_0=td["B1994_autoconversion_coeff_N_0"], acnv_d_low=td["B1994_autoconversion_coeff_d_low"],
                      acnv_d_high=td["B1994_autoconversion_coeff_d_high"], acnv_k=k, accr_A=td["B1994_accretion_coeff_A"]),
        tc1980=f.tc1980(acnv_a=td["TC1980_autoconversion_coeff_a"], acnv_b=td["TC1980_autoconversion_coeff_b"],
                        acnv_D=td["TC1980_autoconversion_coeff_D"], acnv_r_0=td["TC1980_autoconversion_coeff_r_0"],
                        acnv_me_liq=td["TC1980_autoconversion_coeff_me_liq"],
                        acnv_m0_liq_coeff=td["density_liquid_water"] * 4 / 3 * math.pi, acnv_k=k, accr_A=td["TC1980_accretion_coeff_A"]),
        ld2004=f.ld2004(R_6C_0=td["LD2004_R_6C_coeff"], E_0=td["LD2004_E_0_coeff"], rho_w=td["density_liquid_water"], k=k))


def StokesRegimeVelType(FT):
    """CMP.StokesRegimeVelType(FT) — src/parameters/TerminalVelocity.jl:150-164."""
    td = _td(FT)
    return td.fam.stokes_vel(rho_w=td["density_liquid_water"], nu_air=td["kinematic_viscosity_of_air"],
                             grav=td["gravitational_acceleration"])


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

    def __init__(self, FT, with_ice: bool = False, is_limited: bool = True, quadrature_order: int = 16, tau_act: float = 300.0):
        self.warm_rain = WarmRainParams2M(FT, is_limited)
        self.ice = P3IceParams(FT, is_limited=is_limited, quadrature_order=quadrature_order, tau_act=tau_act) if with_ice else None
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


def Mohler2006(FT):
    """CMP.Mohler2006 — src/parameters/IceNucleation.jl:13-28 (values: see DEFAULT_PARAMETERS — parity unpinned)."""
    td = _td(FT)
    return _tag_unpinned(td.fam.mohler2006(S_i_max=td["Mohler2006_maximum_allowed_Si"], T_thr=td["Mohler2006_threshold_T"]), td,
                         ("Mohler2006_maximum_allowed_Si", "Mohler2006_threshold_T"))


def _mohler_dust(FT, name):
    td = _td(FT)
    return _tag_unpinned(td.fam.mohler_dust(S0_warm=td[f"Mohler2006_S0_warm_{name}"], S0_cold=td[f"Mohler2006_S0_cold_{name}"],
                                            a_warm=td[f"Mohler2006_a_warm_{name}"], a_cold=td[f"Mohler2006_a_cold_{name}"]), td,
                         (f"Mohler2006_S0_cold_{name}", f"Mohler2006_a_cold_{name}"))


def DesertDust(FT):
    """The Mohler-2006 deposition fields of CMP.DesertDust — src/parameters/AerosolDesertDust.jl:13-21 (warm branch pinned by KATs)."""
    return _mohler_dust(FT, "DesertDust")


def ArizonaTestDust(FT):
    """The Mohler-2006 deposition fields of CMP.ArizonaTestDust — src/parameters/AerosolATD.jl:12-20 (warm branch pinned by KATs)."""
    return _mohler_dust(FT, "ArizonaTestDust")


def DepositionDust(FT, mineral: str):
    """deposition_m, deposition_c of CMP.Kaolinite / Feldspar / Ferrihydrite (src/parameters/AerosolKaolinite.jl, AerosolFeldspar.jl,
    AerosolFerrihydrite.jl) for CMI_het.deposition_J."""
    td = _td(FT)
    src = {"Kaolinite": "China2017", "Feldspar": "Alpert2022", "Ferrihydrite": "Alpert2022"}[mineral]
    return _tag_unpinned(td.fam.deposition_dust(deposition_m=td[f"{src}_J_deposition_m_{mineral}"], deposition_c=td[f"{src}_J_deposition_c_{mineral}"]), td,
                         (f"{src}_J_deposition_m_{mineral}", f"{src}_J_deposition_c_{mineral}"))


def H2SO4SolutionParameters(FT):
    """CMP.H2SO4SolutionParameters — src/parameters/Aerosol_H2SO4_Solution.jl (Luo et al. 1995; docs/src/WaterActivity.md:28-36)."""
    td = _td(FT)
    g = lambda k: td[f"p_over_sulphuric_acid_solution_{k}"]  # noqa: E731
    return td.fam.h2so4_solution_params(T_max=g("T_max"), T_min=g("T_min"), w_2=g("w_2"), c1=g("c1"), c2=g("c2"), c3=g("c3"), c4=g("c4"),
                                        c5=g("c5"), c6=g("c6"), c7=g("c7"))


def ABIFMDust(FT, ABIFM_m: float, ABIFM_c: float):
    """Any other dust type (DesertDust, ArizonaTestDust, …): the caller supplies its ABIFM m, c (their ClimaParams
    defaults are not in the reference tree and are pinned by no reference test)."""
    return _abi.family(FT).abifm_dust(ABIFM_m=ABIFM_m, ABIFM_c=ABIFM_c)


# ---------------------------------------------------------------------------
# 1-moment scheme
# ---------------------------------------------------------------------------
class _Option:
    """Base of the process-option singletons (CMP.MicrophysicsOption, Microphysics1MOptions.jl:34-41)."""
    flag = 0

    def __repr__(self):
        return type(self).__name__ + "()"


def _opt(name, flag, doc):
    return type(name, (_Option,), {"flag": flag, "__doc__": doc})


CloudLiquidFormation = _opt("CloudLiquidFormation", _abi.CMX_1M_CLOUD_LIQUID_FORMATION, "Microphysics1MOptions.jl:75-82")
ConstantTimescale = _opt("ConstantTimescale", _abi.CMX_1M_CLOUD_ICE_FORMATION_CONST, ":84-91")
TemperatureDependent = _opt("TemperatureDependent", _abi.CMX_1M_CLOUD_ICE_FORMATION_TDEP, ":98-105: Frostenberg (2023) INP timescale for deposition, constant for sublimation")
CloudIceMelt = _opt("CloudIceMelt", _abi.CMX_1M_CLOUD_ICE_MELT, ":199")
Kessler1M = _opt("Kessler1M", _abi.CMX_1M_RAIN_ACNV_KESSLER, ":102-109")
PrescribedNd = _opt("PrescribedNd", _abi.CMX_1M_RAIN_ACNV_PRESCRIBED_ND, ":111-118")
NoSupersaturation = _opt("NoSupersaturation", _abi.CMX_1M_SNOW_ACNV_NO_SUPERSAT, ":120-127")
WithSupersaturation = _opt("WithSupersaturation", _abi.CMX_1M_SNOW_ACNV_WITH_SUPERSAT, ":129-136")
RainEvaporation = _opt("RainEvaporation", _abi.CMX_1M_RAIN_EVAPORATION, ":197")
SublimationOnly = _opt("SublimationOnly", _abi.CMX_1M_SNOW_SUBLIMATION_ONLY, ":193")
DepositionAndSublimation = _opt("DepositionAndSublimation", _abi.CMX_1M_SNOW_DEP_AND_SUBL, ":195")
SnowMelt = _opt("SnowMelt", _abi.CMX_1M_SNOW_MELT, ":201")
CloudLiquidRainAccretion = _opt("CloudLiquidRainAccretion", _abi.CMX_1M_ACCR_LCL_RAI, ":138-145")
CloudLiquidSnowAccretion = _opt("CloudLiquidSnowAccretion", _abi.CMX_1M_ACCR_LCL_SNO, ":147-155")
CloudIceRainAccretion = _opt("CloudIceRainAccretion", _abi.CMX_1M_ACCR_ICL_RAI, ":157-165")
CloudIceSnowAccretion = _opt("CloudIceSnowAccretion", _abi.CMX_1M_ACCR_ICL_SNO, ":167-174")
RainSnowAccretion = _opt("RainSnowAccretion", _abi.CMX_1M_ACCR_RAI_SNO, ":176-184")


class Microphysics1MOptions:
    """CMP.Microphysics1MOptions(; …) — src/parameters/Microphysics1MOptions.jl:257-286.  `None` disables a process."""
    _defaults = dict(
        cloud_liquid_formation=CloudLiquidFormation, cloud_ice_formation=ConstantTimescale, cloud_ice_melt=CloudIceMelt,
        rain_autoconversion=Kessler1M, snow_autoconversion=NoSupersaturation,
        rain_condensation_evaporation=RainEvaporation, snow_deposition_sublimation=DepositionAndSublimation,
        snow_melt=SnowMelt, cloud_liquid_rain_accretion=CloudLiquidRainAccretion,
        cloud_liquid_snow_accretion=CloudLiquidSnowAccretion, cloud_ice_rain_accretion=CloudIceRainAccretion,
        cloud_ice_snow_accretion=CloudIceSnowAccretion, rain_snow_accretion=RainSnowAccretion)
    _allowed = dict(
        cloud_liquid_formation=(CloudLiquidFormation,), cloud_ice_formation=(ConstantTimescale, TemperatureDependent),
        cloud_ice_melt=(CloudIceMelt,), rain_autoconversion=(Kessler1M, PrescribedNd),
        snow_autoconversion=(NoSupersaturation, WithSupersaturation), rain_condensation_evaporation=(RainEvaporation,),
        snow_deposition_sublimation=(SublimationOnly, DepositionAndSublimation), snow_melt=(SnowMelt,),
        cloud_liquid_rain_accretion=(CloudLiquidRainAccretion,), cloud_liquid_snow_accretion=(CloudLiquidSnowAccretion,),
        cloud_ice_rain_accretion=(CloudIceRainAccretion,), cloud_ice_snow_accretion=(CloudIceSnowAccretion,),
        rain_snow_accretion=(RainSnowAccretion,))

    def __init__(self, **kw):
        unknown = set(kw) - set(self._defaults)
        if unknown:
            raise TypeError(f"unknown option field(s) {sorted(unknown)}")
        self.flags = 0
        for name, default in self._defaults.items():
            opt = kw.get(name, default())
            if isinstance(opt, type):
                opt = opt()
            if opt is not None and not isinstance(opt, self._allowed[name]):
                raise TypeError(f"{name}: expected one of {[c.__name__ for c in self._allowed[name]]} or None")
            setattr(self, name, opt)
            self.flags |= opt.flag if opt is not None else 0


def _particle_mass(fam, r0, m0, me, dm, chim):
    return fam.particle_mass(r0=r0, m0=m0, me=me, delta_m=dm, chi_m=chim, gamma_coeff=math.gamma(me + dm + 1))


class Microphysics1MParams:
    """CMP.Microphysics1MParams(FT; options…) — src/parameters/Microphysics1MParams.jl:63-103.

    `.c` is the C struct `cmx_microphysics_1m`, `.processes` the Microphysics1MOptions (→ the `flags` word).
    Host-derived fields follow the reference constructors: m0, a0, gamma_coeff (Microphysics1M.jl params :127-139,
    180-206, 263-289), gamma_vent/term/accr/accr_rain_sink and snow v0 (TerminalVelocity.jl:57-63,121-126)."""

    def __init__(self, FT, **options):
        td = _td(FT)
        fam = self.fam = td.fam
        self.processes = Microphysics1MOptions(**options)
        g = td.__getitem__
        c = self.c = fam.microphysics_1m()
        pi = math.pi
        # Rain
        r0, me, dm, chim = g("rain_drop_length_scale"), g("rain_mass_size_relation_coefficient_me"), g(
            "rain_mass_size_relation_coefficient_delm"), g("rain_mass_size_relation_coefficient_chim")
        ae, da, chia = g("rain_cross_section_size_relation_coefficient_ae"), g(
            "rain_cross_section_size_relation_coefficient_dela"), g("rain_cross_section_size_relation_coefficient_chia")
        ve, dv, chiv = g("rain_terminal_velocity_size_relation_coefficient_ve"), g(
            "rain_terminal_velocity_size_relation_coefficient_delv"), g("rain_terminal_velocity_size_relation_coefficient_chiv")
        c.rain.n0 = g("rain_drop_size_distribution_coefficient_n0")
        c.rain.mass = _particle_mass(fam, r0, g("density_liquid_water") * r0 ** me * pi * 4 / 3, me, dm, chim)
        c.rain.area = fam.particle_area(a0=pi * r0 ** ae, ae=ae, delta_a=da, chi_a=chia)
        c.rain.vent = fam.ventilation(a=g("rain_ventilation_coefficient_a"), b=g("rain_ventilation_coefficient_b"))
        # NB the reference reads r0 of Blk1MVelTypeRain from `snow_flake_length_scale` (TerminalVelocity.jl:39)
        c.vel_rain = fam.blk1m_vel_rain(
            r0=g("snow_flake_length_scale"), ve=ve, delta_v=dv, chi_v=chiv, rho_w=g("density_liquid_water"),
            C_drag=g("rain_drop_drag_coefficient"), grav=g("gravitational_acceleration"),
            gamma_vent=math.gamma((ve + dv + 5) / 2), gamma_term=math.gamma(me + ve + dm + dv + 1),
            gamma_accr=math.gamma(ae + ve + da + dv + 1), gamma_accr_rain_sink=math.gamma(me + ae + ve + dm + da + dv + 1))
        # Snow
        r0, me, dm, chim = g("snow_flake_length_scale"), g("snow_mass_size_relation_coefficient_me"), g(
            "snow_mass_size_relation_coefficient_delm"), g("snow_mass_size_relation_coefficient_chim")
        ae, da, chia = g("snow_cross_section_size_relation_coefficient"), g(
            "snow_cross_section_size_relation_coefficient_dela"), g("snow_cross_section_size_relation_coefficient_chia")
        ve, dv, chiv = g("snow_terminal_velocity_size_relation_coefficient"), g(
            "snow_terminal_velocity_size_relation_coefficient_delv"), g("snow_terminal_velocity_size_relation_coefficient_chiv")
        c.snow.mu, c.snow.nu = g("snow_flake_size_distribution_coefficient_mu"), g("snow_flake_size_distribution_coefficient_nu")
        c.snow.mass = _particle_mass(fam, r0, r0 ** me / 10, me, dm, chim)
        c.snow.area = fam.particle_area(a0=0.3 * pi * r0 ** ae, ae=ae, delta_a=da, chi_a=chia)
        c.snow.vent = fam.ventilation(a=g("snow_ventilation_coefficient_a"), b=g("snow_ventilation_coefficient_b"))
        c.snow.phi, c.snow.kappa, c.snow.rho_i = g("snow_aspect_ratio"), g("snow_aspect_ratio_coefficient"), g("snow_apparent_density")
        a_obl, a_pro = me + dm - 1.5 * (ae + da), 3 * (ae + da) - 2 * (me + dm)
        c.snow.gamma_aspect_oblate = math.gamma(a_obl + 4) / math.gamma(4.0)
        c.snow.gamma_aspect_prolate = math.gamma(a_pro + 4) / math.gamma(4.0)
        c.vel_snow = fam.blk1m_vel_snow(
            r0=r0, ve=ve, delta_v=dv, chi_v=chiv, v0=2 ** (9 / 4) * r0 ** ve, gamma_vent=math.gamma((ve + dv + 5) / 2),
            gamma_term=math.gamma(me + ve + dm + dv + 1), gamma_accr=math.gamma(ae + ve + da + dv + 1))
        # Cloud liquid / ice
        c.cloud_liquid = fam.cloud_liquid(rho_w=g("density_liquid_water"), r_eff=g("liquid_cloud_effective_radius"),
                                          N_0=g("cloud_liquid_sedimentation_number_concentration"))
        r0, me, dm, chim = g("cloud_ice_crystals_length_scale"), g("cloud_ice_mass_size_relation_coefficient_me"), g(
            "cloud_ice_mass_size_relation_coefficient_delm"), g("cloud_ice_mass_size_relation_coefficient_chim")
        c.cloud_ice.n0 = g("cloud_ice_size_distribution_coefficient_n0")
        c.cloud_ice.mass = _particle_mass(fam, r0, g("cloud_ice_apparent_density") * r0 ** me * pi * 4 / 3, me, dm, chim)
        c.cloud_ice.rho_i, c.cloud_ice.r_eff = g("cloud_ice_apparent_density"), g("ice_cloud_effective_radius")
        c.cloud_ice.N_0 = g("cloud_ice_sedimentation_number_concentration")
        c.air_properties = AirProperties(td)
        # process parameters (Microphysics1MOptions.jl:296-395)
        pp = c.process_params
        pp.cloud_liquid_formation_tau_relax = g("condensation_evaporation_timescale")
        pp.cloud_ice_formation_tau_relax = g("sublimation_deposition_timescale")
        pp.cloud_ice_formation_frostenberg = Frostenberg2023(td)     # TemperatureDependent (Microphysics1MOptions.jl:314-318)
        k = g("threshold_smooth_transition_steepness")
        pp.rain_autoconversion = fam.acnv_1m(tau=g("rain_autoconversion_timescale"), k=k, q_threshold=g(
            "cloud_liquid_water_specific_humidity_autoconversion_threshold"))
        pp.rain_autoconversion_nd = fam.var_timescale_acnv(
            tau=g("rain_autoconversion_timescale"), alpha=g("Variable_time_scale_autoconversion_coeff_alpha"),
            Nc=g("prescribed_cloud_droplet_number_concentration"))
        pp.snow_autoconversion = fam.acnv_1m(tau=g("snow_autoconversion_timescale"), k=k, q_threshold=g(
            "cloud_ice_specific_humidity_autoconversion_threshold"))
        pp.r_ice_snow = g("ice_snow_threshold_radius")
        pp.e_lcl_rai, pp.e_lcl_sno = g("cloud_liquid_rain_collision_efficiency"), g("cloud_liquid_snow_collision_efficiency")
        pp.e_icl_rai, pp.e_icl_sno = g("cloud_ice_rain_collision_efficiency"), g("cloud_ice_snow_collision_efficiency")
        pp.e_rai_sno, pp.coeff_disp = g("rain_snow_collision_efficiency"), g("rain_snow_velocity_dispersion_coefficient")

    @property
    def flags(self):
        return self.processes.flags


class ParametersP3:
    """CMP.ParametersP3(FT; slope_law = :powerlaw | :constant) — src/parameters/MicrophysicsP3.jl:267-320 (the fields
    the shape solver reads).  Values: docs/src/P3Scheme.md:56-59,327 (β_va = 1.9, α_va = 7.38e-11·10^(6β−3), γ = 0.2285,
    σ = 1.88, μ = clamp(0.00191 λ^0.8 − 2, 0, 6)), ρ_i = 916.7; pinned by get_ρ_d = 488.9120789986414
    (src/P3_particle_properties.jl:185-188) and the D_m KATs (test/p3_tests.jl:440-447)."""

    def __init__(self, FT, slope_law: str = "powerlaw"):
        td = _td(FT)
        self.fam = td.fam
        if slope_law not in ("powerlaw", "constant"):
            raise ValueError("slope_law must be 'powerlaw' or 'constant'")
        self.flags = _abi.CMX_P3_SLOPE_CONSTANT if slope_law == "constant" else 0
        beta = td["BF1995_mass_exponent_beta"]
        self.c = td.fam.p3_params(
            alpha_va=td["BF1995_mass_coeff_alpha"] * 10 ** (6 * beta - 3), beta_va=beta,
            gamma=td["M1996_area_coeff_gamma"], sigma=td["M1996_area_exponent_sigma"],
            slope_a=td["Heymsfield_mu_coeff1"], slope_b=td["Heymsfield_mu_coeff2"], slope_c=td["Heymsfield_mu_coeff3"],
            mu_max=td["Heymsfield_mu_cutoff"], mu_const=td["P3_constant_slope_parameterization_value"],
            rho_i=td["density_ice_water"], rho_l=td["density_liquid_water"], tau_wet=td["P3_wet_growth_timescale"],
            T_freeze=td["temperature_water_freeze"])


def VentilationFactorP3(FT):
    """CMP.VentilationFactor(FT) of ParametersP3 (src/parameters/MicrophysicsP3.jl:165-180): the SB2006 coefficients
    (a_v, b_v); pinned by the P3 melting KATs (test/p3_tests.jl:650-668)."""
    td = _td(FT)
    return td.fam.ventilation(a=td["SB2006_ventilation_factor_coeff_av"], b=td["SB2006_ventilation_factor_coeff_bv"])


def Chen2022VelTypeIce(FT):
    """The (small_ice, large_ice) part of CMP.Chen2022VelType(FT) — src/parameters/TerminalVelocity.jl:207-275,325-335."""
    td = _td(FT)
    fam = td.fam
    arr = lambda n, key: (fam.ft * n)(*td[key])  # noqa: E731
    small = fam.chen2022_small_ice_vel(A=arr(3, "Chen2022_table_B3_As"), B=arr(3, "Chen2022_table_B3_Bs"),
                                       C=arr(4, "Chen2022_table_B3_Cs"), E=arr(3, "Chen2022_table_B3_Es"),
                                       F=arr(3, "Chen2022_table_B3_Fs"), G=arr(3, "Chen2022_table_B3_Gs"),
                                       cutoff=td["Chen2022_ice_cutoff"])
    large = fam.chen2022_large_ice_vel(A=arr(3, "Chen2022_table_B5_Al"), B=arr(3, "Chen2022_table_B5_Bl"),
                                       C=arr(3, "Chen2022_table_B5_Cl"), E=arr(3, "Chen2022_table_B5_El"),
                                       F=arr(3, "Chen2022_table_B5_Fl"), G=arr(3, "Chen2022_table_B5_Gl"),
                                       H=arr(3, "Chen2022_table_B5_Hl"), cutoff=td["Chen2022_ice_cutoff"])
    return fam.chen2022_ice_vel(small_ice=small, large_ice=large)


def _quadrature(FT, nodes, weights):
    fam = FT.fam if isinstance(FT, ParamDict) else _abi.family(FT)
    n = len(nodes)
    if not 1 <= n <= _abi.CMX_QUAD_MAX:
        raise ValueError(f"quadrature order must be in 1..{_abi.CMX_QUAD_MAX}")
    q = fam.quadrature(n=n)
    for i in range(n):
        q.node[i], q.weight[i] = float(nodes[i]), float(weights[i])
    return q


def ChebyshevGauss(FT, n: int = 100):
    """Quadrature.ChebyshevGauss(n) — src/Quadrature.jl:168-175: yᵢ = cospi((2i−1)/(2n)), total weight √(1−yᵢ²)·π/n."""
    import numpy as np
    i = np.arange(1, n + 1, dtype=np.float64)
    y = np.cos(np.pi * (2 * i - 1) / (2 * n))
    return _quadrature(FT, y, np.sqrt(1 - y * y) * np.pi / n)


def GaussLegendre(FT, n: int):
    """Quadrature.GaussLegendre(FT, n) — src/Quadrature.jl:226-252: Float64 nodes/weights (numpy's leggauss instead of
    FastGaussQuadrature; both are the exact Gauss–Legendre rule to double rounding) converted to FT."""
    import numpy as np
    y, w = np.polynomial.legendre.leggauss(n)
    return _quadrature(FT, y, w)


def build_quadrature(FT, quadrature_order: int):
    """Quadrature.build_quadrature(FT, order) — src/Quadrature.jl:272-278: Gauss–Legendre for 16/32/40/64, else Chebyshev–Gauss."""
    return GaussLegendre(FT, quadrature_order) if quadrature_order in (16, 32, 40, 64) else ChebyshevGauss(FT, quadrature_order)


def Parameters0M(FT, tau_precip=None, qc_0=None, S_0=None):
    """CMP.Parameters0M — src/parameters/Microphysics0M.jl:12-28 (keyword overrides as the reference's constructor takes)."""
    td = _td(FT)
    return td.fam.parameters_0m(tau_precip=td["precipitation_timescale"] if tau_precip is None else tau_precip,
                                qc_0=td["specific_humidity_precipitation_threshold"] if qc_0 is None else qc_0,
                                S_0=td["supersaturation_precipitation_threshold"] if S_0 is None else S_0)


class Microphysics0MParams:
    """CMP.Microphysics0MParams — src/parameters/Microphysics0MParams.jl:4-27: the 0-moment parameter set, `.precip` = Parameters0M."""

    def __init__(self, FT, **kw):
        td = _td(FT)
        self.fam = td.fam
        self.precip = Parameters0M(td, **kw)


def LocalRimeDensity(FT):
    """CMP.LocalRimeDensity — src/parameters/MicrophysicsP3.jl:202-221."""
    td = _td(FT)
    return td.fam.local_rime_density(a=td["CL1993_local_rime_density_constant_coeff"], b=td["CL1993_local_rime_density_linear_coeff"],
                                     c=td["CL1993_local_rime_density_quadratic_coeff"], rho_ice=td["density_ice_water"])


def RainFreezing(FT):
    """CMP.RainFreezing — src/parameters/IceNucleation.jl:129-146."""
    td = _td(FT)
    return td.fam.rain_freezing(het_a=td["BarklieGokhale1959_a_parameter"], het_B=td["BarklieGokhale1959_B_parameter"])


def Frostenberg2023(FT, a=None, b=None):
    """CMP.Frostenberg2023 — src/parameters/IceNucleation.jl:171-193 (log_a = log(a))."""
    import math
    td = _td(FT)
    a = td["Frostenberg2023_a_coefficient"] if a is None else a
    b = td["Frostenberg2023_b_coefficient"] if b is None else b
    return td.fam.frostenberg2023(sigma=td["Frostenberg2023_standard_deviation"], a=a, b=b,
                                  T_freeze=td["temperature_water_freeze"], log_a=math.log(a))


def MorrisonMilbrandt2014(FT):
    """CMP.MorrisonMilbrandt2014 — src/parameters/IceNucleation.jl:80-107."""
    td = _td(FT)
    return td.fam.morrison_milbrandt2014(
        T_dep_thres=td["temperature_homogenous_nucleation"], c1=td["Thompson2004_c1_Cooper"], c2=td["Thompson2004_c2_Cooper"],
        T0=td["temperature_water_freeze"], het_a=td["BarklieGokhale1959_a_parameter"], het_B=td["BarklieGokhale1959_B_parameter"])


class P3IceParams:
    """CMP.P3IceParams(toml_dict; is_limited = true, quadrature_order = 16, inp_depletion_model = NIceProxyDepletion(τ_act = 300))
    — src/parameters/Microphysics2MParams.jl:58-106, flattened into the C layout `cmx_p3_ice_params`."""

    def __init__(self, FT, is_limited: bool = True, quadrature_order: int = 16, tau_act: float = 300.0, slope_law: str = "powerlaw",
                 quad=None):
        td = _td(FT)
        self.fam = td.fam
        self.is_limited = bool(is_limited)
        scheme = ParametersP3(td, slope_law)
        self.flags = scheme.flags | (_abi.CMX_P3_RAIN_PDF_LIMITED if is_limited else 0)
        self.c = td.fam.p3_ice_params()
        self.c.scheme = scheme.c
        self.c.vent = VentilationFactorP3(td)
        self.c.rho_rim_local = LocalRimeDensity(td)
        self.c.vel_rain = Chen2022VelTypeRain(td)
        self.c.vel_ice = Chen2022VelTypeIce(td)
        sb = SB2006(td, is_limited)
        self.c.cloud_pdf = sb.pdf_c
        self.c.rain_pdf = sb.pdf_r
        self.c.ice_nucleation = Frostenberg2023(td)
        self.c.rain_freezing = RainFreezing(td)
        self.c.tau_act = tau_act
        self.c.quad = quad if quad is not None else build_quadrature(td, quadrature_order)


def AerosolActivationParameters(FT):
    """CMP.AerosolActivationParameters — src/parameters/AerosolActivation.jl:12-55."""
    td = _td(FT)
    return td.fam.aerosol_activation_params(
        M_w=td["molar_mass_water"], R=td["universal_gas_constant"], rho_w=td["density_liquid_water"],
        rho_i=td["density_ice_water"], sigma=td["surface_tension_water"], g=td["gravitational_acceleration"],
        f1=td["ARG2000_f_coeff_1"], f2=td["ARG2000_f_coeff_2"], g1=td["ARG2000_g_coeff_1"], g2=td["ARG2000_g_coeff_2"],
        p1=td["ARG2000_pow_1"], p2=td["ARG2000_pow_2"])


class _AerosolSpecies:
    def __init__(self, td, name):
        g = lambda k: td[f"{name}_aerosol_{k}"]  # noqa: E731
        self.M, self.rho, self.phi = g("molar_mass"), g("density"), g("osmotic_coefficient")
        self.nu, self.eps, self.kappa = g("ion_number"), g("water_soluble_mass_fraction"), g("kappa")


def Seasalt(FT):
    """CMP.Seasalt(FT): M, ρ, ϕ, ν, ϵ, κ (src/parameters/AerosolSeasalt.jl)."""
    return _AerosolSpecies(_td(FT), "seasalt")


def Sulfate(FT):
    """CMP.Sulfate(FT) (src/parameters/AerosolSulfate.jl)."""
    return _AerosolSpecies(_td(FT), "sulfate")


def rain_vel_params(FT):
    """Both rain terminal-velocity parameter sets in the C layout `cmx_rain_vel`."""
    td = _td(FT)
    v = td.fam.rain_vel()
    v.sb2006 = SB2006VelType(td)
    v.chen2022 = Chen2022VelTypeRain(td)
    return v
