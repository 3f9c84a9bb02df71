"""
    CMXExt

Reference-side binding of `libcmx.so` (the MI355X / gfx950 array evaluator behind `include/cmx.h`) for
CloudMicrophysics.jl v0.38.1.  The scalar API of the package is untouched; this module adds the ARRAY forms of the
rate functions — one call per fused HIP kernel — taking the package's own parameter structs
(`CMP.Microphysics2MParams`, `CMP.Microphysics1MParams`, `AM.AerosolDistribution`, `CMP.P3IceParams`, …), packing
them into the C layouts of `include/cmx.h` and forwarding device pointers through `ccall`.

    include("CMXExt.jl"); import .CMXExt          # needs CloudMicrophysics, Thermodynamics, Libdl
    ENV["CMX_LIB"] = "/path/to/libcmx.so"         # default: "libcmx.so" on the loader path

Arrays are DEVICE arrays (`AMDGPU.ROCArray{FT}`; anything whose `pointer(x)` is a device `Ptr{FT}`), `FT` is `Float32`
or `Float64`; every call is asynchronous on `stream` (a `hipStream_t` as `Ptr{Cvoid}`, e.g. `AMDGPU.stream().stream`).

How this file is kept honest without a Julia runtime in the build image: `tools/check_julia_shim.py` (run by
`tests/test_julia_shim.py`) parses
  * the `struct` field lists of `/root/reference/src/parameters/*.jl`, `src/AerosolModel.jl`, `src/Quadrature.jl`,
  * every `Cmx*` mirror struct below (field names are the C names of `include/cmx.h`; `# == cmx_…` names the C struct),
  * the `DIRECT_LAYOUT` table (reference structs handed to C as they are),
  * every field access on a type-annotated argument (`x::CMP.T` … `x.field`),
  * every `ccall(_fn("cmx_…", FT), …)` signature,
and fails on a field the reference does not have, on an order or count that differs from the C struct, on a `sizeof`
that differs from `CMX_ASSERT_PARAM_STRUCT_SIZES`, and on a `ccall` whose argument types differ from the C prototype.
Rules the file follows so that this check is sound: packing functions take ONE level of a reference struct per
method (`x.field` on an annotated argument, never `x.a.b`), nested structs go through their own `pack` / accessor
method (Julia's dispatch then checks the nested TYPE at run time), and the argument-type tuple of every `ccall` is a
literal.
"""
module CMXExt

import Libdl
import CloudMicrophysics.Parameters as CMP
import CloudMicrophysics.BulkMicrophysicsTendencies as BMT
import CloudMicrophysics.AerosolModel as AM
import CloudMicrophysics.AerosolActivation as AA
import CloudMicrophysics.Quadrature as QUAD
import Thermodynamics as TD
const TDP = TD.Parameters

# ----------------------------------------------------------------------------------------------------------------
# library handle, version check, status
# ----------------------------------------------------------------------------------------------------------------
const CMX_VERSION_MAJOR = 0
const CMX_VERSION_MINOR = 5        # == include/cmx.h; the minor number moves with every layout change / new entry

const _handle = Ref{Ptr{Cvoid}}(C_NULL)
libpath() = get(ENV, "CMX_LIB", "libcmx.so")

"""Open `libcmx.so` once and refuse a library built from another header (`cmx_version`)."""
function handle()
    h = _handle[]
    if h == C_NULL
        h = Libdl.dlopen(libpath())
        v = ccall(Libdl.dlsym(h, :cmx_version), Int32, ())
        want = (Int32(CMX_VERSION_MAJOR) << 16) | Int32(CMX_VERSION_MINOR)
        v == want || error("libcmx.so reports ABI version $(v >> 16).$(v & 0xffff), this binding mirrors $(CMX_VERSION_MAJOR).$(CMX_VERSION_MINOR)")
        _handle[] = h
    end
    return h
end

_sfx(::Type{Float32}) = "_f32"
_sfx(::Type{Float64}) = "_f64"
_fn(name::String, ::Type{FT}) where {FT} = Libdl.dlsym(handle(), Symbol(name * _sfx(FT)))

last_hip_error() = unsafe_string(ccall(Libdl.dlsym(handle(), :cmx_last_hip_error), Cstring, ()))

function _check(st::Int32, what::String)
    st == 0 && return nothing
    msg = st == -1 ? "bad argument" : st == -2 ? "HIP error: " * last_hip_error() : st == -3 ? "unsupported request" : "status $st"
    error("$what: $msg")
end

# device pointers: `nothing` is the C NULL of an optional column
_dp(::Type{FT}, ::Nothing) where {FT} = Ptr{FT}(C_NULL)
_dp(::Type{FT}, p::Ptr{FT}) where {FT} = p
_dp(::Type{FT}, x) where {FT} = convert(Ptr{FT}, pointer(x))
_ptrs(::Type{FT}, cols) where {FT} = Ptr{FT}[_dp(FT, c) for c in cols]
_opt_ptrs(::Type{FT}, ::Nothing) where {FT} = Ptr{Ptr{FT}}(C_NULL)
_opt_ptrs(::Type{FT}, cols) where {FT} = _ptrs(FT, cols)
_ref_or_null(::Nothing) = C_NULL
_ref_or_null(x) = Ref(x)

# ----------------------------------------------------------------------------------------------------------------
# flags (include/cmx.h)
# ----------------------------------------------------------------------------------------------------------------
const CMX_SB2006_LIMITED = UInt32(1) << 0
const CMX_VEL_SB2006 = UInt32(1) << 1
const CMX_VEL_CHEN2022 = UInt32(1) << 2
const CMX_2M_KK2000 = UInt32(0)
const CMX_2M_B1994 = UInt32(1)
const CMX_2M_TC1980 = UInt32(2)
const CMX_2M_LD2004 = UInt32(3)
const CMX_2M_SMOOTH_TRANSITION = UInt32(1) << 8
const CMX_ICENUC_HOM_LINEAR = UInt32(1) << 0
const CMX_ICENUC_ERR_WORDS = 1024
const CMX_P3_INPUT_IS_STATE = UInt32(1) << 0
const CMX_P3_SLOPE_CONSTANT = UInt32(1) << 1
const CMX_P3_NO_ASPECT_RATIO = UInt32(1) << 2
const CMX_P3_RAIN_PDF_LIMITED = UInt32(1) << 3
const CMX_FREEZE_CLOUD_PSD = UInt32(1) << 4
const CMX_PSD_CLOUD = UInt32(1) << 5
const CMX_1M_CLOUD_LIQUID_FORMATION = UInt32(1) << 0
const CMX_1M_CLOUD_ICE_FORMATION_CONST = UInt32(1) << 1
const CMX_1M_CLOUD_ICE_FORMATION_TDEP = UInt32(1) << 2
const CMX_1M_CLOUD_ICE_MELT = UInt32(1) << 3
const CMX_1M_RAIN_ACNV_KESSLER = UInt32(1) << 4
const CMX_1M_RAIN_ACNV_PRESCRIBED_ND = UInt32(1) << 5
const CMX_1M_SNOW_ACNV_NO_SUPERSAT = UInt32(1) << 6
const CMX_1M_SNOW_ACNV_WITH_SUPERSAT = UInt32(1) << 7
const CMX_1M_RAIN_EVAPORATION = UInt32(1) << 8
const CMX_1M_SNOW_SUBLIMATION_ONLY = UInt32(1) << 9
const CMX_1M_SNOW_DEP_AND_SUBL = UInt32(1) << 10
const CMX_1M_SNOW_MELT = UInt32(1) << 11
const CMX_1M_ACCR_LCL_RAI = UInt32(1) << 12
const CMX_1M_ACCR_LCL_SNO = UInt32(1) << 13
const CMX_1M_ACCR_ICL_RAI = UInt32(1) << 14
const CMX_1M_ACCR_ICL_SNO = UInt32(1) << 15
const CMX_1M_ACCR_RAI_SNO = UInt32(1) << 16
const CMX_SB2006_NPROC = 19
const CMX_MP1M_NSRC = 18
const CMX_ARG_MAX_MODES = 8
const CMX_QUAD_MAX = 128
const CMX_COLUMN_SUMS_MAX_COLS = 16
const CMX_COLUMN_SUMS_PARTIALS = 1024

# ----------------------------------------------------------------------------------------------------------------
# Reference structs whose memory layout IS the C layout (immutable, all-FT, declaration order): passed with Ref(x).
# Nested members are listed with the concrete type the reference's constructors put there.
# ----------------------------------------------------------------------------------------------------------------
const DIRECT_LAYOUT = (
    (CMP.CloudParticlePDF_SB2006, (), :cmx_cloud_pdf_sb2006),
    (CMP.RainParticlePDF_SB2006_limited, (), :cmx_rain_pdf_sb2006),
    (CMP.AcnvSB2006, (), :cmx_acnv_sb2006),
    (CMP.AccrSB2006, (), :cmx_accr_sb2006),
    (CMP.SelfColSB2006, (), :cmx_selfcol_sb2006),
    (CMP.BreakupSB2006, (), :cmx_breakup_sb2006),
    (CMP.EvaporationSB2006, (), :cmx_evap_sb2006),
    (CMP.NumberAdjustmentHorn2012, (), :cmx_numadj_horn2012),
    (CMP.AirProperties, (), :cmx_air_properties),
    (CMP.KK2000, (:acnv => CMP.AcnvKK2000, :accr => CMP.AccrKK2000), :cmx_kk2000),
    (CMP.B1994, (:acnv => CMP.AcnvB1994, :accr => CMP.AccrB1994), :cmx_b1994),
    (CMP.TC1980, (:acnv => CMP.AcnvTC1980, :accr => CMP.AccrTC1980), :cmx_tc1980),
    (CMP.LD2004, (), :cmx_ld2004),
    (CMP.StokesRegimeVelType, (), :cmx_stokes_vel),
    (CMP.SB2006VelType, (), :cmx_sb2006_vel),
    (CMP.Chen2022VelTypeRain, (), :cmx_chen2022_rain_vel),
    (CMP.Chen2022VelTypeSmallIce, (), :cmx_chen2022_small_ice_vel),
    (CMP.Chen2022VelTypeLargeIce, (), :cmx_chen2022_large_ice_vel),
    (CMP.Koop2000, (), :cmx_koop2000),
    (CMP.Mohler2006, (), :cmx_mohler2006),
    (CMP.H2SO4SolutionParameters, (), :cmx_h2so4_solution_params),
    (CMP.ParticleMass, (), :cmx_particle_mass),
    (CMP.ParticleArea, (), :cmx_particle_area),
    (CMP.Ventilation, (), :cmx_ventilation),
    (CMP.Acnv1M, (), :cmx_acnv_1m),
    (CMP.VarTimescaleAcnv, (), :cmx_var_timescale_acnv),
    (CMP.CloudLiquid, (), :cmx_cloud_liquid),
    (CMP.CloudIce, (:pdf => CMP.ParticlePDFIceRain, :mass => CMP.ParticleMass), :cmx_cloud_ice),
    (CMP.Rain, (:pdf => CMP.ParticlePDFIceRain, :mass => CMP.ParticleMass, :area => CMP.ParticleArea, :vent => CMP.Ventilation), :cmx_rain),
    (CMP.Snow, (:pdf => CMP.ParticlePDFSnow, :mass => CMP.ParticleMass, :area => CMP.ParticleArea, :vent => CMP.Ventilation, :aspr => CMP.SnowAspectRatio), :cmx_snow),
    (CMP.Blk1MVelTypeRain, (), :cmx_blk1m_vel_rain),
    (CMP.Blk1MVelTypeSnow, (), :cmx_blk1m_vel_snow),
    (CMP.Frostenberg2023, (), :cmx_frostenberg2023),
    (CMP.AerosolActivationParameters, (), :cmx_aerosol_activation_params),
    (CMP.Parameters0M, (), :cmx_parameters_0m),
    (CMP.LocalRimeDensity, (), :cmx_local_rime_density),
    (CMP.RainFreezing, (), :cmx_rain_freezing),
    (CMP.MorrisonMilbrandt2014, (), :cmx_morrison_milbrandt2014),
)

# ----------------------------------------------------------------------------------------------------------------
# Mirror structs (field names = the C field names of include/cmx.h)
# ----------------------------------------------------------------------------------------------------------------
struct CmxThermo{FT}                         # == cmx_thermo
    R_v::FT
    R_d::FT
    cp_d::FT
    cp_v::FT
    cp_l::FT
    cp_i::FT
    LH_v0::FT
    LH_s0::FT
    T_0::FT
    T_triple::FT
    press_triple::FT
    T_freeze::FT
    cv_l::FT
end
"""Thermodynamics.jl's parameter set flattened through its accessors (src/ThermodynamicsInterface.jl:9-25 uses the same ones)."""
CmxThermo(tps::TDP.ThermodynamicsParameters) = CmxThermo(
    TDP.R_v(tps), TDP.R_d(tps), TDP.cp_d(tps), TDP.cp_v(tps), TDP.cp_l(tps), TDP.cp_i(tps), TDP.LH_v0(tps), TDP.LH_s0(tps),
    TDP.T_0(tps), TDP.T_triple(tps), TDP.press_triple(tps), TDP.T_freeze(tps), TDP.cv_l(tps))

struct CmxRainPDF{FT}                        # == cmx_rain_pdf_sb2006
    nu_r::FT
    mu_r::FT
    xr_min::FT
    xr_max::FT
    N0_min::FT
    N0_max::FT
    lambda_min::FT
    lambda_max::FT
    rho_w::FT
    rho_0::FT
end
pack(p::CMP.RainParticlePDF_SB2006_limited) =
    CmxRainPDF(p.νr, p.μr, p.xr_min, p.xr_max, p.N0_min, p.N0_max, p.λ_min, p.λ_max, p.ρw, p.ρ0)
# the not-limited PSD has six fields (src/parameters/Microphysics2M.jl:362-375); the limiter slots are never read
# when CMX_SB2006_LIMITED / CMX_P3_RAIN_PDF_LIMITED is clear
pack(p::CMP.RainParticlePDF_SB2006_notlimited{FT}) where {FT} =
    CmxRainPDF(p.νr, p.μr, p.xr_min, p.xr_max, zero(FT), zero(FT), zero(FT), zero(FT), p.ρw, p.ρ0)

struct CmxSB2006{PDc, FT, AV, AR, SC, BR, EV, NA}     # == cmx_sb2006
    pdf_c::PDc
    pdf_r::CmxRainPDF{FT}
    acnv::AV
    accr::AR
    self::SC
    brek::BR
    evap::EV
    numadj::NA
end
pack(sb::CMP.SB2006) = CmxSB2006(sb.pdf_c, pack(sb.pdf_r), sb.acnv, sb.accr, sb.self, sb.brek, sb.evap, sb.numadj)
is_limited(sb::CMP.SB2006) = CMP.islimited(sb.pdf_r)

_tau_relax(c::CMP.CondEvap2M) = c.τ_relax
_tau_relax(c::CMP.SubDep2M) = c.τ_relax

struct CmxWarmRain2M{SB, AP, FT}             # == cmx_warm_rain_2m
    seifert_beheng::SB
    air_properties::AP
    condevap_tau_relax::FT
    subdep_tau_relax::FT
end
pack(wr::CMP.WarmRainParams2M) =
    CmxWarmRain2M(pack(wr.seifert_beheng), wr.air_properties, _tau_relax(wr.condevap), _tau_relax(wr.subdep))
is_limited(wr::CMP.WarmRainParams2M) = is_limited(wr.seifert_beheng)
_air_properties(wr::CMP.WarmRainParams2M) = wr.air_properties
_warm_rain(mp::CMP.Microphysics2MParams) = mp.warm_rain
_ice(mp::CMP.Microphysics2MParams) = mp.ice

struct CmxRainVel{V1, V2}                    # == cmx_rain_vel
    sb2006::V1
    chen2022::V2
end
"""Both rain fall-speed parameter sets: `CmxRainVel(CMP.SB2006VelType(FT), CMP.Chen2022VelTypeRain(FT))`."""
CmxRainVel(::Type{FT}) where {FT} = CmxRainVel(CMP.SB2006VelType(FT), CMP.Chen2022VelTypeRain(FT))
_vel_flag(::Nothing) = UInt32(0)
_vel_flag(::CMP.SB2006VelType) = CMX_VEL_SB2006
_vel_flag(::CMP.Chen2022VelTypeRain) = CMX_VEL_CHEN2022
_rain_vel(::Nothing, ::Type{FT}) where {FT} = nothing
_rain_vel(v::CMP.SB2006VelType, ::Type{FT}) where {FT} = CmxRainVel(v, CMP.Chen2022VelTypeRain(FT))
_rain_vel(v::CMP.Chen2022VelTypeRain, ::Type{FT}) where {FT} = CmxRainVel(CMP.SB2006VelType(FT), v)

struct CmxBulk2MSchemes{KK, B, TC, LD}       # == cmx_bulk_2m_schemes
    kk2000::KK
    b1994::B
    tc1980::TC
    ld2004::LD
end
CmxBulk2MSchemes(::Type{FT}) where {FT} = CmxBulk2MSchemes(CMP.KK2000(FT), CMP.B1994(FT), CMP.TC1980(FT), CMP.LD2004(FT))
_scheme_id(::CMP.KK2000) = CMX_2M_KK2000
_scheme_id(::CMP.B1994) = CMX_2M_B1994
_scheme_id(::CMP.TC1980) = CMX_2M_TC1980
_scheme_id(::CMP.LD2004) = CMX_2M_LD2004

struct CmxDust{FT}                           # == cmx_abifm_dust
    ABIFM_m::FT
    ABIFM_c::FT
end
# every dust type with ABIFM coefficients (src/parameters/Aerosol*.jl)
CmxDust(d::CMP.Kaolinite) = CmxDust(d.ABIFM_m, d.ABIFM_c)
CmxDust(d::CMP.Illite) = CmxDust(d.ABIFM_m, d.ABIFM_c)
CmxDust(d::CMP.DesertDust) = CmxDust(d.ABIFM_m, d.ABIFM_c)
CmxDust(d::CMP.ArizonaTestDust) = CmxDust(d.ABIFM_m, d.ABIFM_c)
CmxDust(d::CMP.MiddleEasternDust) = CmxDust(d.ABIFM_m, d.ABIFM_c)
CmxDust(d::CMP.AsianDust) = CmxDust(d.ABIFM_m, d.ABIFM_c)
CmxDust(d::CMP.Dust) = CmxDust(d.ABIFM_m, d.ABIFM_c)

struct CmxMohlerDust{FT}                     # == cmx_mohler_dust
    S0_warm::FT
    S0_cold::FT
    a_warm::FT
    a_cold::FT
end
CmxMohlerDust(d::CMP.DesertDust) = CmxMohlerDust(d.S₀_warm, d.S₀_cold, d.a_warm, d.a_cold)
CmxMohlerDust(d::CMP.ArizonaTestDust) = CmxMohlerDust(d.S₀_warm, d.S₀_cold, d.a_warm, d.a_cold)

struct CmxDepositionDust{FT}                 # == cmx_deposition_dust
    deposition_m::FT
    deposition_c::FT
end
CmxDepositionDust(d::CMP.Kaolinite) = CmxDepositionDust(d.deposition_m, d.deposition_c)
CmxDepositionDust(d::CMP.Feldspar) = CmxDepositionDust(d.deposition_m, d.deposition_c)
CmxDepositionDust(d::CMP.Ferrihydrite) = CmxDepositionDust(d.deposition_m, d.deposition_c)
CmxDepositionDust(d::CMP.Illite) = CmxDepositionDust(d.deposition_m, d.deposition_c)
CmxDepositionDust(d::CMP.ArizonaTestDust) = CmxDepositionDust(d.deposition_m, d.deposition_c)
CmxDepositionDust(d::CMP.SaharanDust) = CmxDepositionDust(d.deposition_m, d.deposition_c)
CmxDepositionDust(d::CMP.AsianDust) = CmxDepositionDust(d.deposition_m, d.deposition_c)
CmxDepositionDust(d::CMP.Dust) = CmxDepositionDust(d.deposition_m, d.deposition_c)

# ---- 1-moment scheme ---------------------------------------------------------------------------------------------
struct CmxFrostenberg{FT}                    # == cmx_frostenberg2023
    sigma::FT
    a::FT
    b::FT
    T_freeze::FT
    log_a::FT
end
CmxFrostenberg(f::CMP.Frostenberg2023) = CmxFrostenberg(f.σ, f.a, f.b, f.T_freeze, f.log_a)
CmxFrostenberg(::Type{FT}) where {FT} = CmxFrostenberg(zero(FT), zero(FT), zero(FT), zero(FT), zero(FT))

struct CmxAcnv1M{FT}                         # == cmx_acnv_1m
    tau::FT
    q_threshold::FT
    k::FT
end
CmxAcnv1M(a::CMP.Acnv1M) = CmxAcnv1M(a.τ, a.q_threshold, a.k)
CmxAcnv1M(::Type{FT}) where {FT} = CmxAcnv1M(zero(FT), zero(FT), zero(FT))

struct CmxVarTimescaleAcnv{FT}               # == cmx_var_timescale_acnv
    tau::FT
    alpha::FT
    Nc::FT
end
CmxVarTimescaleAcnv(a::CMP.VarTimescaleAcnv) = CmxVarTimescaleAcnv(a.τ, a.α, a.Nc)
CmxVarTimescaleAcnv(::Type{FT}) where {FT} = CmxVarTimescaleAcnv(zero(FT), zero(FT), zero(FT))

struct CmxProcessParams1M{FT}                # == cmx_process_params_1m
    cloud_liquid_formation_tau_relax::FT
    cloud_ice_formation_tau_relax::FT
    cloud_ice_formation_frostenberg::CmxFrostenberg{FT}
    rain_autoconversion::CmxAcnv1M{FT}
    rain_autoconversion_nd::CmxVarTimescaleAcnv{FT}
    snow_autoconversion::CmxAcnv1M{FT}
    r_ice_snow::FT
    e_lcl_rai::FT
    e_lcl_sno::FT
    e_icl_rai::FT
    e_icl_sno::FT
    e_rai_sno::FT
    coeff_disp::FT
end

# process_params is the NamedTuple built by microphysics_1m_process_params (src/parameters/Microphysics1MOptions.jl:366-380): one
# entry per option field, `nothing` for a disabled or parameter-free process.  Absent parameters are packed as 0 — the kernel reads
# only what the option bits select.
_pp_tau_relax(::Nothing, ::Type{FT}) where {FT} = zero(FT)
_pp_tau_relax(p::NamedTuple, ::Type{FT}) where {FT} = FT(p.τ_relax)
_pp_frostenberg(::Nothing, ::Type{FT}) where {FT} = CmxFrostenberg(FT)
_pp_frostenberg(p::NamedTuple, ::Type{FT}) where {FT} = hasproperty(p, :frostenberg) ? CmxFrostenberg(p.frostenberg) : CmxFrostenberg(FT)
_pp_kessler(p::CMP.Acnv1M, ::Type{FT}) where {FT} = CmxAcnv1M(p)
_pp_kessler(p, ::Type{FT}) where {FT} = CmxAcnv1M(FT)
_pp_prescribed_nd(p::CMP.VarTimescaleAcnv, ::Type{FT}) where {FT} = CmxVarTimescaleAcnv(p)
_pp_prescribed_nd(p, ::Type{FT}) where {FT} = CmxVarTimescaleAcnv(FT)
_pp_snow_acnv(p::CMP.Acnv1M, ::Type{FT}) where {FT} = CmxAcnv1M(p)
_pp_snow_acnv(p, ::Type{FT}) where {FT} = CmxAcnv1M(FT)
_pp_r_ice_snow(p::NamedTuple, ::Type{FT}) where {FT} = FT(p.r_ice_snow)
_pp_r_ice_snow(p, ::Type{FT}) where {FT} = zero(FT)
_pp_e(::Nothing, ::Type{FT}) where {FT} = zero(FT)
_pp_e(p::NamedTuple, ::Type{FT}) where {FT} = FT(p.e)
_pp_coeff_disp(::Nothing, ::Type{FT}) where {FT} = zero(FT)
_pp_coeff_disp(p::NamedTuple, ::Type{FT}) where {FT} = FT(p.coeff_disp)

function pack_process_params(pp::NamedTuple, ::Type{FT}) where {FT}
    return CmxProcessParams1M{FT}(
        _pp_tau_relax(pp.cloud_liquid_formation, FT),
        _pp_tau_relax(pp.cloud_ice_formation, FT),
        _pp_frostenberg(pp.cloud_ice_formation, FT),
        _pp_kessler(pp.rain_autoconversion, FT),
        _pp_prescribed_nd(pp.rain_autoconversion, FT),
        _pp_snow_acnv(pp.snow_autoconversion, FT),
        _pp_r_ice_snow(pp.snow_autoconversion, FT),
        _pp_e(pp.cloud_liquid_rain_accretion, FT),
        _pp_e(pp.cloud_liquid_snow_accretion, FT),
        _pp_e(pp.cloud_ice_rain_accretion, FT),
        _pp_e(pp.cloud_ice_snow_accretion, FT),
        _pp_e(pp.rain_snow_accretion, FT),
        _pp_coeff_disp(pp.rain_snow_accretion, FT),
    )
end

_bit(::Nothing, ::UInt32) = UInt32(0)
_bit(::Any, b::UInt32) = b
_ice_formation_bit(::Nothing) = UInt32(0)
_ice_formation_bit(::CMP.ConstantTimescale) = CMX_1M_CLOUD_ICE_FORMATION_CONST
_ice_formation_bit(::CMP.TemperatureDependent) = CMX_1M_CLOUD_ICE_FORMATION_TDEP
_rain_acnv_bit(::Nothing) = UInt32(0)
_rain_acnv_bit(::CMP.Kessler1M) = CMX_1M_RAIN_ACNV_KESSLER
_rain_acnv_bit(::CMP.PrescribedNd) = CMX_1M_RAIN_ACNV_PRESCRIBED_ND
_snow_acnv_bit(::Nothing) = UInt32(0)
_snow_acnv_bit(::CMP.NoSupersaturation) = CMX_1M_SNOW_ACNV_NO_SUPERSAT
_snow_acnv_bit(::CMP.WithSupersaturation) = CMX_1M_SNOW_ACNV_WITH_SUPERSAT
_snow_subdep_bit(::Nothing) = UInt32(0)
_snow_subdep_bit(::CMP.SublimationOnly) = CMX_1M_SNOW_SUBLIMATION_ONLY
_snow_subdep_bit(::CMP.DepositionAndSublimation) = CMX_1M_SNOW_DEP_AND_SUBL

"""`Microphysics1MOptions` (src/parameters/Microphysics1MOptions.jl:257-286) → the `CMX_1M_*` flag word; `nothing` clears the bit."""
function option_bits(o::CMP.Microphysics1MOptions)
    return _bit(o.cloud_liquid_formation, CMX_1M_CLOUD_LIQUID_FORMATION) |
           _ice_formation_bit(o.cloud_ice_formation) |
           _bit(o.cloud_ice_melt, CMX_1M_CLOUD_ICE_MELT) |
           _rain_acnv_bit(o.rain_autoconversion) |
           _snow_acnv_bit(o.snow_autoconversion) |
           _bit(o.rain_condensation_evaporation, CMX_1M_RAIN_EVAPORATION) |
           _snow_subdep_bit(o.snow_deposition_sublimation) |
           _bit(o.snow_melt, CMX_1M_SNOW_MELT) |
           _bit(o.cloud_liquid_rain_accretion, CMX_1M_ACCR_LCL_RAI) |
           _bit(o.cloud_liquid_snow_accretion, CMX_1M_ACCR_LCL_SNO) |
           _bit(o.cloud_ice_rain_accretion, CMX_1M_ACCR_ICL_RAI) |
           _bit(o.cloud_ice_snow_accretion, CMX_1M_ACCR_ICL_SNO) |
           _bit(o.rain_snow_accretion, CMX_1M_ACCR_RAI_SNO)
end

struct CmxMicrophysics1M{FT, CL, CI, RA, SN, AP, VR, VS}     # == cmx_microphysics_1m
    process_params::CmxProcessParams1M{FT}
    cloud_liquid::CL
    cloud_ice::CI
    rain::RA
    snow::SN
    air_properties::AP
    vel_rain::VR
    vel_snow::VS
end
_liquid(c::CMP.CloudPhaseParams1M) = c.liquid
_ice(c::CMP.CloudPhaseParams1M) = c.ice
_rain(p::CMP.PrecipPhaseParams1M) = p.rain
_snow(p::CMP.PrecipPhaseParams1M) = p.snow
_rain(v::CMP.Blk1MVelType) = v.rain
_snow(v::CMP.Blk1MVelType) = v.snow
_cloud(mp::CMP.Microphysics1MParams) = mp.cloud
_precip(mp::CMP.Microphysics1MParams) = mp.precip
_terminal_velocity(mp::CMP.Microphysics1MParams) = mp.terminal_velocity
_float_type(::CMP.CloudLiquid{FT}) where {FT} = FT

"""`Microphysics1MParams` (src/parameters/Microphysics1MParams.jl:84-91) → `cmx_microphysics_1m_*` (90 FT) and its flag word."""
function pack(mp::CMP.Microphysics1MParams)
    FT = _float_type(_liquid(mp.cloud))
    return CmxMicrophysics1M(pack_process_params(mp.process_params, FT), _liquid(mp.cloud), _ice(mp.cloud), _rain(mp.precip),
        _snow(mp.precip), mp.air_properties, _rain(mp.terminal_velocity), _snow(mp.terminal_velocity))
end
option_bits(mp::CMP.Microphysics1MParams) = option_bits(mp.processes)

# ---- aerosol activation --------------------------------------------------------------------------------------------
struct CmxAerosolMode{FT}                    # == cmx_aerosol_mode
    r_dry::FT
    stdev::FT
    N::FT
    hygroscopicity::FT
    molar_mass_mix::FT
end
struct CmxAerosolDistribution{FT}            # == cmx_aerosol_distribution
    n_modes::Int32
    pad_::Int32
    modes::NTuple{8, CmxAerosolMode{FT}}
end
# Σ_j M_j w_j of a mode (src/AerosolActivation.jl:313); the component tuples may also be scalars (one component)
_molar_mass_mix(m::AM.Mode_B) = sum(m.molar_mass .* m.mass_mix_ratio)
_molar_mass_mix(m::AM.Mode_κ) = sum(m.molar_mass .* m.mass_mix_ratio)
_mode(m::AM.Mode_B, hyg::FT) where {FT} = CmxAerosolMode{FT}(m.r_dry, m.stdev, m.N, hyg, _molar_mass_mix(m))
_mode(m::AM.Mode_κ, hyg::FT) where {FT} = CmxAerosolMode{FT}(m.r_dry, m.stdev, m.N, hyg, _molar_mass_mix(m))

"""`AM.AerosolDistribution` → `cmx_aerosol_distribution_*`: the per-mode mixture is reduced on the host by the reference's own
`AA.mean_hygroscopicity_parameter` (src/AerosolActivation.jl:61-97: mass-weighted B̄ of `Mode_B`, volume-weighted κ̄ of `Mode_κ`)."""
function pack(ad::AM.AerosolDistribution, ap::CMP.AerosolActivationParameters{FT}) where {FT}
    nm = AM.n_modes(ad)
    nm <= CMX_ARG_MAX_MODES || error("cmx_arg2000_*: at most $(CMX_ARG_MAX_MODES) modes")
    hyg = AA.mean_hygroscopicity_parameter(ap, ad)
    z = CmxAerosolMode{FT}(zero(FT), zero(FT), zero(FT), zero(FT), zero(FT))
    modes = ntuple(i -> i <= nm ? _mode(ad.modes[i], FT(hyg[i])) : z, Val(8))
    return CmxAerosolDistribution{FT}(Int32(nm), Int32(0), modes)
end

# ---- P3 ------------------------------------------------------------------------------------------------------------
struct CmxP3Params{FT}                       # == cmx_p3_params
    alpha_va::FT
    beta_va::FT
    gamma::FT
    sigma::FT
    slope_a::FT
    slope_b::FT
    slope_c::FT
    mu_max::FT
    mu_const::FT
    rho_i::FT
    rho_l::FT
    tau_wet::FT
    T_freeze::FT
end
_alpha_va(m::CMP.MassPowerLaw) = m.α_va
_beta_va(m::CMP.MassPowerLaw) = m.β_va
_gamma(a::CMP.AreaPowerLaw) = a.γ
_sigma(a::CMP.AreaPowerLaw) = a.σ
_slope(s::CMP.SlopePowerLaw{FT}) where {FT} = (s.a, s.b, s.c, s.μ_max, zero(FT))
_slope(s::CMP.SlopeConstant{FT}) where {FT} = (zero(FT), zero(FT), zero(FT), zero(FT), s.μ)
_slope_flag(::CMP.SlopePowerLaw) = UInt32(0)
_slope_flag(::CMP.SlopeConstant) = CMX_P3_SLOPE_CONSTANT
_aspect_flag(::CMP.Oblate) = UInt32(0)
_aspect_flag(::CMP.NoAspectRatio) = CMX_P3_NO_ASPECT_RATIO

"""`ParametersP3` (src/parameters/MicrophysicsP3.jl:267-288) → `cmx_p3_params_*`; the slope law and the aspect-ratio treatment are flags."""
function pack(p::CMP.ParametersP3)
    sa, sb, sc, mu_max, mu_const = _slope(p.slope)
    return CmxP3Params(_alpha_va(p.mass), _beta_va(p.mass), _gamma(p.area), _sigma(p.area), sa, sb, sc, mu_max, mu_const,
        p.ρ_i, p.ρ_l, p.τ_wet, p.T_freeze)
end
p3_flags(p::CMP.ParametersP3) = _slope_flag(p.slope) | _aspect_flag(p.aspect_ratio)

struct CmxVentilation{FT}                    # == cmx_ventilation
    a::FT
    b::FT
end
CmxVentilation(v::CMP.VentilationFactor) = CmxVentilation(v.aᵥ, v.bᵥ)
_vent(p::CMP.ParametersP3) = CmxVentilation(p.vent)
_rho_rim_local(p::CMP.ParametersP3) = p.ρ_rim_local

struct CmxChen2022IceVel{SI, LI}             # == cmx_chen2022_ice_vel
    small_ice::SI
    large_ice::LI
end
"""The (small_ice, large_ice) tables of `CMP.Chen2022VelType` (src/parameters/TerminalVelocity.jl:207-257,325-335)."""
pack(v::CMP.Chen2022VelType) = CmxChen2022IceVel(v.small_ice, v.large_ice)
_rain(v::CMP.Chen2022VelType) = v.rain

struct CmxQuadrature{FT}                     # == cmx_quadrature
    n::Int32
    reserved::Int32
    node::NTuple{128, FT}
    weight::NTuple{128, FT}
end
"""A `QUAD.QuadratureRule` (`ChebyshevGauss(n)` or `GaussLegendre(FT, n)`, src/Quadrature.jl:168-252) → `cmx_quadrature_*`: the nodes
yᵢ and the TOTAL weights `inv_weight_fun(yᵢ) · weight(i)` exactly as `QUAD.integrate` forms them (src/Quadrature.jl:62-83)."""
function CmxQuadrature(quad::QUAD.QuadratureRule, ::Type{FT}) where {FT}
    n = _order(quad)
    n <= CMX_QUAD_MAX || error("cmx_quadrature: order $n > $(CMX_QUAD_MAX)")
    y(i) = FT(QUAD.node(quad, FT(i), n))
    w(i) = FT(QUAD.inv_weight_fun(quad, y(i)) * QUAD.weight(quad, FT(i), n))
    return CmxQuadrature{FT}(Int32(n), Int32(0), ntuple(i -> i <= n ? y(i) : zero(FT), Val(128)),
        ntuple(i -> i <= n ? w(i) : zero(FT), Val(128)))
end
_order(q::QUAD.ChebyshevGauss) = q.n
_order(q::QUAD.GaussLegendre) = q.n

struct CmxP3IceParams{FT, RL, VR, VI, PDc, RF}     # == cmx_p3_ice_params
    scheme::CmxP3Params{FT}
    vent::CmxVentilation{FT}
    rho_rim_local::RL
    vel_rain::VR
    vel_ice::VI
    cloud_pdf::PDc
    rain_pdf::CmxRainPDF{FT}
    ice_nucleation::CmxFrostenberg{FT}
    rain_freezing::RF
    tau_act::FT
    quad::CmxQuadrature{FT}
end
_tau_act(m::CMP.NIceProxyDepletion) = m.τ_act
_float_type(::CMP.RainFreezing{FT}) where {FT} = FT

"""`P3IceParams` (src/parameters/Microphysics2MParams.jl:58-106) → `cmx_p3_ice_params_*` (8 B + 354 FT)."""
function pack(ice::CMP.P3IceParams)
    FT = _float_type(ice.rain_freezing)
    return CmxP3IceParams(pack(ice.scheme), _vent(ice.scheme), _rho_rim_local(ice.scheme), _rain(ice.terminal_velocity),
        pack(ice.terminal_velocity), ice.cloud_pdf, pack(ice.rain_pdf), CmxFrostenberg(ice.ice_nucleation), ice.rain_freezing,
        FT(_tau_act(ice.inp_depletion_model)), CmxQuadrature(ice.quad, FT))
end
"""Flag word of the entries that take `cmx_p3_ice_params_*`: rain-PSD variant, slope law, aspect-ratio treatment."""
p3_flags(ice::CMP.P3IceParams) = (CMP.islimited(ice.rain_pdf) ? CMX_P3_RAIN_PDF_LIMITED : UInt32(0)) | p3_flags(ice.scheme)
_scheme(ice::CMP.P3IceParams) = ice.scheme
_vel_ice(ice::CMP.P3IceParams) = pack(ice.terminal_velocity)
_quad(ice::CMP.P3IceParams, ::Type{FT}) where {FT} = CmxQuadrature(ice.quad, FT)

# ----------------------------------------------------------------------------------------------------------------
# run-time layout check (the static twin is tools/check_julia_shim.py): sizeof of every packed struct against the
# field counts of CMX_ASSERT_PARAM_STRUCT_SIZES
# ----------------------------------------------------------------------------------------------------------------
"""A reference struct handed to C as it is must be listed in `DIRECT_LAYOUT`, with the nested member types recorded there."""
function check_direct(x)
    for (T, nested, _) in DIRECT_LAYOUT
        x isa T || continue
        for (name, NT) in nested
            getfield(x, name) isa NT || error("$(typeof(x)).$name is not a $NT")
        end
        return true
    end
    error("$(typeof(x)) is not in DIRECT_LAYOUT")
end

function check_layouts(::Type{FT}) where {FT}
    s = sizeof(FT)
    tps = TDP.ThermodynamicsParameters(FT)
    @assert sizeof(CmxThermo(tps)) == 13s
    mp2 = CMP.Microphysics2MParams(FT; with_ice = true)
    @assert sizeof(pack(_warm_rain(mp2))) == 50s
    @assert sizeof(pack(_ice(mp2))) == 8 + 354s
    @assert sizeof(CmxRainVel(FT)) == 19s
    @assert sizeof(CmxBulk2MSchemes(FT)) == 28s
    @assert sizeof(pack(CMP.Microphysics1MParams(FT))) == 90s
    @assert sizeof(pack(CMP.Microphysics1MParams(FT; cloud_ice_formation = CMP.TemperatureDependent(), rain_autoconversion = CMP.PrescribedNd(),
        snow_autoconversion = CMP.WithSupersaturation(), cloud_ice_melt = nothing))) == 90s
    @assert sizeof(CmxAerosolDistribution{FT}) == 8 + 40s
    @assert sizeof(pack(CMP.ParametersP3(FT))) == 13s
    @assert sizeof(pack(CMP.Chen2022VelType(FT))) == 42s
    @assert sizeof(CmxQuadrature(QUAD.ChebyshevGauss(100), FT)) == 8 + 256s
    @assert sizeof(CMP.Koop2000(FT)) == 8s && sizeof(CMP.AerosolActivationParameters(FT)) == 12s
    @assert sizeof(CMP.StokesRegimeVelType(FT)) == 3s && sizeof(CMP.Parameters0M(FT)) == 3s
    @assert sizeof(CMP.H2SO4SolutionParameters(FT)) == 10s && sizeof(CMP.Mohler2006(FT)) == 2s
    mp1 = CMP.Microphysics1MParams(FT)
    foreach(check_direct, (CMP.KK2000(FT), CMP.B1994(FT), CMP.TC1980(FT), CMP.LD2004(FT), _liquid(_cloud(mp1)), _ice(_cloud(mp1)),
        _rain(_precip(mp1)), _snow(_precip(mp1)), _rain(_terminal_velocity(mp1)), _snow(_terminal_velocity(mp1))))
    return true
end

# ----------------------------------------------------------------------------------------------------------------
# (1) SB2006 two-moment warm rain — src/BulkMicrophysicsTendencies.jl:820-854
# ----------------------------------------------------------------------------------------------------------------
"""
    bulk_microphysics_tendencies!(out, BMT.Microphysics2Moment(), mp, tps, ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai; vel = nothing, stream = C_NULL)

Array form of `BMT.bulk_microphysics_tendencies(::Microphysics2Moment, mp::Microphysics2MParams{WR, Nothing}, …)`.
`out = (; dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt[, vt_rai_n, vt_rai_m])`; with `vel = CMP.SB2006VelType(FT)` or
`CMP.Chen2022VelTypeRain(FT)` the two fall-speed columns of `CM2.rain_terminal_velocity` are written in the same pass.
"""
function bulk_microphysics_tendencies!(out, ::BMT.Microphysics2Moment, mp::CMP.Microphysics2MParams{WR, Nothing}, tps,
    ρ::AbstractArray{FT}, T, q_tot, q_lcl, n_lcl, q_rai, n_rai; vel = nothing, stream = C_NULL) where {WR, FT}
    wr = _warm_rain(mp)
    flags = (is_limited(wr) ? CMX_SB2006_LIMITED : UInt32(0)) | _vel_flag(vel)
    vt_n = hasproperty(out, :vt_rai_n) ? out.vt_rai_n : nothing
    vt_m = hasproperty(out, :vt_rai_m) ? out.vt_rai_m : nothing
    st = ccall(_fn("cmx_sb2006_warm_rain_tendencies", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64,
            Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT},
            Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(wr)), Ref(CmxThermo(tps)), _ref_or_null(_rain_vel(vel, FT)), flags, length(ρ),
        _dp(FT, ρ), _dp(FT, T), _dp(FT, q_tot), _dp(FT, q_lcl), _dp(FT, n_lcl), _dp(FT, q_rai), _dp(FT, n_rai),
        _dp(FT, out.dq_lcl_dt), _dp(FT, out.dn_lcl_dt), _dp(FT, out.dq_rai_dt), _dp(FT, out.dn_rai_dt), _dp(FT, vt_n), _dp(FT, vt_m), stream)
    _check(st, "cmx_sb2006_warm_rain_tendencies")
    return out
end

"""
    bulk_microphysics_tendencies_fields!(BMT.Microphysics2Moment(), mp, tps, n_seg, seg_len, in, in_stride, out, out_stride, out_aos; stream)

The same tendencies on the host model's storage: `in` = 7 device pointers (ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai), each `n_seg` runs of
`seg_len` elements `in_stride[k]` apart (a component of a ClimaCore `VIJFH` field in place: `seg_len = Nv·Ni·Nj`, stride `Nv·Ni·Nj·Nf`,
`n_seg = Nh`); output either `out` (4 pointers) + `out_stride`, or `out_aos` = a device vector of the reference's 8-field NamedTuple rows.
"""
function bulk_microphysics_tendencies_fields!(::BMT.Microphysics2Moment, mp::CMP.Microphysics2MParams, tps, ::Type{FT}, n_seg::Integer,
    seg_len::Integer, in::Vector{Ptr{FT}}, in_stride, out, out_stride, out_aos; stream = C_NULL) where {FT}
    wr = _warm_rain(mp)
    flags = is_limited(wr) ? CMX_SB2006_LIMITED : UInt32(0)
    st = ccall(_fn("cmx_sb2006_warm_rain_tendencies_fields", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Int64, Ptr{Ptr{FT}}, Ptr{Int64}, Ptr{Ptr{FT}}, Ptr{Int64}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(wr)), Ref(CmxThermo(tps)), flags, n_seg, seg_len, in, _strides(in_stride), _opt_ptrs(FT, out), _strides(out_stride),
        _dp(FT, out_aos), stream)
    _check(st, "cmx_sb2006_warm_rain_tendencies_fields")
    return nothing
end
_strides(::Nothing) = Ptr{Int64}(C_NULL)
_strides(s) = convert(Vector{Int64}, s)

"""
    sb2006_process_rates!(out, mp, tps, q_tot, q_lcl, q_rai, N_lcl, N_rai, ρ, T; vel = nothing, stream)

The individual CM2 process rates (the reference's `SB2006_2M_kernel`, test/gpu_tests.jl:220-244); `out` = `CMX_SB2006_NPROC` (19) device
columns or `nothing`s in the order of `cmx_sb2006_process_column`; N in 1/m³.
"""
function sb2006_process_rates!(out, mp::CMP.Microphysics2MParams, tps, q_tot::AbstractArray{FT}, q_lcl, q_rai, N_lcl, N_rai, ρ, T;
    vel = nothing, stream = C_NULL) where {FT}
    length(out) == CMX_SB2006_NPROC || error("out: $(CMX_SB2006_NPROC) columns (or nothing) expected")
    wr = _warm_rain(mp)
    flags = (is_limited(wr) ? CMX_SB2006_LIMITED : UInt32(0)) | _vel_flag(vel)
    st = ccall(_fn("cmx_sb2006_process_rates", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Ptr{FT}}, Ptr{Cvoid}),
        Ref(pack(wr)), Ref(CmxThermo(tps)), _ref_or_null(_rain_vel(vel, FT)), flags, length(q_tot),
        _dp(FT, q_tot), _dp(FT, q_lcl), _dp(FT, q_rai), _dp(FT, N_lcl), _dp(FT, N_rai), _dp(FT, ρ), _dp(FT, T), _ptrs(FT, out), stream)
    _check(st, "cmx_sb2006_process_rates")
    return out
end

"""`CM2.cloud_terminal_velocity.(Ref(pdf_c), Ref(vel), q_liq, ρ, N_liq)` (src/Microphysics2M.jl:647-664); either output may be `nothing`."""
function cloud_terminal_velocity!(vt_n, vt_m, pdf_c::CMP.CloudParticlePDF_SB2006{FT}, vel::CMP.StokesRegimeVelType{FT}, q_liq, ρ, N_liq;
    stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_sb2006_cloud_terminal_velocity", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pdf_c), Ref(vel), length(q_liq), _dp(FT, q_liq), _dp(FT, ρ), _dp(FT, N_liq), _dp(FT, vt_n), _dp(FT, vt_m), stream)
    _check(st, "cmx_sb2006_cloud_terminal_velocity")
    return nothing
end

"""
    column_tendencies_sedimentation!(out, BMT.Microphysics2Moment(), mp, tps, vel, cloud_vel, n_col, n_lev, inv_dz, ρ, …; precip_flux, stream)

The fused column step (tendencies + fall speeds + first-order upwind sedimentation; include/cmx.h (2b)).  `vel` is required
(`CMP.SB2006VelType` or `CMP.Chen2022VelTypeRain`), `cloud_vel = CMP.StokesRegimeVelType(FT)` or `nothing`.
"""
function column_tendencies_sedimentation!(out, ::BMT.Microphysics2Moment, mp::CMP.Microphysics2MParams, tps, vel, cloud_vel, n_col::Integer,
    n_lev::Integer, inv_dz::AbstractArray{FT}, ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai; precip_flux = nothing, stream = C_NULL) where {FT}
    vel === nothing && error("the column step needs a rain fall-speed scheme (CMP.SB2006VelType or CMP.Chen2022VelTypeRain)")
    wr = _warm_rain(mp)
    flags = (is_limited(wr) ? CMX_SB2006_LIMITED : UInt32(0)) | _vel_flag(vel)
    st = ccall(_fn("cmx_sb2006_column_tendencies_sedimentation", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Int32, Ptr{FT},
            Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(wr)), Ref(CmxThermo(tps)), Ref(_rain_vel(vel, FT)), _ref_or_null(cloud_vel), flags, n_col, n_lev, _dp(FT, inv_dz),
        _dp(FT, ρ), _dp(FT, T), _dp(FT, q_tot), _dp(FT, q_lcl), _dp(FT, n_lcl), _dp(FT, q_rai), _dp(FT, n_rai),
        _dp(FT, out.dq_lcl_dt), _dp(FT, out.dn_lcl_dt), _dp(FT, out.dq_rai_dt), _dp(FT, out.dn_rai_dt), _dp(FT, precip_flux), stream)
    _check(st, "cmx_sb2006_column_tendencies_sedimentation")
    return out
end

"""
    bulk_2m_cloud_to_rain!(acnv, accr, scheme, q_lcl, q_rai, ρ, N_d; smooth_transition = false, stream)

`CM2.conv_q_lcl_to_q_rai.(Ref(scheme), q_lcl, ρ, N_d)` and `CM2.accretion.(Ref(scheme), q_lcl, q_rai, ρ)` for `scheme` a
`CMP.KK2000 | B1994 | TC1980 | LD2004` (src/Microphysics2M.jl:920-1003); `accr` / `q_rai` may be `nothing`.
"""
function bulk_2m_cloud_to_rain!(acnv, accr, scheme, q_lcl::AbstractArray{FT}, q_rai, ρ, N_d; smooth_transition = false, stream = C_NULL) where {FT}
    flags = _scheme_id(scheme) | (smooth_transition ? CMX_2M_SMOOTH_TRANSITION : UInt32(0))
    st = ccall(_fn("cmx_bulk_2m_cloud_to_rain", FT), Int32,
        (Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(_schemes_with(scheme, FT)), flags, length(q_lcl), _dp(FT, q_lcl), _dp(FT, q_rai), _dp(FT, ρ), _dp(FT, N_d), _dp(FT, acnv), _dp(FT, accr), stream)
    _check(st, "cmx_bulk_2m_cloud_to_rain")
    return nothing
end
_schemes_with(s::CMP.KK2000, ::Type{FT}) where {FT} = CmxBulk2MSchemes(s, CMP.B1994(FT), CMP.TC1980(FT), CMP.LD2004(FT))
_schemes_with(s::CMP.B1994, ::Type{FT}) where {FT} = CmxBulk2MSchemes(CMP.KK2000(FT), s, CMP.TC1980(FT), CMP.LD2004(FT))
_schemes_with(s::CMP.TC1980, ::Type{FT}) where {FT} = CmxBulk2MSchemes(CMP.KK2000(FT), CMP.B1994(FT), s, CMP.LD2004(FT))
_schemes_with(s::CMP.LD2004, ::Type{FT}) where {FT} = CmxBulk2MSchemes(CMP.KK2000(FT), CMP.B1994(FT), CMP.TC1980(FT), s)

# ----------------------------------------------------------------------------------------------------------------
# (4) ice nucleation — src/IceNucleation.jl, src/Common.jl
# ----------------------------------------------------------------------------------------------------------------
"""
    ice_nucleation_rates!(out, tps, dust, koop, T, a_w, r; linear = false, errs = nothing, stream)

ABIFM immersion freezing + Koop-2000 homogeneous freezing; `out = (; Δa_w, J_het, J_hom, rate_het, rate_hom)` (any may be `nothing`);
`errs` = zeroed device `Int64[CMX_ICENUC_ERR_WORDS]`: `sum(errs)` after the call = the points where the scalar API throws its domain error.
"""
function ice_nucleation_rates!(out, tps, dust, koop::CMP.Koop2000{FT}, T, a_w, r; linear = false, errs = nothing, stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_ice_nucleation_rates", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Int64}, Ptr{Cvoid}),
        Ref(CmxThermo(tps)), Ref(CmxDust(dust)), Ref(koop), linear ? CMX_ICENUC_HOM_LINEAR : UInt32(0), length(T),
        _dp(FT, T), _dp(FT, a_w), _dp(FT, r), _dp(FT, out.Δa_w), _dp(FT, out.J_het), _dp(FT, out.J_hom), _dp(FT, out.rate_het), _dp(FT, out.rate_hom),
        _dp(Int64, errs), stream)
    _check(st, "cmx_ice_nucleation_rates")
    return out
end

"""The same with a_w of an H2SO4 solution droplet formed in the kernel from the weight fraction `x` (parcel/ParcelTendencies.jl:120-133)."""
function ice_nucleation_rates_xT!(out, tps, dust, koop::CMP.Koop2000{FT}, h2so4::CMP.H2SO4SolutionParameters{FT}, T, x, r; linear = false,
    errs = nothing, stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_ice_nucleation_rates_xT", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Int64},
            Ptr{Cvoid}),
        Ref(CmxThermo(tps)), Ref(CmxDust(dust)), Ref(koop), Ref(h2so4), linear ? CMX_ICENUC_HOM_LINEAR : UInt32(0), length(T),
        _dp(FT, T), _dp(FT, x), _dp(FT, r), _dp(FT, out.Δa_w), _dp(FT, out.J_het), _dp(FT, out.J_hom), _dp(FT, out.rate_het), _dp(FT, out.rate_hom),
        _dp(Int64, errs), stream)
    _check(st, "cmx_ice_nucleation_rates_xT")
    return out
end

"""`CO.H2SO4_soln_saturation_vapor_pressure.(Ref(prs), x, T)` and `CO.a_w_xT.(Ref(prs), Ref(tps), x, T)` (src/Common.jl:188-246)."""
function h2so4_solution!(p_sol, a_w, prs::CMP.H2SO4SolutionParameters{FT}, tps, x, T; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_h2so4_solution", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(prs), Ref(CmxThermo(tps)), length(x), _dp(FT, x), _dp(FT, T), _dp(FT, p_sol), _dp(FT, a_w), stream)
    _check(st, "cmx_h2so4_solution")
    return nothing
end

"""`CMI_het.dust_activated_number_fraction` and `CMI_het.MohlerDepositionRate` (src/IceNucleation.jl:44-79); `errs` = one zeroed device Int64."""
function mohler2006_deposition!(act_frac, dep_rate, dust, ip::CMP.Mohler2006{FT}, S_i, T, dSi_dt, N_aer; errs = nothing, stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_mohler2006_deposition", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Int64}, Ptr{Cvoid}),
        Ref(CmxMohlerDust(dust)), Ref(ip), length(S_i), _dp(FT, S_i), _dp(FT, T), _dp(FT, dSi_dt), _dp(FT, N_aer), _dp(FT, act_frac), _dp(FT, dep_rate),
        _dp(Int64, errs), stream)
    _check(st, "cmx_mohler2006_deposition")
    return nothing
end

"""`CMI_het.deposition_J.(Ref(dust), Δa_w)` (src/IceNucleation.jl:81-102)."""
function deposition_J!(J, dust, Δa_w::AbstractArray{FT}; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_deposition_J", FT), Int32, (Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(CmxDepositionDust(dust)), length(Δa_w), _dp(FT, Δa_w), _dp(FT, J), stream)
    _check(st, "cmx_deposition_J")
    return J
end

"""`CMI_het.INP_concentration_frequency.(Ref(ip), INPC, T)` (src/IceNucleation.jl:219-226)."""
function inp_concentration_frequency!(freq, ip::CMP.Frostenberg2023{FT}, INPC, T; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_inp_concentration_frequency", FT), Int32, (Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(ip), length(T), _dp(FT, INPC), _dp(FT, T), _dp(FT, freq), stream)
    _check(st, "cmx_inp_concentration_frequency")
    return freq
end

"""`CO.a_w_ice.(Ref(tps), T)` and `CO.a_w_eT.(Ref(tps), e, T)` (src/Common.jl:250-271); `e` / `a_w_eT` may be `nothing`."""
function water_activity!(a_w_ice, a_w_eT, tps, T::AbstractArray{FT}, e; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_water_activity", FT), Int32, (Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(CmxThermo(tps)), length(T), _dp(FT, T), _dp(FT, e), _dp(FT, a_w_ice), _dp(FT, a_w_eT), stream)
    _check(st, "cmx_water_activity")
    return nothing
end

# ----------------------------------------------------------------------------------------------------------------
# (5) one-moment scheme — src/BulkMicrophysicsTendencies.jl:141-252,505-632
# ----------------------------------------------------------------------------------------------------------------
"""
    bulk_microphysics_tendencies!(out, BMT.Instantaneous(), BMT.Microphysics1Moment(), mp, tps, ρ, T, q_tot, q_lcl, q_icl, q_rai, q_sno; stream)

`out = (; dq_lcl_dt, dq_icl_dt, dq_rai_dt, dq_sno_dt)`.  Every `Microphysics1MOptions` combination is accepted (`option_bits`).
"""
function bulk_microphysics_tendencies!(out, ::BMT.Instantaneous, ::BMT.Microphysics1Moment, mp::CMP.Microphysics1MParams, tps,
    ρ::AbstractArray{FT}, T, q_tot, q_lcl, q_icl, q_rai, q_sno; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_mp1m_tendencies", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(mp)), Ref(CmxThermo(tps)), option_bits(mp), length(ρ),
        _dp(FT, ρ), _dp(FT, T), _dp(FT, q_tot), _dp(FT, q_lcl), _dp(FT, q_icl), _dp(FT, q_rai), _dp(FT, q_sno),
        _dp(FT, out.dq_lcl_dt), _dp(FT, out.dq_icl_dt), _dp(FT, out.dq_rai_dt), _dp(FT, out.dq_sno_dt), stream)
    _check(st, "cmx_mp1m_tendencies")
    return out
end

"""The Instantaneous 1-moment tendencies on the host model's storage (segmented columns in place; SoA or array-of-NamedTuple output)."""
function bulk_microphysics_tendencies_fields!(::BMT.Instantaneous, ::BMT.Microphysics1Moment, mp::CMP.Microphysics1MParams, tps, ::Type{FT},
    n_seg::Integer, seg_len::Integer, in::Vector{Ptr{FT}}, in_stride, out, out_stride, out_aos; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_mp1m_tendencies_fields", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Int64, Ptr{Ptr{FT}}, Ptr{Int64}, Ptr{Ptr{FT}}, Ptr{Int64}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(mp)), Ref(CmxThermo(tps)), option_bits(mp), n_seg, seg_len, in, _strides(in_stride), _opt_ptrs(FT, out), _strides(out_stride),
        _dp(FT, out_aos), stream)
    _check(st, "cmx_mp1m_tendencies_fields")
    return nothing
end

"""
    bulk_microphysics_tendencies!(out, BMT.LinearizedAverage(), BMT.Microphysics1Moment(), mp, tps, ρ, T, q_tot, q_lcl, q_icl, q_rai, q_sno, Δt, nsub; stream)

The operational mode (src/BulkMicrophysicsTendencies.jl:572-632); `q_min = TD.Parameters.q_min(tps)` as in the reference (:395).
"""
function bulk_microphysics_tendencies!(out, ::BMT.LinearizedAverage, ::BMT.Microphysics1Moment, mp::CMP.Microphysics1MParams, tps,
    ρ::AbstractArray{FT}, T, q_tot, q_lcl, q_icl, q_rai, q_sno, Δt, nsub::Integer; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_mp1m_linearized_average", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, FT, FT, Int32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT},
            Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(mp)), Ref(CmxThermo(tps)), option_bits(mp), FT(TDP.q_min(tps)), FT(Δt), nsub, length(ρ),
        _dp(FT, ρ), _dp(FT, T), _dp(FT, q_tot), _dp(FT, q_lcl), _dp(FT, q_icl), _dp(FT, q_rai), _dp(FT, q_sno),
        _dp(FT, out.dq_lcl_dt), _dp(FT, out.dq_icl_dt), _dp(FT, out.dq_rai_dt), _dp(FT, out.dq_sno_dt), stream)
    _check(st, "cmx_mp1m_linearized_average")
    return out
end

"""LinearizedAverage on the host model's storage."""
function bulk_microphysics_tendencies_fields!(::BMT.LinearizedAverage, ::BMT.Microphysics1Moment, mp::CMP.Microphysics1MParams, tps, ::Type{FT},
    Δt, nsub::Integer, n_seg::Integer, seg_len::Integer, in::Vector{Ptr{FT}}, in_stride, out, out_stride, out_aos; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_mp1m_linearized_average_fields", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, FT, FT, Int32, Int64, Int64, Ptr{Ptr{FT}}, Ptr{Int64}, Ptr{Ptr{FT}}, Ptr{Int64}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(mp)), Ref(CmxThermo(tps)), option_bits(mp), FT(TDP.q_min(tps)), FT(Δt), nsub, n_seg, seg_len, in, _strides(in_stride),
        _opt_ptrs(FT, out), _strides(out_stride), _dp(FT, out_aos), stream)
    _check(st, "cmx_mp1m_linearized_average_fields")
    return nothing
end

"""
    column_tendencies_sedimentation!(out, mode, BMT.Microphysics1Moment(), mp, tps, stokes, chen, n_col, n_lev, inv_dz, cols, Δt, nsub; …)

The operational 1-moment column step in one pass.  `mode = BMT.Instantaneous()` (then `Δt`, `nsub` are ignored) or `BMT.LinearizedAverage()`;
`cols` = the 7 state columns (ρ, T, q_tot, q_lcl, q_icl, q_rai, q_sno), `out` = the 4 tendency columns; `stokes = CMP.StokesRegimeVelType(FT)`,
`chen = CMP.Chen2022VelType(FT)`.
"""
function column_tendencies_sedimentation!(out, mode, ::BMT.Microphysics1Moment, mp::CMP.Microphysics1MParams, tps,
    stokes::CMP.StokesRegimeVelType{FT}, chen::CMP.Chen2022VelType, n_col::Integer, n_lev::Integer, inv_dz, cols, Δt = zero(FT), nsub::Integer = 0;
    precip_rai = nothing, precip_sno = nothing, stream = C_NULL) where {FT}
    ns = mode isa BMT.LinearizedAverage ? Int32(nsub) : Int32(0)
    st = ccall(_fn("cmx_mp1m_column_tendencies_sedimentation", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, FT, FT, Int32, Int64, Int32, Ptr{FT}, Ptr{Ptr{FT}}, Ptr{Ptr{FT}}, Ptr{FT},
            Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(mp)), Ref(CmxThermo(tps)), Ref(stokes), Ref(_rain(chen)), Ref(pack(chen)), option_bits(mp), FT(TDP.q_min(tps)), FT(Δt), ns,
        n_col, n_lev, _dp(FT, inv_dz), _ptrs(FT, cols), _ptrs(FT, out), _dp(FT, precip_rai), _dp(FT, precip_sno), stream)
    _check(st, "cmx_mp1m_column_tendencies_sedimentation")
    return out
end

"""The 18 individual source terms of `_microphysics_source_terms` (src/BulkMicrophysicsTendencies.jl:141-217); `out` = 18 columns or `nothing`s."""
function mp1m_source_terms!(out, mp::CMP.Microphysics1MParams, tps, ρ::AbstractArray{FT}, T, q_tot, q_lcl, q_icl, q_rai, q_sno; stream = C_NULL) where {FT}
    length(out) == CMX_MP1M_NSRC || error("out: $(CMX_MP1M_NSRC) columns (or nothing) expected")
    st = ccall(_fn("cmx_mp1m_source_terms", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Ptr{FT}}, Ptr{Cvoid}),
        Ref(pack(mp)), Ref(CmxThermo(tps)), option_bits(mp), length(ρ),
        _dp(FT, ρ), _dp(FT, T), _dp(FT, q_tot), _dp(FT, q_lcl), _dp(FT, q_icl), _dp(FT, q_rai), _dp(FT, q_sno), _ptrs(FT, out), stream)
    _check(st, "cmx_mp1m_source_terms")
    return out
end

"""`CM1.terminal_velocity.(Ref(rain | snow), Ref(vel), ρ, q)` for Blk1M rain / snow and Chen-2022 rain (src/Microphysics1M.jl:223-270)."""
function mp1m_terminal_velocity!(vt_rai_blk1m, vt_sno_blk1m, vt_rai_chen, mp::CMP.Microphysics1MParams, chen, ρ::AbstractArray{FT}, q_rai, q_sno;
    stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_mp1m_terminal_velocity", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(mp)), _ref_or_null(chen), length(ρ), _dp(FT, ρ), _dp(FT, q_rai), _dp(FT, q_sno), _dp(FT, vt_rai_blk1m), _dp(FT, vt_sno_blk1m),
        _dp(FT, vt_rai_chen), stream)
    _check(st, "cmx_mp1m_terminal_velocity")
    return nothing
end

"""The four bulk sedimentation velocities a host model precomputes (test/gpu_clima_core_test.jl:36-45); any (q, w) pair may be `nothing`."""
function sedimentation_velocities!(w_lcl, w_icl, w_rai, w_sno, mp::CMP.Microphysics1MParams, stokes::CMP.StokesRegimeVelType{FT},
    chen::CMP.Chen2022VelType, ρ, q_lcl, q_icl, q_rai, q_sno; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_sedimentation_velocities", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(mp)), Ref(stokes), Ref(_rain(chen)), Ref(pack(chen)), length(ρ), _dp(FT, ρ), _dp(FT, q_lcl), _dp(FT, q_icl), _dp(FT, q_rai), _dp(FT, q_sno),
        _dp(FT, w_lcl), _dp(FT, w_icl), _dp(FT, w_rai), _dp(FT, w_sno), stream)
    _check(st, "cmx_sedimentation_velocities")
    return nothing
end

# ----------------------------------------------------------------------------------------------------------------
# (6) ARG2000 aerosol activation — src/AerosolActivation.jl:138-433
# ----------------------------------------------------------------------------------------------------------------
"""
    aerosol_activation!(N_act, M_act, S_max, ap, ad, aip, tps, T, p, w, q_tot; q_liq, q_ice, N_liq, N_ice, stream)

`AA.N_activated_per_mode`, `AA.M_activated_per_mode`, `AA.max_supersaturation` for a distribution shared by all states.  `N_act` / `M_act` =
one device column per mode (or `nothing`), `S_max` a column or `nothing`.
"""
function aerosol_activation!(N_act, M_act, S_max, ap::CMP.AerosolActivationParameters{FT}, ad::AM.AerosolDistribution, aip::CMP.AirProperties{FT}, tps,
    T, p, w, q_tot; q_liq = nothing, q_ice = nothing, N_liq = nothing, N_ice = nothing, stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_arg2000_activation", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Ptr{FT}}, Ptr{Ptr{FT}},
            Ptr{FT}, Ptr{Cvoid}),
        Ref(ap), Ref(pack(ad, ap)), Ref(aip), Ref(CmxThermo(tps)), length(T), _dp(FT, T), _dp(FT, p), _dp(FT, w), _dp(FT, q_tot),
        _dp(FT, q_liq), _dp(FT, q_ice), _dp(FT, N_liq), _dp(FT, N_ice), _opt_ptrs(FT, N_act), _opt_ptrs(FT, M_act), _dp(FT, S_max), stream)
    _check(st, "cmx_arg2000_activation")
    return nothing
end

"""`AA.total_N_activated.(…)`, `AA.total_M_activated.(…)` (src/AerosolActivation.jl:355-433)."""
function total_activated!(N_total, M_total, ap::CMP.AerosolActivationParameters{FT}, ad::AM.AerosolDistribution, aip::CMP.AirProperties{FT}, tps,
    T, p, w, q_tot; q_liq = nothing, q_ice = nothing, N_liq = nothing, N_ice = nothing, stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_arg2000_total_activated", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(ap), Ref(pack(ad, ap)), Ref(aip), Ref(CmxThermo(tps)), length(T), _dp(FT, T), _dp(FT, p), _dp(FT, w), _dp(FT, q_tot),
        _dp(FT, q_liq), _dp(FT, q_ice), _dp(FT, N_liq), _dp(FT, N_ice), _dp(FT, N_total), _dp(FT, M_total), stream)
    _check(st, "cmx_arg2000_total_activated")
    return nothing
end

"""
    aerosol_activation_columns!(N_act, M_act, S_max, ap, aip, tps, T, p, w, q_tot, r_dry, stdev, N_mode, hygroscopicity, molar_mass; …)

Aerosol that varies in space — the form of the reference's own GPU test (`aerosol_activation_kernel!`, test/gpu_tests.jl:45-79): each of
`r_dry … molar_mass` is a vector of per-mode device columns (`molar_mass` may be `nothing` unless `M_act` is wanted); `hygroscopicity` is the
mode's B̄ or κ̄ (`AA.mean_hygroscopicity_parameter`).
"""
function aerosol_activation_columns!(N_act, M_act, S_max, ap::CMP.AerosolActivationParameters{FT}, aip::CMP.AirProperties{FT}, tps, T, p, w, q_tot,
    r_dry, stdev, N_mode, hygroscopicity, molar_mass; q_liq = nothing, q_ice = nothing, N_liq = nothing, N_ice = nothing, stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_arg2000_activation_columns", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT},
            Ptr{Ptr{FT}}, Ptr{Ptr{FT}}, Ptr{Ptr{FT}}, Ptr{Ptr{FT}}, Ptr{Ptr{FT}}, Ptr{Ptr{FT}}, Ptr{Ptr{FT}}, Ptr{FT}, Ptr{Cvoid}),
        Ref(ap), Ref(aip), Ref(CmxThermo(tps)), length(r_dry), length(T), _dp(FT, T), _dp(FT, p), _dp(FT, w), _dp(FT, q_tot),
        _dp(FT, q_liq), _dp(FT, q_ice), _dp(FT, N_liq), _dp(FT, N_ice), _ptrs(FT, r_dry), _ptrs(FT, stdev), _ptrs(FT, N_mode), _ptrs(FT, hygroscopicity),
        _opt_ptrs(FT, molar_mass), _opt_ptrs(FT, N_act), _opt_ptrs(FT, M_act), _dp(FT, S_max), stream)
    _check(st, "cmx_arg2000_activation_columns")
    return nothing
end

# ----------------------------------------------------------------------------------------------------------------
# (7)–(9) P3 — src/P3_*.jl, src/BulkMicrophysicsTendencies.jl:898-1083
# ----------------------------------------------------------------------------------------------------------------
"""
    p3_shape!(out, params, ρq_ice, ρn_ice, ρq_rim, ρb_rim; logλ_guess = nothing, brent_iters = 0, stream)

`P3.get_distribution_logλ(P3.state_from_prognostic(params, …))`, `P3.D_m`, `P3.get_logN₀`; `out = (; F_rim, ρ_rim, logλ, D_m, logN₀)`, any may be `nothing`.
"""
function p3_shape!(out, params::CMP.ParametersP3, ρq_ice::AbstractArray{FT}, ρn_ice, ρq_rim, ρb_rim; logλ_guess = nothing, brent_iters::Integer = 0,
    stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_p3_shape", FT), Int32,
        (Ptr{Cvoid}, UInt32, Int32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(params)), p3_flags(params), brent_iters, length(ρq_ice), _dp(FT, ρq_ice), _dp(FT, ρn_ice), _dp(FT, ρq_rim), _dp(FT, ρb_rim),
        _dp(FT, logλ_guess), _dp(FT, out.F_rim), _dp(FT, out.ρ_rim), _dp(FT, out.logλ), _dp(FT, out.D_m), _dp(FT, out.logN₀), stream)
    _check(st, "cmx_p3_shape")
    return out
end

"""`P3.ice_terminal_velocity_number_weighted_from_prognostic` / `_mass_weighted_` (src/P3_terminal_velocity.jl:72-178) at the given `logλ`."""
function p3_terminal_velocities!(v_n, v_m, params::CMP.ParametersP3, vel::CMP.Chen2022VelType, quad::QUAD.QuadratureRule, ρq_ice::AbstractArray{FT},
    ρn_ice, ρq_rim, ρb_rim, ρₐ, logλ; p = 1e-6, stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_p3_terminal_velocities", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, FT, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(params)), Ref(pack(vel)), Ref(CmxQuadrature(quad, FT)), p3_flags(params), FT(p), length(ρq_ice), _dp(FT, ρq_ice), _dp(FT, ρn_ice),
        _dp(FT, ρq_rim), _dp(FT, ρb_rim), _dp(FT, ρₐ), _dp(FT, logλ), _dp(FT, v_n), _dp(FT, v_m), stream)
    _check(st, "cmx_p3_terminal_velocities")
    return nothing
end

"""BASELINE config 5 as one launch: shape solve + D_m + both weighted fall speeds; `out = (; logλ, D_m, v_n, v_m)`."""
function p3_shape_terminal_velocities!(out, params::CMP.ParametersP3, vel::CMP.Chen2022VelType, quad::QUAD.QuadratureRule,
    ρq_ice::AbstractArray{FT}, ρn_ice, ρq_rim, ρb_rim, ρₐ; logλ_guess = nothing, brent_iters::Integer = 0, p = 1e-6, stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_p3_shape_terminal_velocities", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int32, FT, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT},
            Ptr{Cvoid}),
        Ref(pack(params)), Ref(pack(vel)), Ref(CmxQuadrature(quad, FT)), p3_flags(params), brent_iters, FT(p), length(ρq_ice), _dp(FT, ρq_ice),
        _dp(FT, ρn_ice), _dp(FT, ρq_rim), _dp(FT, ρb_rim), _dp(FT, ρₐ), _dp(FT, logλ_guess), _dp(FT, out.logλ), _dp(FT, out.D_m), _dp(FT, out.v_n),
        _dp(FT, out.v_m), stream)
    _check(st, "cmx_p3_shape_terminal_velocities")
    return out
end

"""`P3.ice_melt.(Ref(vel), Ref(aps), Ref(tps), T, ρₐ, state, logλ)` (src/P3_processes.jl:64-94)."""
function p3_ice_melt!(dNdt, dLdt, params::CMP.ParametersP3, vel::CMP.Chen2022VelType, aps::CMP.AirProperties{FT}, tps, quad::QUAD.QuadratureRule,
    ρq_ice, ρn_ice, ρq_rim, ρb_rim, ρₐ, T, logλ; p = 1e-6, stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_p3_ice_melt", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, FT, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT},
            Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(params)), Ref(pack(vel)), Ref(aps), Ref(CmxThermo(tps)), Ref(_vent(params)), Ref(CmxQuadrature(quad, FT)), p3_flags(params), FT(p),
        length(T), _dp(FT, ρq_ice), _dp(FT, ρn_ice), _dp(FT, ρq_rim), _dp(FT, ρb_rim), _dp(FT, ρₐ), _dp(FT, T), _dp(FT, logλ), _dp(FT, dNdt),
        _dp(FT, dLdt), stream)
    _check(st, "cmx_p3_ice_melt")
    return nothing
end

"""`P3.ice_self_collection.(state, logλ, Ref(vel), ρₐ)` (src/P3_processes.jl:676-712; the reference's `benchmark_p3_kernel!`)."""
function p3_ice_self_collection!(dNdt, params::CMP.ParametersP3, vel::CMP.Chen2022VelType, quad::QUAD.QuadratureRule, ρq_ice::AbstractArray{FT},
    ρn_ice, ρq_rim, ρb_rim, ρₐ, logλ; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_p3_ice_self_collection", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(params)), Ref(pack(vel)), Ref(CmxQuadrature(quad, FT)), p3_flags(params), length(ρq_ice), _dp(FT, ρq_ice), _dp(FT, ρn_ice),
        _dp(FT, ρq_rim), _dp(FT, ρb_rim), _dp(FT, ρₐ), _dp(FT, logλ), _dp(FT, dNdt), stream)
    _check(st, "cmx_p3_ice_self_collection")
    return dNdt
end

"""`P3.het_ice_nucleation.(Ref(aerosol), Ref(tps), q_lcl, N_lcl, RH, T, ρₐ)` (src/P3_processes.jl:20-46)."""
function p3_het_ice_nucleation!(dNdt, dLdt, aerosol, tps, q_lcl::AbstractArray{FT}, N_lcl, RH, T, ρₐ; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_p3_het_ice_nucleation", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(CmxDust(aerosol)), Ref(CmxThermo(tps)), length(T), _dp(FT, q_lcl), _dp(FT, N_lcl), _dp(FT, RH), _dp(FT, T), _dp(FT, ρₐ), _dp(FT, dNdt),
        _dp(FT, dLdt), stream)
    _check(st, "cmx_p3_het_ice_nucleation")
    return nothing
end

"""`CMI_het.liquid_freezing_rate.(Ref(rf), Ref(pdf), Ref(tps), q, ρ, N, T)` (src/IceNucleation.jl:274-389); `cloud = true` selects the cloud PSD."""
function liquid_freezing_rate!(dn_frz, dq_frz, ice::CMP.P3IceParams, tps, q::AbstractArray{FT}, ρ, N, T; cloud = false, stream = C_NULL) where {FT}
    flags = p3_flags(ice) | (cloud ? CMX_FREEZE_CLOUD_PSD : UInt32(0))
    st = ccall(_fn("cmx_liquid_freezing_rate", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pack(ice)), Ref(CmxThermo(tps)), flags, length(q), _dp(FT, q), _dp(FT, ρ), _dp(FT, N), _dp(FT, T), _dp(FT, dn_frz), _dp(FT, dq_frz), stream)
    _check(st, "cmx_liquid_freezing_rate")
    return nothing
end

"""
    p3_liquid_ice_collisions!(sources, rates, ice, aps, tps, ρq_ice, ρn_ice, ρq_rim, ρb_rim, L_c, N_c, L_r, N_r, ρₐ, T, logλ; quad, stream)

`P3.bulk_liquid_ice_collision_sources` (7 columns) and `P3.∫liquid_ice_collisions` (10 columns), src/P3_processes.jl:527-655; either may be `nothing`.
"""
function p3_liquid_ice_collisions!(sources, rates, ice::CMP.P3IceParams, aps::CMP.AirProperties{FT}, tps, ρq_ice, ρn_ice, ρq_rim, ρb_rim, L_c, N_c, L_r, N_r,
    ρₐ, T, logλ; quad = nothing, stream = C_NULL) where {FT}
    q = quad === nothing ? _quad(ice, FT) : CmxQuadrature(quad, FT)
    st = ccall(_fn("cmx_p3_liquid_ice_collisions", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT},
            Ptr{FT}, Ptr{Ptr{FT}}, Ptr{Ptr{FT}}, Ptr{Cvoid}),
        Ref(pack(ice)), Ref(aps), Ref(CmxThermo(tps)), Ref(q), p3_flags(ice), length(T), _dp(FT, ρq_ice), _dp(FT, ρn_ice), _dp(FT, ρq_rim), _dp(FT, ρb_rim),
        _dp(FT, L_c), _dp(FT, N_c), _dp(FT, L_r), _dp(FT, N_r), _dp(FT, ρₐ), _dp(FT, T), _dp(FT, logλ), _opt_ptrs(FT, sources), _opt_ptrs(FT, rates), stream)
    _check(st, "cmx_p3_liquid_ice_collisions")
    return nothing
end

"""
    bulk_microphysics_tendencies!(out, BMT.Microphysics2Moment(), mp, tps, ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, logλ[, inpc_log_shift]; stream)

The 2M + P3 method (src/BulkMicrophysicsTendencies.jl:898-1083), `mp::Microphysics2MParams{WR, <:P3IceParams}`.  `out` = the eight tendency columns
in the order of the reference's NamedTuple (dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt, dq_ice_dt, dn_ice_dt, dq_rim_dt, db_rim_dt); the ninth
field `dn_lcl_activation_dt` is identically zero and stays on the Julia side.
"""
function bulk_microphysics_tendencies!(out, ::BMT.Microphysics2Moment, mp::CMP.Microphysics2MParams{WR, <:CMP.P3IceParams}, tps,
    ρ::AbstractArray{FT}, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, logλ, inpc_log_shift = nothing; stream = C_NULL) where {WR, FT}
    length(out) == 8 || error("out: the eight tendency columns")
    st = ccall(_fn("cmx_microphysics_2m_p3_tendencies", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT},
            Ptr{FT}, Ptr{Ptr{FT}}, Ptr{Cvoid}),
        Ref(pack(_warm_rain(mp))), Ref(pack(_ice(mp))), Ref(CmxThermo(tps)), p3_flags(_ice(mp)), length(ρ), _dp(FT, ρ), _dp(FT, T), _dp(FT, q_tot),
        _dp(FT, q_lcl), _dp(FT, n_lcl), _dp(FT, q_rai), _dp(FT, n_rai), _dp(FT, q_ice), _dp(FT, n_ice), _dp(FT, q_rim), _dp(FT, b_rim), _dp(FT, logλ),
        _dp(FT, inpc_log_shift), _ptrs(FT, out), stream)
    _check(st, "cmx_microphysics_2m_p3_tendencies")
    return out
end

"""The 2M + P3 method on the host model's storage: `in` = 13 pointers (the last, inpc_log_shift, may be NULL), `out` = 8 pointers, per-column run strides."""
function bulk_microphysics_tendencies_p3_fields!(::BMT.Microphysics2Moment, mp::CMP.Microphysics2MParams{WR, <:CMP.P3IceParams}, tps, ::Type{FT},
    n_seg::Integer, seg_len::Integer, in::Vector{Ptr{FT}}, in_stride, out::Vector{Ptr{FT}}, out_stride; stream = C_NULL) where {WR, FT}
    st = ccall(_fn("cmx_microphysics_2m_p3_tendencies_fields", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Int64, Int64, Ptr{Ptr{FT}}, Ptr{Int64}, Ptr{Ptr{FT}}, Ptr{Int64}, Ptr{Cvoid}),
        Ref(pack(_warm_rain(mp))), Ref(pack(_ice(mp))), Ref(CmxThermo(tps)), p3_flags(_ice(mp)), n_seg, seg_len, in, _strides(in_stride), out,
        _strides(out_stride), stream)
    _check(st, "cmx_microphysics_2m_p3_tendencies_fields")
    return nothing
end

# ----------------------------------------------------------------------------------------------------------------
# (0) 0-moment scheme, (3) diagnostic sums, diagnostic function evaluation
# ----------------------------------------------------------------------------------------------------------------
"""`BMT.bulk_microphysics_tendencies.(Microphysics0Moment(), mp, tps, T, q_lcl, q_icl[, q_vap_sat])` (src/BulkMicrophysicsTendencies.jl:658-680)."""
function bulk_microphysics_tendencies!(dq_tot_dt, ::BMT.Microphysics0Moment, mp::CMP.Microphysics0MParams, tps, T, q_lcl::AbstractArray{FT}, q_icl,
    q_vap_sat = nothing; ddq_dq_tot = nothing, stream = C_NULL) where {FT}   # tps and T are not read by the reference either (BMT:658-680)
    st = ccall(_fn("cmx_mp0m_tendencies", FT), Int32, (Ptr{Cvoid}, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(mp.precip), length(q_lcl), _dp(FT, q_lcl), _dp(FT, q_icl), _dp(FT, q_vap_sat), _dp(FT, dq_tot_dt), _dp(FT, ddq_dq_tot), stream)
    _check(st, "cmx_mp0m_tendencies")
    return dq_tot_dt
end

"""`UT.gamma_inc.(a, x)` → (P, Q) (src/Utilities.jl:54-61,93-144; `test_gamma_inc_kernel!`, test/gpu_tests.jl:456-461); either output may be `nothing`."""
function gamma_inc!(P, Q, a::AbstractArray{FT}, x; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_gamma_inc", FT), Int32, (Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        length(a), _dp(FT, a), _dp(FT, x), _dp(FT, P), _dp(FT, Q), stream)
    _check(st, "cmx_gamma_inc")
    return nothing
end

"""`UT.gamma_inc_inv.(a, p, q)` (src/Utilities.jl:162-165,205-252)."""
function gamma_inc_inv!(x, a::AbstractArray{FT}, p, q; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_gamma_inc_inv", FT), Int32, (Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        length(a), _dp(FT, a), _dp(FT, p), _dp(FT, q), _dp(FT, x), stream)
    _check(st, "cmx_gamma_inc_inv")
    return x
end

"""`DT.generalized_gamma_quantile.(ν, μ, B, Y)` / `DT.generalized_gamma_cdf.(ν, μ, B, x)` (src/DistributionTools.jl:44-82); an output and its input may be `nothing`."""
function generalized_gamma!(quantile, cdf, ν, μ, B::AbstractArray{FT}, Y, x; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_generalized_gamma", FT), Int32, (FT, FT, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        FT(ν), FT(μ), length(B), _dp(FT, B), _dp(FT, Y), _dp(FT, x), _dp(FT, quantile), _dp(FT, cdf), stream)
    _check(st, "cmx_generalized_gamma")
    return nothing
end

"""`DT.exponential_quantile.(D_mean, Y)` / `DT.exponential_cdf.(D_mean, D)` (src/DistributionTools.jl:124-151)."""
function exponential_distribution!(quantile, cdf, D_mean::AbstractArray{FT}, Y, D; stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_exponential_distribution", FT), Int32, (Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        length(D_mean), _dp(FT, D_mean), _dp(FT, Y), _dp(FT, D), _dp(FT, quantile), _dp(FT, cdf), stream)
    _check(st, "cmx_exponential_distribution")
    return nothing
end

"""
    size_distribution!(n_D, D_min, D_max, pdf, q, ρₐ, N, D; p = eps(FT), stream)

`CM2.size_distribution_value.(Ref(pdf), q, ρₐ, N, D)` and `CM2.get_size_distribution_bounds.(Ref(pdf), q, ρₐ, N, p)` (src/Microphysics2M.jl:270-354)
for `pdf` a `CMP.CloudParticlePDF_SB2006` or a `CMP.RainParticlePDF_SB2006_limited` / `_notlimited`; `D` is needed for `n_D` only.
"""
function size_distribution!(n_D, D_min, D_max, pdf::CMP.CloudParticlePDF_SB2006{FT}, q, ρₐ, N, D; p = eps(FT), stream = C_NULL) where {FT}
    st = ccall(_fn("cmx_sb2006_size_distribution", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, FT, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        Ref(pdf), C_NULL, CMX_PSD_CLOUD, FT(p), length(q), _dp(FT, q), _dp(FT, ρₐ), _dp(FT, N), _dp(FT, D), _dp(FT, n_D), _dp(FT, D_min), _dp(FT, D_max), stream)
    _check(st, "cmx_sb2006_size_distribution")
    return nothing
end
function size_distribution!(n_D, D_min, D_max, pdf::CMP.RainParticlePDF_SB2006, q::AbstractArray{FT}, ρₐ, N, D; p = eps(FT), stream = C_NULL) where {FT}
    flags = CMP.islimited(pdf) ? CMX_SB2006_LIMITED : UInt32(0)
    st = ccall(_fn("cmx_sb2006_size_distribution", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, FT, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        C_NULL, Ref(pack(pdf)), flags, FT(p), length(q), _dp(FT, q), _dp(FT, ρₐ), _dp(FT, N), _dp(FT, D), _dp(FT, n_D), _dp(FT, D_min), _dp(FT, D_max), stream)
    _check(st, "cmx_sb2006_size_distribution")
    return nothing
end

"""
    cloud_diagnostics!(Z_1m, Z_2m, reff_2m, reff_lh97, rain, sb, wtr, ρ, q_lcl, q_rai, N_lcl, N_rai; stream)

`CMD.radar_reflectivity_1M.(Ref(rain), q_rai, ρ)`, `CMD.radar_reflectivity_2M.(Ref(sb), q_lcl, q_rai, N_lcl, N_rai, ρ)`,
`CMD.effective_radius_2M.(…)` and `CMD.effective_radius_Liu_Hallet_97.(Ref(wtr), ρ, q_lcl, N_lcl, q_rai, N_rai)` (src/CloudDiagnostics.jl:31-163) in one pass
over the state columns.  An output may be `nothing`, and so may the parameter struct only it needs (`rain::CMP.Rain`, `sb::CMP.SB2006`,
`wtr::Union{CMP.WaterProperties, CMP.CloudLiquid}`); the three-argument Liu–Hallett method when `N_lcl`, `q_rai`, `N_rai` are all `nothing`.
"""
function cloud_diagnostics!(Z_1m, Z_2m, reff_2m, reff_lh97, rain, sb, wtr, ρ::AbstractArray{FT}, q_lcl, q_rai, N_lcl, N_rai; stream = C_NULL) where {FT}
    flags = (sb !== nothing && is_limited(sb)) ? CMX_SB2006_LIMITED : UInt32(0)
    p_rain = rain === nothing ? C_NULL : Ref(rain)          # CMP.Rain is layout-compatible (DIRECT_LAYOUT)
    p_c = sb === nothing ? C_NULL : Ref(sb.pdf_c)
    p_r = sb === nothing ? C_NULL : Ref(pack(sb.pdf_r))
    ρw = wtr === nothing ? FT(0) : FT(wtr.ρw)
    st = ccall(_fn("cmx_cloud_diagnostics", FT), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, FT, UInt32, Int64, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{FT}, Ptr{Cvoid}),
        p_rain, p_c, p_r, ρw, flags, length(ρ), _dp(FT, ρ), _dp(FT, q_lcl), _dp(FT, q_rai), _dp(FT, N_lcl), _dp(FT, N_rai), _dp(FT, Z_1m), _dp(FT, Z_2m),
        _dp(FT, reff_2m), _dp(FT, reff_lh97), stream)
    _check(st, "cmx_cloud_diagnostics")
    return nothing
end

"""
    column_sums!(sums, workspace, cols, FT, n; stream)

Σ of up to 16 device columns into `sums` (device `Float64[ncols]`); `workspace` = device `Float64[ncols · CMX_COLUMN_SUMS_PARTIALS]`.
Deterministic (no atomics): bit-identical from run to run; the caller all-reduces the doubles (MPI / RCCL).
"""
function column_sums!(sums, workspace, cols, ::Type{FT}, n::Integer; stream = C_NULL) where {FT}
    length(cols) <= CMX_COLUMN_SUMS_MAX_COLS || error("at most $(CMX_COLUMN_SUMS_MAX_COLS) columns per call")
    st = ccall(_fn("cmx_column_sums", FT), Int32, (Int32, Ptr{Ptr{FT}}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
        length(cols), _ptrs(FT, cols), n, _dp(Float64, sums), _dp(Float64, workspace), stream)
    _check(st, "cmx_column_sums")
    return sums
end

end # module CMXExt
