// point_host.cpp — TEST INFRASTRUCTURE: the device point functions of csrc/ compiled by g++ for the HOST (CMX_HOST_BUILD: libm stand-ins
// for the hardware transcendental instructions, no kernel-argument tricks) so that their algebra — host-folded constants, log2-domain
// rewrites, gates — can be checked against the oracle in the CPU test suite (tests/test_point_host.py), without a GPU.  It is NOT part
// of libcmx.so, nothing in the package loads it, and it is no CPU fallback: the product has none (cmx._lib raises without libcmx.so).
// Rounding differs from the device (libm exp2f/log2f vs v_exp_f32/v_log_f32, 1/x vs v_rcp_f32), so this checks formulas at the parity
// tolerances, not bits; the -m gpu suite remains the parity gate.
#define CMX_HOST_BUILD 1
#include <cstdint>

#include "../../cloudmicrophysics.jl_amd/csrc/cmx_mp1m.hpp"
#include "../../cloudmicrophysics.jl_amd/csrc/cmx_mp1m_vel.hpp"
#include "../../cloudmicrophysics.jl_amd/csrc/cmx_sb2006.hpp"

namespace {
using namespace cmx;

template <typename FT, typename MP, typename TH>
int32_t tendencies(const MP *mp, const TH *tps, uint32_t flags, int64_t n, const FT *const *x, FT *const *y) {
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    const bool def = flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c);
    for (int64_t i = 0; i < n; ++i) {
        if (def) mp1m_tendencies_point<FT, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>(c, x[0][i], x[1][i], x[2][i], x[3][i], x[4][i], x[5][i], x[6][i], y[0][i], y[1][i], y[2][i], y[3][i]);
        else mp1m_tendencies_point<FT>(c, x[0][i], x[1][i], x[2][i], x[3][i], x[4][i], x[5][i], x[6][i], y[0][i], y[1][i], y[2][i], y[3][i]);
    }
    return def ? 1 : 0;
}
template <typename FT, typename MP, typename TH>
int32_t sources(const MP *mp, const TH *tps, uint32_t flags, int64_t n, const FT *const *x, FT *const *s) {
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    const bool def = flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c);
    for (int64_t i = 0; i < n; ++i) {
        const Mp1mSrc<FT> p = def ? mp1m_point<FT, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>(c, x[0][i], x[1][i], x[2][i], x[3][i], x[4][i], x[5][i], x[6][i])
                                  : mp1m_point<FT>(c, x[0][i], x[1][i], x[2][i], x[3][i], x[4][i], x[5][i], x[6][i]);
        FT e[CMX_MP1M_NSRC];
        mp1m_expand<FT>(p, e);
        for (int k = 0; k < CMX_MP1M_NSRC; ++k) s[k][i] = e[k];
    }
    return def ? 1 : 0;
}
template <typename FT, typename MP, typename TH>
int32_t linearized(const MP *mp, const TH *tps, uint32_t flags, FT q_min, FT dt, int32_t nsub, int64_t n, const FT *const *x, FT *const *y) {
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    const Mp1mLinArgs<FT> a = make_mp1m_lin_args<FT>(q_min, dt, nsub, (FT)tps->LH_v0, (FT)tps->LH_s0, (FT)tps->cp_d);
    const bool def = flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c);
    auto args = [&](FT) -> const Mp1mLinArgs<FT> & { return a; };
    for (int64_t i = 0; i < n; ++i) {
        if (def) mp1m_linearized_point<FT, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>(c, args, nsub, x[0][i], x[1][i], x[2][i], x[3][i], x[4][i], x[5][i], x[6][i], y[0][i], y[1][i], y[2][i], y[3][i]);
        else mp1m_linearized_point<FT>(c, args, nsub, x[0][i], x[1][i], x[2][i], x[3][i], x[4][i], x[5][i], x[6][i], y[0][i], y[1][i], y[2][i], y[3][i]);
    }
    return def ? 1 : 0;
}
// the four sedimentation velocities (w_lcl, w_icl, w_rai, w_sno) of (rho, q_lcl, q_icl, q_rai, q_sno)
template <typename FT, typename MP, typename ST, typename CH, typename CI>
int32_t sedimentation(const MP *mp, const ST *stokes, const CH *chen_rain, const CI *chen_ice, int general_gamma, int64_t n, const FT *const *x, FT *const *w) {
    bool fit_general = false;
    Vel1mConsts<FT> c = make_vel1m_consts<FT>(*mp, chen_rain, &fit_general);
    add_sedimentation_consts<FT>(c, *mp, stokes, chen_ice);
    if (general_gamma < 0) general_gamma = fit_general;      // −1: what the entry points would pick
    for (int64_t i = 0; i < n; ++i) {
        const FT rho = x[0][i], rp = Math<FT>::max(FT(0), rho);
        w[0][i] = vel_lcl_stokes<FT>(c, rho, x[1][i]);
        w[1][i] = vel_icl_chen<FT>(c, rho, rp, x[2][i]);
        const FT l2r = vel_l2_li_rain<FT>(c, rp, x[3][i]);
        w[2][i] = general_gamma ? vel_rain_chen<FT, true>(c, rp, l2r, x[3][i]) : vel_rain_chen<FT, false>(c, rp, l2r, x[3][i]);
        w[3][i] = vel_snow_chen<FT>(c, rp, vel_l2_li_snow<FT>(c, rp, x[4][i]), x[4][i]);
    }
    return fit_general ? 1 : 0;
}
// the fused 1-moment column step, level by level from the model top (the kernel's per-point functions in a plain loop)
template <typename FT, typename MP, typename TH, typename ST, typename CH, typename CI>
int32_t column(const MP *mp, const TH *tps, const ST *stokes, const CH *chen_rain, const CI *chen_ice, uint32_t flags, FT q_min, FT dt, int32_t nsub,
               int64_t n_col, int32_t n_lev, const FT *inv_dz, const FT *const *x, FT *const *y, FT *precip_rai, FT *precip_sno) {
    using M = Math<FT>;
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)M::eps_1m());
    const Mp1mLinArgs<FT> a = nsub > 0 ? make_mp1m_lin_args<FT>(q_min, dt, nsub, (FT)tps->LH_v0, (FT)tps->LH_s0, (FT)tps->cp_d) : Mp1mLinArgs<FT>{};
    bool general = false;
    Vel1mConsts<FT> vc = make_vel1m_consts<FT>(*mp, chen_rain, &general);
    add_sedimentation_consts<FT>(vc, *mp, stokes, chen_ice);
    auto args = [&](FT) -> const Mp1mLinArgs<FT> & { return a; };
    for (int64_t col = 0; col < n_col; ++col) {
        SedFlux4<FT> up{{FT(0), FT(0), FT(0), FT(0)}};
        for (int32_t k = n_lev - 1; k >= 0; --k) {
            const int64_t i = col * n_lev + k;
            FT t[4];
            if (nsub > 0) mp1m_linearized_point<FT>(c, args, nsub, x[0][i], x[1][i], x[2][i], x[3][i], x[4][i], x[5][i], x[6][i], t[0], t[1], t[2], t[3]);
            else mp1m_tendencies_point<FT>(c, x[0][i], x[1][i], x[2][i], x[3][i], x[4][i], x[5][i], x[6][i], t[0], t[1], t[2], t[3]);
            const SedFlux4<FT> F = general ? mp1m_sed_fluxes<FT, true>(vc, x[0][i], x[3][i], x[4][i], x[5][i], x[6][i])
                                           : mp1m_sed_fluxes<FT, false>(vc, x[0][i], x[3][i], x[4][i], x[5][i], x[6][i]);
            const FT g = inv_dz[k] * M::rcp(max0(x[0][i]));
            for (int s = 0; s < 4; ++s) {
                const FT A = M::fma(-F.f[s], g, t[s]);
                y[s][i] = k == n_lev - 1 ? A : M::fma(up.f[s], g, A);
            }
            if (k == 0 && precip_rai) precip_rai[col] = F.f[2];
            if (k == 0 && precip_sno) precip_sno[col] = F.f[3];
            up = F;
        }
    }
    return 0;
}
// ---- round 5: the SB2006 point function on the host — the fused sums of sb2006_tendencies_kernel (csrc/cmx_sb2006_kernels.hpp) around sb2006_point, for
// a value type VT (one point, or a pair).  x = (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai), y = (dq_lcl, dn_lcl, dq_rai, dn_rai, vt_n, vt_m)
template <typename VT, bool LIMITED, int VEL, bool INTPOW, typename C>
void sb_fused(const C &c, const VT (&x)[7], VT (&y)[6]) {
    using MV = Math<VT>;
    const VT r_ = max0(x[0]), qt = max0(x[2]), ql = max0(x[3]), nl = max0(x[4]), qr = max0(x[5]), nr = max0(x[6]);
    const SbRates<VT> p = sb2006_point<VT, LIMITED, VEL, false, INTPOW>(c, r_, x[1], qt, ql, qr, r_ * nl, r_ * nr, nl, nr);
    y[0] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
    y[1] = MV::fma(p.lsc_plus_au + p.ac_dN_lcl, p.inv_rho, p.na_lcl);
    y[2] = (p.evq + p.au_dq_rai) + p.ac_dq_rai;
    y[3] = MV::fma(((p.evN + p.au_dN_rai) + p.rsc) + p.rbr, p.inv_rho, p.na_rai);
    y[4] = p.vt_n; y[5] = p.vt_m;
    const VT poison = nan_mask(x[0], x[2], x[3], x[4], x[5], x[6], x[1]) ? MV::nan() : VT(0);
    for (int q = 0; q < 6; ++q) y[q] += poison;
}
template <typename FT, typename WR, typename TH, typename VL>
int32_t sb2006_host(const WR *wr, const TH *tps, const VL *vel, uint32_t flags, int pairs, int64_t n, const FT *const *x, FT *const *y) {
    const SbConsts<FT> c = make_sb_consts<FT>(*wr, *tps, vel, (double)Math<FT>::eps_1m());
    const bool limited = flags & CMX_SB2006_LIMITED, chen = flags & CMX_VEL_CHEN2022, intpow = sb_integer_exponents(*wr);
    if (limited && !sb_limiters_ok(*wr)) return CMX_ERR_BAD_ARG;
    if (chen && chen_vel_kind<FT>(vel->chen2022) != VEL_CHEN) return CMX_ERR_UNSUPPORTED;      // (the general-Γ instantiation is not built here)
    auto run = [&](auto lim, auto vk, auto ip) {
        constexpr bool L = decltype(lim)::value, I = decltype(ip)::value;
        constexpr int V = decltype(vk)::value;
#if CMX_HAVE_PACKED
        if constexpr (sizeof(FT) == 4) {
            if (pairs) {
                for (int64_t i = 0; i + 1 < n; i += 2) {
                    f32x2 xi[7], yi[6];
                    for (int k = 0; k < 7; ++k) xi[k] = f32x2{x[k][i], x[k][i + 1]};
                    sb_fused<f32x2, L, V, I>(c, xi, yi);
                    for (int k = 0; k < 6; ++k) { y[k][i] = yi[k].x; y[k][i + 1] = yi[k].y; }
                }
                return;
            }
        }
#endif
        for (int64_t i = 0; i < n; ++i) {
            FT xi[7], yi[6];
            for (int k = 0; k < 7; ++k) xi[k] = x[k][i];
            sb_fused<FT, L, V, I>(c, xi, yi);
            for (int k = 0; k < 6; ++k) y[k][i] = yi[k];
        }
    };
    auto pick_ip = [&](auto lim, auto vk) { if (intpow) run(lim, vk, std::true_type{}); else run(lim, vk, std::false_type{}); };
    auto pick_v = [&](auto lim) { if (chen) pick_ip(lim, std::integral_constant<int, VEL_CHEN>{}); else pick_ip(lim, std::integral_constant<int, VEL_SB>{}); };
    if (limited) pick_v(std::true_type{}); else pick_v(std::false_type{});
    return intpow ? 1 : 0;
}

#if CMX_HAVE_PACKED
// ---- round 5: the PACKED instantiations (value type f32x2, cmx_math.hpp) next to the one-point ones, pair by pair — for the bit-identity test of
// tests/test_point_host.py (built with clang++: g++ has no ext_vector_type and compiles this file without the block) --------------------------------
int32_t tendencies_pairs(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n, const float *const *x, float *const *y) {
    const Mp1mConsts<float> c = make_mp1m_consts<float>(*mp, *tps, flags, (double)Math<float>::eps_1m());
    const bool def = flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c);
    for (int64_t i = 0; i + 1 < n; i += 2) {
        f32x2 v[7], o[4];
        for (int k = 0; k < 7; ++k) v[k] = f32x2{x[k][i], x[k][i + 1]};
        if (def) mp1m_tendencies_point<f32x2, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>(c, v[0], v[1], v[2], v[3], v[4], v[5], v[6], o[0], o[1], o[2], o[3]);
        else mp1m_tendencies_point<f32x2>(c, v[0], v[1], v[2], v[3], v[4], v[5], v[6], o[0], o[1], o[2], o[3]);
        for (int k = 0; k < 4; ++k) { y[k][i] = o[k].x; y[k][i + 1] = o[k].y; }
    }
    return def ? 1 : 0;
}
int32_t linearized_pairs(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, float q_min, float dt, int32_t nsub, int64_t n,
                         const float *const *x, float *const *y) {
    const Mp1mConsts<float> c = make_mp1m_consts<float>(*mp, *tps, flags, (double)Math<float>::eps_1m());
    const Mp1mLinArgs<float> a = make_mp1m_lin_args<float>(q_min, dt, nsub, (float)tps->LH_v0, (float)tps->LH_s0, (float)tps->cp_d);
    const bool def = flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c);
    auto args = [&](f32x2) -> const Mp1mLinArgs<float> & { return a; };
    for (int64_t i = 0; i + 1 < n; i += 2) {
        f32x2 v[7], o[4];
        for (int k = 0; k < 7; ++k) v[k] = f32x2{x[k][i], x[k][i + 1]};
        if (def) mp1m_linearized_point<f32x2, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>(c, args, nsub, v[0], v[1], v[2], v[3], v[4], v[5], v[6], o[0], o[1], o[2], o[3]);
        else mp1m_linearized_point<f32x2>(c, args, nsub, v[0], v[1], v[2], v[3], v[4], v[5], v[6], o[0], o[1], o[2], o[3]);
        for (int k = 0; k < 4; ++k) { y[k][i] = o[k].x; y[k][i + 1] = o[k].y; }
    }
    return def ? 1 : 0;
}
// the four sedimentation fluxes of (rho, q_lcl, q_icl, q_rai, q_sno), one point at a time (pairs == 0) or pair by pair
int32_t sed_fluxes(const cmx_microphysics_1m_f32 *mp, const cmx_stokes_vel_f32 *stokes, const cmx_chen2022_rain_vel_f32 *chen_rain, const cmx_chen2022_ice_vel_f32 *chen_ice,
                   int pairs, int64_t n, const float *const *x, float *const *w) {
    bool general = false;
    Vel1mConsts<float> vc = make_vel1m_consts<float>(*mp, chen_rain, &general);
    add_sedimentation_consts<float>(vc, *mp, stokes, chen_ice);
    if (general) return -1;
    if (!pairs) {
        for (int64_t i = 0; i < n; ++i) {
            const SedFlux4<float> F = mp1m_sed_fluxes<float, false>(vc, x[0][i], x[1][i], x[2][i], x[3][i], x[4][i]);
            for (int k = 0; k < 4; ++k) w[k][i] = F.f[k];
        }
    } else {
        for (int64_t i = 0; i + 1 < n; i += 2) {
            f32x2 v[5];
            for (int k = 0; k < 5; ++k) v[k] = f32x2{x[k][i], x[k][i + 1]};
            const SedFlux4<f32x2> F = mp1m_sed_fluxes<f32x2, false>(vc, v[0], v[1], v[2], v[3], v[4]);
            for (int k = 0; k < 4; ++k) { w[k][i] = F.f[k].x; w[k][i + 1] = F.f[k].y; }
        }
    }
    return 0;
}
#endif
}  // namespace

extern "C" {
int32_t host_have_packed(void) { return CMX_HAVE_PACKED; }
int32_t host_sb2006_f32(const cmx_warm_rain_2m_f32 *wr, const cmx_thermo_f32 *tps, const cmx_rain_vel_f32 *vel, uint32_t flags, int pairs, int64_t n, const float *const *x, float *const *y) { return sb2006_host<float>(wr, tps, vel, flags, pairs, n, x, y); }
int32_t host_sb2006_f64(const cmx_warm_rain_2m_f64 *wr, const cmx_thermo_f64 *tps, const cmx_rain_vel_f64 *vel, uint32_t flags, int pairs, int64_t n, const double *const *x, double *const *y) { return sb2006_host<double>(wr, tps, vel, flags, pairs, n, x, y); }
#if CMX_HAVE_PACKED
int32_t host_mp1m_tendencies_pairs_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n, const float *const *x, float *const *y) { return tendencies_pairs(mp, tps, flags, n, x, y); }
int32_t host_mp1m_linearized_pairs_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, float q_min, float dt, int32_t nsub, int64_t n, const float *const *x, float *const *y) { return linearized_pairs(mp, tps, flags, q_min, dt, nsub, n, x, y); }
int32_t host_sed_fluxes_f32(const cmx_microphysics_1m_f32 *mp, const cmx_stokes_vel_f32 *st, const cmx_chen2022_rain_vel_f32 *cr, const cmx_chen2022_ice_vel_f32 *ci, int pairs, int64_t n, const float *const *x, float *const *w) { return sed_fluxes(mp, st, cr, ci, pairs, n, x, w); }
#endif
int32_t host_mp1m_tendencies_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n, const float *const *x, float *const *y) { return tendencies<float>(mp, tps, flags, n, x, y); }
int32_t host_mp1m_tendencies_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n, const double *const *x, double *const *y) { return tendencies<double>(mp, tps, flags, n, x, y); }
int32_t host_mp1m_sources_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n, const float *const *x, float *const *s) { return sources<float>(mp, tps, flags, n, x, s); }
int32_t host_mp1m_sources_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n, const double *const *x, double *const *s) { return sources<double>(mp, tps, flags, n, x, s); }
int32_t host_mp1m_linearized_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, float q_min, float dt, int32_t nsub, int64_t n, const float *const *x, float *const *y) { return linearized<float>(mp, tps, flags, q_min, dt, nsub, n, x, y); }
int32_t host_mp1m_linearized_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, double q_min, double dt, int32_t nsub, int64_t n, const double *const *x, double *const *y) { return linearized<double>(mp, tps, flags, q_min, dt, nsub, n, x, y); }
int32_t host_sedimentation_f32(const cmx_microphysics_1m_f32 *mp, const cmx_stokes_vel_f32 *st, const cmx_chen2022_rain_vel_f32 *cr, const cmx_chen2022_ice_vel_f32 *ci, int general_gamma, int64_t n, const float *const *x, float *const *w) { return sedimentation<float>(mp, st, cr, ci, general_gamma, n, x, w); }
int32_t host_sedimentation_f64(const cmx_microphysics_1m_f64 *mp, const cmx_stokes_vel_f64 *st, const cmx_chen2022_rain_vel_f64 *cr, const cmx_chen2022_ice_vel_f64 *ci, int general_gamma, int64_t n, const double *const *x, double *const *w) { return sedimentation<double>(mp, st, cr, ci, general_gamma, n, x, w); }
int32_t host_mp1m_column_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, const cmx_stokes_vel_f32 *st, const cmx_chen2022_rain_vel_f32 *cr, const cmx_chen2022_ice_vel_f32 *ci, uint32_t flags, float q_min, float dt, int32_t nsub, int64_t n_col, int32_t n_lev, const float *inv_dz, const float *const *x, float *const *y, float *pr, float *ps) { return column<float>(mp, tps, st, cr, ci, flags, q_min, dt, nsub, n_col, n_lev, inv_dz, x, y, pr, ps); }
int32_t host_mp1m_column_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, const cmx_stokes_vel_f64 *st, const cmx_chen2022_rain_vel_f64 *cr, const cmx_chen2022_ice_vel_f64 *ci, uint32_t flags, double q_min, double dt, int32_t nsub, int64_t n_col, int32_t n_lev, const double *inv_dz, const double *const *x, double *const *y, double *pr, double *ps) { return column<double>(mp, tps, st, cr, ci, flags, q_min, dt, nsub, n_col, n_lev, inv_dz, x, y, pr, ps); }
}
