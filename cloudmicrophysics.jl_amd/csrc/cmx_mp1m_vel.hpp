// cmx_mp1m_vel.hpp — bulk fall speeds of the 1-moment species, per point: Blk1M rain / snow (CM1.terminal_velocity,
// /root/reference/src/Microphysics1M.jl:223-249), Chen-2022 rain and snow mass-weighted over the Marshall–Palmer PSD (CM1:251-297,
// Common.jl:290-350,414-422), cloud liquid in the Stokes regime and cloud ice by Chen-2022 small ice at the mean-volume diameter
// (CMNonEq.terminal_velocity, MicrophysicsNonEq.jl:250-281).  Shared by the velocity / sedimentation kernels (cmx_mp1m_kernels.hip)
// and the fused 1-moment column step (cmx_mp1m_column.hip).
#pragma once
#include "cmx_mp1m.hpp"

namespace cmx {

template <typename FT> struct Vel1mConsts {
    FT eps_1m, l2_eps, lam_a_rai, lam_b_rai, lam_floor_rai, lam_a_sno, lam_b_sno, lam_floor_sno, sno_l2_mu, sno_nu;
    FT rho_w, vt_k_rai, vt_e_rai, vt_k_sno, vt_e_sno;
    FT ch_rho0_l2e, ch_a[3], ch_a3_pow, ch_b[3], ch_b_rho, ch_c1000[3], l2_1000;
    ChenGamma<FT> chg;   // Γ(b_i(ρ) + 4)/3! as polynomials in ρ (cmx_math.hpp)
    // cloud liquid, Stokes (NonEq:250-265): v = st_pref (ρw/ρ − 1) D², D³ = st_D3 ρ q
    FT st_pref, st_rho_w, st_D3;
    // cloud ice, Chen-2022 small ice reduced at ρᵢ(cloud ice) (NonEq:267-281, Common.jl:304-325): D³ = ci_D3 ρ q
    FT ci_D3, ci_A, ci_B, ci_C, ci_E, ci_F, ci_c2;
    // snow, Chen-2022 large ice reduced at ρᵢ(snow), mass-weighted over the Marshall–Palmer PSD (CM1:272-297):
    // ϕ^κ Γ(b+4)/3! folded into the amplitudes
    FT sn_A, sn_a1, sn_b1, sn_a2, sn_H, sn_b2, sn_c2;
};

// (VC below: Vel1mConsts<FT>, possibly in the constant address space — kernels whose constants overflow the SGPR file read them through the
// kernel-argument pointer, cmx_math.hpp front_consts)
// log2 λ⁻¹ of rain / snow (CM1.lambda_inverse :126-152, get_n0 :83-86) from ρ⁺ = max(0, ρ) and q
template <typename FT, typename VC> __device__ __forceinline__ FT vel_l2_li_rain(const VC &c, FT rp, FT q) {
    using M = Math<FT>;
    return M::max(c.lam_floor_rai, M::fma(log2_floored(rp * M::max(FT(0), q)), c.lam_a_rai, c.lam_b_rai));
}
template <typename FT, typename VC> __device__ __forceinline__ FT vel_l2_li_snow(const VC &c, FT rp, FT q) {
    using M = Math<FT>;
    const FT l2_rq = log2_floored(rp * M::max(FT(0), q));
    const FT l2_n0 = q > c.eps_1m ? FT(M::fma(c.sno_nu, l2_rq, c.sno_l2_mu)) : FT(c.l2_eps);
    return M::max(c.lam_floor_sno, M::fma(l2_rq - M::max(l2_n0, c.l2_eps), c.lam_a_sno, c.lam_b_sno));
}
// CM1.terminal_velocity(::Rain / ::Snow, ::Blk1MVelType, ρ, q) — CM1:223-249
template <typename FT, typename VC> __device__ __forceinline__ FT vel_rain_blk1m(const VC &c, FT rho, FT l2_li, FT q) {
    using M = Math<FT>;
    const FT sq = M::sqrt(M::max(c.rho_w * M::rcp(rho) - FT(1), FT(0)));
    return q > c.eps_1m ? FT((c.vt_k_rai * sq) * M::exp2_fin(c.vt_e_rai * l2_li)) : FT(0);      // l2_li: floored, finite
}
template <typename FT, typename VC> __device__ __forceinline__ FT vel_snow_blk1m(const VC &c, FT l2_li, FT q) {
    using M = Math<FT>;
    return q > c.eps_1m ? FT(c.vt_k_sno * M::exp2_fin(c.vt_e_sno * l2_li)) : FT(0);
}
// Chen 2022 rain, mass-weighted (k = 3), diameter slope = 2 λ⁻¹ — CM1:251-270, Common.jl:290-302,414-422.  Γ(b+1) from the host-fitted
// polynomials in ρ (NaN fall speed beyond their range, ρ > 2 kg/m³); GENERAL: run-time Γ for parameter sets the fit cannot represent
// l2_rho = log2_floored(ρ⁺) (cmx_math.hpp), shared by the three Chen fall speeds of a point: with it and the floored slope parameter every exponent below is finite
template <typename FT, bool GENERAL = false, typename VC> __device__ __forceinline__ FT vel_rain_chen(const VC &c, FT rp, FT l2_li, FT q, FT l2_rho) {
    using M = Math<FT>;
    const FT l2_lam_inv = l2_li + FT(1);
    const FT lam = M::exp2_fin(-l2_lam_inv);
    const FT l2_q = c.ch_rho0_l2e * rp;
    FT w = FT(0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const FT bi = M::fma(-c.ch_b_rho, rp, c.ch_b[k]);
        FT l2_mag = M::fma(bi, c.l2_1000, l2_q);
        if (k == 2) l2_mag = M::fma(c.ch_a3_pow, l2_rho, l2_mag);
        const FT l2_den = M::log2_pn(lam + c.ch_c1000[k]);                  // λ = 2^finite > 0, c ≥ 0
        const FT e3 = M::exp2_fin(l2_mag - FT(4) * l2_lam_inv - (bi + FT(4)) * l2_den);
        // Γ(b+4)/3!: the fitted polynomial in ρ, or (b+3)(b+2)(b+1)·Γ(b+1)/6 with the run-time Γ
        const FT g = GENERAL ? FT(tgamma_general<FT>(bi + FT(1)) * (bi + FT(3)) * (bi + FT(2)) * (bi + FT(1)) * FT(1.0 / 6.0)) : chen_gamma_eval<typename M::Scalar>(c.chg, k, rp);
        w = k == 0 ? FT((c.ch_a[k] * e3) * g) : M::fma(c.ch_a[k] * e3, g, w);
    }
    w = M::max(FT(0), w);
    if constexpr (!GENERAL) w = rp > FT(kChenGammaRhoMax) ? M::nan() : w;       // outside the range of the fitted Γ: no silent extrapolation
    return q > c.eps_1m ? w : FT(0);
}
template <typename FT, bool GENERAL = false, typename VC> __device__ __forceinline__ FT vel_rain_chen(const VC &c, FT rp, FT l2_li, FT q) {
    return vel_rain_chen<FT, GENERAL>(c, rp, l2_li, q, log2_floored(rp));
}
// CMNonEq.terminal_velocity(::CloudLiquid, ::StokesRegimeVelType, ρ, q): Stokes at the mean-volume diameter — NonEq:250-265
template <typename FT, typename VC> __device__ __forceinline__ FT vel_lcl_stokes(const VC &c, FT rho, FT q) {
    using M = Math<FT>;
    const FT D2 = M::exp2_fin(FT(2.0 / 3.0) * log2_floored(c.st_D3 * rho * M::max(FT(0), q)));
    return q > c.eps_1m ? FT(c.st_pref * (c.st_rho_w * M::rcp(rho) - FT(1)) * D2) : FT(0);
}
// CMNonEq.terminal_velocity(::CloudIce, ::Chen2022VelTypeSmallIce, ρ, q): Σ aₖ D^bₖ e^{−cₖD} at that diameter — NonEq:267-281
template <typename FT, typename VC> __device__ __forceinline__ FT vel_icl_chen(const VC &c, FT rho, FT rp, FT q, FT l2_rho) {
    using M = Math<FT>;
    const FT l2_D = FT(1.0 / 3.0) * log2_floored(c.ci_D3 * rho * M::max(FT(0), q));
    const FT D = M::exp2_fin(l2_D);
    const FT b = M::fma(rp, c.ci_C, c.ci_B);
    const FT common = M::exp2_fin(c.ci_A * l2_rho + b * (c.l2_1000 + l2_D));             // ρₐ^As · (1000 D)^b
    const FT w = common * M::fma(c.ci_F, M::exp2_fin(-c.ci_c2 * D * FT(1.4426950408889634)), c.ci_E);
    return q > c.eps_1m ? FT(M::max(FT(0), w)) : FT(0);
}
template <typename FT, typename VC> __device__ __forceinline__ FT vel_icl_chen(const VC &c, FT rho, FT rp, FT q) {
    return vel_icl_chen<FT>(c, rho, rp, q, log2_floored(rp));
}
// CM1.terminal_velocity(::Snow, ::Chen2022VelTypeLargeIce, ρ, q): mass-weighted (k = 3), λ_D⁻¹ = 2 λ⁻¹ — CM1:272-297
template <typename FT, typename VC> __device__ __forceinline__ FT vel_snow_chen(const VC &c, FT rp, FT l2_li, FT q, FT l2_rho) {
    using M = Math<FT>;
    const FT l2_ld = l2_li + FT(1), lam = M::exp2_fin(-l2_ld);
    const FT l2_ra = c.sn_A * l2_rho;
    // aₖ e^{−4 ln λ_D⁻¹ − (bₖ+4) ln(λ_D + cₖ)}: term 1 has c = 0 → λ_D^{−b₁}·… collapses to one power
    const FT t1 = c.sn_a1 * M::exp2_fin(l2_ra + c.sn_b1 * l2_ld);
    const FT t2 = c.sn_a2 * M::exp2_fin(l2_ra + c.sn_H * rp * FT(1.4426950408889634) - FT(4) * l2_ld - (c.sn_b2 + FT(4)) * M::log2_pn(lam + c.sn_c2));
    return q > c.eps_1m ? FT(M::max(FT(0), t1 + t2)) : FT(0);
}
template <typename FT, typename VC> __device__ __forceinline__ FT vel_snow_chen(const VC &c, FT rp, FT l2_li, FT q) {
    return vel_snow_chen<FT>(c, rp, l2_li, q, log2_floored(rp));
}

// ---- sedimentation fluxes of the fused column step (cmx_mp1m_column.hip) ------------------------------------------------------------
template <typename FT> struct SedFlux4 { FT f[4]; };   // lcl, icl, rai, sno

// F = ρ⁺ χ⁺ w(ρ⁺, χ⁺) of the four species of one raw point
template <typename FT, bool GENERAL_GAMMA, typename VC>
__device__ __forceinline__ SedFlux4<FT> mp1m_sed_fluxes(const VC &vc, FT rho, FT q_lcl, FT q_icl, FT q_rai, FT q_sno) {
    using M = Math<FT>;
    const FT r_ = max0(rho), ql = max0(q_lcl), qi = max0(q_icl), qr = max0(q_rai), qs = max0(q_sno);
    SedFlux4<FT> F;
    // one species after the other (consts_after: only one species' constants live where they are read through the kernel-argument pointer)
    F.f[0] = (r_ * ql) * vel_lcl_stokes<FT>(vc, r_, ql);
    const VC &v1 = consts_after(vc, F.f[0]);
    const FT l2_rho = log2_floored(r_);      // once for the three Chen fall speeds
    F.f[1] = (r_ * qi) * vel_icl_chen<FT>(v1, r_, r_, qi, l2_rho);
    const VC &v2 = consts_after(v1, F.f[1]);
    // (the general-Γ instantiation forms its own log2 ρ: with the shared one live across its run-time Γ calls the LinearizedAverage column kernel is left
    // with a 100-byte private segment that no instruction touches — tests/test_kernel_resources.py)
    F.f[2] = (r_ * qr) * vel_rain_chen<FT, GENERAL_GAMMA>(v2, r_, vel_l2_li_rain<FT>(v2, r_, qr), qr, GENERAL_GAMMA ? log2_floored(keep(r_)) : l2_rho);
    const VC &v3 = consts_after(v2, F.f[2]);
    F.f[3] = (r_ * qs) * vel_snow_chen<FT>(v3, r_, vel_l2_li_snow<FT>(v3, r_, qs), qs, l2_rho);
    // NaN in → NaN out per species; Float64: ρ ≤ 0 poisons the point, as in the tendencies (cmx_math.hpp bad_density)
    auto poison = [&](FT q_s, FT f) -> FT {
        typename M::Mask bad = nan_mask(rho, q_s);
        if constexpr (M::IS_F64) bad = (bool)((int)bad | (int)bad_density(rho));
        return bad ? M::nan() : f;
    };
    F.f[0] = poison(q_lcl, F.f[0]); F.f[1] = poison(q_icl, F.f[1]); F.f[2] = poison(q_rai, F.f[2]); F.f[3] = poison(q_sno, F.f[3]);
    return F;
}

// *chen_general (optional): true when the Chen-2022 rain table needs the GENERAL instantiation (the Γ fit is not accurate for it)
template <typename FT, typename MP, typename CH>
static Vel1mConsts<FT> make_vel1m_consts(const MP &mp, const CH *chen, bool *chen_general = nullptr) {
    // reuse the folding of the tendencies kernel (thermo part unused): a neutral thermo struct keeps it well-defined
    cmx_thermo_f64 tp{461.5, 287.0, 1004.5, 1859.0, 4181.0, 2070.0, 2.5008e6, 2.8344e6, 273.16, 273.16, 611.657, 273.15, 4181.0};
    const Mp1mConsts<FT> m = make_mp1m_consts<FT>(mp, tp, 0u, (double)Math<FT>::eps_1m());
    Vel1mConsts<FT> c{};
    c.eps_1m = m.eps_1m; c.l2_eps = m.l2_eps; c.lam_a_rai = m.lam_a_rai; c.lam_b_rai = m.lam_b_rai; c.lam_floor_rai = m.lam_floor_rai;
    c.lam_a_sno = m.lam_a_sno; c.lam_b_sno = m.lam_b_sno; c.lam_floor_sno = m.lam_floor_sno; c.sno_l2_mu = m.sno_l2_mu; c.sno_nu = m.sno_nu;
    c.rho_w = m.rho_w; c.vt_k_rai = m.vt_k_rai; c.vt_e_rai = m.vt_e_rai; c.vt_k_sno = m.vt_k_sno; c.vt_e_sno = m.vt_e_sno;
    c.l2_1000 = (FT)std::log2(1000.0);
    if (chen) {
        c.ch_rho0_l2e = (FT)((double)chen->rho_0 * 1.4426950408889634074);
        for (int k = 0; k < 3; ++k) { c.ch_a[k] = (FT)chen->a[k]; c.ch_b[k] = (FT)chen->b[k]; c.ch_c1000[k] = (FT)((double)chen->c[k] * 1000.0); }
        c.ch_a3_pow = (FT)chen->a3_pow; c.ch_b_rho = (FT)chen->b_rho;
        const bool fit_ok = make_chen_gamma<FT>(*chen, c.chg, 3);
        if (chen_general) *chen_general = !fit_ok;
    } else if (chen_general) {
        *chen_general = false;
    }
    return c;
}

// adds the cloud-liquid (Stokes) and Chen-2022 ice constants of the four sedimentation velocities a host model precomputes
// (ClimaAtmos set_sedimentation_precomputed_quantities; test/gpu_clima_core_test.jl:36-45, KA kernel test/gpu_tests.jl:608-630)
template <typename FT, typename MP, typename ST, typename CI>
static void add_sedimentation_consts(Vel1mConsts<FT> &c, const MP &mp, const ST *stokes, const CI *chen_ice) {
    const double pi = 3.14159265358979323846;
    if (stokes) {
        c.st_pref = (FT)((double)stokes->grav / (18.0 * (double)stokes->nu_air));
        c.st_rho_w = (FT)stokes->rho_w;
        c.st_D3 = (FT)(6.0 / pi / ((double)mp.cloud_liquid.N_0 * (double)mp.cloud_liquid.rho_w));
    }
    if (chen_ice) {
        {   // small ice reduced at the cloud-ice apparent density — Common.jl:304-325
            const auto &t = chen_ice->small_ice;
            const double ri = (double)mp.cloud_ice.rho_i, l = std::log(ri), sq = std::sqrt(ri);
            c.ci_D3 = (FT)(6.0 / pi / ((double)mp.cloud_ice.N_0 * ri));
            c.ci_A = (FT)((double)t.A[1] * l * l - (double)t.A[2] * l + (double)t.A[0]);
            c.ci_B = (FT)(1.0 / ((double)t.B[0] + (double)t.B[1] * l + (double)t.B[2] / sq));
            c.ci_C = (FT)((double)t.C[0] + (double)t.C[1] * std::exp((double)t.C[2] * ri) + (double)t.C[3] * sq);
            c.ci_E = (FT)((double)t.E[0] - (double)t.E[1] * l * l + (double)t.E[2] * sq);
            c.ci_F = (FT)(-std::exp((double)t.F[0] - (double)t.F[1] * l * l + (double)t.F[2] * l));
            c.ci_c2 = (FT)(1000.0 / ((double)t.G[0] + (double)t.G[1] / l - (double)t.G[2] * l / ri));
        }
        {   // large ice reduced at the snow apparent density — Common.jl:327-350; ϕ^κ Γ(b+4)/3! folded in (CM1:287-295)
            const auto &t = chen_ice->large_ice;
            const double ri = (double)mp.snow.rho_i, l = std::log(ri), sq = std::sqrt(ri);
            const double Al = (double)t.A[0] + (double)t.A[1] * l + (double)t.A[2] / (ri * sq);
            const double Bl = std::exp((double)t.B[0] + (double)t.B[1] * l * l + (double)t.B[2] * l);
            const double Cl = std::exp((double)t.C[0] + (double)t.C[1] / l + (double)t.C[2] / ri);
            const double El = (double)t.E[0] + (double)t.E[1] * l * sq + (double)t.E[2] * sq;
            const double Fl = (double)t.F[0] + (double)t.F[1] * l - std::exp(std::log(-(double)t.F[2]) - ri);
            const double Gl = 1.0 / ((double)t.G[0] + (double)t.G[1] * l * sq + (double)t.G[2] / sq);
            const double Hl = (double)t.H[0] + (double)t.H[1] * ri * ri * sq + std::exp(std::log(-(double)t.H[2]) - ri);
            const double pk = std::pow((double)mp.snow.phi, (double)mp.snow.kappa);
            c.sn_A = (FT)Al; c.sn_b1 = (FT)Cl; c.sn_b2 = (FT)Fl; c.sn_H = (FT)Hl; c.sn_c2 = (FT)(1000.0 * Gl);
            c.sn_a1 = (FT)(pk * Bl * std::pow(1000.0, Cl) * std::tgamma(Cl + 4.0) / 6.0);
            c.sn_a2 = (FT)(pk * El * std::pow(1000.0, Fl) * std::tgamma(Fl + 4.0) / 6.0);
        }
    }
}

}  // namespace cmx
