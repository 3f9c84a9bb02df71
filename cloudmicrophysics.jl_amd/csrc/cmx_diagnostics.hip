// cmx_diagnostics.hip — CloudDiagnostics over columns for gfx950; C-ABI entry points cmx_cloud_diagnostics_* (include/cmx.h §(10)).
//
// Reference (src = /root/reference/src): CloudDiagnostics.jl — radar_reflectivity_1M :31-46, radar_reflectivity_2M :64-84, effective_radius_2M :100-125,
// effective_radius_Liu_Hallet_97 :143-163 — with CM1.get_n0 / lambda_inverse Microphysics1M.jl:83-152, CM2.pdf_rain_parameters(_mass) Microphysics2M.jl:67-146,
// log_pdf_cloud_parameters_mass :176-192, DT.generalized_gamma_Mⁿ DistributionTools.jl:109-112.  These are the diagnostics a host model computes from the SAME
// state columns the tendency kernels read (ClimaAtmos cloud / radiation diagnostics): a pure stream, 5 columns in, up to 4 out (36 B per Float32 point).
//
// Everything is a power law of the mean particle masses, so the kernel works in the log2 domain like the rate kernels: the generalized-gamma moments
// Mⁿ = N B^(−n/μ) Γ((ν+1+n)/μ)/Γ((ν+1)/μ) become  log2 N + (n/μ)·(−log2 B) + a host-folded constant, with −log2 B linear in log2 of the mean mass (the rain
// mean mass comes from the same limited / not-limited PSD routine as the rate kernels, sb2006_rain_psd).  The reference's gates are kept as selects:
// N < ϵ / q < ϵ (absent species → that species' moments are 0), notvalid(B) (B = 0 or not finite IN THE FLOAT TYPE'S RANGE), the −150 dBZ clip, the
// M2 ≤ ϵ gate of the effective radius.  A NaN input gives NaN (Julia's max(-150, NaN); the hardware max would hide it).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "../../include/cmx.h"
#include "cmx_launch.hpp"
#include "cmx_math.hpp"
#include "cmx_sb2006.hpp"

namespace cmx {

template <typename FT> struct DiagConsts {
    // radar_reflectivity_1M: Z = max(−150, z1_c0 + z1_c1·log2 λ⁻¹), log2 λ⁻¹ = max(lam_floor, lam_a·log2(ρ⁺q⁺) + lam_b)
    FT lam_a, lam_b, lam_floor, z1_c0, z1_c1;
    // rain PSD (sb2006_rain_psd reads these names)
    FT l2_pi_rho_w, l2_xr_min, l2_xr_max, l2_N0_min, l2_N0_max, l2_lam_min, l2_lam_max;
    // moments: log2 Mⁿ_r = log2 N_r + pr_n·(log2 x̄_r − log2 6) + kr_n ;  log2 Mⁿ_c = log2 N_c + pc_n·(log2 x̄_c + dlg) + kc_n   (n = 2, 1, 2/3; the k's
    // carry log2 of the Γ ratio and of C^n, C = 4/3 π ρw)
    FT l2_6, pr2, pr1, pr23, kr2, kr1, kr23;
    FT dlg, mu_c, pc2, pc1, pc23, kc2, kc1, kc23;
    FT lnB_hi, lnB_lo;          // log(floatmax(FT)), log(nextfloat(0)): exp(logB) is finite and non-zero strictly between them
    FT eps, eps_1m;
    FT lh_l2k;                  // Liu–Hallett: log2(3/(4π ρw)) , r = 2^((log2(q ρ/N) + lh_l2k)/3) / k^(1/3)
    FT lh_inv_k3;
};

template <typename FT> struct DiagIO { const FT *rho, *q_lcl, *q_rai, *N_lcl, *N_rai; FT *Z_1m, *Z_2m, *reff_2m, *reff_lh97; };

template <typename FT, bool LIMITED, typename C>
__device__ __forceinline__ void diag_point(const C &c, bool w1, bool w2, bool wr, bool wl, bool lh_defaults, FT rho, FT q_lcl, FT q_rai, FT N_lcl, FT N_rai, FT &Z_1m,
                                           FT &Z_2m, FT &reff_2m, FT &reff_lh97) {
    using M = Math<FT>;
    const FT log10_2 = FT(0.30102999566398119521);
    if (w1) {   // CMD :31-46 with CM1.lambda_inverse :126-152 (q, ρ clamped to ≥ 0 there; the floor r0·1e-5 makes the floored log2 of an absent species harmless: cmx_math.hpp log2_floored)
        const FT l2_li = M::max(c.lam_floor, M::fma(log2_floored(max0(rho) * max0(q_rai)), c.lam_a, c.lam_b));
        const FT z = M::max(FT(-150), M::fma(c.z1_c1, l2_li, c.z1_c0));
        Z_1m = any_nan(rho, q_rai) ? M::nan() : z;
    }
    if (w2 || wr) {
        const FT eps = c.eps;
        // rain: x̄_r from pdf_rain_parameters (the same routine as the rate kernels, log2 domain); gate as in CM2:84-88 / :103-108
        const FT sq_r = M::max(q_rai, eps), sN_r = M::max(N_rai, eps);
        const SbRainPsd<FT> psd = sb2006_rain_psd<FT, LIMITED>(c, rho * sq_r, sN_r);
        const bool no_rain = LIMITED ? (N_rai < eps && q_rai < eps) : (N_rai < eps || q_rai < eps);
        // the moments are N·B^(−n/μ): a NEGATIVE rain number (left by advection; the limited PSD only gates on N < ϵ AND q < ϵ) gives a finite negative
        // moment in the reference — log2 of the magnitude, the sign folded back below (log2 of the raw column was a NaN there)
        const FT l2_Nr = log2_floored(N_rai < FT(0) ? -N_rai : N_rai), sgn_r = N_rai < FT(0) ? FT(-1) : FT(1), dxr = psd.l2_xr - c.l2_6;
        // cloud: log x̄_c and logB = −μc (log x̄ + lgΓ(z1) − lgΓ(z2)) (CM2:176-192); notvalid(Bc) where exp(logB) leaves the float type's range
        const FT sq_c = M::max(q_lcl, eps), sN_c = M::max(N_lcl, eps);
        const FT l2_xc = M::log2(rho * sq_c * M::rcp(sN_c));
        const FT lnB = -c.mu_c * (l2_xc * FT(0.69314718055994530942) + c.dlg);
        const bool no_cloud = (N_lcl < eps || q_lcl < eps) || !(lnB < c.lnB_hi) || !(lnB > c.lnB_lo);
        const FT l2_Nc = log2_floored(N_lcl), dxc = M::fma(c.dlg, FT(1.4426950408889634074), l2_xc);
        const bool poisoned = any_nan(rho, q_lcl, q_rai, N_lcl, N_rai);
        if (w2) {   // CMD :64-84
            const FT Zc = no_cloud ? FT(0) : M::exp2(l2_Nc + M::fma(c.pc2, dxc, c.kc2));
            const FT Zr = no_rain ? FT(0) : sgn_r * M::exp2(l2_Nr + M::fma(c.pr2, dxr, c.kr2));
            const FT z = M::max(FT(-150), FT(10) * M::fma(log2_floored(M::max(FT(0), Zc + Zr)), log10_2, FT(18)));
            Z_2m = poisoned ? M::nan() : z;
        }
        if (wr) {   // CMD :100-125
            const FT M3c = no_cloud ? FT(0) : M::exp2(l2_Nc + M::fma(c.pc1, dxc, c.kc1)), M3r = no_rain ? FT(0) : sgn_r * M::exp2(l2_Nr + M::fma(c.pr1, dxr, c.kr1));
            const FT M2c = no_cloud ? FT(0) : M::exp2(l2_Nc + M::fma(c.pc23, dxc, c.kc23)), M2r = no_rain ? FT(0) : sgn_r * M::exp2(l2_Nr + M::fma(c.pr23, dxr, c.kr23));
            const FT M2 = M2c + M2r;
            const FT r = M2 <= c.eps_1m ? FT(0) : (M3c + M3r) * M::rcp(M2);
            reff_2m = poisoned ? M::nan() : r;
        }
    }
    if (wl) {   // CMD :143-163 (the three-argument method: N_lcl = 100, no rain — lh_defaults)
        const FT Nl = lh_defaults ? FT(100) : N_lcl, qr = lh_defaults ? FT(0) : q_rai, Nr = lh_defaults ? FT(0) : N_rai;
        const FT N = Nl + Nr, q = q_lcl + qr;
        // q = 0 (no condensate: the ordinary case) keeps its exact 0 through the select instead of through log2(0) = −Inf (cmx_math.hpp log2_floored)
        const FT x = q * rho * M::rcp(N);
        const FT r = M::exp2((log2_floored(x) + c.lh_l2k) * FT(1.0 / 3.0)) * c.lh_inv_k3;
        const FT r1 = x > FT(0) ? r : (x == FT(0) ? FT(0) : M::nan());      // a negative argument: the reference's ^(1/3) throws — NaN here, as before
        const FT r0 = N < c.eps_1m ? FT(0) : r1;
        reff_lh97 = (any_nan(rho, q_lcl) || any_nan(Nl, qr, Nr)) ? M::nan() : r0;
    }
}

template <typename FT, int VEC, bool LIMITED>
__global__ __launch_bounds__(kBlock) void cloud_diagnostics_kernel(const DiagConsts<FT> c, const DiagIO<FT> io, const bool lh_defaults, const int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    FT rho[VEC], ql[VEC], qr[VEC], Nl[VEC], Nr[VEC];
    const bool w1 = io.Z_1m, w2 = io.Z_2m, wr = io.reff_2m, wl = io.reff_lh97;
    if (i < nvec) {
        load_col<FT, VEC>(io.rho, i, rho);
        if (io.q_lcl) load_col<FT, VEC>(io.q_lcl, i, ql);
        if (io.q_rai) load_col<FT, VEC>(io.q_rai, i, qr);
        if (io.N_lcl) load_col<FT, VEC>(io.N_lcl, i, Nl);
        if (io.N_rai) load_col<FT, VEC>(io.N_rai, i, Nr);
    }
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (every lane of the workgroup reaches the barrier inside); no-op for Float32
    if (i >= nvec) return;
    FT o1[VEC], o2[VEC], o3[VEC], o4[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        o1[k] = o2[k] = o3[k] = o4[k] = FT(0);
        diag_point<FT, LIMITED>(c, w1, w2, wr, wl, lh_defaults, rho[k], io.q_lcl ? ql[k] : FT(0), io.q_rai ? qr[k] : FT(0), io.N_lcl ? Nl[k] : FT(0), io.N_rai ? Nr[k] : FT(0),
                                o1[k], o2[k], o3[k], o4[k]);
    }
    if (w1) store_col<FT, VEC>(io.Z_1m, i, o1);
    if (w2) store_col<FT, VEC>(io.Z_2m, i, o2);
    if (wr) store_col<FT, VEC>(io.reff_2m, i, o3);
    if (wl) store_col<FT, VEC>(io.reff_lh97, i, o4);
}

template <typename FT, typename RN, typename PC, typename PR>
static int32_t cloud_diagnostics_entry(const RN *rain, const PC *pdf_c, const PR *pdf_r, FT rho_w, uint32_t flags, int64_t n, const FT *rho, const FT *q_lcl,
                                       const FT *q_rai, const FT *N_lcl, const FT *N_rai, FT *Z_1m, FT *Z_2m, FT *reff_2m, FT *reff_lh97, void *stream) {
    if (n < 0 || (flags & ~(uint32_t)CMX_SB2006_LIMITED)) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (!Z_1m && !Z_2m && !reff_2m && !reff_lh97) return CMX_ERR_BAD_ARG;
    if ((Z_1m && !rain) || ((Z_2m || reff_2m) && (!pdf_c || !pdf_r)) || (reff_lh97 && !(rho_w > FT(0)))) return CMX_ERR_BAD_ARG;
    const bool limited = flags & CMX_SB2006_LIMITED;
    if ((Z_2m || reff_2m) && limited &&
        !(pdf_r->xr_min > 0 && pdf_r->xr_min <= pdf_r->xr_max && pdf_r->N0_min > 0 && pdf_r->N0_min <= pdf_r->N0_max && pdf_r->lambda_min > 0 &&
          pdf_r->lambda_min <= pdf_r->lambda_max))
        return CMX_ERR_BAD_ARG;      // the limited PSD clamps with its limiter pairs (as the rate entries: sb_limiters_ok)
    if (n == 0) return CMX_OK;
    if (!rho) return CMX_ERR_BAD_ARG;
    if (Z_1m && !q_rai) return CMX_ERR_BAD_ARG;
    if ((Z_2m || reff_2m) && (!q_lcl || !q_rai || !N_lcl || !N_rai)) return CMX_ERR_BAD_ARG;
    if (reff_lh97 && !q_lcl) return CMX_ERR_BAD_ARG;
    // the three-argument Liu–Hallett method (N_lcl = 100, no rain): all three of N_lcl, q_rai, N_rai NULL — only when nothing else needs them
    const bool lh_defaults = reff_lh97 && !N_lcl && !q_rai && !N_rai;
    if (reff_lh97 && !lh_defaults && (!N_lcl || !q_rai || !N_rai)) return CMX_ERR_BAD_ARG;

    DiagConsts<FT> c{};
    const double pi = 3.14159265358979323846, l2e = 1.4426950408889634074, l10_2 = 0.30102999566398119521;
    c.eps = Math<FT>::eps(); c.eps_1m = Math<FT>::eps_1m();
    if (Z_1m) {
        const auto &m = rain->mass;
        const double d = (double)m.me + (double)m.delta_m, e = 1.0 / (d + 1.0);
        const double denom = (double)m.chi_m * (double)m.m0 * std::fmax((double)rain->n0, (double)Math<FT>::eps_1m()) * (double)m.gamma_coeff;
        c.lam_a = (FT)e; c.lam_b = (FT)(e * std::log2(std::pow((double)m.r0, d) / denom)); c.lam_floor = (FT)std::log2((double)m.r0 * 1e-5);
        // Z = 720 n0·1e-12 (λ⁻¹/1e-3)⁷ ;  10 (log10 Z + 18 − 9) = 10 (log10(720 n0 1e-12) + 21 + 9) + 70 log10(2)·log2 λ⁻¹
        c.z1_c0 = (FT)(10.0 * (std::log10(720.0 * (double)rain->n0 * 1e-12) + 21.0 + 9.0)); c.z1_c1 = (FT)(70.0 * l10_2);
    }
    if (Z_2m || reff_2m) {
        const double C = (double)(FT)(4.0 / 3.0 * pi * (double)pdf_r->rho_w);       // FT(4/3 π ρw) as the reference rounds it
        c.l2_pi_rho_w = (FT)std::log2(pi * (double)pdf_r->rho_w);
        c.l2_xr_min = (FT)std::log2((double)pdf_r->xr_min); c.l2_xr_max = (FT)std::log2((double)pdf_r->xr_max);
        c.l2_N0_min = (FT)std::log2(std::fmax((double)pdf_r->N0_min, 1e-300)); c.l2_N0_max = (FT)std::log2(std::fmax((double)pdf_r->N0_max, 1e-300));
        c.l2_lam_min = (FT)std::log2(std::fmax((double)pdf_r->lambda_min, 1e-300)); c.l2_lam_max = (FT)std::log2(std::fmax((double)pdf_r->lambda_max, 1e-300));
        c.l2_6 = (FT)std::log2(6.0);
        // rain: Br = ∛(6/x̄) → B^(−n/μ) = (x̄/6)^(n/(3μ));  Γ((ν+1+n)/μ)/Γ((ν+1)/μ) and C^n folded into k
        const double nu_r = pdf_r->nu_r, mu_r = pdf_r->mu_r, nu_c = pdf_c->nu_c, mu_c = pdf_c->mu_c;
        auto kk = [&](double nu, double mu, double nn) { return (std::lgamma((nu + 1 + nn) / mu) - std::lgamma((nu + 1) / mu)) * l2e - nn * std::log2(C); };
        c.pr2 = (FT)(2.0 / (3.0 * mu_r)); c.pr1 = (FT)(1.0 / (3.0 * mu_r)); c.pr23 = (FT)((2.0 / 3.0) / (3.0 * mu_r));
        c.kr2 = (FT)kk(nu_r, mu_r, 2.0); c.kr1 = (FT)kk(nu_r, mu_r, 1.0); c.kr23 = (FT)kk(nu_r, mu_r, 2.0 / 3.0);
        // cloud: B = (x̄ Γ(z1)/Γ(z2))^(−μ) → B^(−n/μ) = (x̄ e^{dlg})^n
        c.dlg = (FT)((double)pdf_c->loggamma_z1 - (double)pdf_c->loggamma_z2); c.mu_c = (FT)mu_c;
        c.pc2 = FT(2); c.pc1 = FT(1); c.pc23 = (FT)(2.0 / 3.0);
        c.kc2 = (FT)kk(nu_c, mu_c, 2.0); c.kc1 = (FT)kk(nu_c, mu_c, 1.0); c.kc23 = (FT)kk(nu_c, mu_c, 2.0 / 3.0);
        c.lnB_hi = (FT)std::log((double)std::numeric_limits<FT>::max()); c.lnB_lo = (FT)std::log((double)std::numeric_limits<FT>::denorm_min());
    }
    if (reff_lh97) {
        c.lh_l2k = (FT)std::log2(3.0 / (4.0 * pi * (double)rho_w));
        c.lh_inv_k3 = (FT)(1.0 / std::cbrt(0.8));
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int VEC = sizeof(FT) == 8 ? 1 : Math<FT>::VEC;
    const void *ptrs[] = {rho, q_lcl, q_rai, N_lcl, N_rai, Z_1m, Z_2m, reff_2m, reff_lh97};
    bool vec_ok = true;
    for (const void *q : ptrs) vec_ok = vec_ok && (!q || aligned16(q));
    auto launch = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        auto off = [lo](auto *p) { return p ? p + lo : p; };
        const DiagIO<FT> io{rho + lo, off(q_lcl), off(q_rai), off(N_lcl), off(N_rai), off(Z_1m), off(Z_2m), off(reff_2m), off(reff_lh97)};
        const int64_t nvec = count / V;
        const dim3 grid((unsigned)((nvec + kBlock - 1) / kBlock));
        if (limited) hipLaunchKernelGGL((cloud_diagnostics_kernel<FT, V, true>), grid, dim3(kBlock), 0, s, c, io, lh_defaults, nvec);
        else hipLaunchKernelGGL((cloud_diagnostics_kernel<FT, V, false>), grid, dim3(kBlock), 0, s, c, io, lh_defaults, nvec);
    };
    const int64_t body = vec_ok ? (n / VEC) * VEC : 0;
    launch(std::integral_constant<int, VEC>{}, 0, body);
    launch(std::integral_constant<int, 1>{}, body, n - body);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {
int32_t cmx_cloud_diagnostics_f32(const cmx_rain_f32 *rain, const cmx_cloud_pdf_sb2006_f32 *pdf_c, const cmx_rain_pdf_sb2006_f32 *pdf_r, float rho_w, uint32_t flags,
                                  int64_t n, const float *rho, const float *q_lcl, const float *q_rai, const float *N_lcl, const float *N_rai, float *Z_1m, float *Z_2m,
                                  float *reff_2m, float *reff_lh97, void *stream) {
    return cmx::cloud_diagnostics_entry<float>(rain, pdf_c, pdf_r, rho_w, flags, n, rho, q_lcl, q_rai, N_lcl, N_rai, Z_1m, Z_2m, reff_2m, reff_lh97, stream);
}
int32_t cmx_cloud_diagnostics_f64(const cmx_rain_f64 *rain, const cmx_cloud_pdf_sb2006_f64 *pdf_c, const cmx_rain_pdf_sb2006_f64 *pdf_r, double rho_w, uint32_t flags,
                                  int64_t n, const double *rho, const double *q_lcl, const double *q_rai, const double *N_lcl, const double *N_rai, double *Z_1m,
                                  double *Z_2m, double *reff_2m, double *reff_lh97, void *stream) {
    return cmx::cloud_diagnostics_entry<double>(rain, pdf_c, pdf_r, rho_w, flags, n, rho, q_lcl, q_rai, N_lcl, N_rai, Z_1m, Z_2m, reff_2m, reff_lh97, stream);
}
}
