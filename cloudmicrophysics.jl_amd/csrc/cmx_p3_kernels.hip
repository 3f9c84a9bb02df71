// cmx_p3_kernels.hip — P3 ice scheme: state construction, regime thresholds, size-distribution shape solver
// (Brent on logλ ∈ [2, 17]) and mass-weighted mean diameter, one point per lane, for gfx950; C-ABI entry points of
// include/cmx.h §(7).
//
// Reference (src = /root/reference/src): Utilities.jl gamma_inc :93-144, regularised ratios :445-509;
// P3_particle_properties.jl P3State :43-56, state_from_prognostic :101-106, exprel / get_ρ_d :159-199, thresholds
// :222-272, regime_value / ice_mass_coeffs :320-356; P3_size_distribution.jl loggamma_inc_moment :97-109,
// get_μ :171-173, logmass_gamma_moment :193-200, logLdivN :211-216, get_logN₀ :233-237, get_distribution_logλ :284-320;
// P3_integral_properties.jl D_m :56-61.
//
// COMPUTE-bound: ≈12 residual evaluations per point × 8 incomplete-gamma evaluations × 20/30 fixed
// iterations ≈ 1e5 flops per point against 32–72 B of HBM traffic — the roofline is the FP64 (FP32) vector rate,
// not HBM.  What this kernel does about it:
//   * the four mass-regime coefficients (a_k, b_k), log a_k and the segment boundaries are per-point invariants
//     hoisted out of the solver; each residual evaluation needs lgamma for only TWO distinct z (b ∈ {3, β_va}) plus
//     μ+1 instead of the reference's five calls; e^{logλ} is formed once per evaluation;
//   * only the half (P or Q) of each incomplete-gamma pair that the segment difference needs is formed;
//   * the Brent iteration count is fixed (as in the reference: no data-dependent exit → no divergence from it).
// The solver is the same algorithm as the oracle's (Brent: inverse quadratic interpolation / secant with bisection
// safeguards), so both land on the same root where the SlopePowerLaw makes the residual multi-rooted.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "cmx_launch.hpp"
#include "cmx_math.hpp"

namespace cmx {

// accurate (OCML) elementary functions for the solver: the residual is a log-sum-exp of incomplete-gamma moments
template <typename FT> struct PM;
template <> struct PM<double> {
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double lgamma(double x) { return ::lgamma(x); }
    static __device__ __forceinline__ double expm1(double x) { return ::expm1(x); }
    static __device__ __forceinline__ double log1p(double x) { return ::log1p(x); }
    static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
    static __device__ __forceinline__ double tanh(double x) { return ::tanh(x); }
    static __device__ __forceinline__ double atanh(double x) { return ::atanh(x); }
    static __device__ __forceinline__ double log2(double x) { return ::log2(x); }
    static __device__ __forceinline__ double abs(double x) { return __builtin_fabs(x); }
    static constexpr int kBrent = 10, kGammaIters = 30;      // P3_size_distribution.jl:311, Utilities.jl:104
    static constexpr double eps() { return 2.220446049250313e-16; }
};
template <> struct PM<float> {
    static __device__ __forceinline__ float log(float x) { return ::logf(x); }
    static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
    static __device__ __forceinline__ float lgamma(float x) { return ::lgammaf(x); }
    static __device__ __forceinline__ float expm1(float x) { return ::expm1f(x); }
    static __device__ __forceinline__ float log1p(float x) { return ::log1pf(x); }
    static __device__ __forceinline__ float pow(float x, float y) { return ::powf(x, y); }
    static __device__ __forceinline__ float tanh(float x) { return ::tanhf(x); }
    static __device__ __forceinline__ float atanh(float x) { return ::atanhf(x); }
    static __device__ __forceinline__ float log2(float x) { return ::log2f(x); }
    static __device__ __forceinline__ float abs(float x) { return __builtin_fabsf(x); }
    static constexpr int kBrent = 8, kGammaIters = 20;
    static constexpr float eps() { return 1.1920928955078125e-07f; }
};

template <typename FT> struct P3Consts {
    uint32_t flags;
    int32_t brent_iters;   // fixed Brent iteration budget (P3_size_distribution.jl:311: 8 Float32 / 10 Float64)
    FT alpha_va, beta_va, slope_a, slope_b, slope_c, mu_max, mu_const, rho_i, rho_l_08;
    FT p_inv;            // 1/(3 − β_va)
    FT six_alpha_pi;     // 6 α_va / π
    FT a_sph_i, D_th;    // ρ_i π/6, (6 α_va/(π ρ_i))^(1/(3−β_va))
    FT pi_6;
};

template <typename FT, typename PR> static P3Consts<FT> make_p3_consts(const PR &p, uint32_t flags) {
    P3Consts<FT> c{};
    const double pi = 3.14159265358979323846;
    c.flags = flags;
    c.alpha_va = (FT)p.alpha_va; c.beta_va = (FT)p.beta_va;
    c.slope_a = (FT)p.slope_a; c.slope_b = (FT)p.slope_b; c.slope_c = (FT)p.slope_c; c.mu_max = (FT)p.mu_max; c.mu_const = (FT)p.mu_const;
    c.rho_i = (FT)p.rho_i; c.rho_l_08 = (FT)(0.8 * (double)p.rho_l);
    c.p_inv = (FT)(1.0 / (3.0 - (double)p.beta_va));
    c.six_alpha_pi = (FT)(6.0 * (double)p.alpha_va / pi);
    c.a_sph_i = (FT)((double)p.rho_i * pi / 6.0);
    c.D_th = (FT)std::pow(6.0 * (double)p.alpha_va / (pi * (double)p.rho_i), 1.0 / (3.0 - (double)p.beta_va));
    c.pi_6 = (FT)(pi / 6.0);
    return c;
}

// UT.gamma_inc — Utilities.jl:93-144.  Returns P if want_P else Q (the caller knows which one it will difference).
template <typename FT> __device__ FT gamma_inc_dev(FT a, FT x, FT lgam_a, bool want_P) {
    using P = PM<FT>;
    if (x <= FT(0)) return want_P ? FT(0) : FT(1);
    if (isinf(x)) return want_P ? FT(1) : FT(0);
    const FT factor = P::exp(a * P::log(x) - x - lgam_a);
    FT pq;   // P on the series branch, Q on the continued-fraction branch
    const bool series = x < a + FT(1);
    if (series) {
        FT term = FT(1) / a, sum = term;
        for (int k = 1; k <= P::kGammaIters; ++k) { term *= x / (a + FT(k)); sum += term; }
        pq = Math<FT>::min(Math<FT>::max(factor * sum, FT(0)), FT(1));
    } else {
        const FT tiny = FT(1e-30);
        const FT b1 = x + FT(1) - a;
        FT c = b1 + FT(1) / tiny, d = FT(1) / b1, h = d;
        for (int k = 1; k <= P::kGammaIters; ++k) {
            const FT ak = -FT(k) * (FT(k) - a), bk = x + FT(2 * k + 1) - a;
            const FT dt = bk + ak * d;
            d = P::abs(dt) < tiny ? tiny : dt;
            const FT ct = bk + ak / c;
            c = P::abs(ct) < tiny ? tiny : ct;
            d = FT(1) / d;
            h *= c * d;
        }
        pq = Math<FT>::min(Math<FT>::max(factor * h, FT(0)), FT(1));
    }
    return (series == want_P) ? pq : FT(1) - pq;
}

template <typename FT> struct P3Point {
    FT rho_q, rho_n, F_rim, rho_rim, rho_g;
    FT bnd[5];          // 0, D_th, D_gr, D_cr, ∞          segment_boundaries :280-291
    FT log_a[4], b[4];  // ice_mass_coeffs at each segment's midpoint :346-356
};

template <typename FT> __device__ __forceinline__ FT p3_mu(const P3Consts<FT> &c, FT loglam) {   // get_μ :171-173
    using P = PM<FT>;
    if (c.flags & CMX_P3_SLOPE_CONSTANT) return c.mu_const;
    return Math<FT>::min(Math<FT>::max(c.slope_a * P::exp(c.slope_b * loglam) - c.slope_c, FT(0)), c.mu_max);
}

// logmass_gamma_moment(state, μ, logλ; n) — :193-200 with loggamma_inc_moment :97-109 and unrolled_logsumexp
template <typename FT> __device__ FT p3_logmass_moment(const P3Consts<FT> &c, const P3Point<FT> &s, FT mu, FT loglam, FT n) {
    using P = PM<FT>;
    const FT lam = P::exp(loglam);
    // lgamma for the two distinct z: b = 3 (spherical regimes) and b = β_va (power-law regimes)
    const FT z_sph = FT(3) + n + mu + FT(1), z_pow = c.beta_va + n + mu + FT(1);
    const FT lg_sph = P::lgamma(z_sph), lg_pow = P::lgamma(z_pow);
    FT m[4];
    FT xmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const FT D1 = s.bnd[i], D2 = s.bnd[i + 1];
        FT val = -INFINITY;
        if (D1 < D2) {
            const bool sph = s.b[i] == FT(3);
            const FT z = sph ? z_sph : z_pow, lg = sph ? lg_sph : lg_pow;
            const FT x1 = D1 * lam, x2 = D2 * lam;
            const bool use_P = x2 < z + FT(1);
            const FT g1 = gamma_inc_dev<FT>(z, x1, lg, use_P), g2 = gamma_inc_dev<FT>(z, x2, lg, use_P);
            FT dq = use_P ? g2 - g1 : g1 - g2;
            dq = Math<FT>::max(dq, P::eps());
            val = -z * loglam + lg + P::log(dq) + s.log_a[i];
        }
        m[i] = val;
        xmax = (val > xmax || isnan(val)) ? val : xmax;
    }
    if (!isfinite(xmax)) return xmax;
    FT sum = FT(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += P::exp(m[i] - xmax);
    return xmax + P::log(sum);
}

template <typename FT> __device__ __forceinline__ FT p3_logLdivN(const P3Consts<FT> &c, const P3Point<FT> &s, FT loglam) {   // :211-216
    const FT mu = p3_mu<FT>(c, loglam);
    return p3_logmass_moment<FT>(c, s, mu, loglam, FT(0)) - (-(mu + FT(1)) * loglam + PM<FT>::lgamma(mu + FT(1)));
}

template <typename FT> __device__ __forceinline__ FT exprel1(FT x) { return PM<FT>::expm1(x) / x; }
template <typename FT> __device__ __forceinline__ FT exprel2(FT x) {   // P3_particle_properties.jl:161-166
    using P = PM<FT>;
    if (P::abs(x) < FT(0.2)) {
        FT r = FT(1.0 / 362880);
        r = r * x + FT(1.0 / 40320); r = r * x + FT(1.0 / 5040); r = r * x + FT(1.0 / 720); r = r * x + FT(1.0 / 120);
        r = r * x + FT(1.0 / 24); r = r * x + FT(1.0 / 6); r = r * x + FT(0.5);
        return r;
    }
    return (P::expm1(x) - x) / (x * x);
}
template <typename FT> __device__ __forceinline__ FT regularised_ratio(FT num, FT den) {   // Utilities.jl:445-488
    using P = PM<FT>;
    const FT half = P::eps();
    FT w;
    if (den < FT(0)) w = FT(0);
    else if (den > Math<FT>::min(FT(1), FT(42) * half)) w = FT(1);
    else if (FT(4) * den < P::eps()) w = FT(0);
    else w = (FT(1) + P::tanh(FT(2) * P::atanh(FT(1) - FT(2) * P::pow(FT(1) - den, FT(-1) / P::log2(FT(1) - half))))) / FT(2);
    return den < P::eps() * P::eps() ? FT(0) : w * num / den;
}

template <typename FT> struct P3IO { const FT *rho_q, *rho_n, *x3, *x4; FT *F_rim, *rho_rim, *loglam, *D_m, *logN0; };

template <typename FT>
__global__ __launch_bounds__(kBlock) void p3_shape_kernel(const P3Consts<FT> c, const P3IO<FT> io, const int64_t n) {
    using P = PM<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    P3Point<FT> s;
    s.rho_q = io.rho_q[i]; s.rho_n = io.rho_n[i];
    if (c.flags & CMX_P3_INPUT_IS_STATE) {
        s.F_rim = io.x3[i]; s.rho_rim = io.x4[i];
    } else {   // state_from_prognostic :101-106
        const FT q_rim = io.x3[i], b_rim = io.x4[i];
        s.F_rim = Math<FT>::min(regularised_ratio<FT>(Math<FT>::min(q_rim, s.rho_q), s.rho_q), FT(1) - P::eps());
        s.rho_rim = Math<FT>::min(regularised_ratio<FT>(q_rim, b_rim), c.rho_l_08);
    }
    // P3State :43-56 — ρ_d (exact solution :191-199), ρ_g, thresholds
    {
        const FT p = c.p_inv, logFu = P::log1p(-s.F_rim);
        const FT phi1 = exprel1<FT>(logFu), phi1mp = exprel1<FT>((FT(1) - p) * logFu);
        const FT H = -p * exprel2<FT>(-p * logFu) - (FT(1) - p) * exprel2<FT>((FT(1) - p) * logFu);
        const FT rho_d = -(s.rho_rim * phi1 * phi1mp) / (H - phi1mp * phi1);
        s.rho_g = s.F_rim * s.rho_rim + (FT(1) - s.F_rim) * rho_d;
    }
    const bool unrimed = s.F_rim == FT(0);
    const FT D_gr = unrimed ? FT(INFINITY) : P::pow(c.six_alpha_pi / s.rho_g, c.p_inv);
    const FT D_cr = unrimed ? FT(INFINITY) : P::pow(c.six_alpha_pi / (s.rho_g * (FT(1) - s.F_rim)), c.p_inv);
    s.bnd[0] = FT(0); s.bnd[1] = c.D_th; s.bnd[2] = D_gr; s.bnd[3] = D_cr; s.bnd[4] = FT(INFINITY);
    const FT Fu = Math<FT>::max(FT(1) - s.F_rim, P::eps());
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // regime_value at the segment midpoint :320-332
        const FT D = (s.bnd[k] + s.bnd[k + 1]) / FT(2);
        FT a, b;
        if (D < c.D_th) { a = c.a_sph_i; b = FT(3); }
        else if (unrimed) { a = c.alpha_va; b = c.beta_va; }
        else if (D < D_gr) { a = c.alpha_va; b = c.beta_va; }
        else if (D < D_cr) { a = s.rho_g * c.pi_6; b = FT(3); }
        else { a = c.alpha_va / Fu; b = c.beta_va; }
        s.log_a[k] = P::log(a); s.b[k] = b;
    }
    // get_distribution_logλ :284-320
    FT loglam;
    if (s.rho_n < P::eps() || s.rho_q < P::eps()) {
        loglam = -INFINITY;
    } else {
        const FT target = P::log(s.rho_q) - P::log(s.rho_n);
        FT a = FT(2), b = FT(17);
        FT fa = p3_logLdivN<FT>(c, s, a) - target, fb = p3_logLdivN<FT>(c, s, b) - target;
        if (!isfinite(fa) || !isfinite(fb) || fa * fb > FT(0)) {
            loglam = P::abs(fa) <= P::abs(fb) ? a : b;
        } else {
            if (P::abs(fa) < P::abs(fb)) { FT t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
            FT cc = a, fc = fa, d = FT(0);
            bool mflag = true;
            for (int it = 0; it < c.brent_iters; ++it) {
                if (fb == FT(0) || a == b) break;
                FT sx;
                if (fa != fc && fb != fc)
                    sx = a * fb * fc / ((fa - fb) * (fa - fc)) + b * fa * fc / ((fb - fa) * (fb - fc)) + cc * fa * fb / ((fc - fa) * (fc - fb));
                else
                    sx = b - fb * (b - a) / (fb - fa);
                const FT lo3 = (FT(3) * a + b) / FT(4);
                const bool out_of_range = !((sx > Math<FT>::min(lo3, b)) && (sx < Math<FT>::max(lo3, b)));
                if (out_of_range || (mflag && P::abs(sx - b) >= P::abs(b - cc) / FT(2)) || (!mflag && P::abs(sx - b) >= P::abs(cc - d) / FT(2))) {
                    sx = (a + b) / FT(2);
                    mflag = true;
                } else {
                    mflag = false;
                }
                const FT fs = p3_logLdivN<FT>(c, s, sx) - target;
                d = cc; cc = b; fc = fb;
                if (fa * fs < FT(0)) { b = sx; fb = fs; } else { a = sx; fa = fs; }
                if (P::abs(fa) < P::abs(fb)) { FT t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
            }
            loglam = b;
        }
    }
    if (io.F_rim) io.F_rim[i] = s.F_rim;
    if (io.rho_rim) io.rho_rim[i] = s.rho_rim;
    if (io.loglam) io.loglam[i] = loglam;
    if (io.D_m || io.logN0) {
        const FT mu = p3_mu<FT>(c, loglam);
        const FT logN0 = P::log(s.rho_n) - (-(mu + FT(1)) * loglam + P::lgamma(mu + FT(1)));   // get_logN₀ :233-237
        if (io.logN0) io.logN0[i] = logN0;
        if (io.D_m) io.D_m[i] = P::exp(logN0 + p3_logmass_moment<FT>(c, s, mu, loglam, FT(1))) / s.rho_q;   // D_m :56-61
    }
}

template <typename FT, typename PR>
static int32_t p3_entry(const PR *params, uint32_t flags, int32_t brent_iters, int64_t n, const FT *rho_q, const FT *rho_n, const FT *x3, const FT *x4,
                        FT *F_rim, FT *rho_rim, FT *loglam, FT *D_m, FT *logN0, void *stream) {
    if (!params || n < 0 || (flags & ~(CMX_P3_INPUT_IS_STATE | CMX_P3_SLOPE_CONSTANT))) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(*params, flags);
    c.brent_iters = brent_iters > 0 ? brent_iters : PM<FT>::kBrent;
    P3IO<FT> io{rho_q, rho_n, x3, x4, F_rim, rho_rim, loglam, D_m, logN0};
    hipLaunchKernelGGL((p3_shape_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_p3_shape_f32(const cmx_p3_params_f32 *params, uint32_t flags, int32_t brent_iters, int64_t n, const float *rho_q_ice, const float *rho_n_ice,
                         const float *x3, const float *x4, float *F_rim, float *rho_rim, float *log_lambda, float *D_m,
                         float *log_N0, void *stream) {
    return cmx::p3_entry<float>(params, flags, brent_iters, n, rho_q_ice, rho_n_ice, x3, x4, F_rim, rho_rim, log_lambda, D_m, log_N0, stream);
}
int32_t cmx_p3_shape_f64(const cmx_p3_params_f64 *params, uint32_t flags, int32_t brent_iters, int64_t n, const double *rho_q_ice, const double *rho_n_ice,
                         const double *x3, const double *x4, double *F_rim, double *rho_rim, double *log_lambda, double *D_m,
                         double *log_N0, void *stream) {
    return cmx::p3_entry<double>(params, flags, brent_iters, n, rho_q_ice, rho_n_ice, x3, x4, F_rim, rho_rim, log_lambda, D_m, log_N0, stream);
}

}  // extern "C"
