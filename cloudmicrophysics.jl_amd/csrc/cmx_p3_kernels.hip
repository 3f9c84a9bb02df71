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

// elementary functions for the solver (the residual is a log-sum-exp of incomplete-gamma moments): lean exp/log
// (cmx_lean_f64.hpp) in Float64, OCML for the rest and for Float32
template <typename FT> struct PM;
template <> struct PM<double> {
    // one-argument forms: OCML.  In the shape solver (a dozen call sites around the incomplete-gamma loops) the inlined lean
    // routines push the kernel from 2 waves/SIMD to 1 (24 → 30 ms per 1e7 columns); the quadrature loops use the
    // register-pinned lean forms below.
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
    // 1/d to ≈1 ulp: v_rcp_f64 (≈2⁻²⁴ relative) + two Newton steps — no div_scale / div_fmas / div_fixup sequence.
    // Only for finite, normal d (the incomplete-gamma loops: d = a + k, or a rescaled continued-fraction denominator).
    static __device__ __forceinline__ double rcp(double d) { return lean::rcp_finite(d); }
    // hot-loop variants with register-pinned constants (cmx_lean_f64.hpp)
    using Coefs = lean::PinnedCoefs;
    static __device__ __forceinline__ Coefs coefs() { return lean::pinned_coefs(); }
    static __device__ __forceinline__ void pin(double &x) { lean::pin(x); }
    static __device__ __forceinline__ double exp(double x, const Coefs &k) { return lean::exp(x, k); }
    static __device__ __forceinline__ double log(double x, const Coefs &k) { return lean::log(x, k); }
    static constexpr int kBrent = 10, kGammaIters = 30;      // P3_size_distribution.jl:311, Utilities.jl:104
    static constexpr int kRescale = 6;                        // continued-fraction rescale period (b ≤ 1e8 → 1e48 growth)
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
    struct Coefs {};
    static __device__ __forceinline__ Coefs coefs() { return {}; }
    static __device__ __forceinline__ void pin(float &) {}
    static __device__ __forceinline__ float exp(float x, const Coefs &) { return ::expf(x); }
    static __device__ __forceinline__ float log(float x, const Coefs &) { return ::logf(x); }
    static __device__ __forceinline__ float rcp(float d) {
        const float r = __builtin_amdgcn_rcpf(d);
        return __builtin_fmaf(r, __builtin_fmaf(-d, r, 1.0f), r);
    }
    static constexpr int kBrent = 8, kGammaIters = 20;
    static constexpr int kRescale = 2;                        // Float32: (1e8)² < 3e38
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

// UT.gamma_inc — Utilities.jl:93-144: series for x < a+1, Lentz continued fraction otherwise, both with a FIXED
// number of terms (20 Float32 / 30 Float64).  The device evaluates the same truncations in cheaper arithmetic:
//   * series: Σ_k x^k / (a(a+1)…(a+k)) with the reciprocal of (a+k) from rcp() instead of an IEEE division;
//   * continued fraction: the n-th convergent h_n = A_n/B_n of  1/(b₀+ a₁/(b₁+ a₂/(b₂+…)))  by the forward (Wallis)
//     recurrence A_n = b_n A_{n−1} + a_n A_{n−2} — the value modified Lentz produces with two divisions per term —
//     rescaled every kRescale terms; one reciprocal at the end.
// `gamma_series` / `gamma_cf` return the bracketed sums WITHOUT the prefactor x^a e^{−x}/Γ(a).
template <typename FT> __device__ __forceinline__ FT gamma_series_sum(FT a, FT x) {
    using P = PM<FT>;
    FT term = P::rcp(a), sum = term;
#pragma unroll 2
    for (int k = 1; k <= P::kGammaIters; ++k) { term *= x * P::rcp(a + FT(k)); sum += term; }
    return sum;
}
template <typename FT> __device__ __forceinline__ FT gamma_cf_value(FT a, FT x) {
    using P = PM<FT>;
    static_assert(P::kGammaIters % P::kRescale == 0, "rescale period must divide the term count");
    // h₀ = 1/b₀:  A₀ = 1, B₀ = b₀;  A₋₁ = 0, B₋₁ = 1
    const FT b0 = x + FT(1) - a;
    FT Am = FT(0), Bm = FT(1), A = FT(1), B = b0;
#pragma unroll 1
    for (int k0 = 0; k0 < P::kGammaIters; k0 += P::kRescale) {
#pragma unroll
        for (int j = 1; j <= P::kRescale; ++j) {
            const FT kk = FT(k0 + j);
            const FT ak = -kk * (kk - a), bk = b0 + FT(2) * kk;
            const FT An = bk * A + ak * Am, Bn = bk * B + ak * Bm;
            Am = A; Bm = B; A = An; B = Bn;
        }
        const FT r = P::rcp(B);
        A *= r; Am *= r; Bm *= r; B = FT(1);
    }
    return A;
}
template <typename FT> __device__ FT gamma_inc_dev(FT a, FT x, FT lgam_a, bool want_P) {
    using P = PM<FT>;
    if (x <= FT(0)) return want_P ? FT(0) : FT(1);
    if (isinf(x)) return want_P ? FT(1) : FT(0);
    const FT factor = P::exp(a * P::log(x) - x - lgam_a);
    const bool series = x < a + FT(1);
    const FT body = series ? gamma_series_sum<FT>(a, x) : gamma_cf_value<FT>(a, x);
    const FT pq = Math<FT>::min(Math<FT>::max(factor * body, FT(0)), FT(1));   // P on the series branch, Q on the other
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

template <typename FT> struct P3IO { const FT *rho_q, *rho_n, *x3, *x4, *guess; FT *F_rim, *rho_rim, *loglam, *D_m, *logN0; };

// state_from_prognostic :101-106 (or P3State from (F_rim, ρ_rim)) → P3State :43-56: ρ_d (exact solution :191-199), ρ_g,
// thresholds, and the per-segment mass-law coefficients (regime_value at the segment midpoint :320-332)
template <typename FT>
__device__ __forceinline__ void p3_make_point(const P3Consts<FT> &c, FT rho_q, FT rho_n, FT x3, FT x4, P3Point<FT> &s) {
    using P = PM<FT>;
    s.rho_q = rho_q; s.rho_n = rho_n;
    if (c.flags & CMX_P3_INPUT_IS_STATE) {
        s.F_rim = x3; s.rho_rim = x4;
    } else {
        s.F_rim = Math<FT>::min(regularised_ratio<FT>(Math<FT>::min(x3, s.rho_q), s.rho_q), FT(1) - P::eps());
        s.rho_rim = Math<FT>::min(regularised_ratio<FT>(x3, x4), c.rho_l_08);
    }
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
    for (int k = 0; k < 4; ++k) {
        const FT D = (s.bnd[k] + s.bnd[k + 1]) / FT(2);
        FT a, b;
        if (D < c.D_th) { a = c.a_sph_i; b = FT(3); }
        else if (unrimed) { a = c.alpha_va; b = c.beta_va; }
        else if (D < D_gr) { a = c.alpha_va; b = c.beta_va; }
        else if (D < D_cr) { a = s.rho_g * c.pi_6; b = FT(3); }
        else { a = c.alpha_va / Fu; b = c.beta_va; }
        s.log_a[k] = P::log(a); s.b[k] = b;
    }
}

template <typename FT>
__global__ __launch_bounds__(kBlock) void p3_shape_kernel(const P3Consts<FT> c, const P3IO<FT> io, const int64_t n) {
    using P = PM<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    P3Point<FT> s;
    p3_make_point<FT>(c, io.rho_q[i], io.rho_n[i], io.x3[i], io.x4[i], s);
    // get_distribution_logλ :284-320.  All residual evaluations (the two bracket ends, the optional warm-start guess
    // of _narrow_bracket :336-353, the Brent iterations) go through ONE inlined call site: steps −3, −2, −1, 0 … —
    // three separate inlined copies of the residual doubled the VGPR count (255, occupancy 1).
    FT loglam;
    if (s.rho_n < P::eps() || s.rho_q < P::eps()) {
        loglam = -INFINITY;
    } else {
        const FT target = P::log(s.rho_q) - P::log(s.rho_n);
        FT a = FT(2), b = FT(17), fa = FT(0), fb = FT(0), cc = FT(0), fc = FT(0), d = FT(0);
        bool mflag = true, active = true, guess_valid = false;
        const int first = io.guess ? -3 : -2;          // wave-uniform
        for (int it = first; it < c.brent_iters; ++it) {
            // which abscissa this step evaluates
            FT sx;
            const int step = it == first ? -3 : (it == first + 1 ? -2 : it);   // −3: lo end, −2: hi end, −1: guess
            if (step == -3) sx = a;
            else if (step == -2) sx = b;
            else if (step == -1) {
                const FT pg = io.guess[i];
                guess_valid = isfinite(pg) && (a < pg && pg < b);
                sx = guess_valid ? pg : a;
            } else {
                if (!active || fb == FT(0) || a == b) { active = false; continue; }
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
            }
            const FT fs = p3_logLdivN<FT>(c, s, sx) - target;
            // what the step does with the value
            if (step == -3) {
                fa = fs;
            } else if (step == -2) {
                fb = fs;
                if (!isfinite(fa) || !isfinite(fb) || fa * fb > FT(0)) {        // no sign change: nearer end (:295-297)
                    const bool lo_end = P::abs(fa) <= P::abs(fb);
                    b = lo_end ? a : b; fb = lo_end ? fa : fb;
                    active = false;
                }
            } else if (step == -1) {
                if (active) {
                    const bool valid = guess_valid && isfinite(fs);
                    const bool left = valid && (fa * fs < FT(0)), right = valid && !left;
                    if (left) { b = sx; fb = fs; }
                    if (right) { a = sx; fa = fs; }
                }
            } else {
                d = cc; cc = b; fc = fb;
                if (fa * fs < FT(0)) { b = sx; fb = fs; } else { a = sx; fa = fs; }
                if (P::abs(fa) < P::abs(fb)) { FT t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
            }
            // entering the Brent phase: order the bracket so that b is the better end, c = a
            if (step < 0 && (io.guess ? step == -1 : step == -2) && active) {
                if (P::abs(fa) < P::abs(fb)) { FT t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
                cc = a; fc = fa; d = FT(0); mflag = true;
            }
        }
        loglam = b;
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
                        const FT *guess, FT *F_rim, FT *rho_rim, FT *loglam, FT *D_m, FT *logN0, void *stream) {
    if (!params || n < 0 || (flags & ~(CMX_P3_INPUT_IS_STATE | CMX_P3_SLOPE_CONSTANT | CMX_P3_NO_ASPECT_RATIO))) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(*params, flags);
    c.brent_iters = brent_iters > 0 ? brent_iters : PM<FT>::kBrent;
    P3IO<FT> io{rho_q, rho_n, x3, x4, guess, F_rim, rho_rim, loglam, D_m, logN0};
    hipLaunchKernelGGL((p3_shape_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Number- and mass-weighted fall speeds — src/P3_terminal_velocity.jl:72-91,118-137.
//
// Per quadrature node the reference evaluates n(D)·v(D)[·m(D)] with ≈12 pow/exp/log/cbrt; here every factor is a
// power law or an exponential of D inside one mass-regime segment, so the whole integrand is assembled in the log
// domain from ONE log(D):  n·v = Σ_k A_k exp(base + e_k + b_k logD − c_k D),  base = logN₀ + μ logD − λD, where
//   * (A_k, e_k, b_k, c_k) are the Chen-2022 small- or large-ice terms (selected per node by D ≤ cutoff) with the
//     parameter-only table reductions at ρᵢ = 916.7 folded on the host and the ρₐ-dependent prefactors once per point;
//     the two terms have opposite signs and cancel to ≈1/200 of their size for small D, so the sum is formed as
//     e^{base+E₁}·(A₁ + A₂ e^{E₂−E₁}) — the large shared factor stays OUTSIDE the difference, as D^b does in the reference;
//   * the aspect factor cbrt(ϕᵢ) is exactly 1 on the two spherical segments (small ice, graupel), a pure power law
//     of D on the unrimed / dense-rimed segment (folded into e_k, b_k: no extra transcendental), and needs the mixed
//     area F·πD²/4 + (1−F)·γD^σ only on the partially-rimed segment (one exp + one log more);
//   * m(D) = exp(log a_seg + b_seg logD) with the per-segment mass law of the shape solver.
// → 3–4 transcendentals per node (log D, exp of the shared factor, exp of the term ratio[, D^β]) instead of ≈12; nodes/weights are wave-uniform scalar loads from the kernel arguments.
template <typename FT> struct P3VelConsts {
    // small ice (table B3 reduced at ρᵢ): aᵢ = (Es, Fs)·ρₐ^As·1000^b, b = Bs + ρₐ Cs, c = (0, 1000 Gs)
    FT s_A, s_B, s_C, s_E, s_F, s_c2;
    // large ice (table B5 reduced): a = (Bl ρₐ^Al 1000^Cl, El ρₐ^Al e^{Hl ρₐ} 1000^Fl), b = (Cl, Fl), c = (0, 1000 Gl)
    FT l_A, l_a1, l_b1, l_a2, l_H, l_b2, l_c2;
    FT cutoff, ln1000;
    FT g0, g1;          // unrimed / dense-rimed aspect factor: cbrt ϕ = exp(g0 + g1 logD)
    FT h0_num;          // partially rimed: cbrt ϕ = exp((h0_num − log Fu)/3 + β/3 logD − ½ log area)
    FT pi_4, gamma_area, sigma_area;
    FT p_lo, p_hi;      // FT(p), FT(1 − p)
    // ice_melt (P3_processes.jl:64-94): F_v = vent_a + vent_bc √(D v), vent_bc = b_v ∛(ν/D_v)/√ν; L_f(T) = LH_f0 + dcp_f (T − T_0)
    FT vent_a, vent_bc, K4, LH_f0, dcp_f, T_0, T_freeze;
};

template <typename FT, typename PR, typename VR>
static P3VelConsts<FT> make_p3_vel_consts(const PR &pr, const VR &vel, double p) {
    P3VelConsts<FT> v{};
    const double pi = 3.14159265358979323846, rho_i = 916.7;   // src/P3_terminal_velocity.jl:41
    const double l = std::log(rho_i), sq = std::sqrt(rho_i);
    const auto &s = vel.small_ice;
    const auto &g = vel.large_ice;
    v.s_A = (FT)((double)s.A[1] * l * l - (double)s.A[2] * l + (double)s.A[0]);
    v.s_B = (FT)(1.0 / ((double)s.B[0] + (double)s.B[1] * l + (double)s.B[2] / sq));
    v.s_C = (FT)((double)s.C[0] + (double)s.C[1] * std::exp((double)s.C[2] * rho_i) + (double)s.C[3] * sq);
    v.s_E = (FT)((double)s.E[0] - (double)s.E[1] * l * l + (double)s.E[2] * sq);
    v.s_F = (FT)(-std::exp((double)s.F[0] - (double)s.F[1] * l * l + (double)s.F[2] * l));
    v.s_c2 = (FT)(1000.0 / ((double)s.G[0] + (double)s.G[1] / l - (double)s.G[2] * l / rho_i));
    const double Al = (double)g.A[0] + (double)g.A[1] * l + (double)g.A[2] / (rho_i * sq);
    const double Bl = std::exp((double)g.B[0] + (double)g.B[1] * l * l + (double)g.B[2] * l);
    const double Cl = std::exp((double)g.C[0] + (double)g.C[1] / l + (double)g.C[2] / rho_i);
    const double El = (double)g.E[0] + (double)g.E[1] * l * sq + (double)g.E[2] * sq;
    const double Fl = (double)g.F[0] + (double)g.F[1] * l - std::exp(std::log(-(double)g.F[2]) - rho_i);
    const double Gl = 1.0 / ((double)g.G[0] + (double)g.G[1] * l * sq + (double)g.G[2] / sq);
    const double Hl = (double)g.H[0] + (double)g.H[1] * rho_i * rho_i * sq + std::exp(std::log(-(double)g.H[2]) - rho_i);
    v.l_A = (FT)Al; v.l_a1 = (FT)(Bl * std::pow(1000.0, Cl)); v.l_b1 = (FT)Cl;
    v.l_a2 = (FT)(El * std::pow(1000.0, Fl)); v.l_H = (FT)Hl; v.l_b2 = (FT)Fl; v.l_c2 = (FT)(1000.0 * Gl);
    v.cutoff = (FT)s.cutoff; v.ln1000 = (FT)std::log(1000.0);
    const double al = (double)pr.alpha_va, be = (double)pr.beta_va, ga = (double)pr.gamma, si = (double)pr.sigma, ri = (double)pr.rho_i;
    v.g0 = (FT)(std::log(3.0 * std::sqrt(pi) * al / (4.0 * ri * ga * std::sqrt(ga))) / 3.0);
    v.g1 = (FT)((be - 1.5 * si) / 3.0);
    v.h0_num = (FT)std::log(3.0 * std::sqrt(pi) * al / (4.0 * ri));
    v.pi_4 = (FT)(pi / 4.0); v.gamma_area = (FT)ga; v.sigma_area = (FT)si;
    v.p_lo = (FT)p; v.p_hi = (FT)(1.0 - p);
    return v;
}

// UT._gamma_inc_inv — Utilities.jl:205-252 (Halley on P − p or Q − q, ≤ 15 iterations)
template <typename FT> __device__ FT gamma_inc_inv_dev(FT a, FT p, FT q) {
    using P = PM<FT>;
    if (p <= FT(0)) return FT(0);
    if (q <= FT(0)) return FT(INFINITY);
    const FT lg = P::lgamma(a);
    FT x = p < FT(0.5) ? P::exp((P::log(p) + lg + P::log(a)) / a) : a - P::log(q);   // (p Γ(a+1))^{1/a}
    const bool use_q = p > FT(0.5);
    for (int it = 0; it < 15; ++it) {
        const FT g = gamma_inc_dev<FT>(a, x, lg, !use_q);
        const FT f = use_q ? g - q : g - p;
        FT fprime = P::exp((a - FT(1)) * P::log(x) - x - lg);
        if (use_q) fprime = -fprime;
        if (fprime == FT(0)) break;
        const FT r = (a - FT(1) - x) / x;
        FT step = f / (fprime * (FT(1) - FT(0.5) * f * r / fprime));
        if (x - step <= FT(0)) step = FT(0.5) * x;
        x = x - step;
        if (P::abs(step) < P::eps() * x) break;
    }
    return x;
}

template <typename FT, typename QUAD> struct P3VelIO {
    const FT *rho_q, *rho_n, *x3, *x4, *rho_a, *loglam; FT *v_n, *v_m;
    const FT *T; FT *dNdt, *dLdt;     // MELT mode
};

template <typename FT, typename QUAD, bool ASPECT, bool MELT = false>
__global__ __launch_bounds__(kBlock) void p3_velocity_kernel(const P3Consts<FT> c, const P3VelConsts<FT> v, const QUAD quad,
                                                            const P3VelIO<FT, QUAD> io, const int64_t n) {
    using P = PM<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    P3Point<FT> s;
    p3_make_point<FT>(c, io.rho_q[i], io.rho_n[i], io.x3[i], io.x4[i], s);
    FT vn = FT(0), vm = FT(0);
    if (!(s.rho_n < P::eps() || s.rho_q < P::eps())) {
        const FT loglam = io.loglam[i], lam = P::exp(loglam), mu = p3_mu<FT>(c, loglam);
        const FT logN0 = P::log(s.rho_n) - (-(mu + FT(1)) * loglam + P::lgamma(mu + FT(1)));
        // Chen-2022 coefficients at this air density — Common.jl:304-350
        const FT rho_a = Math<FT>::max(io.rho_a[i], FT(0)), lra = P::log(rho_a);
        const FT sb = v.s_B + rho_a * v.s_C, se = v.s_A * lra + sb * v.ln1000;          // small: both terms share e, b
        const FT le1 = v.l_A * lra, le2 = le1 + v.l_H * rho_a;                           // large
        // integral_bounds — P3_integral_properties.jl:34-46
        const FT D_min = gamma_inc_inv_dev<FT>(mu + FT(1), v.p_lo, FT(1) - v.p_lo) / lam;
        const FT D_max = gamma_inc_inv_dev<FT>(mu + FT(1), v.p_hi, FT(1) - v.p_hi) / lam;
        FT bnd[5];
        bnd[0] = D_min; bnd[4] = D_max;
#pragma unroll
        for (int k = 1; k < 4; ++k) bnd[k] = Math<FT>::min(Math<FT>::max(s.bnd[k], D_min), D_max);
        const FT Fu = Math<FT>::max(FT(1) - s.F_rim, P::eps());
        const FT h0 = (v.h0_num - P::log(Fu)) / FT(3), h1 = c.beta_va / FT(3);
        FT sum_n = FT(0), sum_m = FT(0);
        const typename P::Coefs kc = P::coefs();          // exp / log constants pinned in VGPRs for the node loops
        // … and so are the Chen-2022 / area constants the node loop reads (Float64 kernel arguments are SGPR pairs too)
        FT k_cut = v.cutoff, k_sc2 = v.s_c2, k_lb1 = v.l_b1, k_db = v.l_b2 - v.l_b1, k_lc2 = v.l_c2, k_sE = v.s_E, k_sF = v.s_F,
           k_la1 = v.l_a1, k_la2 = v.l_a2, k_pi4 = v.pi_4, k_ga = v.gamma_area, k_sa = v.sigma_area;
        P::pin(k_cut); P::pin(k_sc2); P::pin(k_lb1); P::pin(k_db); P::pin(k_lc2); P::pin(k_sE); P::pin(k_sF); P::pin(k_la1); P::pin(k_la2);
        P::pin(k_pi4); P::pin(k_ga); P::pin(k_sa);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const FT a = bnd[k], b = bnd[k + 1];
            if (!(a < b)) continue;
            const FT scale = (b - a) / FT(2), shift = (a + b) / FT(2);
            // aspect factor of this segment as exp(q0 + q1 logD [− ½ log area])
            FT q0 = FT(0), q1 = FT(0);
            const bool mixed_area = ASPECT && k == 3;
            if (ASPECT && k == 1) { q0 = v.g0; q1 = v.g1; }
            if (mixed_area) { q0 = h0; q1 = h1; }
            const FT lam_ = lam, mb = s.b[k], mla = s.log_a[k];
            const bool sph_mass = mb == FT(3);
            const FT ma = P::exp(mla);
            FT rn = FT(0), rm = FT(0);
            for (int j = 0; j < quad.n; ++j) {
                const FT x = scale * quad.node[j] + shift, w = quad.weight[j];
                const FT logD = P::log(x, kc);
                const FT eN = logN0 + mu * logD - lam_ * x;                 // log n(D)
                FT eA = q0 + q1 * logD;                                      // log of the aspect factor
                if (mixed_area) {
                    const FT area = s.F_rim * k_pi4 * x * x + (FT(1) - s.F_rim) * k_ga * P::exp(k_sa * logD, kc);
                    eA -= FT(0.5) * P::log(area, kc);
                }
                // Chen-2022 particle speed Σ aₖ D^bₖ e^{−cₖD}: the two terms have opposite signs and cancel to ≈1/200 of
                // their size for small D, so the shared factor stays OUTSIDE the difference (as D^b does in the reference)
                const bool small = x <= k_cut;
                const FT E1 = small ? se + sb * logD : le1 + k_lb1 * logD;
                const FT dE = small ? -k_sc2 * x : (le2 - le1) + k_db * logD - k_lc2 * x;   // E2 − E1
                const FT A1 = small ? k_sE : k_la1, A2 = small ? k_sF : k_la2;
                const FT S = A1 + A2 * P::exp(dE, kc);
                if constexpr (!MELT) {
                    const FT nv = P::exp(eN + eA + E1, kc) * S;
                    // m(D) = a D^b: b = 3 on the spherical segments (no transcendental), β_va otherwise
                    const FT mD = sph_mass ? ma * (x * x * x) : P::exp(mla + mb * logD, kc);
                    rn += nv * w;
                    rm += nv * mD * w;
                } else {
                    // ∂m/∂D · F_v(D) · N′(D) / D,  ∂m/∂D = a b D^(b−1)
                    const FT vD = P::exp(eA + E1, kc) * S;                                     // fall speed incl. aspect factor
                    const FT Fv = v.vent_a + v.vent_bc * Math<FT>::sqrt(Math<FT>::max(x * vD, FT(0)));
                    const FT dm_over_D = sph_mass ? ma * mb * x : ma * mb * P::exp((mb - FT(2)) * logD, kc);
                    rn += dm_over_D * Fv * P::exp(eN, kc) * w;
                }
            }
            sum_n += scale * rn; sum_m += scale * rm;
        }
        if constexpr (!MELT) {
            vn = sum_n / s.rho_n; vm = sum_m / s.rho_q;
        } else {
            const FT T = io.T[i];
            const FT L_f = v.LH_f0 + v.dcp_f * (T - v.T_0);
            vm = Math<FT>::max(FT(0), v.K4 / L_f * (T - v.T_freeze) * sum_n);                 // dL/dt
            vn = s.rho_n / s.rho_q * vm;                                                        // dN/dt
        }
    }
    if constexpr (!MELT) {
        if (io.v_n) io.v_n[i] = vn;
        if (io.v_m) io.v_m[i] = vm;
    } else {
        if (io.dNdt) io.dNdt[i] = vn;
        if (io.dLdt) io.dLdt[i] = vm;
    }
}

template <typename FT, typename PR, typename VR, typename QUAD>
static int32_t p3_velocity_entry(const PR *params, const VR *vel, const QUAD *quad, uint32_t flags, FT p, int64_t n, const FT *rho_q,
                                 const FT *rho_n, const FT *x3, const FT *x4, const FT *rho_a, const FT *loglam, FT *v_n, FT *v_m,
                                 void *stream) {
    if (!params || !vel || !quad || n < 0 || (flags & ~(CMX_P3_INPUT_IS_STATE | CMX_P3_SLOPE_CONSTANT | CMX_P3_NO_ASPECT_RATIO)))
        return CMX_ERR_BAD_ARG;
    if (quad->n < 1 || quad->n > CMX_QUAD_MAX || !(p > FT(0) && p < FT(0.5))) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4 || !rho_a || !loglam) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(*params, flags);
    c.brent_iters = 0;
    const P3VelConsts<FT> v = make_p3_vel_consts<FT>(*params, *vel, (double)p);
    P3VelIO<FT, QUAD> io{rho_q, rho_n, x3, x4, rho_a, loglam, v_n, v_m, nullptr, nullptr, nullptr};
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (flags & CMX_P3_NO_ASPECT_RATIO)
        hipLaunchKernelGGL((p3_velocity_kernel<FT, QUAD, false>), grid, block, 0, st, c, v, *quad, io, n);
    else
        hipLaunchKernelGGL((p3_velocity_kernel<FT, QUAD, true>), grid, block, 0, st, c, v, *quad, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

template <typename FT, typename PR, typename VR, typename AP, typename TH, typename VT, typename QUAD>
static int32_t p3_melt_entry(const PR *params, const VR *vel, const AP *aps, const TH *tps, const VT *vent, const QUAD *quad, uint32_t flags,
                             FT p, int64_t n, const FT *rho_q, const FT *rho_n, const FT *x3, const FT *x4, const FT *rho_a, const FT *T,
                             const FT *loglam, FT *dNdt, FT *dLdt, void *stream) {
    if (!params || !vel || !aps || !tps || !vent || !quad || n < 0 ||
        (flags & ~(CMX_P3_INPUT_IS_STATE | CMX_P3_SLOPE_CONSTANT | CMX_P3_NO_ASPECT_RATIO)))
        return CMX_ERR_BAD_ARG;
    if (quad->n < 1 || quad->n > CMX_QUAD_MAX || !(p > FT(0) && p < FT(0.5))) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4 || !rho_a || !T || !loglam) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(*params, flags);
    c.brent_iters = 0;
    P3VelConsts<FT> v = make_p3_vel_consts<FT>(*params, *vel, (double)p);
    v.vent_a = (FT)vent->a;
    v.vent_bc = (FT)((double)vent->b * std::cbrt((double)aps->nu_air / (double)aps->D_vapor) / std::sqrt((double)aps->nu_air));
    v.K4 = (FT)(4.0 * (double)aps->K_therm);
    v.LH_f0 = (FT)((double)tps->LH_s0 - (double)tps->LH_v0); v.dcp_f = (FT)((double)tps->cp_l - (double)tps->cp_i);
    v.T_0 = (FT)tps->T_0; v.T_freeze = (FT)params->T_freeze;
    P3VelIO<FT, QUAD> io{rho_q, rho_n, x3, x4, rho_a, loglam, nullptr, nullptr, T, dNdt, dLdt};
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (flags & CMX_P3_NO_ASPECT_RATIO)
        hipLaunchKernelGGL((p3_velocity_kernel<FT, QUAD, false, true>), grid, block, 0, st, c, v, *quad, io, n);
    else
        hipLaunchKernelGGL((p3_velocity_kernel<FT, QUAD, true, true>), grid, block, 0, st, c, v, *quad, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ice_self_collection — P3_processes.jl:676-712: dN/dt = ½ ∫∫ π (r₁+r₂)² |v₁ − v₂| n(D₁) n(D₂) dD₂ dD₁ (E = 1, r = √(area/π)).
// Outer integral over the four regime segments between the eps(FT) and 1 − eps(FT) quantiles, inner integral over
// (D_lo, D₁) and (D₁, D_hi) — split at the |v₁ − v₂| cusp, not at the regime thresholds, so the regime of every inner
// node is found by comparison.  8 n² integrand evaluations per point (n = quadrature order): the heaviest P3 process.
template <typename FT, typename QUAD> struct P3SelfIO { const FT *rho_q, *rho_n, *x3, *x4, *rho_a, *loglam; FT *dNdt; };

template <typename FT, typename QUAD, bool ASPECT>
__global__ __launch_bounds__(kBlock) void p3_self_collection_kernel(const P3Consts<FT> c, const P3VelConsts<FT> v, const QUAD quad,
                                                                   const P3SelfIO<FT, QUAD> io, const int64_t n) {
    using P = PM<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    P3Point<FT> s;
    p3_make_point<FT>(c, io.rho_q[i], io.rho_n[i], io.x3[i], io.x4[i], s);
    FT out = FT(0);
    if (!(s.rho_n < P::eps() || s.rho_q < P::eps())) {
        const FT loglam = io.loglam[i], lam = P::exp(loglam), mu = p3_mu<FT>(c, loglam);
        const FT logN0 = P::log(s.rho_n) - (-(mu + FT(1)) * loglam + P::lgamma(mu + FT(1)));
        const FT rho_a = Math<FT>::max(io.rho_a[i], FT(0)), lra = P::log(rho_a);
        const FT sb = v.s_B + rho_a * v.s_C, se = v.s_A * lra + sb * v.ln1000;
        const FT le1 = v.l_A * lra, le2 = le1 + v.l_H * rho_a;
        const FT D_lo = gamma_inc_inv_dev<FT>(mu + FT(1), P::eps(), FT(1) - P::eps()) / lam;      // p = eps(one(ρₐ))
        const FT D_hi = gamma_inc_inv_dev<FT>(mu + FT(1), FT(1) - P::eps(), FT(1) - (FT(1) - P::eps())) / lam;
        FT bnd[5];
        bnd[0] = D_lo; bnd[4] = D_hi;
#pragma unroll
        for (int k = 1; k < 4; ++k) bnd[k] = Math<FT>::min(Math<FT>::max(s.bnd[k], D_lo), D_hi);
        const FT Fu = Math<FT>::max(FT(1) - s.F_rim, P::eps());
        const FT h0 = (v.h0_num - P::log(Fu)) / FT(3), h1 = c.beta_va / FT(3);
        const bool unrimed = s.F_rim == FT(0);
        const FT inv_pi = FT(0.3183098861837907);
        const typename P::Coefs kc = P::coefs();
        // fall speed (incl. aspect factor), collision radius and number density at diameter x
        auto eval = [&](FT x, FT &vv, FT &rr, FT &nn) {
            const FT logD = P::log(x, kc);
            const int reg = x < s.bnd[1] ? 0 : (unrimed ? 1 : (x < s.bnd[2] ? 1 : (x < s.bnd[3] ? 2 : 3)));
            const FT sph = v.pi_4 * x * x;
            FT area = sph, eA = FT(0);
            if (reg == 1 || reg == 3) {
                const FT non = v.gamma_area * P::exp(v.sigma_area * logD, kc);
                area = reg == 1 ? non : s.F_rim * sph + (FT(1) - s.F_rim) * non;
                if (ASPECT) eA = reg == 1 ? v.g0 + v.g1 * logD : h0 + h1 * logD - FT(0.5) * P::log(area, kc);
            }
            const bool small = x <= v.cutoff;
            const FT E1 = small ? se + sb * logD : le1 + v.l_b1 * logD;
            const FT dE = small ? -v.s_c2 * x : (le2 - le1) + (v.l_b2 - v.l_b1) * logD - v.l_c2 * x;
            const FT A1 = small ? v.s_E : v.l_a1, A2 = small ? v.s_F : v.l_a2;
            vv = P::exp(eA + E1, kc) * (A1 + A2 * P::exp(dE, kc));
            rr = Math<FT>::sqrt(area * inv_pi);
            nn = P::exp(logN0 + mu * logD - lam * x, kc);
        };
        FT total = FT(0);
        for (int k = 0; k < 4; ++k) {
            const FT a = bnd[k], b = bnd[k + 1];
            if (!(a < b)) continue;
            const FT scale = (b - a) / FT(2), shift = (a + b) / FT(2);
            FT r_out = FT(0);
            for (int j1 = 0; j1 < quad.n; ++j1) {
                const FT D1 = scale * quad.node[j1] + shift;
                FT v1, r1, n1;
                eval(D1, v1, r1, n1);
                FT inner = FT(0);
                for (int h = 0; h < 2; ++h) {
                    const FT ia = h == 0 ? D_lo : D1, ib = h == 0 ? D1 : D_hi;
                    if (!(ia < ib)) continue;
                    const FT sc2 = (ib - ia) / FT(2), sh2 = (ia + ib) / FT(2);
                    FT r_in = FT(0);
                    for (int j2 = 0; j2 < quad.n; ++j2) {
                        FT v2, r2, n2;
                        eval(sc2 * quad.node[j2] + sh2, v2, r2, n2);
                        const FT rs = r1 + r2;
                        r_in += rs * rs * P::abs(v1 - v2) * n2 * quad.weight[j2];
                    }
                    inner += sc2 * r_in;
                }
                r_out += inner * n1 * quad.weight[j1];
            }
            total += scale * r_out;
        }
        out = FT(0.5) * FT(3.14159265358979323846) * total;       // the π of K = π (r₁+r₂)² taken out of the sums
    }
    io.dNdt[i] = out;
}

template <typename FT, typename PR, typename VR, typename QUAD>
static int32_t p3_self_collection_entry(const PR *params, const VR *vel, const QUAD *quad, uint32_t flags, int64_t n, const FT *rho_q,
                                        const FT *rho_n, const FT *x3, const FT *x4, const FT *rho_a, const FT *loglam, FT *dNdt,
                                        void *stream) {
    if (!params || !vel || !quad || n < 0 || (flags & ~(CMX_P3_INPUT_IS_STATE | CMX_P3_SLOPE_CONSTANT | CMX_P3_NO_ASPECT_RATIO)))
        return CMX_ERR_BAD_ARG;
    if (quad->n < 1 || quad->n > CMX_QUAD_MAX) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4 || !rho_a || !loglam || !dNdt) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(*params, flags);
    c.brent_iters = 0;
    const P3VelConsts<FT> v = make_p3_vel_consts<FT>(*params, *vel, 1e-6);
    P3SelfIO<FT, QUAD> io{rho_q, rho_n, x3, x4, rho_a, loglam, dNdt};
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (flags & CMX_P3_NO_ASPECT_RATIO)
        hipLaunchKernelGGL((p3_self_collection_kernel<FT, QUAD, false>), grid, block, 0, st, c, v, *quad, io, n);
    else
        hipLaunchKernelGGL((p3_self_collection_kernel<FT, QUAD, true>), grid, block, 0, st, c, v, *quad, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_p3_shape_f32(const cmx_p3_params_f32 *params, uint32_t flags, int32_t brent_iters, int64_t n, const float *rho_q_ice, const float *rho_n_ice,
                         const float *x3, const float *x4, const float *log_lambda_guess, float *F_rim, float *rho_rim,
                         float *log_lambda, float *D_m, float *log_N0, void *stream) {
    return cmx::p3_entry<float>(params, flags, brent_iters, n, rho_q_ice, rho_n_ice, x3, x4, log_lambda_guess, F_rim, rho_rim, log_lambda, D_m,
                                log_N0, stream);
}
int32_t cmx_p3_shape_f64(const cmx_p3_params_f64 *params, uint32_t flags, int32_t brent_iters, int64_t n, const double *rho_q_ice, const double *rho_n_ice,
                         const double *x3, const double *x4, const double *log_lambda_guess, double *F_rim, double *rho_rim,
                         double *log_lambda, double *D_m, double *log_N0, void *stream) {
    return cmx::p3_entry<double>(params, flags, brent_iters, n, rho_q_ice, rho_n_ice, x3, x4, log_lambda_guess, F_rim, rho_rim, log_lambda, D_m,
                                 log_N0, stream);
}

int32_t cmx_p3_terminal_velocities_f32(const cmx_p3_params_f32 *params, const cmx_chen2022_ice_vel_f32 *vel,
                                       const cmx_quadrature_f32 *quad, uint32_t flags, float p, int64_t n, const float *rho_q_ice,
                                       const float *rho_n_ice, const float *x3, const float *x4, const float *rho_air,
                                       const float *log_lambda, float *v_n, float *v_m, void *stream) {
    return cmx::p3_velocity_entry<float>(params, vel, quad, flags, p, n, rho_q_ice, rho_n_ice, x3, x4, rho_air, log_lambda, v_n, v_m, stream);
}
int32_t cmx_p3_terminal_velocities_f64(const cmx_p3_params_f64 *params, const cmx_chen2022_ice_vel_f64 *vel,
                                       const cmx_quadrature_f64 *quad, uint32_t flags, double p, int64_t n, const double *rho_q_ice,
                                       const double *rho_n_ice, const double *x3, const double *x4, const double *rho_air,
                                       const double *log_lambda, double *v_n, double *v_m, void *stream) {
    return cmx::p3_velocity_entry<double>(params, vel, quad, flags, p, n, rho_q_ice, rho_n_ice, x3, x4, rho_air, log_lambda, v_n, v_m, stream);
}

int32_t cmx_p3_ice_melt_f32(const cmx_p3_params_f32 *params, const cmx_chen2022_ice_vel_f32 *vel, const cmx_air_properties_f32 *aps,
                            const cmx_thermo_f32 *tps, const cmx_ventilation_f32 *vent, const cmx_quadrature_f32 *quad, uint32_t flags,
                            float p, int64_t n, const float *rho_q_ice, const float *rho_n_ice, const float *x3, const float *x4,
                            const float *rho_air, const float *T, const float *log_lambda, float *dNdt, float *dLdt, void *stream) {
    return cmx::p3_melt_entry<float>(params, vel, aps, tps, vent, quad, flags, p, n, rho_q_ice, rho_n_ice, x3, x4, rho_air, T, log_lambda,
                                     dNdt, dLdt, stream);
}
int32_t cmx_p3_ice_melt_f64(const cmx_p3_params_f64 *params, const cmx_chen2022_ice_vel_f64 *vel, const cmx_air_properties_f64 *aps,
                            const cmx_thermo_f64 *tps, const cmx_ventilation_f64 *vent, const cmx_quadrature_f64 *quad, uint32_t flags,
                            double p, int64_t n, const double *rho_q_ice, const double *rho_n_ice, const double *x3, const double *x4,
                            const double *rho_air, const double *T, const double *log_lambda, double *dNdt, double *dLdt, void *stream) {
    return cmx::p3_melt_entry<double>(params, vel, aps, tps, vent, quad, flags, p, n, rho_q_ice, rho_n_ice, x3, x4, rho_air, T, log_lambda,
                                      dNdt, dLdt, stream);
}

int32_t cmx_p3_ice_self_collection_f32(const cmx_p3_params_f32 *params, const cmx_chen2022_ice_vel_f32 *vel, const cmx_quadrature_f32 *quad,
                                       uint32_t flags, int64_t n, const float *rho_q_ice, const float *rho_n_ice, const float *x3,
                                       const float *x4, const float *rho_air, const float *log_lambda, float *dNdt, void *stream) {
    return cmx::p3_self_collection_entry<float>(params, vel, quad, flags, n, rho_q_ice, rho_n_ice, x3, x4, rho_air, log_lambda, dNdt, stream);
}
int32_t cmx_p3_ice_self_collection_f64(const cmx_p3_params_f64 *params, const cmx_chen2022_ice_vel_f64 *vel, const cmx_quadrature_f64 *quad,
                                       uint32_t flags, int64_t n, const double *rho_q_ice, const double *rho_n_ice, const double *x3,
                                       const double *x4, const double *rho_air, const double *log_lambda, double *dNdt, void *stream) {
    return cmx::p3_self_collection_entry<double>(params, vel, quad, flags, n, rho_q_ice, rho_n_ice, x3, x4, rho_air, log_lambda, dNdt, stream);
}

}  // extern "C"
