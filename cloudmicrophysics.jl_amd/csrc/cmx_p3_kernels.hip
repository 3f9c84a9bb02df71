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
// COMPUTE-bound (HISTORY.md §4.5): ≈12 residual evaluations per point × 8 incomplete-gamma evaluations × 20/30 fixed
// iterations ≈ 1e5 flops per point against 32–72 B of HBM traffic — the roofline is the FP64 (FP32) vector rate,
// not HBM.  What this kernel does about it:
//   * the four mass-regime coefficients (a_k, b_k), log a_k and the segment boundaries are per-point invariants
//     hoisted out of the solver; each residual evaluation needs lgamma for only TWO distinct z (b ∈ {3, β_va}) plus
//     μ+1 instead of the reference's five calls; e^{logλ} is formed once per evaluation;
//   * only the half (P or Q) of each incomplete-gamma pair that the segment difference needs is formed;
//   * the Brent iteration count is fixed (as in the reference: no data-dependent exit → no divergence from it).
// The solver is the same algorithm as the oracle's (cmx_p3.hpp Zeroin: Brent's zeroin under a fixed evaluation budget), so both land on the same
// root where the SlopePowerLaw makes the residual multi-rooted.
#include <hip/hip_runtime.h>

#include "cmx_p3.hpp"

namespace cmx {
CMX_P3_CONTRACT_BEGIN      // the whole translation unit is P3 quadrature / solver code (cmx_p3.hpp)

#ifndef CMX_P3_BS
#define CMX_P3_BS 256
#endif
#ifndef CMX_P3_F32_PACKED_NODES
#define CMX_P3_F32_PACKED_NODES 1      // Float32 fall-speed / melting node loops two nodes at a time in packed arithmetic (0: one node at a time, A/B switch)
#endif
constexpr int kP3BS = CMX_P3_BS;      // lanes per workgroup of the kernels of this file, one state per lane (A/B switch)


// get_distribution_logλ :284-320 for one state (shared by the shape kernel and the fused shape → fall-speed kernel).  `guess` may be
// nullptr (no warm start).  All residual evaluations (the two bracket ends, the optional warm-start guess of _narrow_bracket :336-353,
// the Brent iterations) go through ONE inlined call site: steps −3, −2, −1, 0 … — three separate inlined copies of the residual
// doubled the VGPR count (255, occupancy 1).
template <typename FT>
__device__ __forceinline__ FT p3_solve_loglam(const P3Consts<FT> &c, const P3Point<FT> &s, const FT *__restrict__ guess, const int64_t i) {
    using P = PM<FT>;
    FT loglam;
    if (s.rho_n < P::eps() || s.rho_q < P::eps()) {
        loglam = -INFINITY;
    } else {
        const FT target = P::log(s.rho_q) - P::log(s.rho_n);
        FT a = FT(2), b = FT(17), fa = FT(0), fb = FT(0);
        Zeroin<FT> z;
        bool active = true, guess_valid = false;
        const int first = guess ? -3 : -2;          // wave-uniform
        for (int it = first; it < c.brent_iters; ++it) {
            // which abscissa this step evaluates
            FT sx;
            const int step = it == first ? -3 : (it == first + 1 ? -2 : it);   // −3: lo end, −2: hi end, −1: guess
            if (step == -3) sx = a;
            else if (step == -2) sx = b;
            else if (step == -1) {
                const FT pg = guess[i];
                guess_valid = isfinite(pg) && (a < pg && pg < b);
                sx = guess_valid ? pg : a;
            } else {
                if (!active || !z.order()) { active = false; continue; }
                sx = z.propose();
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
                z.accept(fs);
            }
            // entering the Brent phase with the (possibly narrowed) bracket
            if (step < 0 && (guess ? step == -1 : step == -2)) {
                z.start(a, b, fa, fb);       // (without a sign change z.b = b is already the answer and `active` is off)
            }
        }
        if (active) z.order();               // after the last evaluation: b is the better end
        loglam = z.b;
    }
    return loglam;
}

template <typename FT>
__global__ __launch_bounds__(kP3BS) void p3_shape_kernel(const P3Consts<FT> c, const P3IO<FT> io, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using P = PM<FT>;
    const int64_t i = (int64_t)blockIdx.x * kP3BS + threadIdx.x;
    if (i >= n) return;
    P3Point<FT> s;
    p3_make_point<FT>(c, io.rho_q[i], io.rho_n[i], io.x3[i], io.x4[i], s);
    FT loglam = p3_solve_loglam<FT>(c, s, io.guess, i);
    // NaN in → NaN out (cmx_math.hpp any_nan): the gates and the regularised ratios above would map a NaN input to "no ice"
    if (any_nan(io.rho_q[i], io.rho_n[i], io.x3[i], io.x4[i])) loglam = s.F_rim = s.rho_rim = Math<FT>::nan();
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
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(*params, flags);
    c.brent_iters = brent_iters > 0 ? brent_iters : PM<FT>::kBrent;
    P3IO<FT> io{rho_q, rho_n, x3, x4, guess, F_rim, rho_rim, loglam, D_m, logN0};
    hipLaunchKernelGGL((p3_shape_kernel<FT>), dim3((unsigned)((n + kP3BS - 1) / kP3BS)), dim3(kP3BS), 0,
                       reinterpret_cast<hipStream_t>(stream), c, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}


template <typename FT, typename QUAD> struct P3VelIO {
    const FT *rho_q, *rho_n, *x3, *x4, *rho_a, *loglam; FT *v_n, *v_m;
    const FT *T; FT *dNdt, *dLdt;     // MELT mode
    const FT *guess; FT *loglam_out, *D_m_out;   // SOLVE mode (fused shape → fall speeds): optional warm start, optional logλ / D_m outputs
};

// SOLVE: the shape solve (get_distribution_logλ) runs in this launch instead of reading a log λ column — one launch for the whole
// BASELINE config-5 pass: no second read of the state, no log λ round trip through HBM (cmx_p3_shape_terminal_velocities_*).
template <typename FT, typename QUAD, bool ASPECT, bool MELT = false, bool SOLVE = false>
__global__ __launch_bounds__(kP3BS) void p3_velocity_kernel(const P3Consts<FT> c, const P3VelConsts<FT> v, const QUAD quad,
                                                            const P3VelIO<FT, QUAD> io, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using P = PM<FT>;
    const int64_t i = (int64_t)blockIdx.x * kP3BS + threadIdx.x;
    if (i >= n) return;
    P3Point<FT> s;
    p3_make_point<FT>(c, io.rho_q[i], io.rho_n[i], io.x3[i], io.x4[i], s);
    FT vn = FT(0), vm = FT(0);
    FT loglam_in = FT(0);
    if constexpr (SOLVE) {
        loglam_in = p3_solve_loglam<FT>(c, s, io.guess, i);
        // NaN in → NaN out, as in p3_shape_kernel
        if (any_nan(io.rho_q[i], io.rho_n[i], io.x3[i], io.x4[i])) loglam_in = s.F_rim = s.rho_rim = Math<FT>::nan();
        if (io.loglam_out) io.loglam_out[i] = loglam_in;
        if (io.D_m_out) {
            const FT mu = p3_mu<FT>(c, loglam_in);
            const FT logN0 = P::log(s.rho_n) - (-(mu + FT(1)) * loglam_in + P::lgamma(mu + FT(1)));   // get_logN₀ :233-237
            io.D_m_out[i] = P::exp(logN0 + p3_logmass_moment<FT>(c, s, mu, loglam_in, FT(1))) / s.rho_q;   // D_m :56-61
        }
    }
    if (!(s.rho_n < P::eps() || s.rho_q < P::eps())) {
        const FT loglam = SOLVE ? loglam_in : io.loglam[i], lam = P::exp(loglam), mu = p3_mu<FT>(c, loglam);
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
            // one node — or, Float32, a PAIR of nodes (VT = f32x2: packed arithmetic, cmx_p3.hpp PM<f32x2>) — of the segment's integrand
            auto node = [&](auto x, auto w, auto &an, auto &am) {
                using VT = decltype(x);
                using PV = PM<VT>;
                using MVT = Math<VT>;
                const VT logD = PV::log_pos(x, kc);          // an interior quadrature node: positive, normal, finite
                const VT eN = logN0 + mu * logD - lam_ * x;                 // log n(D)
                const VT eA = q0 + q1 * logD;                                // log of the aspect factor — but for the area^(−½) of the partially rimed segment:
                VT mA = VT(1);                                               // a reciprocal square root (≈ 9 Float64 instructions) where eA −= ½ ln(area) cost a logarithm (≈ 20)
                if (mixed_area) {
                    const VT area = s.F_rim * k_pi4 * x * x + (FT(1) - s.F_rim) * k_ga * PV::exp(k_sa * logD, kc);
                    mA = MVT::rsqrt_pos(area);
                }
                // Chen-2022 particle speed Σ aₖ D^bₖ e^{−cₖD}: the two terms have opposite signs and cancel to ≈1/200 of
                // their size for small D, so the shared factor stays OUTSIDE the difference (as D^b does in the reference)
                const typename MVT::Mask small = x <= VT(k_cut);
                const VT E1 = small ? se + sb * logD : le1 + k_lb1 * logD;
                const VT dE = small ? -k_sc2 * x : (le2 - le1) + k_db * logD - k_lc2 * x;   // E2 − E1
                const VT A1 = small ? VT(k_sE) : VT(k_la1), A2 = small ? VT(k_sF) : VT(k_la2);
                const VT S = mixed_area ? mA * (A1 + A2 * PV::exp(dE, kc)) : A1 + A2 * PV::exp(dE, kc);
                if constexpr (!MELT) {
                    const VT nv = PV::exp(eN + eA + E1, kc) * S;
                    // m(D) = a D^b: b = 3 on the spherical segments (no transcendental), β_va otherwise
                    const VT mD = sph_mass ? ma * (x * x * x) : PV::exp(mla + mb * logD, kc);
                    an += nv * w;
                    am += nv * mD * w;
                } else {
                    // ∂m/∂D · F_v(D) · N′(D) / D,  ∂m/∂D = a b D^(b−1)
                    const VT vD = PV::exp(eA + E1, kc) * S;                                     // fall speed incl. aspect factor
                    const VT Fv = v.vent_a + v.vent_bc * MVT::sqrt(MVT::max(x * vD, VT(0)));
                    const VT dm_over_D = sph_mass ? ma * mb * x : ma * mb * PV::exp((mb - FT(2)) * logD, kc);
                    an += dm_over_D * Fv * PV::exp(eN, kc) * w;
                }
            };
            int j = 0;
#if CMX_HAVE_PACKED
            if constexpr (sizeof(FT) == 4 && CMX_P3_F32_PACKED_NODES) {
                f32x2 rn2 = f32x2(0.0f), rm2 = f32x2(0.0f);
                for (; j + 1 < quad.n; j += 2) {
                    const f32x2 xn = f32x2{quad.node[j], quad.node[j + 1]}, wn = f32x2{quad.weight[j], quad.weight[j + 1]};
                    node(scale * xn + shift, wn, rn2, rm2);
                }
                rn = rn2.x + rn2.y; rm = rm2.x + rm2.y;
            }
#endif
            for (; j < quad.n; ++j) node(FT(scale * quad.node[j] + shift), FT(quad.weight[j]), rn, rm);
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
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (quad->n < 1 || quad->n > CMX_QUAD_MAX || !(p > FT(0) && p < FT(0.5))) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4 || !rho_a || !loglam) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(*params, flags);
    c.brent_iters = 0;
    const P3VelConsts<FT> v = make_p3_vel_consts<FT>(*params, *vel, (double)p);
    P3VelIO<FT, QUAD> io{rho_q, rho_n, x3, x4, rho_a, loglam, v_n, v_m, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const dim3 grid((unsigned)((n + kP3BS - 1) / kP3BS)), block(kP3BS);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (flags & CMX_P3_NO_ASPECT_RATIO)
        hipLaunchKernelGGL((p3_velocity_kernel<FT, QUAD, false>), grid, block, 0, st, c, v, *quad, io, n);
    else
        hipLaunchKernelGGL((p3_velocity_kernel<FT, QUAD, true>), grid, block, 0, st, c, v, *quad, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// cmx_p3_shape_terminal_velocities_*: shape solve + D_m + number- / mass-weighted fall speeds in ONE launch (BASELINE config 5)
template <typename FT, typename PR, typename VR, typename QUAD>
static int32_t p3_shape_velocity_entry(const PR *params, const VR *vel, const QUAD *quad, uint32_t flags, int32_t brent_iters, FT p, int64_t n,
                                       const FT *rho_q, const FT *rho_n, const FT *x3, const FT *x4, const FT *rho_a, const FT *guess,
                                       FT *loglam, FT *D_m, FT *v_n, FT *v_m, void *stream) {
    if (!params || !vel || !quad || n < 0 || (flags & ~(CMX_P3_INPUT_IS_STATE | CMX_P3_SLOPE_CONSTANT | CMX_P3_NO_ASPECT_RATIO)))
        return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (quad->n < 1 || quad->n > CMX_QUAD_MAX || !(p > FT(0) && p < FT(0.5))) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4 || !rho_a) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(*params, flags);
    c.brent_iters = brent_iters > 0 ? brent_iters : PM<FT>::kBrent;
    const P3VelConsts<FT> v = make_p3_vel_consts<FT>(*params, *vel, (double)p);
    P3VelIO<FT, QUAD> io{rho_q, rho_n, x3, x4, rho_a, nullptr, v_n, v_m, nullptr, nullptr, nullptr, guess, loglam, D_m};
    const dim3 grid((unsigned)((n + kP3BS - 1) / kP3BS)), block(kP3BS);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (flags & CMX_P3_NO_ASPECT_RATIO)
        hipLaunchKernelGGL((p3_velocity_kernel<FT, QUAD, false, false, true>), grid, block, 0, st, c, v, *quad, io, n);
    else
        hipLaunchKernelGGL((p3_velocity_kernel<FT, QUAD, true, false, true>), grid, block, 0, st, c, v, *quad, io, n);
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
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
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
    P3VelIO<FT, QUAD> io{rho_q, rho_n, x3, x4, rho_a, loglam, nullptr, nullptr, T, dNdt, dLdt, nullptr, nullptr, nullptr};
    const dim3 grid((unsigned)((n + kP3BS - 1) / kP3BS)), block(kP3BS);
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
__global__ __launch_bounds__(kP3BS) void p3_self_collection_kernel(const P3Consts<FT> c, const P3VelConsts<FT> v, const QUAD quad,
                                                                   const P3SelfIO<FT, QUAD> io, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using P = PM<FT>;
    const int64_t i = (int64_t)blockIdx.x * kP3BS + threadIdx.x;
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
        const FT h0 = (v.h0_num - P::log(Fu)) / FT(3) - FT(0.5723649429247001), h1 = c.beta_va / FT(3);      // incl. the −½ ln π of area^(−½) = y/√π (eval below)
        const bool unrimed = s.F_rim == FT(0);
        const FT inv_pi = FT(0.3183098861837907);
        const typename P::Coefs kc = P::coefs();
        // fall speed (incl. aspect factor), collision radius and number density at diameter x
        auto eval = [&](FT x, FT &vv, FT &rr, FT &nn) {
            const FT logD = P::log_pos(x, kc);          // an interior quadrature node: positive, normal, finite
            const int reg = x < s.bnd[1] ? 0 : (unrimed ? 1 : (x < s.bnd[2] ? 1 : (x < s.bnd[3] ? 2 : 3)));
            // collision radius r = √(area/π) and aspect factor: spherical regimes r = D/2 exactly; unrimed non-spherical area = γ D^σ:
            // r = √(γ/π)·D^(σ/2) (one exponential, no square root); only the partially rimed regime needs the mixed area and its root.  D^(σ/2) serves
            // both non-spherical laws (γ D^σ is its square): lanes of one wave sit in both regimes, so the wave evaluates one exponential, not two
            FT eA = FT(0), mA = FT(1);      // aspect factor = exp(eA)·mA: the partially rimed regime's area^(−½) is the reciprocal root of its √(area/π) (cmx_p3_collisions.hip eval_ice)
            rr = FT(0.5) * x;
            if (reg == 1 || reg == 3) {
                const FT dh = P::exp(v.half_sigma * logD, kc);
                if (reg == 1) {
                    rr = v.sqrt_gamma_pi * dh;
                    if (ASPECT) eA = v.g0 + v.g1 * logD;
                } else {
                    const FT area = s.F_rim * (v.pi_4 * x * x) + (FT(1) - s.F_rim) * (v.gamma_area * (dh * dh));
                    const FT ap = area * inv_pi;
                    if (ASPECT) { const FT y = Math<FT>::rsqrt_pos(ap); rr = ap * y; mA = y; eA = h0 + h1 * logD; }
                    else rr = Math<FT>::sqrt(ap);
                }
            }
            const bool small = x <= v.cutoff;
            const FT E1 = small ? se + sb * logD : le1 + v.l_b1 * logD;
            const FT dE = small ? -v.s_c2 * x : (le2 - le1) + (v.l_b2 - v.l_b1) * logD - v.l_c2 * x;
            const FT A1 = small ? kpin(v.s_E) : kpin(v.l_a1), A2 = small ? kpin(v.s_F) : kpin(v.l_a2);
            const FT S = A1 + A2 * P::exp(dE, kc);
            vv = ASPECT ? P::exp(eA + E1, kc) * (mA * S) : P::exp(E1, kc) * S;
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
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (quad->n < 1 || quad->n > CMX_QUAD_MAX) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4 || !rho_a || !loglam || !dNdt) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(*params, flags);
    c.brent_iters = 0;
    const P3VelConsts<FT> v = make_p3_vel_consts<FT>(*params, *vel, 1e-6);
    P3SelfIO<FT, QUAD> io{rho_q, rho_n, x3, x4, rho_a, loglam, dNdt};
    const dim3 grid((unsigned)((n + kP3BS - 1) / kP3BS)), block(kP3BS);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (flags & CMX_P3_NO_ASPECT_RATIO)
        hipLaunchKernelGGL((p3_self_collection_kernel<FT, QUAD, false>), grid, block, 0, st, c, v, *quad, io, n);
    else
        hipLaunchKernelGGL((p3_self_collection_kernel<FT, QUAD, true>), grid, block, 0, st, c, v, *quad, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// ---- UT.gamma_inc / UT.gamma_inc_inv over columns (src/Utilities.jl:54-61,93-144,205-252; KA wrapper test_gamma_inc_kernel!,
// test/gpu_tests.jl:456-461) — the very functions the shape solver, the quantile bounds and the collision kernels call
template <typename FT>
__global__ __launch_bounds__(kP3BS) void gamma_inc_kernel(const int64_t n, const FT *__restrict__ a, const FT *__restrict__ x, FT *__restrict__ Pout,
                                                           FT *__restrict__ Qout) {
    Math<FT>::prepare();
    const int64_t i = (int64_t)blockIdx.x * kP3BS + threadIdx.x;
    if (i >= n) return;
    const FT ai = a[i], xi = x[i];
    const FT lg = PM<FT>::lgamma(ai);
    // one evaluation, as the reference returns (P, 1 − P) on the series branch and (1 − Q, Q) on the continued-fraction branch
    const bool series = xi < ai + FT(1);
    const FT g = gamma_inc_dev<FT>(ai, xi, lg, series);
    if (Pout) Pout[i] = series ? g : FT(1) - g;
    if (Qout) Qout[i] = series ? FT(1) - g : g;
}
template <typename FT>
__global__ __launch_bounds__(kP3BS) void gamma_inc_inv_kernel(const int64_t n, const FT *__restrict__ a, const FT *__restrict__ p, const FT *__restrict__ q,
                                                               FT *__restrict__ xout) {
    Math<FT>::prepare();
    const int64_t i = (int64_t)blockIdx.x * kP3BS + threadIdx.x;
    if (i >= n) return;
    xout[i] = gamma_inc_inv_dev<FT>(a[i], p[i], q[i]);
}

template <typename FT> static int32_t gamma_inc_entry(int64_t n, const FT *a, const FT *x, FT *P, FT *Q, void *stream) {
    if (n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!a || !x || (!P && !Q)) return CMX_ERR_BAD_ARG;
    hipLaunchKernelGGL((gamma_inc_kernel<FT>), dim3((unsigned)((n + kP3BS - 1) / kP3BS)), dim3(kP3BS), 0, reinterpret_cast<hipStream_t>(stream), n, a, x, P, Q);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}
template <typename FT> static int32_t gamma_inc_inv_entry(int64_t n, const FT *a, const FT *p, const FT *q, FT *x, void *stream) {
    if (n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!a || !p || !q || !x) return CMX_ERR_BAD_ARG;
    hipLaunchKernelGGL((gamma_inc_inv_kernel<FT>), dim3((unsigned)((n + kP3BS - 1) / kP3BS)), dim3(kP3BS), 0, reinterpret_cast<hipStream_t>(stream), n, a, p, q,
                       x);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// ---- size-distribution helpers over columns (src/DistributionTools.jl:44-151, src/Microphysics2M.jl:270-354) ----------------------------
// Helper entries (diagnostics, integration bounds a host model asks for): plain one-point-per-lane kernels on the OCML functions; the
// incomplete gamma functions are the device routines of cmx_p3.hpp (the reference's DT functions call UT.gamma_inc / UT.gamma_inc_inv).
template <typename FT> struct DistMath;
template <> struct DistMath<float> {
    static __device__ __forceinline__ float log(float x) { return ::logf(x); }
    static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
    static __device__ __forceinline__ float pow(float x, float y) { return ::powf(x, y); }
    static __device__ __forceinline__ float cbrt(float x) { return ::cbrtf(x); }
    static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
    static __device__ __forceinline__ float log1p(float x) { return ::log1pf(x); }
    static __device__ __forceinline__ float expm1(float x) { return ::expm1f(x); }
};
template <> struct DistMath<double> {
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
    static __device__ __forceinline__ double cbrt(double x) { return ::cbrt(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double log1p(double x) { return ::log1p(x); }
    static __device__ __forceinline__ double expm1(double x) { return ::expm1(x); }
};
// DT.generalized_gamma_quantile(ν, μ, B, Y) — :44-47
template <typename FT> __device__ __forceinline__ FT gg_quantile_dev(FT nu, FT mu, FT B, FT Y) {
    return DistMath<FT>::pow(gamma_inc_inv_dev<FT>((nu + FT(1)) / mu, Y, FT(1) - Y) / B, FT(1) / mu);
}
// DT.generalized_gamma_cdf(ν, μ, B, x) — :75-82 (its DomainErrors, μ ≤ 0 or B ≤ 0, give NaN)
template <typename FT> __device__ __forceinline__ FT gg_cdf_dev(FT nu, FT mu, FT B, FT x) {
    if (!(mu > FT(0)) || !(B > FT(0))) return Math<FT>::nan();
    if (x <= FT(0)) return FT(0);
    const FT a = (nu + FT(1)) / mu;
    return gamma_inc_dev<FT>(a, B * DistMath<FT>::pow(x, mu), PM<FT>::lgamma(a), true);
}
// DT.exponential_cdf(D_mean, D) = exp(log1mexp(−D/D_mean)) — :124-129; DT.exponential_quantile(D_mean, Y) = exp(log D_mean + cloglog Y) — :146-151
template <typename FT> __device__ __forceinline__ FT exp_cdf_dev(FT D_mean, FT D) {
    using DM = DistMath<FT>;
    if (!(D_mean > FT(0))) return Math<FT>::nan();
    if (D < FT(0)) return FT(0);
    const FT x = -D / D_mean;
    return DM::exp(x > FT(-0.6931471805599453) ? DM::log(-DM::expm1(x)) : DM::log1p(-DM::exp(x)));      // LogExpFunctions.log1mexp
}
template <typename FT> __device__ __forceinline__ FT exp_quantile_dev(FT D_mean, FT Y) {
    using DM = DistMath<FT>;
    if (!(Y >= FT(0) && Y <= FT(1)) || !(D_mean > FT(0))) return Math<FT>::nan();
    return DM::exp(DM::log(D_mean) + DM::log(-DM::log1p(-Y)));                                              // LogExpFunctions.cloglog
}
template <typename FT>
__global__ __launch_bounds__(kP3BS) void generalized_gamma_kernel(const FT nu, const FT mu, const int64_t n, const FT *__restrict__ B,
                                                                  const FT *__restrict__ Y, const FT *__restrict__ x, FT *__restrict__ quantile,
                                                                  FT *__restrict__ cdf) {
    Math<FT>::prepare();
    const int64_t i = (int64_t)blockIdx.x * kP3BS + threadIdx.x;
    if (i >= n) return;
    if (quantile) quantile[i] = gg_quantile_dev<FT>(nu, mu, B[i], Y[i]);
    if (cdf) cdf[i] = gg_cdf_dev<FT>(nu, mu, B[i], x[i]);
}
template <typename FT>
__global__ __launch_bounds__(kP3BS) void exponential_distribution_kernel(const int64_t n, const FT *__restrict__ D_mean, const FT *__restrict__ Y,
                                                                         const FT *__restrict__ D, FT *__restrict__ quantile, FT *__restrict__ cdf) {
    const int64_t i = (int64_t)blockIdx.x * kP3BS + threadIdx.x;
    if (i >= n) return;
    if (quantile) quantile[i] = exp_quantile_dev<FT>(D_mean[i], Y[i]);
    if (cdf) cdf[i] = exp_cdf_dev<FT>(D_mean[i], D[i]);
}
// pdf parameters of the two SB2006 size distributions, restated as the reference writes them (CM2:67-110, 176-191, 227-236): these are
// diagnostics — the rate kernels carry their own log2-domain forms
template <typename FT> struct PsdPar {
    FT nu_c, mu_c, rho_w_c, lg_z1, lg_z2;                                   // CloudParticlePDF_SB2006
    FT xr_min, xr_max, N0_min, N0_max, lam_min, lam_max, rho_w_r;            // RainParticlePDF_SB2006 (limited fields used iff LIMITED)
    FT p;
};
template <typename FT, bool CLOUD, bool LIMITED>
__global__ __launch_bounds__(kP3BS) void sb2006_size_distribution_kernel(const PsdPar<FT> k, const int64_t n, const FT *__restrict__ q,
                                                                         const FT *__restrict__ rho, const FT *__restrict__ N, const FT *__restrict__ D,
                                                                         FT *__restrict__ n_D, FT *__restrict__ D_min, FT *__restrict__ D_max) {
    Math<FT>::prepare();
    using DM = DistMath<FT>;
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kP3BS + threadIdx.x;
    if (i >= n) return;
    const FT eps = M::eps(), pi = FT(3.14159265358979323846);
    const FT qi = q[i], ri = rho[i], Ni = N[i];
    const FT sq = M::max(qi, eps), sN = M::max(Ni, eps), L = ri * sq;
    if constexpr (CLOUD) {
        // log_pdf_cloud_parameters_mass CM2:176-191, pdf_cloud_parameters :227-236
        const FT z1 = (k.nu_c + FT(1)) / k.mu_c;
        FT logB = -k.mu_c * (DM::log(L / sN) + k.lg_z1 - k.lg_z2);
        FT logA = DM::log(k.mu_c) + DM::log(sN) + z1 * logB - k.lg_z1;
        if (Ni < eps || qi < eps) { logA = -INFINITY; logB = INFINITY; }
        const FT k_m = k.rho_w_c * pi / FT(6);
        const FT logN0c = logA + DM::log(FT(3)) + (k.nu_c + FT(1)) * DM::log(k_m);
        const FT lam_c = DM::exp(logB) * DM::pow(k_m, k.mu_c);
        const FT nu_D = FT(3) * k.nu_c + FT(2), mu_D = FT(3) * k.mu_c;
        if (n_D) n_D[i] = logN0c == -INFINITY ? FT(0) : DM::exp(logN0c + nu_D * DM::log(D[i]) - lam_c * DM::pow(D[i], mu_D));     // CM2:295-303
        if (D_min) D_min[i] = gg_quantile_dev<FT>(nu_D, mu_D, lam_c, k.p);                                                        // CM2:346-354
        if (D_max) D_max[i] = gg_quantile_dev<FT>(nu_D, mu_D, lam_c, FT(1) - k.p);
    } else {
        // pdf_rain_parameters CM2:67-110
        FT N0r, Dr_mean;
        bool cond;
        if constexpr (!LIMITED) {
            const FT lam = DM::cbrt(pi * k.rho_w_r / (L / sN));
            N0r = lam * sN; Dr_mean = FT(1) / lam;
            cond = Ni < eps || qi < eps;
        } else {
            const FT xt = clampv(L / sN, k.xr_min, k.xr_max);                                            // Eq. 94
            N0r = clampv(sN * DM::cbrt(pi * k.rho_w_r / xt), k.N0_min, k.N0_max);                        // Eq. 95
            const FT lam = clampv(DM::sqrt(DM::sqrt(pi * k.rho_w_r * N0r / L)), k.lam_min, k.lam_max);   // Eq. 96
            Dr_mean = FT(1) / lam;
            cond = Ni < eps && qi < eps;
        }
        if (cond) { N0r = FT(0); Dr_mean = FT(0); }
        // CM2:270-277.  Float32 forms N₀r·exp(−D/D̄) as exp(log N₀r − D/D̄): the hardware exponential flushes results below 2⁻¹²⁶ to zero, and N₀r (up
        // to 1e13) times such a value is still a normal Float32 number in the reference's arithmetic; Float64 keeps the reference's product (its
        // exponential delivers subnormal results, and the product underflows exactly where the reference's does)
        if (n_D) {
            if constexpr (sizeof(FT) == 4) n_D[i] = N0r == FT(0) ? FT(0) : DM::exp(DM::log(N0r) - D[i] / Dr_mean);
            else n_D[i] = N0r == FT(0) ? FT(0) : N0r * DM::exp(-D[i] / Dr_mean);
        }
        const bool none = Dr_mean == FT(0);                                                              // CM2:336-345
        if (D_min) D_min[i] = none ? FT(0) : exp_quantile_dev<FT>(Dr_mean, k.p);
        if (D_max) D_max[i] = none ? FT(0) : exp_quantile_dev<FT>(Dr_mean, FT(1) - k.p);
    }
}

template <typename FT> static int32_t generalized_gamma_entry(FT nu, FT mu, int64_t n, const FT *B, const FT *Y, const FT *x, FT *quantile, FT *cdf, void *stream) {
    if (n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!B || (!quantile && !cdf) || (quantile && !Y) || (cdf && !x)) return CMX_ERR_BAD_ARG;
    hipLaunchKernelGGL((generalized_gamma_kernel<FT>), dim3((unsigned)((n + kP3BS - 1) / kP3BS)), dim3(kP3BS), 0, reinterpret_cast<hipStream_t>(stream), nu, mu, n,
                       B, Y, x, quantile, cdf);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}
template <typename FT> static int32_t exponential_distribution_entry(int64_t n, const FT *D_mean, const FT *Y, const FT *D, FT *quantile, FT *cdf, void *stream) {
    if (n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!D_mean || (!quantile && !cdf) || (quantile && !Y) || (cdf && !D)) return CMX_ERR_BAD_ARG;
    hipLaunchKernelGGL((exponential_distribution_kernel<FT>), dim3((unsigned)((n + kP3BS - 1) / kP3BS)), dim3(kP3BS), 0, reinterpret_cast<hipStream_t>(stream), n,
                       D_mean, Y, D, quantile, cdf);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}
template <typename FT, typename PC, typename PR>
static int32_t sb2006_size_distribution_entry(const PC *pdf_c, const PR *pdf_r, uint32_t flags, FT p, int64_t n, const FT *q, const FT *rho, const FT *N,
                                              const FT *D, FT *n_D, FT *D_min, FT *D_max, void *stream) {
    const bool cloud = flags & CMX_PSD_CLOUD, limited = flags & CMX_SB2006_LIMITED;
    if (n < 0 || (flags & ~(uint32_t)(CMX_PSD_CLOUD | CMX_SB2006_LIMITED)) || (cloud ? !pdf_c : !pdf_r)) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!q || !rho || !N || (!n_D && !D_min && !D_max) || (n_D && !D)) return CMX_ERR_BAD_ARG;
    // the limited rain PSD clamps with its limiter pairs: a struct of the NOT-limited variant (limiters all zero, cmx.h) passed with the flag set
    // would clamp N0 and λ to 0 and return Inf / NaN silently — the same rule as the rate and column entries (sb_limiters_ok)
    if (!cloud && limited &&
        !(pdf_r->xr_min > 0 && pdf_r->xr_min <= pdf_r->xr_max && pdf_r->N0_min > 0 && pdf_r->N0_min <= pdf_r->N0_max && pdf_r->lambda_min > 0 &&
          pdf_r->lambda_min <= pdf_r->lambda_max))
        return CMX_ERR_BAD_ARG;
    PsdPar<FT> k{};
    if (cloud) { k.nu_c = pdf_c->nu_c; k.mu_c = pdf_c->mu_c; k.rho_w_c = pdf_c->rho_w; k.lg_z1 = pdf_c->loggamma_z1; k.lg_z2 = pdf_c->loggamma_z2; }
    else {
        k.xr_min = pdf_r->xr_min; k.xr_max = pdf_r->xr_max; k.N0_min = pdf_r->N0_min; k.N0_max = pdf_r->N0_max; k.lam_min = pdf_r->lambda_min;
        k.lam_max = pdf_r->lambda_max; k.rho_w_r = pdf_r->rho_w;
    }
    k.p = p;
    const dim3 grid((unsigned)((n + kP3BS - 1) / kP3BS)), block(kP3BS);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (cloud) hipLaunchKernelGGL((sb2006_size_distribution_kernel<FT, true, false>), grid, block, 0, st, k, n, q, rho, N, D, n_D, D_min, D_max);
    else if (limited) hipLaunchKernelGGL((sb2006_size_distribution_kernel<FT, false, true>), grid, block, 0, st, k, n, q, rho, N, D, n_D, D_min, D_max);
    else hipLaunchKernelGGL((sb2006_size_distribution_kernel<FT, false, false>), grid, block, 0, st, k, n, q, rho, N, D, n_D, D_min, D_max);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

CMX_P3_CONTRACT_END
}  // namespace cmx

extern "C" {

int32_t cmx_generalized_gamma_f32(float nu, float mu, int64_t n, const float *B, const float *Y, const float *x, float *quantile, float *cdf, void *stream) {
    return cmx::generalized_gamma_entry<float>(nu, mu, n, B, Y, x, quantile, cdf, stream);
}
int32_t cmx_generalized_gamma_f64(double nu, double mu, int64_t n, const double *B, const double *Y, const double *x, double *quantile, double *cdf, void *stream) {
    return cmx::generalized_gamma_entry<double>(nu, mu, n, B, Y, x, quantile, cdf, stream);
}
int32_t cmx_exponential_distribution_f32(int64_t n, const float *D_mean, const float *Y, const float *D, float *quantile, float *cdf, void *stream) {
    return cmx::exponential_distribution_entry<float>(n, D_mean, Y, D, quantile, cdf, stream);
}
int32_t cmx_exponential_distribution_f64(int64_t n, const double *D_mean, const double *Y, const double *D, double *quantile, double *cdf, void *stream) {
    return cmx::exponential_distribution_entry<double>(n, D_mean, Y, D, quantile, cdf, stream);
}
int32_t cmx_sb2006_size_distribution_f32(const cmx_cloud_pdf_sb2006_f32 *pdf_c, const cmx_rain_pdf_sb2006_f32 *pdf_r, uint32_t flags, float p, int64_t n,
                                         const float *q, const float *rho, const float *N, const float *D, float *n_D, float *D_min, float *D_max,
                                         void *stream) {
    return cmx::sb2006_size_distribution_entry<float>(pdf_c, pdf_r, flags, p, n, q, rho, N, D, n_D, D_min, D_max, stream);
}
int32_t cmx_sb2006_size_distribution_f64(const cmx_cloud_pdf_sb2006_f64 *pdf_c, const cmx_rain_pdf_sb2006_f64 *pdf_r, uint32_t flags, double p, int64_t n,
                                         const double *q, const double *rho, const double *N, const double *D, double *n_D, double *D_min, double *D_max,
                                         void *stream) {
    return cmx::sb2006_size_distribution_entry<double>(pdf_c, pdf_r, flags, p, n, q, rho, N, D, n_D, D_min, D_max, stream);
}


int32_t cmx_gamma_inc_f32(int64_t n, const float *a, const float *x, float *P, float *Q, void *stream) { return cmx::gamma_inc_entry<float>(n, a, x, P, Q, stream); }
int32_t cmx_gamma_inc_f64(int64_t n, const double *a, const double *x, double *P, double *Q, void *stream) { return cmx::gamma_inc_entry<double>(n, a, x, P, Q, stream); }
int32_t cmx_gamma_inc_inv_f32(int64_t n, const float *a, const float *p, const float *q, float *x, void *stream) {
    return cmx::gamma_inc_inv_entry<float>(n, a, p, q, x, stream);
}
int32_t cmx_gamma_inc_inv_f64(int64_t n, const double *a, const double *p, const double *q, double *x, void *stream) {
    return cmx::gamma_inc_inv_entry<double>(n, a, p, q, x, stream);
}


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

int32_t cmx_p3_shape_terminal_velocities_f32(const cmx_p3_params_f32 *params, const cmx_chen2022_ice_vel_f32 *vel, const cmx_quadrature_f32 *quad,
                                             uint32_t flags, int32_t brent_iters, float p, int64_t n, const float *rho_q_ice, const float *rho_n_ice,
                                             const float *x3, const float *x4, const float *rho_air, const float *log_lambda_guess,
                                             float *log_lambda, float *D_m, float *v_n, float *v_m, void *stream) {
    return cmx::p3_shape_velocity_entry<float>(params, vel, quad, flags, brent_iters, p, n, rho_q_ice, rho_n_ice, x3, x4, rho_air, log_lambda_guess,
                                               log_lambda, D_m, v_n, v_m, stream);
}
int32_t cmx_p3_shape_terminal_velocities_f64(const cmx_p3_params_f64 *params, const cmx_chen2022_ice_vel_f64 *vel, const cmx_quadrature_f64 *quad,
                                             uint32_t flags, int32_t brent_iters, double p, int64_t n, const double *rho_q_ice,
                                             const double *rho_n_ice, const double *x3, const double *x4, const double *rho_air,
                                             const double *log_lambda_guess, double *log_lambda, double *D_m, double *v_n, double *v_m,
                                             void *stream) {
    return cmx::p3_shape_velocity_entry<double>(params, vel, quad, flags, brent_iters, p, n, rho_q_ice, rho_n_ice, x3, x4, rho_air,
                                                log_lambda_guess, log_lambda, D_m, v_n, v_m, stream);
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
