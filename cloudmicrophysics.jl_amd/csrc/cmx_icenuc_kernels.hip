// cmx_icenuc_kernels.hip — ABIFM immersion-freezing + Koop-2000 homogeneous-freezing rates and the
// water-activity helpers, fused per point, for gfx950; C-ABI entry points of include/cmx.h §(4).
//
// Reference (src = /root/reference/src): CO.a_w_ice / a_w_eT  Common.jl:250-271; CMI_het.ABIFM_J
// IceNucleation.jl:124-134; CMI_hom.homogeneous_J_cubic / _linear  IceNucleation.jl:557-584; products with
// the droplet area / volume as formed in parcel/ParcelTendencies.jl:120-133,194-205.
//
// HBM-bound pointwise map: 3 input columns, up to 5 output columns (20 B/point for the (T, a_w, r) →
// (rate_het, rate_hom) configuration, f32).  Same launch shape as the SB2006 kernel: one 16-byte vector per
// lane, one short-lived 256-lane workgroup per tile, non-temporal accesses.
//
// Numerics: a_w_ice = p_sat,ice / p_sat,liq is ONE exp2 of the difference of the two Rankine–Kirchhoff
// exponents (host-folded coefficient differences), not a ratio of two exponentials.  The Koop cubic
// log10 J = c1 + c2 Δ − c3 Δ² + c4 Δ³ cancels from O(2500) terms to O(6); in Float32 that alone would cost
// ≈3e-4 relative on J, so the four-term Horner form is evaluated in double (4 FMAs per point on an
// HBM-bound kernel) and only the result is rounded to FT.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "cmx_launch.hpp"
#include "cmx_math.hpp"
#include "cmx_lean_eval.hpp"

namespace cmx {

template <typename FT> struct IceNucConsts {
    // log2 a_w_ice(T) = d0 + da·log2(T/T_tr) + db·(1/T_tr − 1/T)
    FT aw_d0, aw_da, aw_db, inv_T_tr;
    // log2 p_sat,liq(T) = c0 + a·log2(T/T_tr) + b·(1/T_tr − 1/T)   (a_w_eT)
    FT ps_c0, ps_a, ps_b;
    FT abifm_m_l2, abifm_c_l2;        // (m Δ + c + 4)·log2(10)
    double c1, c2, c3, c4;            // Koop cubic, evaluated in double
    FT lin_c1_l2, lin_c2_l2;          // (linear_c2 Δ + linear_c1 + 6)·log2(10)
    FT d_min, d_max;
    FT l2_four_pi, l2_four_thirds_pi;
    // H2SO4 solution (Common.jl:188-220): log2 p_sol[Pa] = hp1(x) + hp2(x)/T,  hp1 = Σ hp1[m] x^m (m ≤ 3), hp2 = Σ hp2[m] x^m (m ≤ 2)
    FT hp1[4], hp2[3];
};

template <typename FT, typename HS> static void add_h2so4_consts(IceNucConsts<FT> &c, const HS &h) {
    const double l2e = 1.4426950408889634074, w2 = h.w_2;
    c.hp1[0] = (FT)((double)h.c1 * l2e + std::log2(100.0));   // ·100: mbar → Pa
    c.hp1[1] = (FT)(-(double)h.c2 * l2e);
    c.hp1[2] = (FT)((double)h.c3 * w2 * l2e);                  // c3 x w_h, w_h = w_2 x
    c.hp1[3] = (FT)(-(double)h.c4 * w2 * w2 * l2e);            // −c4 x w_h²
    c.hp2[0] = (FT)((double)h.c5 * l2e);
    c.hp2[1] = (FT)((double)h.c6 * l2e);
    c.hp2[2] = (FT)(-(double)h.c7 * w2 * l2e);                 // −c7 x w_h
}
// log2 of CO.H2SO4_soln_saturation_vapor_pressure(prs, x, T) [Pa]
template <typename FT> __device__ __forceinline__ FT l2_p_sol_dev(const IceNucConsts<FT> &c, FT x, FT inv_T) {
    using M = Math<FT>;
    const FT p1 = M::fma(M::fma(M::fma(c.hp1[3], x, c.hp1[2]), x, c.hp1[1]), x, c.hp1[0]);
    const FT p2 = M::fma(M::fma(c.hp2[2], x, c.hp2[1]), x, c.hp2[0]);
    return M::fma(p2, inv_T, p1);
}
template <typename FT> __device__ __forceinline__ FT l2_p_sat_liq_dev(const IceNucConsts<FT> &c, FT T, FT inv_T) {
    using M = Math<FT>;
    return M::fma(c.ps_a, M::log2(T * c.inv_T_tr), M::fma(c.ps_b, c.inv_T_tr - inv_T, c.ps_c0));
}

template <typename FT, typename TH, typename DU, typename KO>
static IceNucConsts<FT> make_icenuc_consts(const TH &tp, const DU *dust, const KO *koop) {
    IceNucConsts<FT> c{};
    const double l2e = 1.4426950408889634074, l2_10 = 3.3219280948873623479, pi = 3.14159265358979323846;
    const double Rv = tp.R_v, Ttr = tp.T_triple, T0 = tp.T_0;
    const double dcp_l = (double)tp.cp_v - (double)tp.cp_l, dcp_i = (double)tp.cp_v - (double)tp.cp_i;
    const double a_l = dcp_l / Rv, a_i = dcp_i / Rv;
    const double b_l = ((double)tp.LH_v0 - dcp_l * T0) / Rv * l2e, b_i = ((double)tp.LH_s0 - dcp_i * T0) / Rv * l2e;
    c.aw_d0 = (FT)0;                                   // same triple-point pressure for both phases
    c.aw_da = (FT)(a_i - a_l);
    c.aw_db = (FT)(b_i - b_l);
    c.inv_T_tr = (FT)(1.0 / Ttr);
    c.ps_c0 = (FT)std::log2((double)tp.press_triple);
    c.ps_a = (FT)a_l;
    c.ps_b = (FT)b_l;
    if (dust) {
        c.abifm_m_l2 = (FT)((double)dust->ABIFM_m * l2_10);
        c.abifm_c_l2 = (FT)(((double)dust->ABIFM_c + 4.0) * l2_10);
    }
    if (koop) {
        c.c1 = koop->c1; c.c2 = koop->c2; c.c3 = koop->c3; c.c4 = koop->c4;
        c.lin_c1_l2 = (FT)(((double)koop->linear_c1 + 6.0) * l2_10);
        c.lin_c2_l2 = (FT)((double)koop->linear_c2 * l2_10);
        c.d_min = (FT)koop->delta_a_w_min;
        c.d_max = (FT)koop->delta_a_w_max;
    }
    c.l2_four_pi = (FT)std::log2(4.0 * pi);
    c.l2_four_thirds_pi = (FT)std::log2(4.0 / 3.0 * pi);
    return c;
}

template <typename FT> struct IceNucIO {
    const FT *T, *a_w, *r;
    FT *delta_a_w, *J_het, *J_hom, *rate_het, *rate_hom;
    unsigned long long *n_err;
};

template <typename FT> __device__ __forceinline__ FT a_w_ice_dev(const IceNucConsts<FT> &c, FT T, FT inv_T) {
    using M = Math<FT>;
    return M::exp2(M::fma(c.aw_da, M::log2(T * c.inv_T_tr), M::fma(c.aw_db, c.inv_T_tr - inv_T, c.aw_d0)));
}

// RATES_ONLY: only the per-droplet rates J·4πr² / J·4⁄3πr³ are requested (the BASELINE configuration): the two exp2 for the bare
// rate coefficients are then dead code — as a compile-time fact (with run-time nullable pointers they are always evaluated)
// FROM_X: the second input column is the H2SO4 weight fraction x of the solution droplets and a_w = CO.a_w_xT(prs, tps, x, T) is formed here
// (cmx_ice_nucleation_rates_xT_*: config 4 as the parcel model drives it, parcel/ParcelTendencies.jl:120-133)
template <typename FT, bool LINEAR, int VEC, bool RATES_ONLY = false, bool FROM_X = false>
__global__ __launch_bounds__(kBlock) void ice_nucleation_kernel(const IceNucConsts<FT> c, const IceNucIO<FT> io,
                                                                const int64_t nvec) {
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = i < nvec;
    int nerr = 0;
    const bool want_rates = io.rate_het || io.rate_hom;
    FT T[VEC], aw[VEC], r[VEC] = {};
    if (active) {
        load_col<FT, VEC>(io.T, i, T);
        load_col<FT, VEC>(io.a_w, i, aw);
        if (want_rates) load_col<FT, VEC>(io.r, i, r);
    }
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly (every lane of the workgroup reaches the barrier inside); no-op for Float32
    if (active) {
        FT d[VEC], jh[VEC], jo[VEC], rh[VEC], ro[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const FT inv_T = M::rcp(T[k]);
            if constexpr (FROM_X) aw[k] = M::exp2(l2_p_sol_dev<FT>(c, aw[k], inv_T) - l2_p_sat_liq_dev<FT>(c, T[k], inv_T));   // a_w_xT  Common.jl:235-246
            d[k] = aw[k] - a_w_ice_dev<FT>(c, T[k], inv_T);                        // Δa_w
            const FT l2_jh = M::fma(c.abifm_m_l2, d[k], c.abifm_c_l2);            // ABIFM_J  IceNucleation.jl:124-134
            FT l2_jo;
            bool ok = true;
            if constexpr (LINEAR) {
                l2_jo = M::fma(c.lin_c2_l2, d[k], c.lin_c1_l2);                   // homogeneous_J_linear :581-584
            } else {
                const double dd = (double)d[k];
                const double logJ = __builtin_fma(dd, __builtin_fma(dd, __builtin_fma(dd, c.c4, -c.c3), c.c2), c.c1);
                l2_jo = (FT)((logJ + 6.0) * 3.3219280948873623479);               // homogeneous_J_cubic :557-565
                ok = (c.d_min <= d[k]) && (d[k] <= c.d_max);                       // DomainError → NaN  :558-562
                nerr += ok ? 0 : 1;
            }
            if constexpr (!RATES_ONLY) {
                jh[k] = M::exp2(l2_jh);
                jo[k] = ok ? M::exp2(l2_jo) : FT(__builtin_nan(""));
            }
            // J·4πr² and J·4/3πr³ formed in the log2 domain: J_hom alone reaches 1e39 (Float32 overflow) for
            // Δa_w ≈ 0.4 with the linear fit while the per-droplet rate J·V stays O(1e16)
            const FT l2_r = M::log2(r[k]);
            rh[k] = M::exp2(l2_jh + M::fma(FT(2), l2_r, c.l2_four_pi));
            ro[k] = ok ? M::exp2(l2_jo + M::fma(FT(3), l2_r, c.l2_four_thirds_pi)) : FT(__builtin_nan(""));
        }
        if constexpr (!RATES_ONLY) {
            if (io.delta_a_w) store_col<FT, VEC>(io.delta_a_w, i, d);
            if (io.J_het) store_col<FT, VEC>(io.J_het, i, jh);
            if (io.J_hom) store_col<FT, VEC>(io.J_hom, i, jo);
        }
        if (io.rate_het) store_col<FT, VEC>(io.rate_het, i, rh);
        if (io.rate_hom) store_col<FT, VEC>(io.rate_hom, i, ro);
    }
    if constexpr (!LINEAR) {
        // Domain-error count: LDS reduce per workgroup, then ONE global atomic per workgroup into one of
        // CMX_ICENUC_ERR_SLOTS counters, each on its own 128-byte line.  (One atomic per wave into a single word
        // serialises on one L2 atomic unit at ≈12 ns each: measured 4.7 ms for 1e8 points vs 0.3 ms of HBM time.)
        if (io.n_err) {   // wave-uniform
            __shared__ int blk_err;
            if (threadIdx.x == 0) blk_err = 0;
            __syncthreads();
            if (nerr) atomicAdd(&blk_err, nerr);
            __syncthreads();
            if (threadIdx.x == 0 && blk_err)
                atomicAdd(io.n_err + (size_t)(blockIdx.x % CMX_ICENUC_ERR_SLOTS) * (CMX_ICENUC_ERR_WORDS / CMX_ICENUC_ERR_SLOTS),
                          (unsigned long long)blk_err);
        }
    }
}

template <typename FT> struct WaterActIO { const FT *T, *e; FT *a_w_ice, *a_w_eT; };

template <typename FT>
__global__ __launch_bounds__(kBlock) void water_activity_kernel(const IceNucConsts<FT> c, const WaterActIO<FT> io,
                                                                const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FT T = io.T[i], inv_T = M::rcp(T);
    if (io.a_w_ice) io.a_w_ice[i] = a_w_ice_dev<FT>(c, T, inv_T);                  // Common.jl:267-271
    if (io.a_w_eT) {                                                               // Common.jl:250-253
        const FT l2_ps = M::fma(c.ps_a, M::log2(T * c.inv_T_tr), M::fma(c.ps_b, c.inv_T_tr - inv_T, c.ps_c0));
        io.a_w_eT[i] = io.e[i] * M::exp2(-l2_ps);
    }
}

template <typename FT, typename TH, typename DU, typename KO, typename HS>
static int32_t icenuc_entry(const TH *tps, const DU *dust, const KO *koop, const HS *h2so4, bool from_x, uint32_t flags, int64_t n, const FT *T,
                            const FT *a_w, const FT *r, FT *delta_a_w, FT *J_het, FT *J_hom, FT *rate_het, FT *rate_hom,
                            int64_t *n_domain_errors, void *stream) {
    if (!tps || !dust || !koop || n < 0 || (flags & ~CMX_ICENUC_HOM_LINEAR) || (from_x && !h2so4)) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!T || !a_w || ((rate_het || rate_hom) && !r)) return CMX_ERR_BAD_ARG;
    IceNucConsts<FT> c = make_icenuc_consts<FT>(*tps, dust, koop);
    if (from_x) add_h2so4_consts<FT>(c, *h2so4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int VEC = Math<FT>::VEC;
    const void *ptrs[] = {T, a_w, r, delta_a_w, J_het, J_hom, rate_het, rate_hom};
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(T) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (const void *p : ptrs)
        if (p) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(p) & 15u) == mis0);
    const bool linear = flags & CMX_ICENUC_HOM_LINEAR;
    auto off = [](auto *p, int64_t lo) { return p ? p + lo : p; };
    auto launch_range = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        IceNucIO<FT> io{T + lo, a_w + lo, off(r, lo), off(delta_a_w, lo), off(J_het, lo), off(J_hom, lo),
                        off(rate_het, lo), off(rate_hom, lo), reinterpret_cast<unsigned long long *>(n_domain_errors)};
        const int64_t nv = count / V;
        const unsigned grid = (unsigned)((nv + kBlock - 1) / kBlock);
        const bool rates_only = !delta_a_w && !J_het && !J_hom && rate_het && rate_hom;
        if (from_x) {   // the solution-droplet input mode: one instantiation per homogeneous-rate form
            if (linear) hipLaunchKernelGGL((ice_nucleation_kernel<FT, true, V, false, true>), dim3(grid), dim3(kBlock), 0, s, c, io, nv);
            else hipLaunchKernelGGL((ice_nucleation_kernel<FT, false, V, false, true>), dim3(grid), dim3(kBlock), 0, s, c, io, nv);
        } else if (rates_only) {
            if (linear) hipLaunchKernelGGL((ice_nucleation_kernel<FT, true, V, true>), dim3(grid), dim3(kBlock), 0, s, c, io, nv);
            else hipLaunchKernelGGL((ice_nucleation_kernel<FT, false, V, true>), dim3(grid), dim3(kBlock), 0, s, c, io, nv);
        } else if (linear) hipLaunchKernelGGL((ice_nucleation_kernel<FT, true, V>), dim3(grid), dim3(kBlock), 0, s, c, io, nv);
        else hipLaunchKernelGGL((ice_nucleation_kernel<FT, false, V>), dim3(grid), dim3(kBlock), 0, s, c, io, nv);
    };
    if (same_mis) {
        const int64_t head = std::min<int64_t>(n, mis0 ? (int64_t)((16 - mis0) / sizeof(FT)) : 0);
        const int64_t body = ((n - head) / VEC) * VEC;
        launch_range(std::integral_constant<int, 1>{}, 0, head);
        launch_range(std::integral_constant<int, VEC>{}, head, body);
        launch_range(std::integral_constant<int, 1>{}, head + body, n - head - body);
    } else {
        launch_range(std::integral_constant<int, 1>{}, 0, n);
    }
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

template <typename FT, typename TH>
static int32_t water_activity_entry(const TH *tps, int64_t n, const FT *T, const FT *e, FT *a_w_ice, FT *a_w_eT,
                                    void *stream) {
    if (!tps || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!T || (a_w_eT && !e)) return CMX_ERR_BAD_ARG;
    const IceNucConsts<FT> c =
        make_icenuc_consts<FT>(*tps, (const cmx_abifm_dust_f64 *)nullptr, (const cmx_koop2000_f64 *)nullptr);
    WaterActIO<FT> io{T, e, a_w_ice, a_w_eT};
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((water_activity_kernel<FT>), dim3(grid), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), c, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// P3.het_ice_nucleation(aerosol, tps, q_lcl, N_lcl, RH, T, ρₐ) — src/P3_processes.jl:20-46: ABIFM immersion freezing on an assumed
// aerosol surface A_aer = 1e-10 m² per droplet; a non-finite J counts as no nucleation.  32 B/point (f32), one point per lane.
template <typename FT>
__global__ __launch_bounds__(kBlock) void p3_het_nucleation_kernel(const IceNucConsts<FT> c, const FT *__restrict__ q_lcl,
                                                                  const FT *__restrict__ N_lcl, const FT *__restrict__ RH,
                                                                  const FT *__restrict__ T, const FT *__restrict__ rho, FT *__restrict__ dNdt,
                                                                  FT *__restrict__ dLdt, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FT Ti = T[i];
    const FT d = RH[i] - a_w_ice_dev<FT>(c, Ti, M::rcp(Ti));
    const FT J = M::exp2(M::fma(c.abifm_m_l2, d, c.abifm_c_l2));              // ABIFM_J  IceNucleation.jl:124-134 [1/m²/s]
    const FT JA = (J - J == FT(0)) ? J * FT(1e-10) : FT(0);                    // isfinite(J) ? J·A_aer : 0
    if (dNdt) dNdt[i] = M::max(FT(0), JA * N_lcl[i]);
    if (dLdt) dLdt[i] = M::max(FT(0), JA * q_lcl[i] * rho[i]);
}
template <typename FT, typename TH, typename DU>
static int32_t p3_het_nucleation_entry(const DU *dust, const TH *tps, int64_t n, const FT *q_lcl, const FT *N_lcl, const FT *RH, const FT *T,
                                       const FT *rho, FT *dNdt, FT *dLdt, void *stream) {
    if (!dust || !tps || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!q_lcl || !N_lcl || !RH || !T || !rho || (!dNdt && !dLdt)) return CMX_ERR_BAD_ARG;
    using KO = std::conditional_t<std::is_same_v<FT, float>, cmx_koop2000_f32, cmx_koop2000_f64>;
    const IceNucConsts<FT> c = make_icenuc_consts<FT>(*tps, dust, (const KO *)nullptr);
    hipLaunchKernelGGL((p3_het_nucleation_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c, q_lcl, N_lcl, RH, T, rho, dNdt, dLdt, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// ---- round 3: the remaining public functions of the subsystems this library replaces (VERDICT r02 row g) ------------------------------
// One point per lane; each is a handful of instructions per point.

// CO.H2SO4_soln_saturation_vapor_pressure / CO.a_w_xT — Common.jl:188-246
template <typename FT>
__global__ __launch_bounds__(kBlock) void h2so4_solution_kernel(const IceNucConsts<FT> c, const FT *__restrict__ x, const FT *__restrict__ T,
                                                               FT *__restrict__ p_sol, FT *__restrict__ a_w, const int64_t n) {
    Math<FT>::prepare();
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FT Ti = T[i], inv_T = M::rcp(Ti);
    const FT l2p = l2_p_sol_dev<FT>(c, x[i], inv_T);
    if (p_sol) p_sol[i] = M::exp2(l2p);
    if (a_w) a_w[i] = M::exp2(l2p - l2_p_sat_liq_dev<FT>(c, Ti, inv_T));
}

// CMI_het.dust_activated_number_fraction / MohlerDepositionRate — IceNucleation.jl:44-79
template <typename FT> struct MohlerConsts { FT S_i_max, T_thr, S0_warm, S0_cold, a_warm_l2e, a_cold_l2e, a_warm, a_cold; };
template <typename FT>
__global__ __launch_bounds__(kBlock) void mohler_deposition_kernel(const MohlerConsts<FT> c, const FT *__restrict__ S_i, const FT *__restrict__ T,
                                                                  const FT *__restrict__ dSi_dt, const FT *__restrict__ N_aer,
                                                                  FT *__restrict__ act_frac, FT *__restrict__ dep_rate,
                                                                  unsigned long long *__restrict__ n_err, const int64_t n) {
    Math<FT>::prepare();
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool bad = false;
    if (i < n) {
        const FT Si = S_i[i];
        const bool warm = T[i] > c.T_thr;
        bad = !(Si < c.S_i_max);                                  // the reference's @assert Si < ip.Sᵢ_max
        if (act_frac) {
            const FT f = M::max(FT(0), M::exp2((warm ? c.a_warm_l2e : c.a_cold_l2e) * (Si - (warm ? c.S0_warm : c.S0_cold))) - FT(1));
            act_frac[i] = bad ? M::nan() : f;
        }
        if (dep_rate) {
            const FT rate = M::max(FT(0), N_aer[i] * (warm ? c.a_warm : c.a_cold) * dSi_dt[i]);
            dep_rate[i] = bad ? M::nan() : rate;
        }
    }
    if (n_err) {   // wave-uniform; one atomic per wave that holds a domain error (rare)
        const unsigned long long m = __ballot(bad);
        if (m && (threadIdx.x & 63) == 0) atomicAdd(n_err, (unsigned long long)__popcll(m));
    }
}

// CMI_het.deposition_J — IceNucleation.jl:81-102
template <typename FT>
__global__ __launch_bounds__(kBlock) void deposition_J_kernel(const FT m_l2, const FT c_l2, const FT *__restrict__ d, FT *__restrict__ J, const int64_t n) {
    Math<FT>::prepare();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) J[i] = Math<FT>::exp2(Math<FT>::fma(m_l2, d[i], c_l2));
}

// CMI_het.INP_concentration_frequency — IceNucleation.jl:219-226 with INP_concentration_mean :250-253
template <typename FT> struct InpFreqConsts { FT T_freeze, b10, log_a, ln2, inv_2s2_l2e, inv_norm; };
template <typename FT>
__global__ __launch_bounds__(kBlock) void inp_frequency_kernel(const InpFreqConsts<FT> c, const FT *__restrict__ INPC, const FT *__restrict__ T,
                                                              FT *__restrict__ freq, const int64_t n) {
    Math<FT>::prepare();
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FT Ti = T[i];
    const FT Tc = M::min(Ti - c.T_freeze, FT(0));
    const FT mu = M::fma(FT(9) * c.ln2, M::log2(-(c.b10 * Tc)), -c.log_a);          // 9 log(−b T_c/10) − log a
    const FT dl = M::fma(c.ln2, M::log2(INPC[i]), -mu);                              // log(INPC) − μ
    const FT f = M::exp2(-(dl * dl) * c.inv_2s2_l2e) * c.inv_norm;                   // exp(−(…)²/2σ²)/√(2πσ²)
    freq[i] = Ti >= c.T_freeze ? FT(0) : f;
}

template <typename FT, typename HS, typename TH>
static int32_t h2so4_entry(const HS *prs, const TH *tps, int64_t n, const FT *x, const FT *T, FT *p_sol, FT *a_w, void *stream) {
    if (!prs || n < 0 || (a_w && !tps) || (!p_sol && !a_w)) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!x || !T) return CMX_ERR_BAD_ARG;
    IceNucConsts<FT> c{};
    if (tps) c = make_icenuc_consts<FT>(*tps, (const cmx_abifm_dust_f64 *)nullptr, (const cmx_koop2000_f64 *)nullptr);
    add_h2so4_consts<FT>(c, *prs);
    hipLaunchKernelGGL((h2so4_solution_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), c,
                       x, T, p_sol, a_w, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}
template <typename FT, typename DU, typename IP>
static int32_t mohler_entry(const DU *dust, const IP *ip, int64_t n, const FT *S_i, const FT *T, const FT *dSi_dt, const FT *N_aer, FT *act_frac,
                            FT *dep_rate, int64_t *n_err, void *stream) {
    if (!dust || !ip || n < 0 || (!act_frac && !dep_rate)) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!S_i || !T || (dep_rate && (!dSi_dt || !N_aer))) return CMX_ERR_BAD_ARG;
    const double l2e = 1.4426950408889634074;
    const MohlerConsts<FT> c{(FT)ip->S_i_max, (FT)ip->T_thr, (FT)dust->S0_warm, (FT)dust->S0_cold, (FT)((double)dust->a_warm * l2e),
                             (FT)((double)dust->a_cold * l2e), (FT)dust->a_warm, (FT)dust->a_cold};
    hipLaunchKernelGGL((mohler_deposition_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream),
                       c, S_i, T, dSi_dt, N_aer, act_frac, dep_rate, reinterpret_cast<unsigned long long *>(n_err), n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}
template <typename FT, typename DU> static int32_t deposition_J_entry(const DU *dust, int64_t n, const FT *d, FT *J, void *stream) {
    if (!dust || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!d || !J) return CMX_ERR_BAD_ARG;
    const double l2_10 = 3.3219280948873623479;
    hipLaunchKernelGGL((deposition_J_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream),
                       (FT)((double)dust->deposition_m * l2_10), (FT)(((double)dust->deposition_c + 4.0) * l2_10), d, J, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}
template <typename FT, typename IP> static int32_t inp_frequency_entry(const IP *ip, int64_t n, const FT *INPC, const FT *T, FT *freq, void *stream) {
    if (!ip || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!INPC || !T || !freq) return CMX_ERR_BAD_ARG;
    const double l2e = 1.4426950408889634074, pi = 3.14159265358979323846, two_s2 = 2.0 * (double)ip->sigma * (double)ip->sigma;
    const InpFreqConsts<FT> c{(FT)ip->T_freeze, (FT)((double)ip->b / 10.0), (FT)ip->log_a, (FT)0.69314718055994530942, (FT)(l2e / two_s2),
                              (FT)(1.0 / std::sqrt(pi * two_s2))};
    hipLaunchKernelGGL((inp_frequency_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), c,
                       INPC, T, freq, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_ice_nucleation_rates_f32(const cmx_thermo_f32 *tps, const cmx_abifm_dust_f32 *dust, const cmx_koop2000_f32 *koop,
                                     uint32_t flags, int64_t n, const float *T, const float *a_w, const float *r,
                                     float *delta_a_w, float *J_het, float *J_hom, float *rate_het, float *rate_hom,
                                     int64_t *n_domain_errors, void *stream) {
    return cmx::icenuc_entry<float>(tps, dust, koop, (const cmx_h2so4_solution_params_f32 *)nullptr, false, flags, n, T, a_w, r, delta_a_w, J_het, J_hom,
                                    rate_het, rate_hom, n_domain_errors, stream);
}
int32_t cmx_ice_nucleation_rates_f64(const cmx_thermo_f64 *tps, const cmx_abifm_dust_f64 *dust, const cmx_koop2000_f64 *koop,
                                     uint32_t flags, int64_t n, const double *T, const double *a_w, const double *r,
                                     double *delta_a_w, double *J_het, double *J_hom, double *rate_het, double *rate_hom,
                                     int64_t *n_domain_errors, void *stream) {
    return cmx::icenuc_entry<double>(tps, dust, koop, (const cmx_h2so4_solution_params_f64 *)nullptr, false, flags, n, T, a_w, r, delta_a_w, J_het, J_hom,
                                     rate_het, rate_hom, n_domain_errors, stream);
}
int32_t cmx_water_activity_f32(const cmx_thermo_f32 *tps, int64_t n, const float *T, const float *e, float *a_w_ice,
                               float *a_w_eT, void *stream) {
    return cmx::water_activity_entry<float>(tps, n, T, e, a_w_ice, a_w_eT, stream);
}
int32_t cmx_water_activity_f64(const cmx_thermo_f64 *tps, int64_t n, const double *T, const double *e, double *a_w_ice,
                               double *a_w_eT, void *stream) {
    return cmx::water_activity_entry<double>(tps, n, T, e, a_w_ice, a_w_eT, stream);
}

int32_t cmx_p3_het_ice_nucleation_f32(const cmx_abifm_dust_f32 *dust, const cmx_thermo_f32 *tps, int64_t n, const float *q_lcl,
                                      const float *N_lcl, const float *RH, const float *T, const float *rho_air, float *dNdt, float *dLdt,
                                      void *stream) {
    return cmx::p3_het_nucleation_entry<float>(dust, tps, n, q_lcl, N_lcl, RH, T, rho_air, dNdt, dLdt, stream);
}
int32_t cmx_p3_het_ice_nucleation_f64(const cmx_abifm_dust_f64 *dust, const cmx_thermo_f64 *tps, int64_t n, const double *q_lcl,
                                      const double *N_lcl, const double *RH, const double *T, const double *rho_air, double *dNdt,
                                      double *dLdt, void *stream) {
    return cmx::p3_het_nucleation_entry<double>(dust, tps, n, q_lcl, N_lcl, RH, T, rho_air, dNdt, dLdt, stream);
}

int32_t cmx_ice_nucleation_rates_xT_f32(const cmx_thermo_f32 *tps, const cmx_abifm_dust_f32 *dust, const cmx_koop2000_f32 *koop,
                                        const cmx_h2so4_solution_params_f32 *h2so4, uint32_t flags, int64_t n, const float *T, const float *x_sulph,
                                        const float *r, float *delta_a_w, float *J_het, float *J_hom, float *rate_het, float *rate_hom,
                                        int64_t *n_domain_errors, void *stream) {
    return cmx::icenuc_entry<float>(tps, dust, koop, h2so4, true, flags, n, T, x_sulph, r, delta_a_w, J_het, J_hom, rate_het, rate_hom, n_domain_errors, stream);
}
int32_t cmx_ice_nucleation_rates_xT_f64(const cmx_thermo_f64 *tps, const cmx_abifm_dust_f64 *dust, const cmx_koop2000_f64 *koop,
                                        const cmx_h2so4_solution_params_f64 *h2so4, uint32_t flags, int64_t n, const double *T, const double *x_sulph,
                                        const double *r, double *delta_a_w, double *J_het, double *J_hom, double *rate_het, double *rate_hom,
                                        int64_t *n_domain_errors, void *stream) {
    return cmx::icenuc_entry<double>(tps, dust, koop, h2so4, true, flags, n, T, x_sulph, r, delta_a_w, J_het, J_hom, rate_het, rate_hom, n_domain_errors, stream);
}
int32_t cmx_h2so4_solution_f32(const cmx_h2so4_solution_params_f32 *prs, const cmx_thermo_f32 *tps, int64_t n, const float *x_sulph, const float *T,
                               float *p_sol, float *a_w, void *stream) {
    return cmx::h2so4_entry<float>(prs, tps, n, x_sulph, T, p_sol, a_w, stream);
}
int32_t cmx_h2so4_solution_f64(const cmx_h2so4_solution_params_f64 *prs, const cmx_thermo_f64 *tps, int64_t n, const double *x_sulph, const double *T,
                               double *p_sol, double *a_w, void *stream) {
    return cmx::h2so4_entry<double>(prs, tps, n, x_sulph, T, p_sol, a_w, stream);
}
int32_t cmx_mohler2006_deposition_f32(const cmx_mohler_dust_f32 *dust, const cmx_mohler2006_f32 *ip, int64_t n, const float *S_i, const float *T,
                                      const float *dSi_dt, const float *N_aer, float *act_frac, float *dep_rate, int64_t *n_domain_errors,
                                      void *stream) {
    return cmx::mohler_entry<float>(dust, ip, n, S_i, T, dSi_dt, N_aer, act_frac, dep_rate, n_domain_errors, stream);
}
int32_t cmx_mohler2006_deposition_f64(const cmx_mohler_dust_f64 *dust, const cmx_mohler2006_f64 *ip, int64_t n, const double *S_i, const double *T,
                                      const double *dSi_dt, const double *N_aer, double *act_frac, double *dep_rate, int64_t *n_domain_errors,
                                      void *stream) {
    return cmx::mohler_entry<double>(dust, ip, n, S_i, T, dSi_dt, N_aer, act_frac, dep_rate, n_domain_errors, stream);
}
int32_t cmx_deposition_J_f32(const cmx_deposition_dust_f32 *dust, int64_t n, const float *delta_a_w, float *J, void *stream) {
    return cmx::deposition_J_entry<float>(dust, n, delta_a_w, J, stream);
}
int32_t cmx_deposition_J_f64(const cmx_deposition_dust_f64 *dust, int64_t n, const double *delta_a_w, double *J, void *stream) {
    return cmx::deposition_J_entry<double>(dust, n, delta_a_w, J, stream);
}
int32_t cmx_inp_concentration_frequency_f32(const cmx_frostenberg2023_f32 *ip, int64_t n, const float *INPC, const float *T, float *freq, void *stream) {
    return cmx::inp_frequency_entry<float>(ip, n, INPC, T, freq, stream);
}
int32_t cmx_inp_concentration_frequency_f64(const cmx_frostenberg2023_f64 *ip, int64_t n, const double *INPC, const double *T, double *freq, void *stream) {
    return cmx::inp_frequency_entry<double>(ip, n, INPC, T, freq, stream);
}

// the same diagnostic as cmx_lean_eval_f64, compiled in this literal-coefficient translation unit (cmx_lean_eval.hpp)
int32_t cmx_lean_eval_literal_f64(int32_t which, int64_t n, const double *x, double *y, void *stream) { return cmx::lean_eval_entry<1>(which, n, x, y, stream); }

}  // extern "C"
