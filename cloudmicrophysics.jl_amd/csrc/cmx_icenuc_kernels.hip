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
};

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
template <typename FT, bool LINEAR, int VEC, bool RATES_ONLY = false>
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

template <typename FT, typename TH, typename DU, typename KO>
static int32_t icenuc_entry(const TH *tps, const DU *dust, const KO *koop, uint32_t flags, int64_t n, const FT *T,
                            const FT *a_w, const FT *r, FT *delta_a_w, FT *J_het, FT *J_hom, FT *rate_het, FT *rate_hom,
                            int64_t *n_domain_errors, void *stream) {
    if (!tps || !dust || !koop || n < 0 || (flags & ~CMX_ICENUC_HOM_LINEAR)) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!T || !a_w || ((rate_het || rate_hom) && !r)) return CMX_ERR_BAD_ARG;
    const IceNucConsts<FT> c = make_icenuc_consts<FT>(*tps, dust, koop);
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
        if (rates_only) {
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

}  // namespace cmx

extern "C" {

int32_t cmx_ice_nucleation_rates_f32(const cmx_thermo_f32 *tps, const cmx_abifm_dust_f32 *dust, const cmx_koop2000_f32 *koop,
                                     uint32_t flags, int64_t n, const float *T, const float *a_w, const float *r,
                                     float *delta_a_w, float *J_het, float *J_hom, float *rate_het, float *rate_hom,
                                     int64_t *n_domain_errors, void *stream) {
    return cmx::icenuc_entry<float>(tps, dust, koop, flags, n, T, a_w, r, delta_a_w, J_het, J_hom, rate_het, rate_hom,
                                    n_domain_errors, stream);
}
int32_t cmx_ice_nucleation_rates_f64(const cmx_thermo_f64 *tps, const cmx_abifm_dust_f64 *dust, const cmx_koop2000_f64 *koop,
                                     uint32_t flags, int64_t n, const double *T, const double *a_w, const double *r,
                                     double *delta_a_w, double *J_het, double *J_hom, double *rate_het, double *rate_hom,
                                     int64_t *n_domain_errors, void *stream) {
    return cmx::icenuc_entry<double>(tps, dust, koop, flags, n, T, a_w, r, delta_a_w, J_het, J_hom, rate_het, rate_hom,
                                     n_domain_errors, stream);
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

}  // extern "C"
