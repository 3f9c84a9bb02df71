// cmx_sb2006_kernels.hip — C-ABI entry points (include/cmx.h) of the fused SB2006 warm-rain kernels:
// argument validation, host-side constant folding, alignment dispatch, launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "cmx_sb2006_kernels.hpp"

namespace cmx {

// ---- host side -----------------------------------------------------------------------------
static int decode_vel(uint32_t flags, bool want) {
    if (!want) return VEL_NONE;
    const bool sb = flags & CMX_VEL_SB2006, ch = flags & CMX_VEL_CHEN2022;
    if (sb == ch) return -1;   // none or both
    return sb ? VEL_SB : VEL_CHEN;
}

#ifndef CMX_TEND_BS
#define CMX_TEND_BS 128
#endif
constexpr int kTendBS = CMX_TEND_BS;   // lanes per workgroup of the fused tendency kernel (tuning: -DCMX_TEND_BS=64|256|512).  Measured with the
                                       // 263-instruction point function, 1e8 f32 points, three runs on one box: 128 → 0.883–0.885 ms, 64 → 0.887–0.902,
                                       // 512 → 0.894–0.923, 256 → 0.915–0.962 (256 was the optimum of the 350-instruction version)

#ifndef CMX_SB_INTPOW
#define CMX_SB_INTPOW 1      // A/B switch for the integer-exponent instantiations (cmx_sb2006.hpp INTPOW)
#endif
template <typename FT, int VEC>
static void launch_tendencies(bool limited, int vel, bool intpow, const SbConsts<FT> &c, const SbIn<FT> &in, const SbOut<FT> &out,
                              int64_t nvec, hipStream_t s) {
    // one short-lived workgroup per tile of kTendBS lanes (see the kernel's header comment)
    const int64_t grid = (nvec + kTendBS - 1) / kTendBS;
#define CMX_LAUNCH(L, V)                                                                                                                   \
    do {                                                                                                                                   \
        if (intpow && CMX_SB_INTPOW)                                                                                                       \
            CMX_LAUNCH_FRONT((sb2006_tendencies_kernel<FT, L, V, VEC, kTendBS, 1, true, true>), dim3((unsigned)grid), dim3(kTendBS), 0, s, c, in, \
                               out, nvec);                                                                                                 \
        else                                                                                                                               \
            CMX_LAUNCH_FRONT((sb2006_tendencies_kernel<FT, L, V, VEC, kTendBS>), dim3((unsigned)grid), dim3(kTendBS), 0, s, c, in, out,  \
                               nvec);                                                                                                      \
    } while (0)
    if (limited) {
        if (vel == VEL_NONE) CMX_LAUNCH(true, VEL_NONE);
        else if (vel == VEL_SB) CMX_LAUNCH(true, VEL_SB);
        else if (vel == VEL_CHEN) CMX_LAUNCH(true, VEL_CHEN);
        else CMX_LAUNCH(true, VEL_CHEN_GEN);
    } else {
        if (vel == VEL_NONE) CMX_LAUNCH(false, VEL_NONE);
        else if (vel == VEL_SB) CMX_LAUNCH(false, VEL_SB);
        else if (vel == VEL_CHEN) CMX_LAUNCH(false, VEL_CHEN);
        else CMX_LAUNCH(false, VEL_CHEN_GEN);
    }
#undef CMX_LAUNCH
}

template <typename FT, typename WR, typename TH, typename VL>
static int32_t tendencies_entry(const WR *wr, const TH *tps, const VL *vel, uint32_t flags, int64_t n, const FT *rho,
                                const FT *T, const FT *q_tot, const FT *q_lcl, const FT *n_lcl, const FT *q_rai,
                                const FT *n_rai, FT *dq_lcl, FT *dn_lcl, FT *dq_rai, FT *dn_rai, FT *vt_n, FT *vt_m,
                                void *stream) {
    if (!wr || !tps || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!rho || !T || !q_tot || !q_lcl || !n_lcl || !q_rai || !n_rai || !dq_lcl || !dn_lcl || !dq_rai || !dn_rai)
        return CMX_ERR_BAD_ARG;
    const bool want_vel = vt_n || vt_m;
    int velk = decode_vel(flags, want_vel);
    if (velk < 0 || (want_vel && !vel)) return CMX_ERR_BAD_ARG;
    if (velk == VEL_CHEN) velk = chen_vel_kind<FT>(vel->chen2022);   // fitted Γ(b(ρ)+1), or the general instantiation (cmx_math.hpp ChenGamma)
    const bool limited = flags & CMX_SB2006_LIMITED;
    if ((flags & CMX_SB2006_LIMITED) && !sb_limiters_ok(*wr)) return CMX_ERR_BAD_ARG;      // clamp_ordered needs ordered limiter pairs
    const SbConsts<FT> c = make_sb_consts<FT>(*wr, *tps, vel, (double)Math<FT>::eps_1m());
    const bool intpow = sb_integer_exponents(*wr);        // b = 3, c = 4, d = −5: the integer-power instantiation (cmx_sb2006.hpp)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // Float64: one point per lane (8-byte loads).  The kernel is VALU-bound there and the two-points-per-lane variant spends 14 % of
    // its instructions on SGPR spill traffic (60 Float64 constants = 120 SGPRs): 1307 vs 1048 VALU instructions per point.
    constexpr int VEC = sizeof(FT) == 8 ? 1 : Math<FT>::VEC;
    // Alignment dispatch.  The 16-byte body needs every column at the same offset modulo 16 B (true for
    // any common slice [lo:hi] of aligned columns): `head` points are peeled so the body starts aligned,
    // the body runs VEC points per lane, the ragged tail one point per lane.  Columns with mixed
    // misalignment take the one-point-per-lane kernel throughout.  All three compute the identical
    // instruction sequence per point (-ffp-contract=off), so results do not depend on alignment.
    const void *ptrs[] = {rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, dq_lcl, dn_lcl, dq_rai, dn_rai, vt_n, vt_m};
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(rho) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (const void *p : ptrs)
        if (p) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(p) & 15u) == mis0);
    auto launch_range = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        SbIn<FT> in{rho + lo, T + lo, q_tot + lo, q_lcl + lo, n_lcl + lo, q_rai + lo, n_rai + lo};
        SbOut<FT> out{dq_lcl + lo, dn_lcl + lo, dq_rai + lo, dn_rai + lo, vt_n ? vt_n + lo : nullptr,
                      vt_m ? vt_m + lo : nullptr};
        launch_tendencies<FT, V>(limited, velk, intpow, c, in, out, count / V, s);
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

// cmx_sb2006_warm_rain_tendencies_fields_*: segmented (ClimaCore field) columns in, segmented columns or the reference's
// array-of-NamedTuples out — cmx_layout.hpp with Sb2006LayoutPolicy
template <typename FT, typename WR, typename TH>
static int32_t fields_entry(const WR *wr, const TH *tps, uint32_t flags, int64_t n_seg, int64_t seg_len, const FT *const *in,
                            const int64_t *in_stride, FT *const *out, const int64_t *out_stride, FT *aos, void *stream) {
    if (!wr || !tps || (flags & ~(uint32_t)CMX_SB2006_LIMITED)) return CMX_ERR_BAD_ARG;
    if ((flags & CMX_SB2006_LIMITED) && !sb_limiters_ok(*wr)) return CMX_ERR_BAD_ARG;      // clamp_ordered needs ordered limiter pairs
    const SbConsts<FT> c = make_sb_consts<FT>(*wr, *tps, (const std::conditional_t<std::is_same_v<FT, float>, cmx_rain_vel_f32, cmx_rain_vel_f64> *)nullptr,
                                              (double)Math<FT>::eps_1m());
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (sb_integer_exponents(*wr) && CMX_SB_INTPOW) {     // the integer-exponent instantiation of the point function (cmx_sb2006.hpp INTPOW)
        if (flags & CMX_SB2006_LIMITED) return launch_layout<FT, Sb2006LayoutPolicy<FT, true, true>>(c, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
        return launch_layout<FT, Sb2006LayoutPolicy<FT, false, true>>(c, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
    }
    if (flags & CMX_SB2006_LIMITED) return launch_layout<FT, Sb2006LayoutPolicy<FT, true>>(c, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
    return launch_layout<FT, Sb2006LayoutPolicy<FT, false>>(c, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
}

template <typename FT, typename WR, typename TH, typename VL>
static int32_t process_entry(const WR *wr, const TH *tps, const VL *vel, uint32_t flags, int64_t n, const FT *q_tot,
                             const FT *q_lcl, const FT *q_rai, const FT *N_lcl, const FT *N_rai, const FT *rho,
                             const FT *T, FT *const out[CMX_SB2006_NPROC], void *stream) {
    if (!wr || !tps || !out || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!q_tot || !q_lcl || !q_rai || !N_lcl || !N_rai || !rho || !T) return CMX_ERR_BAD_ARG;
    const bool want_vel = out[CMX_SB_RAI_VEL_N] || out[CMX_SB_RAI_VEL_M];
    int velk = decode_vel(flags, want_vel);
    if (velk < 0 || (want_vel && !vel)) return CMX_ERR_BAD_ARG;
    if (velk == VEL_CHEN) velk = chen_vel_kind<FT>(vel->chen2022);   // fitted Γ(b(ρ)+1), or the general instantiation (cmx_math.hpp ChenGamma)
    const bool limited = flags & CMX_SB2006_LIMITED;
    if ((flags & CMX_SB2006_LIMITED) && !sb_limiters_ok(*wr)) return CMX_ERR_BAD_ARG;      // clamp_ordered needs ordered limiter pairs
    const SbConsts<FT> c = make_sb_consts<FT>(*wr, *tps, vel, (double)Math<FT>::eps_1m());
    SbProcOut<FT> o;
    for (int k = 0; k < CMX_SB2006_NPROC; ++k) o.col[k] = out[k];
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_for(n);
#define CMX_LAUNCH(L, V)                                                                                        \
    CMX_LAUNCH_FRONT((sb2006_process_kernel<FT, L, V>), dim3(grid), dim3(kBlock), 0, s, c, q_tot, q_lcl, q_rai, \
                       N_lcl, N_rai, rho, T, o, n)
    if (limited) {
        if (velk == VEL_NONE) CMX_LAUNCH(true, VEL_NONE);
        else if (velk == VEL_SB) CMX_LAUNCH(true, VEL_SB);
        else if (velk == VEL_CHEN) CMX_LAUNCH(true, VEL_CHEN);
        else CMX_LAUNCH(true, VEL_CHEN_GEN);
    } else {
        if (velk == VEL_NONE) CMX_LAUNCH(false, VEL_NONE);
        else if (velk == VEL_SB) CMX_LAUNCH(false, VEL_SB);
        else if (velk == VEL_CHEN) CMX_LAUNCH(false, VEL_CHEN);
        else CMX_LAUNCH(false, VEL_CHEN_GEN);
    }
#undef CMX_LAUNCH
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// CM2.cloud_terminal_velocity — Microphysics2M.jl:647-664 (point function and constants: cmx_sb2006.hpp).  20 B/point (f32), one point per lane.
template <typename FT>
__global__ __launch_bounds__(kBlock) void sb2006_cloud_velocity_kernel(const CloudVelConsts<FT> c, const FT *__restrict__ q_liq,
                                                                       const FT *__restrict__ rho, const FT *__restrict__ N_liq,
                                                                       FT *__restrict__ vt_n, FT *__restrict__ vt_m, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    FT v0, v1;
    sb2006_cloud_velocity(c, q_liq[i], rho[i], N_liq[i], v0, v1);
    if (vt_n) vt_n[i] = v0;
    if (vt_m) vt_m[i] = v1;
}

template <typename FT, typename PDF, typename VEL>
static int32_t cloud_velocity_entry(const PDF *pdf, const VEL *vel, int64_t n, const FT *q_liq, const FT *rho, const FT *N_liq, FT *vt_n,
                                    FT *vt_m, void *stream) {
    if (!pdf || !vel || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!q_liq || !rho || !N_liq) return CMX_ERR_BAD_ARG;
    const CloudVelConsts<FT> c = make_cloud_vel_consts<FT>(*pdf, *vel);
    hipLaunchKernelGGL((sb2006_cloud_velocity_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c, q_liq, rho, N_liq, vt_n, vt_m, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// ---- bulk 2M cloud → rain conversion of KK2000 / B1994 / TC1980 / LD2004 — Microphysics2M.jl:920-1003 -------------------
// Four 1-line power laws; every power in the log2 domain on the hardware units; one point per lane (16–24 B/point).
template <typename FT> struct Bulk2mConsts {
    uint32_t scheme; FT eps_1m, eps_m;
    FT kk_l2A, kk_a, kk_b, kk_c, kk_accr_A, kk_accr_a, kk_accr_b;
    FT b_l2C, b_a, b_b, b_c, b_N_0, b_d_low, b_d_high, b_k, b_accr_A;
    FT t_a, t_b, t_D, t_thr_c, t_k, t_accr_A;                    // threshold = t_thr_c · N_d / ρ
    FT l_rvol_c, l_E_0, l_R_6C_0, l_k;                           // r_vol³ = l_rvol_c · q ρ / N_d  [µm³]
};
template <typename FT> __device__ __forceinline__ FT logistic_dev(FT x, FT x_0, FT k, FT eps) {   // Common.jl:125-139
    using M = Math<FT>;
    x = M::max(FT(0), x);
    const FT xs = M::max(x, eps), x0s = M::max(x_0, eps);
    const FT z = k * (xs * M::rcp(x0s) - x0s * M::rcp(xs));
    // σ(z) = 1/(1 + e^{−z}); e^{−z} may overflow to +Inf → σ = 0, the correct limit
    const FT r = M::rcp(FT(1) + M::exp2(-z * FT(1.4426950408889634)));
    return x < eps ? FT(0) : (x_0 < eps ? FT(1) : r);
}
template <typename FT>
__global__ __launch_bounds__(kBlock) void bulk_2m_cloud_to_rain_kernel(const Bulk2mConsts<FT> c, const FT *__restrict__ q_lcl,
                                                                       const FT *__restrict__ q_rai, const FT *__restrict__ rho,
                                                                       const FT *__restrict__ N_d, FT *__restrict__ acnv,
                                                                       FT *__restrict__ accr, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t sch = c.scheme & 0xffu;
    const bool smooth = c.scheme & CMX_2M_SMOOTH_TRANSITION;
    const FT ql_raw = q_lcl[i], r = rho[i];
    const FT ql = M::max(FT(0), ql_raw);
    if (acnv) {
        const FT Nd = N_d[i];
        const FT l2q = M::log2(ql), l2N = M::log2(Nd), l2r = M::log2(r);
        FT v;
        if (sch == CMX_2M_KK2000) {
            v = M::exp2(c.kk_l2A + c.kk_a * l2q + c.kk_b * l2N + c.kk_c * l2r);
        } else if (sch == CMX_2M_B1994) {
            FT d;
            if (smooth) {
                const FT f = logistic_dev<FT>(Nd, c.b_N_0, c.b_k, c.eps_1m);
                d = f * c.b_d_low + (FT(1) - f) * c.b_d_high;
            } else {
                d = Nd >= c.b_N_0 ? c.b_d_low : c.b_d_high;
            }
            v = M::exp2(c.b_l2C + c.b_a * M::log2(d) + c.b_b * (l2q + l2r) + c.b_c * l2N - l2r);
        } else if (sch == CMX_2M_TC1980) {
            const FT thr = c.t_thr_c * Nd * M::rcp(r);
            const FT out = smooth ? logistic_dev<FT>(ql, thr, c.t_k, c.eps_1m) : (ql - thr > FT(0) ? FT(1) : FT(0));
            v = c.t_D * M::exp2(c.t_a * l2q + c.t_b * l2N) * out;
        } else {   // LD2004 (:948-972): everything from r_vol [µm]
            const FT l2_rv3 = M::log2(c.l_rvol_c * ql_raw * r * M::rcp(Nd));
            const FT r_vol = M::exp2(l2_rv3 * FT(1.0 / 3.0));
            const FT l2_b6 = FT(1.0 / 3.0) * M::log2((r_vol + FT(3)) * M::rcp(r_vol));
            const FT R_6 = M::exp2(l2_b6) * r_vol;
            const FT l2_qr = M::log2(ql_raw * r);
            const FT R_6C = c.l_R_6C_0 * M::exp2(FT(-1.0 / 6.0) * l2_qr) * M::rsqrt(R_6);
            const FT out = smooth ? logistic_dev<FT>(R_6, R_6C, c.l_k, c.eps_1m) : (R_6 - R_6C > FT(0) ? FT(1) : FT(0));
            const FT val = c.l_E_0 * M::exp2(FT(6) * l2_b6 + FT(3) * l2_qr - l2N - l2r) * out;
            v = ql_raw <= c.eps_m ? FT(0) : val;
        }
        acnv[i] = v;
    }
    if (accr) {
        const FT qr = M::max(FT(0), q_rai[i]);
        FT v;
        if (sch == CMX_2M_KK2000) v = c.kk_accr_A * M::exp2(c.kk_accr_a * M::log2(ql * qr) + c.kk_accr_b * M::log2(r));
        else if (sch == CMX_2M_B1994) v = c.b_accr_A * ql * r * qr;
        else v = c.t_accr_A * ql * qr;
        accr[i] = v;
    }
}

template <typename FT, typename SC>
static int32_t bulk_2m_entry(const SC *p, uint32_t scheme, int64_t n, const FT *q_lcl, const FT *q_rai, const FT *rho, const FT *N_d,
                             FT *acnv, FT *accr, void *stream) {
    const uint32_t sch = scheme & 0xffu;
    if (!p || n < 0 || sch > CMX_2M_LD2004 || (scheme & ~(0xffu | CMX_2M_SMOOTH_TRANSITION))) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (sch == CMX_2M_LD2004 && accr) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!q_lcl || !rho || (acnv && !N_d) || (accr && !q_rai) || (!acnv && !accr)) return CMX_ERR_BAD_ARG;
    const double pi = 3.14159265358979323846;
    Bulk2mConsts<FT> c{};
    c.scheme = scheme; c.eps_1m = Math<FT>::eps_1m(); c.eps_m = Math<FT>::eps();
    c.kk_l2A = (FT)std::log2((double)p->kk2000.acnv_A); c.kk_a = p->kk2000.acnv_a; c.kk_b = p->kk2000.acnv_b; c.kk_c = p->kk2000.acnv_c;
    c.kk_accr_A = p->kk2000.accr_A; c.kk_accr_a = p->kk2000.accr_a; c.kk_accr_b = p->kk2000.accr_b;
    c.b_l2C = (FT)std::log2((double)p->b1994.acnv_C); c.b_a = p->b1994.acnv_a; c.b_b = p->b1994.acnv_b; c.b_c = p->b1994.acnv_c;
    c.b_N_0 = p->b1994.acnv_N_0; c.b_d_low = p->b1994.acnv_d_low; c.b_d_high = p->b1994.acnv_d_high; c.b_k = p->b1994.acnv_k;
    c.b_accr_A = p->b1994.accr_A;
    c.t_a = p->tc1980.acnv_a; c.t_b = p->tc1980.acnv_b; c.t_D = p->tc1980.acnv_D; c.t_k = p->tc1980.acnv_k; c.t_accr_A = p->tc1980.accr_A;
    c.t_thr_c = (FT)((double)p->tc1980.acnv_m0_liq_coeff * std::pow((double)p->tc1980.acnv_r_0, (double)p->tc1980.acnv_me_liq));
    c.l_rvol_c = (FT)(3.0 / 4.0 / pi / (double)p->ld2004.rho_w * 1e18);
    c.l_E_0 = p->ld2004.E_0; c.l_R_6C_0 = p->ld2004.R_6C_0; c.l_k = p->ld2004.k;
    hipLaunchKernelGGL((bulk_2m_cloud_to_rain_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c, q_lcl, q_rai, rho, N_d, acnv, accr, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_bulk_2m_cloud_to_rain_f32(const cmx_bulk_2m_schemes_f32 *schemes, uint32_t scheme, int64_t n, const float *q_lcl,
                                      const float *q_rai, const float *rho, const float *N_d, float *acnv, float *accr, void *stream) {
    return cmx::bulk_2m_entry<float>(schemes, scheme, n, q_lcl, q_rai, rho, N_d, acnv, accr, stream);
}
int32_t cmx_bulk_2m_cloud_to_rain_f64(const cmx_bulk_2m_schemes_f64 *schemes, uint32_t scheme, int64_t n, const double *q_lcl,
                                      const double *q_rai, const double *rho, const double *N_d, double *acnv, double *accr,
                                      void *stream) {
    return cmx::bulk_2m_entry<double>(schemes, scheme, n, q_lcl, q_rai, rho, N_d, acnv, accr, stream);
}

int32_t cmx_sb2006_cloud_terminal_velocity_f32(const cmx_cloud_pdf_sb2006_f32 *pdf_c, const cmx_stokes_vel_f32 *vel, int64_t n,
                                               const float *q_liq, const float *rho, const float *N_liq, float *vt_n, float *vt_m,
                                               void *stream) {
    return cmx::cloud_velocity_entry<float>(pdf_c, vel, n, q_liq, rho, N_liq, vt_n, vt_m, stream);
}
int32_t cmx_sb2006_cloud_terminal_velocity_f64(const cmx_cloud_pdf_sb2006_f64 *pdf_c, const cmx_stokes_vel_f64 *vel, int64_t n,
                                               const double *q_liq, const double *rho, const double *N_liq, double *vt_n, double *vt_m,
                                               void *stream) {
    return cmx::cloud_velocity_entry<double>(pdf_c, vel, n, q_liq, rho, N_liq, vt_n, vt_m, stream);
}

int32_t cmx_sb2006_warm_rain_tendencies_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps,
                                            const cmx_rain_vel_f32 *vel, uint32_t flags, int64_t n, const float *rho,
                                            const float *T, const float *q_tot, const float *q_lcl, const float *n_lcl,
                                            const float *q_rai, const float *n_rai, float *dq_lcl_dt, float *dn_lcl_dt,
                                            float *dq_rai_dt, float *dn_rai_dt, float *vt_rai_n, float *vt_rai_m,
                                            void *stream) {
    return cmx::tendencies_entry<float>(warm_rain, tps, vel, flags, n, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai,
                                        dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt, vt_rai_n, vt_rai_m, stream);
}

int32_t cmx_sb2006_warm_rain_tendencies_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps,
                                            const cmx_rain_vel_f64 *vel, uint32_t flags, int64_t n, const double *rho,
                                            const double *T, const double *q_tot, const double *q_lcl,
                                            const double *n_lcl, const double *q_rai, const double *n_rai,
                                            double *dq_lcl_dt, double *dn_lcl_dt, double *dq_rai_dt, double *dn_rai_dt,
                                            double *vt_rai_n, double *vt_rai_m, void *stream) {
    return cmx::tendencies_entry<double>(warm_rain, tps, vel, flags, n, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai,
                                         dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt, vt_rai_n, vt_rai_m, stream);
}

int32_t cmx_sb2006_process_rates_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps,
                                     const cmx_rain_vel_f32 *vel, uint32_t flags, int64_t n, const float *q_tot,
                                     const float *q_lcl, const float *q_rai, const float *N_lcl, const float *N_rai,
                                     const float *rho, const float *T, float *const out[CMX_SB2006_NPROC],
                                     void *stream) {
    return cmx::process_entry<float>(warm_rain, tps, vel, flags, n, q_tot, q_lcl, q_rai, N_lcl, N_rai, rho, T, out,
                                     stream);
}

int32_t cmx_sb2006_process_rates_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps,
                                     const cmx_rain_vel_f64 *vel, uint32_t flags, int64_t n, const double *q_tot,
                                     const double *q_lcl, const double *q_rai, const double *N_lcl,
                                     const double *N_rai, const double *rho, const double *T,
                                     double *const out[CMX_SB2006_NPROC], void *stream) {
    return cmx::process_entry<double>(warm_rain, tps, vel, flags, n, q_tot, q_lcl, q_rai, N_lcl, N_rai, rho, T, out,
                                      stream);
}

int32_t cmx_sb2006_warm_rain_tendencies_fields_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n_seg,
                                                   int64_t seg_len, const float *const *in, const int64_t *in_seg_stride, float *const *out,
                                                   const int64_t *out_seg_stride, float *out_aos, void *stream) {
    return cmx::fields_entry<float>(warm_rain, tps, flags, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, out_aos, stream);
}
int32_t cmx_sb2006_warm_rain_tendencies_fields_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n_seg,
                                                   int64_t seg_len, const double *const *in, const int64_t *in_seg_stride, double *const *out,
                                                   const int64_t *out_seg_stride, double *out_aos, void *stream) {
    return cmx::fields_entry<double>(warm_rain, tps, flags, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, out_aos, stream);
}

}  // extern "C"
