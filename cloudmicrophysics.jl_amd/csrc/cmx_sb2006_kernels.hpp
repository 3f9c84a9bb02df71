// cmx_sb2006_kernels.hpp — the fused SB2006 warm-rain device kernels (templates), shared by the C-ABI
// translation unit (cmx_sb2006_kernels.hip) and the roofline probe (tools/sb2006_probe.hip).
//
// Kernel shape: pointwise map over structure-of-arrays state columns, HBM-bound.
// One lane owns VEC consecutive points (VEC·sizeof(FT) = 16 B), so each of the 7 input columns is
// read with one global_load_dwordx4 per lane (1 KiB per wave-instruction, fully coalesced) and each
// of the 4–6 output columns written with one global_store_dwordx4; loads and stores carry the
// non-temporal hint (every byte is touched exactly once).  128-thread workgroups (kTendBS), one short-lived workgroup per
// tile (non-persistent, see below).  No LDS, no MFMA: there is no data reuse and no contraction on this path.
#pragma once
#include <hip/hip_runtime.h>

#include "cmx_launch.hpp"
#include "cmx_layout.hpp"
#include "cmx_sb2006.hpp"

#ifndef CMX_POINT_FENCE
#define CMX_POINT_FENCE 0      // A/B switch (round 2): a scheduling fence between the points of a lane did not pay (0.89 vs 0.86 ms)
#endif

namespace cmx {

template <typename FT> struct SbIn { const FT *rho, *T, *q_tot, *q_lcl, *n_lcl, *q_rai, *n_rai; };
template <typename FT> struct SbOut { FT *dq_lcl, *dn_lcl, *dq_rai, *dn_rai, *vt_n, *vt_m; };
template <typename FT> struct SbProcOut { FT *col[CMX_SB2006_NPROC]; };

// bulk_microphysics_tendencies(::Microphysics2Moment, …) over columns — BMT:820-854 + :707-782
//
// Launch shape: NON-persistent — workgroup b owns C consecutive tiles of BS lanes × VEC points, issues all of
// its 7·C vector loads up front, computes, stores, retires.  Measured on MI355X (tools/sb2006_probe, 1e8 f32
// points, 13 concurrent HBM streams): a grid-stride loop over CUs×k resident workgroups keeps every wave of the
// chip in the same load→compute→store phase and sustains only ≈59 % of the 8 TB/s peak, the same bytes moved by
// short-lived workgroups ≈70–73 % (new waves start loading while older ones store; profiles/r01_probe.txt).
// Float64 is VALU-bound (≈ 800 dependent instructions per point at 4 waves per SIMD).  A software-pipelined shape for it — a workgroup
// walking 4 or 8 tiles in a loop, tile t+1's loads issued before tile t is computed, the table copy paid once — was measured in round
// 2 and rejected: the loop carries the prefetched columns and hoisted invariants, 167–198 VGPRs against 125, 2–3 waves per SIMD
// instead of 4, 3.50 ms against 3.06 (tools/valu_probe: a dependent v_fma_f64 issues every 11 cycles; what hides that is waves).
// CMX_SB_WAVES (A/B switch, default 1 = no request): waves per SIMD the register allocator is asked to keep for the instantiations without a Chen-2022
// velocity (they sit at 124–136 VGPRs, on either side of the 128-register step between four and three waves; the Chen variants need 200+ and are left alone)
#ifndef CMX_SB_WAVES
#define CMX_SB_WAVES 1
#endif
template <typename FT, bool LIMITED, int VEL, int VEC, int BS = kBlock, int C = 1, bool NT = true, bool INTPOW = false>
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu((VEL <= VEL_SB && VEC * (int)sizeof(FT) <= 16 && C == 1) ? CMX_SB_WAVES : 1)))
void sb2006_tendencies_kernel(const SbConsts<FT> c, const SbIn<FT> in,
                                                               const SbOut<FT> out, const int64_t nvec) {
    const int64_t base = ((int64_t)blockIdx.x * C) * BS + threadIdx.x;
    FT rho[C][VEC], T[C][VEC], q_tot[C][VEC], q_lcl[C][VEC], n_lcl[C][VEC], q_rai[C][VEC], n_rai[C][VEC];
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int64_t i = base + (int64_t)t * BS;
        if (i < nvec) {
            load_col<FT, VEC, NT>(in.rho, i, rho[t]);
            load_col<FT, VEC, NT>(in.T, i, T[t]);
            load_col<FT, VEC, NT>(in.q_tot, i, q_tot[t]);
            load_col<FT, VEC, NT>(in.q_lcl, i, q_lcl[t]);
            load_col<FT, VEC, NT>(in.n_lcl, i, n_lcl[t]);
            load_col<FT, VEC, NT>(in.q_rai, i, q_rai[t]);
            load_col<FT, VEC, NT>(in.n_rai, i, n_rai[t]);
        }
    }
    // Float64: exp2 / log2 tables → LDS while the loads fly (every lane of the workgroup reaches the barrier inside; same-box A/B
    // 3.18 → 3.06 ms against the copy in front of the loads); no-op for Float32
    Math<FT>::prepare();
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int64_t i = base + (int64_t)t * BS;
        if (i >= nvec) continue;
        FT dq_lcl[VEC], dn_lcl[VEC], dq_rai[VEC], dn_rai[VEC], vt_n[VEC], vt_m[VEC];
        // one value of the value type VT at a time: a point, or — Float32 with four points per lane — a PAIR of points in packed arithmetic (cmx_math.hpp f32x2)
        // Packed for the Chen-2022 instantiation only (same-box A/B, round 5, profiles/r05_ab_sessions.txt session 3, ms per 1e8 points: Chen 0.955 → 0.93 packed,
        // 0.922 with the constants left in SGPRs; the SB2006-velocity instantiation — the north star — is bound by its 13 HBM streams, not by issue: 0.856 one point
        // at a time, 0.862 packed at four waves per SIMD, 0.93 packed at three).  Not the general Chen instantiation: its run-time Γ is an OCML call per lane.
        constexpr int L = (sizeof(FT) == 4 && VEC % 2 == 0 && CMX_F32_PACKED && (VEL == VEL_CHEN || CMX_F32_PACKED > 1)) ? 2 : 1;
        if constexpr (L == 1) {      // one point at a time — rounds 1–4 verbatim (routing it through the generic form below costs the compiler 20–60 registers)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                // clamp_to_nonneg — BMT:828-837 (T is not clamped)
                const FT r_ = max0(rho[t][k]);
                const FT qt = max0(q_tot[t][k]);
                const FT ql = max0(q_lcl[t][k]);
                const FT qr = max0(q_rai[t][k]);
                const FT nl = max0(n_lcl[t][k]);
                const FT nr = max0(n_rai[t][k]);
                // a NaN in any input column poisons every output of the point (cmx_math.hpp any_nan)
                const bool poisoned = any_nan(rho[t][k], q_tot[t][k], q_lcl[t][k], n_lcl[t][k], q_rai[t][k], n_rai[t][k], T[t][k]);
                // N = ρ n — BMT:718-719
                const SbRates<FT> p = sb2006_point<FT, LIMITED, VEL, false, INTPOW>(front_consts<FT>(c), r_, T[t][k], qt, ql, qr, r_ * nl, r_ * nr, nl, nr);
                // sums of warm_rain_tendencies_2m — BMT:738-779.  The per-m³ number rates are added first and divided by ρ once (the
                // reference divides each term: same value to rounding, five multiplies fewer); autoconversion's −2·dN_rai cancels
                // against the same term inside cloud self-collection (CM2:499), so their sum is formed directly.
                dq_lcl[k] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
                dn_lcl[k] = Math<FT>::fma(p.lsc_plus_au + p.ac_dN_lcl, p.inv_rho, p.na_lcl);
                dq_rai[k] = (p.evq + p.au_dq_rai) + p.ac_dq_rai;
                dn_rai[k] = Math<FT>::fma(((p.evN + p.au_dN_rai) + p.rsc) + p.rbr, p.inv_rho, p.na_rai);
                vt_n[k] = p.vt_n;
                vt_m[k] = p.vt_m;
                // branch-free: x + NaN = NaN, x + 0 = x (a −0 result becomes +0, the same in every variant of this kernel)
                const FT poison = poisoned ? Math<FT>::nan() : FT(0);
                dq_lcl[k] += poison; dn_lcl[k] += poison; dq_rai[k] += poison; dn_rai[k] += poison; vt_n[k] += poison; vt_m[k] += poison;
#if CMX_POINT_FENCE
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        } else {
            using VT = std::conditional_t<L == 2, f32x2, FT>;
            using MV = Math<VT>;
#pragma unroll
            for (int k = 0; k < VEC; k += L) {
                auto val = [k](const FT (&a)[VEC]) -> VT {
                    if constexpr (L == 2) return VT{a[k], a[k + 1]};
                    else return a[k];
                };
                const VT rho_k = val(rho[t]), T_k = val(T[t]), qt_k = val(q_tot[t]), ql_k = val(q_lcl[t]), nl_k = val(n_lcl[t]), qr_k = val(q_rai[t]), nr_k = val(n_rai[t]);
                // clamp_to_nonneg — BMT:828-837 (T is not clamped)
                const VT r_ = max0(rho_k);
                const VT qt = max0(qt_k);
                const VT ql = max0(ql_k);
                const VT qr = max0(qr_k);
                const VT nl = max0(nl_k);
                const VT nr = max0(nr_k);
                // a NaN in any input column poisons every output of the point (cmx_math.hpp any_nan)
                const typename MV::Mask poisoned = nan_mask(rho_k, qt_k, ql_k, nl_k, qr_k, nr_k, T_k);
                // N = ρ n — BMT:718-719
                const SbRates<VT> p = sb2006_point<VT, LIMITED, VEL, false, INTPOW>(front_consts<FT, (L == 2 && CMX_F32_PACKED_PHASE_CONSTS > 1)>(c), r_, T_k, qt, ql, qr, r_ * nl, r_ * nr, nl, nr);
                // sums of warm_rain_tendencies_2m — BMT:738-779.  The per-m³ number rates are added first and divided by ρ once (the
                // reference divides each term: same value to rounding, five multiplies fewer); autoconversion's −2·dN_rai cancels
                // against the same term inside cloud self-collection (CM2:499), so their sum is formed directly.
                VT o[6];
                o[0] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
                o[1] = MV::fma(p.lsc_plus_au + p.ac_dN_lcl, p.inv_rho, p.na_lcl);
                o[2] = (p.evq + p.au_dq_rai) + p.ac_dq_rai;
                o[3] = MV::fma(((p.evN + p.au_dN_rai) + p.rsc) + p.rbr, p.inv_rho, p.na_rai);
                o[4] = p.vt_n;
                o[5] = p.vt_m;
                // branch-free: x + NaN = NaN, x + 0 = x (a −0 result becomes +0, the same in every variant of this kernel)
                const VT poison = poisoned ? MV::nan() : VT(0);
#pragma unroll
                for (int q = 0; q < 6; ++q) o[q] += poison;
                auto put = [k](FT (&a)[VEC], VT x) {
                    if constexpr (L == 2) { a[k] = x.x; a[k + 1] = x.y; }
                    else a[k] = x;
                };
                put(dq_lcl, o[0]); put(dn_lcl, o[1]); put(dq_rai, o[2]); put(dn_rai, o[3]); put(vt_n, o[4]); put(vt_m, o[5]);
#if CMX_POINT_FENCE
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
        store_col<FT, VEC, NT>(out.dq_lcl, i, dq_lcl);
        store_col<FT, VEC, NT>(out.dn_lcl, i, dn_lcl);
        store_col<FT, VEC, NT>(out.dq_rai, i, dq_rai);
        store_col<FT, VEC, NT>(out.dn_rai, i, dn_rai);
        if constexpr (VEL != VEL_NONE) {
            if (out.vt_n) store_col<FT, VEC, NT>(out.vt_n, i, vt_n);
            if (out.vt_m) store_col<FT, VEC, NT>(out.vt_m, i, vt_m);
        }
    }
}

// ---- host-model layouts: the VEL_NONE tendencies as a policy of the generic adapter kernel (cmx_layout.hpp) ---
template <typename FT, bool LIMITED, bool INTPOW = false> struct Sb2006LayoutPolicy {
    static constexpr int NIN = 7, NOUT = 4, NAOS = 8;       // rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai → dq_lcl, dn_lcl, dq_rai, dn_rai (+4 zero fields)
    using Consts = SbConsts<FT>;
    static constexpr bool PACKABLE = true;      // point() also takes the packed pair type (cmx_layout.hpp evaluates two points per call then)
    template <typename C, typename VT> static __device__ __forceinline__ void point(const C &c, const VT (&x)[NIN], VT (&y)[NOUT]) {
        using M = Math<VT>;
        const VT r_ = max0(x[0]), qt = max0(x[2]), ql = max0(x[3]);
        const VT nl = max0(x[4]), qr = max0(x[5]), nr = max0(x[6]);
        const SbRates<VT> p = sb2006_point<VT, LIMITED, VEL_NONE, false, INTPOW>(c, r_, x[1], qt, ql, qr, r_ * nl, r_ * nr, nl, nr);
        y[0] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
        y[1] = M::fma(p.lsc_plus_au + p.ac_dN_lcl, p.inv_rho, p.na_lcl);
        y[2] = (p.evq + p.au_dq_rai) + p.ac_dq_rai;
        y[3] = M::fma(((p.evN + p.au_dN_rai) + p.rsc) + p.rbr, p.inv_rho, p.na_rai);
        const VT poison = nan_mask(x[0], x[2], x[3], x[4], x[5], x[6], x[1]) ? M::nan() : VT(0);
        y[0] += poison; y[1] += poison; y[2] += poison; y[3] += poison;
    }
};

// SB2006_2M_kernel (test/gpu_tests.jl:220-235): the individual process rates, N per m³, no clamping
template <typename FT, bool LIMITED, int VEL>
__global__ __launch_bounds__(kBlock) void sb2006_process_kernel(const SbConsts<FT> c, const FT *__restrict__ q_tot,
                                                                const FT *__restrict__ q_lcl, const FT *__restrict__ q_rai,
                                                                const FT *__restrict__ N_lcl, const FT *__restrict__ N_rai,
                                                                const FT *__restrict__ rho, const FT *__restrict__ T,
                                                                const SbProcOut<FT> out, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using M = Math<FT>;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const FT r_ = rho[i];
        const FT inv_r = M::rcp(r_);
        const SbRates<FT> p = sb2006_point<FT, LIMITED, VEL>(front_consts<FT>(c), r_, T[i], q_tot[i], q_lcl[i], q_rai[i], N_lcl[i],
                                                             N_rai[i], N_lcl[i] * inv_r, N_rai[i] * inv_r);
        // NaN in → NaN out (cmx_math.hpp any_nan): the max(x, ϵ) floors of the process functions would hide it
        const FT poison = any_nan(q_tot[i], q_lcl[i], q_rai[i], N_lcl[i], N_rai[i], r_, T[i]) ? M::nan() : FT(0);
#define CMX_PUT(colid, v) if (out.col[colid]) out.col[colid][i] = (v) + poison
        CMX_PUT(CMX_SB_ACNV_DQ_LCL, p.au_dq_lcl);
        CMX_PUT(CMX_SB_ACNV_DN_LCL, p.au_dN_lcl);
        CMX_PUT(CMX_SB_ACNV_DQ_RAI, p.au_dq_rai);
        CMX_PUT(CMX_SB_ACNV_DN_RAI, p.au_dN_rai);
        CMX_PUT(CMX_SB_LCL_SELFCOL, p.lsc);
        CMX_PUT(CMX_SB_ACCR_DQ_LCL, p.ac_dq_lcl);
        CMX_PUT(CMX_SB_ACCR_DN_LCL, p.ac_dN_lcl);
        CMX_PUT(CMX_SB_ACCR_DQ_RAI, p.ac_dq_rai);
        CMX_PUT(CMX_SB_RAI_SELFCOL, p.rsc);
        CMX_PUT(CMX_SB_RAI_BREAKUP, p.rbr);
        CMX_PUT(CMX_SB_RAI_VEL_N, p.vt_n);
        CMX_PUT(CMX_SB_RAI_VEL_M, p.vt_m);
        CMX_PUT(CMX_SB_EVAP_DN_RAI, p.evN);
        CMX_PUT(CMX_SB_EVAP_DQ_RAI, p.evq);
        CMX_PUT(CMX_SB_NUMADJ_RAI, p.na_rai);
        CMX_PUT(CMX_SB_NUMADJ_LCL, p.na_lcl);
        CMX_PUT(CMX_SB_CONDEVAP, p.cond);
        // ∂rain_evaporation_∂N_rai_∂q_rai — CM2:844-855: the evaporation tendencies over N_rai / q_rai above ϵ, 0 otherwise
        CMX_PUT(CMX_SB_DEVAP_DN_RAI, N_rai[i] > M::eps() ? p.evN * M::rcp(N_rai[i]) : FT(0));
        CMX_PUT(CMX_SB_DEVAP_DQ_RAI, q_rai[i] > M::eps() ? p.evq * M::rcp(q_rai[i]) : FT(0));
#undef CMX_PUT
    }
}

}  // namespace cmx
