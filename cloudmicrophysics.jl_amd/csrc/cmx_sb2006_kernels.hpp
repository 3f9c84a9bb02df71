// cmx_sb2006_kernels.hpp — the fused SB2006 warm-rain device kernels (templates), shared by the C-ABI
// translation unit (cmx_sb2006_kernels.hip) and the roofline probe (tools/sb2006_probe.hip).
//
// Kernel shape: pointwise map over structure-of-arrays state columns, HBM-bound.
// One lane owns VEC consecutive points (VEC·sizeof(FT) = 16 B), so each of the 7 input columns is
// read with one global_load_dwordx4 per lane (1 KiB per wave-instruction, fully coalesced) and each
// of the 4–6 output columns written with one global_store_dwordx4; loads and stores carry the
// non-temporal hint (every byte is touched exactly once).  256-thread workgroups, grid-stride over
// CUs × k workgroups.  No LDS, no MFMA: there is no data reuse and no contraction on this path.
#pragma once
#include <hip/hip_runtime.h>

#include "cmx_launch.hpp"
#include "cmx_sb2006.hpp"

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
template <typename FT, bool LIMITED, int VEL, int VEC, int BS = kBlock, int C = 1, bool NT = true>
__global__ __launch_bounds__(BS) void sb2006_tendencies_kernel(const SbConsts<FT> c, const SbIn<FT> in,
                                                               const SbOut<FT> out, const int64_t nvec) {
    using M = Math<FT>;
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
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int64_t i = base + (int64_t)t * BS;
        if (i >= nvec) continue;
        FT dq_lcl[VEC], dn_lcl[VEC], dq_rai[VEC], dn_rai[VEC], vt_n[VEC], vt_m[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            // clamp_to_nonneg — BMT:828-837 (T is not clamped)
            const FT r_ = M::max(FT(0), rho[t][k]);
            const FT qt = M::max(FT(0), q_tot[t][k]);
            const FT ql = M::max(FT(0), q_lcl[t][k]);
            const FT qr = M::max(FT(0), q_rai[t][k]);
            const FT nl = M::max(FT(0), n_lcl[t][k]);
            const FT nr = M::max(FT(0), n_rai[t][k]);
            // N = ρ n — BMT:718-719
            const SbRates<FT> p = sb2006_point<FT, LIMITED, VEL>(c, r_, T[t][k], qt, ql, qr, r_ * nl, r_ * nr, nl, nr);
            // sums of warm_rain_tendencies_2m — BMT:738-779.  The per-m³ number rates are added first and divided by ρ once (the
            // reference divides each term: same value to rounding, five multiplies fewer); autoconversion's −2·dN_rai cancels
            // against the same term inside cloud self-collection (CM2:499), so their sum is formed directly.
            dq_lcl[k] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
            dn_lcl[k] = M::fma(p.lsc_plus_au + p.ac_dN_lcl, p.inv_rho, p.na_lcl);
            dq_rai[k] = (p.evq + p.au_dq_rai) + p.ac_dq_rai;
            dn_rai[k] = M::fma(((p.evN + p.au_dN_rai) + p.rsc) + p.rbr, p.inv_rho, p.na_rai);
            vt_n[k] = p.vt_n;
            vt_m[k] = p.vt_m;
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

// ---- host-model layouts -------------------------------------------------------------------------------
// The same per-point arithmetic as sb2006_tendencies_kernel (VEL_NONE) behind two layout adapters:
//   * SEGMENTED columns: column k is n_seg runs of seg_len contiguous elements, run s starting at base_k + s·stride_k.
//     That is a ClimaCore field in its storage: a DataLayouts.VIJFH array (Nv, Ni, Nj, Nf, Nh) holds component f of
//     element h as the contiguous run [v + Nv (i + Ni j)] of length Nv·Ni·Nj at offset Nv·Ni·Nj·(f + Nf h); VF columns
//     and VIJHF are the single-run case.  Every column has its own stride (state and tendency fields differ in Nf).
//   * AoS output: the reference's result type, an array of 8-field NamedTuples (dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt,
//     dq_ice_dt, dq_rim_dt, db_rim_dt, dn_lcl_activation_dt — BMT:852-853; test/gpu_performance.jl:212-216), the last
//     four identically zero.  A lane owns VEC consecutive points = 128 B of the output; written directly that is 8
//     store instructions of 16 B at a 128-B lane stride (64 partial cache lines each).  The tile goes through LDS
//     instead and leaves as 8 fully coalesced 1-KiB-per-wave stores.
template <typename FT> struct SbLayoutIO {
    const FT *in[7];          // rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai
    int64_t in_stride[7];
    FT *out[4];               // dq_lcl, dn_lcl, dq_rai, dn_rai (SoA mode)
    int64_t out_stride[4];
    FT *aos;                  // n × 8 (AoS mode)
    int64_t seg_len;
    double inv_seg_len;
};
template <typename FT, bool LIMITED, int VEC, bool SEG, bool AOS, int BS = kBlock>
__global__ __launch_bounds__(BS) void sb2006_tendencies_layout_kernel(const SbConsts<FT> c, const SbLayoutIO<FT> io, const int64_t nvec) {
    using M = Math<FT>;
    constexpr int CH = 16 / (int)sizeof(FT);                  // elements per 16-byte chunk
    constexpr int ROW = VEC * 8 + CH;                         // LDS row of one lane (+1 chunk of padding against bank conflicts)
    const int64_t tile0 = (int64_t)blockIdx.x * BS;
    const int64_t i = tile0 + threadIdx.x;
    const bool active = i < nvec;
    FT dq_lcl[VEC], dn_lcl[VEC], dq_rai[VEC], dn_rai[VEC];
    int64_t seg = 0, off = i * VEC;                           // element index → (run, offset in run)
    if constexpr (SEG) {
        const int64_t e = i * VEC;
        seg = (int64_t)((double)e * io.inv_seg_len);
        off = e - seg * io.seg_len;
        if (off < 0) { --seg; off += io.seg_len; }
        if (off >= io.seg_len) { ++seg; off -= io.seg_len; }
    }
    if (active) {
        FT x[7][VEC];
#pragma unroll
        for (int k = 0; k < 7; ++k) load_col<FT, VEC, true>(io.in[k] + (SEG ? seg * io.in_stride[k] : 0), off / VEC, x[k]);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const FT r_ = M::max(FT(0), x[0][k]), qt = M::max(FT(0), x[2][k]), ql = M::max(FT(0), x[3][k]);
            const FT nl = M::max(FT(0), x[4][k]), qr = M::max(FT(0), x[5][k]), nr = M::max(FT(0), x[6][k]);
            const SbRates<FT> p = sb2006_point<FT, LIMITED, VEL_NONE>(c, r_, x[1][k], qt, ql, qr, r_ * nl, r_ * nr, nl, nr);
            dq_lcl[k] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
            dn_lcl[k] = M::fma(p.lsc_plus_au + p.ac_dN_lcl, p.inv_rho, p.na_lcl);
            dq_rai[k] = (p.evq + p.au_dq_rai) + p.ac_dq_rai;
            dn_rai[k] = M::fma(((p.evN + p.au_dN_rai) + p.rsc) + p.rbr, p.inv_rho, p.na_rai);
        }
    }
    if constexpr (!AOS) {
        if (!active) return;
        store_col<FT, VEC, true>(io.out[0] + (SEG ? seg * io.out_stride[0] : 0), off / VEC, dq_lcl);
        store_col<FT, VEC, true>(io.out[1] + (SEG ? seg * io.out_stride[1] : 0), off / VEC, dn_lcl);
        store_col<FT, VEC, true>(io.out[2] + (SEG ? seg * io.out_stride[2] : 0), off / VEC, dq_rai);
        store_col<FT, VEC, true>(io.out[3] + (SEG ? seg * io.out_stride[3] : 0), off / VEC, dn_rai);
    } else {
        extern __shared__ __align__(16) unsigned char lds_raw[];
        FT *lds = reinterpret_cast<FT *>(lds_raw);
        using V16 = typename VecT<FT, CH>::type;
        if (active) {
            FT *row = lds + threadIdx.x * ROW;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                row[k * 8 + 0] = dq_lcl[k]; row[k * 8 + 1] = dn_lcl[k]; row[k * 8 + 2] = dq_rai[k]; row[k * 8 + 3] = dn_rai[k];
                row[k * 8 + 4] = FT(0); row[k * 8 + 5] = FT(0); row[k * 8 + 6] = FT(0); row[k * 8 + 7] = FT(0);
            }
        }
        __syncthreads();
        constexpr int CPL = VEC * 8 / CH;                     // 16-byte chunks per lane
        const int64_t nvalid = nvec - tile0 < BS ? nvec - tile0 : BS;
        V16 *dst = reinterpret_cast<V16 *>(io.aos) + tile0 * CPL;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int t = j * BS + threadIdx.x;                // chunk of the tile, in output order
            const int src = t / CPL, sub = t % CPL;
            if (src < nvalid) __builtin_nontemporal_store(*reinterpret_cast<const V16 *>(lds + src * ROW + sub * CH), dst + t);
        }
    }
}

// SB2006_2M_kernel (test/gpu_tests.jl:220-235): the individual process rates, N per m³, no clamping
template <typename FT, bool LIMITED, int VEL>
__global__ __launch_bounds__(kBlock) void sb2006_process_kernel(const SbConsts<FT> c, const FT *__restrict__ q_tot,
                                                                const FT *__restrict__ q_lcl, const FT *__restrict__ q_rai,
                                                                const FT *__restrict__ N_lcl, const FT *__restrict__ N_rai,
                                                                const FT *__restrict__ rho, const FT *__restrict__ T,
                                                                const SbProcOut<FT> out, const int64_t n) {
    using M = Math<FT>;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const FT r_ = rho[i];
        const FT inv_r = M::rcp(r_);
        const SbRates<FT> p = sb2006_point<FT, LIMITED, VEL>(c, r_, T[i], q_tot[i], q_lcl[i], q_rai[i], N_lcl[i],
                                                             N_rai[i], N_lcl[i] * inv_r, N_rai[i] * inv_r);
#define CMX_PUT(colid, v) if (out.col[colid]) out.col[colid][i] = (v)
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
#undef CMX_PUT
    }
}

}  // namespace cmx
