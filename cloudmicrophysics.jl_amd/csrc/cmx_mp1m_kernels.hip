// cmx_mp1m_kernels.hip — one-moment (Marshall–Palmer) bulk scheme, fused per point, for gfx950; C-ABI entry
// points of include/cmx.h §(5).
//
// Reference (src = /root/reference/src): BulkMicrophysicsTendencies.jl:141-252 (`_microphysics_source_terms`,
// `_aggregate_tendencies`), Microphysics1M.jl (CM1), MicrophysicsNonEq.jl:104-193 (NonEq), Common.jl:47-102,
// 157-173.  13 option-dispatched processes, 18 source terms, 4 tendencies.
//
// HBM-bound pointwise map: 7 state columns in, 4 tendency columns out = 44 B/point (f32).  Same launch shape as
// the SB2006 kernel (one 16-byte vector per lane, one short-lived 256-lane workgroup per tile, non-temporal
// accesses).  Per point: the two saturation pressures, the three Marshall–Palmer slope parameters λ⁻¹ and the
// rain v0 are evaluated ONCE (the reference recomputes them per process through `size_distr_parameters` at
// best once, p_sat six times); every power of λ⁻¹ comes from its one log2; parameter-only factors (Γ terms,
// a0·χa·χv·E…, r0 powers) are folded on the host in double.  The reference's Microphysics1MOptions arrive as a
// flags word in an SGPR: disabled processes are skipped by wave-uniform branches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "cmx_launch.hpp"
#include "cmx_layout.hpp"
#include "cmx_math.hpp"
#include "cmx_math.hpp"
#include "cmx_mp1m.hpp"
#include "cmx_mp1m_vel.hpp"

namespace cmx {


// bulk_microphysics_tendencies(Instantaneous(), Microphysics1Moment(), …) over columns — BMT:505-514
// Lanes per workgroup of the tendencies kernel, per float type (same-box A/Bs, round 4, ms per 1e8 points: Float32 256 lanes 0.804, 128 lanes
// 0.758 — and on another box 128 lanes 0.723, 64 lanes 0.714; Float64 256 lanes 2.375, 128 lanes 2.396 — profiles/r04_ab_sessions.txt, sessions
// 13-15): short-lived one-wave workgroups free their slots as soon as their own wave has stored.  -DCMX_1M_BLOCK=n forces one size for both.
#ifdef CMX_1M_BLOCK
template <typename FT> constexpr int kBlock1m = CMX_1M_BLOCK;
#else
template <typename FT> constexpr int kBlock1m = sizeof(FT) == 4 ? 64 : 256;
#endif
#ifndef CMX_1M_PACKED
#define CMX_1M_PACKED 1      // A/B switch: 0 = one point at a time (rounds 1–4)
#endif
template <typename FT, int VEC, uint32_t FLAGS = kRuntimeFlags>
__global__ __launch_bounds__(kBlock1m<FT>) void mp1m_tendencies_kernel(const Mp1mConsts<FT> c, const Mp1mIn<FT> in,
                                                                       const Mp1mOut<FT> out, const int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock1m<FT> + threadIdx.x;
    FT rho[VEC], T[VEC], q_tot[VEC], q_lcl[VEC], q_icl[VEC], q_rai[VEC], q_sno[VEC];
    if (i < nvec) {
        load_col<FT, VEC>(in.rho, i, rho); load_col<FT, VEC>(in.T, i, T); load_col<FT, VEC>(in.q_tot, i, q_tot);
        load_col<FT, VEC>(in.q_lcl, i, q_lcl); load_col<FT, VEC>(in.q_icl, i, q_icl); load_col<FT, VEC>(in.q_rai, i, q_rai);
        load_col<FT, VEC>(in.q_sno, i, q_sno);
    }
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly (every lane of the workgroup reaches the barrier inside); no-op for Float32
    if (i >= nvec) return;
    FT dl[VEC], di[VEC], dr[VEC], ds[VEC];
    if constexpr (CMX_1M_PACKED && sizeof(FT) == 4 && VEC == 4) {
        // Float32, four points per lane: two PAIRS of points in packed arithmetic (cmx_math.hpp f32x2); the constants are read phase by phase through the
        // kernel-argument pointer (a packed instruction takes no literal: the point function's literals occupy SGPRs next to the constants)
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
            f32x2 l, ic, r, sn;
            mp1m_tendencies_point<f32x2, FLAGS>(front_consts<FT, true>(c), f32x2{rho[k], rho[k + 1]}, f32x2{T[k], T[k + 1]}, f32x2{q_tot[k], q_tot[k + 1]},
                                                f32x2{q_lcl[k], q_lcl[k + 1]}, f32x2{q_icl[k], q_icl[k + 1]}, f32x2{q_rai[k], q_rai[k + 1]},
                                                f32x2{q_sno[k], q_sno[k + 1]}, l, ic, r, sn);
            dl[k] = l.x; dl[k + 1] = l.y; di[k] = ic.x; di[k + 1] = ic.y; dr[k] = r.x; dr[k + 1] = r.y; ds[k] = sn.x; ds[k + 1] = sn.y;
        }
    } else if constexpr (CMX_1M_HOIST && sizeof(FT) == 4 && VEC == 4 && FLAGS != kRuntimeFlags) {
        Mp1mConsts<FT> ch = c;
        mp1m_hoist_consts<FT>(ch);
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            mp1m_tendencies_point<FT, FLAGS>(ch, rho[k], T[k], q_tot[k], q_lcl[k], q_icl[k], q_rai[k], q_sno[k], dl[k], di[k], dr[k], ds[k]);
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            mp1m_tendencies_point<FT, FLAGS>(front_consts<FT, FLAGS == kRuntimeFlags>(c), rho[k], T[k], q_tot[k], q_lcl[k], q_icl[k], q_rai[k], q_sno[k], dl[k],
                                             di[k], dr[k], ds[k]);
    }
    store_col<FT, VEC>(out.dq_lcl, i, dl); store_col<FT, VEC>(out.dq_icl, i, di);
    store_col<FT, VEC>(out.dq_rai, i, dr); store_col<FT, VEC>(out.dq_sno, i, ds);
}

// bulk_microphysics_tendencies(LinearizedAverage(), Microphysics1Moment(), …, Δt, nsub) over columns — BMT:572-632 (cmx_mp1m.hpp
// mp1m_linearized_point).  One point per lane (the substep loop carries five state variables).  Both constant structs travel as ONE
// by-value kernel argument so that the phase-local reads of the Float64 instantiation (front_consts: the FIRST kernel argument)
// address members of one struct — no hand-computed offsets into the kernel-argument segment.
template <typename FT> struct Mp1mLinIO { const FT *in[7]; FT *out[4]; };

// `at(column pointer, k)`: element of column k (0–6 state, 7–10 tendencies) this lane owns
template <typename FT, uint32_t FLAGS, typename IDX>
__device__ __forceinline__ void mp1m_linearized_lane(const Mp1mLinKernArgs<FT> &k0, const Mp1mLinIO<FT> &io, IDX at) {
    const auto &k = front_consts<FT, FLAGS == kRuntimeFlags>(k0);
    FT x[7], dl, di, dr, ds;
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = at(io.in[j], j);
    mp1m_linearized_point<FT, FLAGS>(k.c, [&](FT dep) -> decltype(auto) { return (consts_after(k, dep).a); }, k0.a.nsub, x[0], x[1], x[2], x[3], x[4], x[5], x[6],
                                     dl, di, dr, ds);
    at(io.out[0], 7) = dl; at(io.out[1], 8) = di; at(io.out[2], 9) = dr; at(io.out[3], 10) = ds;
}

#ifndef CMX_1M_LIN_BLOCK
#define CMX_1M_LIN_BLOCK 256      // lanes per workgroup of the LinearizedAverage kernel (A/B switch)
#endif
constexpr int kBlockLin = CMX_1M_LIN_BLOCK;
template <typename FT, uint32_t FLAGS = kRuntimeFlags>
__global__ __launch_bounds__(kBlockLin) void mp1m_linearized_kernel(const Mp1mLinKernArgs<FT> k0, const Mp1mLinIO<FT> io, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    const int64_t i = (int64_t)blockIdx.x * kBlockLin + threadIdx.x;
    if (i >= n) return;
    mp1m_linearized_lane<FT, FLAGS>(k0, io, [i](auto *p, int) -> decltype(auto) { return (p[i]); });
}

// Float32, round 5: TWO consecutive points per lane as one packed pair (cmx_math.hpp f32x2) — 8-byte loads and stores, the substep loop in packed
// arithmetic.  Needs 8-byte-aligned columns; `npair` pairs.  (The host launches the one-point kernel for an odd last point or misaligned columns.)
#ifndef CMX_1M_LIN_PACKED
#define CMX_1M_LIN_PACKED 1      // A/B switch
#endif
template <uint32_t FLAGS = kRuntimeFlags>
__global__ __launch_bounds__(kBlockLin) void mp1m_linearized_pair_kernel(const Mp1mLinKernArgs<float> k0, const Mp1mLinIO<float> io, const int64_t npair) {
    const int64_t i = (int64_t)blockIdx.x * kBlockLin + threadIdx.x;
    if (i >= npair) return;
    const auto &k = front_consts<float, true>(k0);
    f32x2 x[7], d[4];
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = reinterpret_cast<const f32x2 *>(io.in[j])[i];
    mp1m_linearized_point<f32x2, FLAGS>(k.c, [&](f32x2 dep) -> decltype(auto) { return (consts_after(k, dep).a); }, k0.a.nsub, x[0], x[1], x[2], x[3], x[4], x[5], x[6],
                                        d[0], d[1], d[2], d[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) reinterpret_cast<f32x2 *>(io.out[j])[i] = d[j];
}

// _microphysics_source_terms over columns (KAT / diagnostics harness): one point per lane
template <typename FT>
__global__ __launch_bounds__(kBlock) void mp1m_sources_kernel(const Mp1mConsts<FT> c, const Mp1mIn<FT> in,
                                                              const Mp1mSrcOut<FT> out, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FT rho = in.rho[i], T = in.T[i], q_tot = in.q_tot[i], q_lcl = in.q_lcl[i], q_icl = in.q_icl[i], q_rai = in.q_rai[i], q_sno = in.q_sno[i];
    const Mp1mSrc<FT> p = mp1m_point<FT>(front_consts<FT>(c), rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno);
    FT s[CMX_MP1M_NSRC];
    mp1m_expand<FT>(p, s);
    // NaN in → NaN out, as in the tendencies kernels (cmx_math.hpp any_nan): the clamps and the max0 gates of the point function return 0
    // for a NaN operand (for the Float32 integer form: for a NaN with the sign bit set), which would turn a NaN input into a
    // zero melt / accretion term here
    const FT poison = ((int)any_nan(rho, q_tot, q_lcl, q_icl, q_rai, q_sno, T) | (int)bad_density(rho)) ? Math<FT>::nan() : FT(0);
#pragma unroll
    for (int k = 0; k < CMX_MP1M_NSRC; ++k)
        if (out.col[k]) out.col[k][i] = s[k] + poison;
}

// ---- terminal velocities over (ρ, q) columns — CM1:223-270 (point functions: cmx_mp1m_vel.hpp) ---------------------
template <typename FT> struct Vel1mIO {
    const FT *rho, *q_rai, *q_sno; FT *vt_rai, *vt_sno, *vt_chen;
    const FT *q_lcl, *q_icl; FT *w_lcl, *w_icl, *w_sno_chen;
};

// GENERAL_GAMMA: the Chen-2022 rain table's exponents leave the polynomial-Γ window → run-time Γ for any argument
template <typename FT, bool GENERAL_GAMMA>
__global__ __launch_bounds__(kBlock) void mp1m_velocity_kernel(const Vel1mConsts<FT> c, const Vel1mIO<FT> io, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FT rho = io.rho[i];
    const FT rp = M::max(FT(0), rho);
    // Float64: ρ ≤ 0 (outside the domain, include/cmx.h) gives NaN in every output, as in the tendency entries — the floored logarithms of the fall
    // speeds (cmx_math.hpp log2_floored) would otherwise return finite numbers where the reference's are ±Inf or NaN
    auto out = [&](FT v) -> FT {
        if constexpr (M::IS_F64) return bad_density(rho) ? M::nan() : v;
        else return v;
    };
    if (io.vt_rai || io.vt_chen) {
        const FT q = io.q_rai[i];
        const FT l2_li = vel_l2_li_rain<FT>(c, rp, q);
        if (io.vt_rai) io.vt_rai[i] = out(vel_rain_blk1m<FT>(c, rho, l2_li, q));
        if (io.vt_chen) io.vt_chen[i] = out(vel_rain_chen<FT, GENERAL_GAMMA>(c, rp, l2_li, q));
    }
    if (io.vt_sno || io.w_sno_chen) {
        const FT q = io.q_sno[i];
        const FT l2_li = vel_l2_li_snow<FT>(c, rp, q);
        if (io.vt_sno) io.vt_sno[i] = out(vel_snow_blk1m<FT>(c, l2_li, q));
        if (io.w_sno_chen) io.w_sno_chen[i] = out(vel_snow_chen<FT>(c, rp, l2_li, q));
    }
    if (io.w_lcl) io.w_lcl[i] = out(vel_lcl_stokes<FT>(c, rho, io.q_lcl[i]));
    if (io.w_icl) io.w_icl[i] = out(vel_icl_chen<FT>(c, rho, rp, io.q_icl[i]));
}

// ---- host side -----------------------------------------------------------------------------------------------------
template <typename FT, typename MP, typename TH>
static int32_t tendencies_1m_entry(const MP *mp, const TH *tps, uint32_t flags, int64_t n, const FT *rho, const FT *T,
                                   const FT *q_tot, const FT *q_lcl, const FT *q_icl, const FT *q_rai, const FT *q_sno,
                                   FT *dq_lcl, FT *dq_icl, FT *dq_rai, FT *dq_sno, void *stream) {
    if (!mp || !tps || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (const int32_t st = check_flags_1m(flags)) return st;
    if (n == 0) return CMX_OK;
    if (!rho || !T || !q_tot || !q_lcl || !q_icl || !q_rai || !q_sno || !dq_lcl || !dq_icl || !dq_rai || !dq_sno) return CMX_ERR_BAD_ARG;
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int VEC = (sizeof(FT) == 8 && CMX_F64_ONE_POINT_PER_LANE) ? 1 : Math<FT>::VEC;
    const void *ptrs[] = {rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl, dq_icl, dq_rai, dq_sno};
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(rho) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (const void *p : ptrs) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(p) & 15u) == mis0);
    auto launch_range = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        Mp1mIn<FT> in{rho + lo, T + lo, q_tot + lo, q_lcl + lo, q_icl + lo, q_rai + lo, q_sno + lo};
        Mp1mOut<FT> out{dq_lcl + lo, dq_icl + lo, dq_rai + lo, dq_sno + lo};
        const int64_t nv = count / V;
        const dim3 grid((unsigned)((nv + kBlock1m<FT> - 1) / kBlock1m<FT>));
        if (flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c))
            CMX_LAUNCH_FRONT((mp1m_tendencies_kernel<FT, V, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>), grid, dim3(kBlock1m<FT>), 0, s, c, in, out, nv);
        else
            CMX_LAUNCH_FRONT((mp1m_tendencies_kernel<FT, V>), grid, dim3(kBlock1m<FT>), 0, s, c, in, out, nv);
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

// host-model layouts: the Instantaneous tendencies as a policy of the generic adapter kernel (cmx_layout.hpp)
template <typename FT, uint32_t FLAGS> struct Mp1mLayoutPolicy {
    static constexpr int NIN = 7, NOUT = 4, NAOS = 4;   // rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno → (dq_lcl_dt, dq_icl_dt, dq_rai_dt, dq_sno_dt)
    using Consts = Mp1mConsts<FT>;
    static constexpr bool PACKABLE = CMX_1M_PACKED;      // point() also takes the packed pair type (cmx_layout.hpp)
    template <typename C, typename VT> static __device__ __forceinline__ void point(const C &c, const VT (&x)[NIN], VT (&y)[NOUT]) {
        mp1m_tendencies_point<VT, FLAGS>(c, x[0], x[1], x[2], x[3], x[4], x[5], x[6], y[0], y[1], y[2], y[3]);
    }
};
// … and the LinearizedAverage tendencies (both constant structs as the one kernel argument)
template <typename FT, uint32_t FLAGS> struct Mp1mLinLayoutPolicy {
    static constexpr int NIN = 7, NOUT = 4, NAOS = 4;
    using Consts = Mp1mLinKernArgs<FT>;
    static constexpr bool PACKABLE = CMX_1M_LIN_PACKED;
    static constexpr bool PHASE_CONSTS = FLAGS == kRuntimeFlags;
    template <typename C, typename VT> static __device__ __forceinline__ void point(const C &k, const VT (&x)[NIN], VT (&y)[NOUT]) {
        mp1m_linearized_point<VT, FLAGS>(k.c, [&](VT dep) -> decltype(auto) { return (consts_after(k, dep).a); }, k.a.nsub, x[0], x[1], x[2], x[3], x[4], x[5],
                                         x[6], y[0], y[1], y[2], y[3]);
    }
};
template <typename FT, typename MP, typename TH>
static int32_t fields_1m_entry(const MP *mp, const TH *tps, uint32_t flags, int64_t n_seg, int64_t seg_len, const FT *const *in,
                               const int64_t *in_stride, FT *const *out, const int64_t *out_stride, FT *aos, void *stream) {
    if (!mp || !tps) return CMX_ERR_BAD_ARG;
    if (const int32_t st = check_flags_1m(flags)) return st;
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c))
        return launch_layout<FT, Mp1mLayoutPolicy<FT, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>>(c, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
    return launch_layout<FT, Mp1mLayoutPolicy<FT, kRuntimeFlags>>(c, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
}

template <typename FT, typename MP, typename TH>
static int32_t make_lin_kernargs(const MP *mp, const TH *tps, uint32_t flags, FT q_min, FT dt, int32_t nsub, Mp1mLinKernArgs<FT> &k) {
    if (!mp || !tps || nsub < 1 || !(dt > FT(0)) || !(q_min >= FT(0))) return CMX_ERR_BAD_ARG;
    if (const int32_t st = check_flags_1m(flags)) return st;
    k.c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    k.a = make_mp1m_lin_args<FT>(q_min, dt, nsub, (FT)tps->LH_v0, (FT)tps->LH_s0, (FT)tps->cp_d);
    return CMX_OK;
}

template <typename FT, typename MP, typename TH>
static int32_t linearized_1m_entry(const MP *mp, const TH *tps, uint32_t flags, FT q_min, FT dt, int32_t nsub, int64_t n, const FT *rho,
                                   const FT *T, const FT *q_tot, const FT *q_lcl, const FT *q_icl, const FT *q_rai, const FT *q_sno,
                                   FT *dq_lcl, FT *dq_icl, FT *dq_rai, FT *dq_sno, void *stream) {
    if (n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    Mp1mLinKernArgs<FT> k{};
    if (const int32_t st = make_lin_kernargs<FT>(mp, tps, flags, q_min, dt, nsub, k)) return st;
    if (n == 0) return CMX_OK;
    if (!rho || !T || !q_tot || !q_lcl || !q_icl || !q_rai || !q_sno || !dq_lcl || !dq_icl || !dq_rai || !dq_sno) return CMX_ERR_BAD_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool def = flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(k.c);
    auto launch_points = [&](int64_t lo, int64_t count) {       // one point per lane
        if (count <= 0) return;
        Mp1mLinIO<FT> io{{rho + lo, T + lo, q_tot + lo, q_lcl + lo, q_icl + lo, q_rai + lo, q_sno + lo}, {dq_lcl + lo, dq_icl + lo, dq_rai + lo, dq_sno + lo}};
        const dim3 grid((unsigned)((count + kBlockLin - 1) / kBlockLin));
        if (def) CMX_LAUNCH_FRONT((mp1m_linearized_kernel<FT, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>), grid, dim3(kBlockLin), 0, s, k, io, count);
        else CMX_LAUNCH_FRONT((mp1m_linearized_kernel<FT>), grid, dim3(kBlockLin), 0, s, k, io, count);
    };
    if constexpr (sizeof(FT) == 4 && CMX_1M_LIN_PACKED) {
        // Float32: pairs of points in packed arithmetic wherever all eleven columns share their offset modulo 8 bytes: an odd first point (and an
        // odd last one) goes through the one-point kernel — the same per-point operations, so the result does not depend on the split
        const void *ptrs[] = {rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl, dq_icl, dq_rai, dq_sno};
        const uintptr_t mis0 = reinterpret_cast<uintptr_t>(rho) & 7u;
        bool same_mis = (mis0 % sizeof(FT)) == 0;
        for (const void *p : ptrs) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(p) & 7u) == mis0);
        if (same_mis && n >= 2) {
            const int64_t head = mis0 ? 1 : 0, npair = (n - head) / 2;
            launch_points(0, head);
            if (npair > 0) {
                const int64_t lo = head;
                Mp1mLinIO<float> io{{rho + lo, T + lo, q_tot + lo, q_lcl + lo, q_icl + lo, q_rai + lo, q_sno + lo}, {dq_lcl + lo, dq_icl + lo, dq_rai + lo, dq_sno + lo}};
                const dim3 grid((unsigned)((npair + kBlockLin - 1) / kBlockLin));
                if (def) CMX_LAUNCH_FRONT((mp1m_linearized_pair_kernel<CMX_1M_DEFAULT_OPTIONS | kDefExpBit>), grid, dim3(kBlockLin), 0, s, k, io, npair);
                else CMX_LAUNCH_FRONT((mp1m_linearized_pair_kernel<kRuntimeFlags>), grid, dim3(kBlockLin), 0, s, k, io, npair);
            }
            launch_points(head + 2 * npair, n - head - 2 * npair);
        } else
            launch_points(0, n);
    } else
        launch_points(0, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

template <typename FT, typename MP, typename TH>
static int32_t linearized_fields_1m_entry(const MP *mp, const TH *tps, uint32_t flags, FT q_min, FT dt, int32_t nsub, int64_t n_seg, int64_t seg_len,
                                          const FT *const *in, const int64_t *in_stride, FT *const *out, const int64_t *out_stride, FT *aos, void *stream) {
    Mp1mLinKernArgs<FT> k{};
    if (const int32_t st = make_lin_kernargs<FT>(mp, tps, flags, q_min, dt, nsub, k)) return st;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(k.c))
        return launch_layout<FT, Mp1mLinLayoutPolicy<FT, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>>(k, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
    return launch_layout<FT, Mp1mLinLayoutPolicy<FT, kRuntimeFlags>>(k, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
}

template <typename FT, typename MP, typename TH>
static int32_t sources_1m_entry(const MP *mp, const TH *tps, uint32_t flags, int64_t n, const FT *rho, const FT *T,
                                const FT *q_tot, const FT *q_lcl, const FT *q_icl, const FT *q_rai, const FT *q_sno,
                                FT *const out[CMX_MP1M_NSRC], void *stream) {
    if (!mp || !tps || !out || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (const int32_t st = check_flags_1m(flags)) return st;
    if (n == 0) return CMX_OK;
    if (!rho || !T || !q_tot || !q_lcl || !q_icl || !q_rai || !q_sno) return CMX_ERR_BAD_ARG;
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    Mp1mIn<FT> in{rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno};
    Mp1mSrcOut<FT> o;
    for (int k = 0; k < CMX_MP1M_NSRC; ++k) o.col[k] = out[k];
    CMX_LAUNCH_FRONT((mp1m_sources_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c, in, o, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// the Chen-2022 rain table selects the instantiation: polynomial Γ on its window (fast) or run-time Γ for any exponent (general)
template <typename FT> static void launch_velocity(bool general_gamma, const Vel1mConsts<FT> &c, const Vel1mIO<FT> &io, int64_t n, void *stream) {
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (general_gamma) hipLaunchKernelGGL((mp1m_velocity_kernel<FT, true>), grid, dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), c, io, n);
    else hipLaunchKernelGGL((mp1m_velocity_kernel<FT, false>), grid, dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), c, io, n);
}

template <typename FT, typename MP, typename CH>
static int32_t velocity_1m_entry(const MP *mp, const CH *chen, int64_t n, const FT *rho, const FT *q_rai, const FT *q_sno,
                                 FT *vt_rai, FT *vt_sno, FT *vt_chen, void *stream) {
    if (!mp || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!rho || ((vt_rai || vt_chen) && !q_rai) || (vt_sno && !q_sno) || (vt_chen && !chen)) return CMX_ERR_BAD_ARG;
    bool general = false;
    const Vel1mConsts<FT> c = make_vel1m_consts<FT>(*mp, chen, &general);
    Vel1mIO<FT> io{rho, q_rai, q_sno, vt_rai, vt_sno, vt_chen, nullptr, nullptr, nullptr, nullptr, nullptr};
    launch_velocity<FT>(general, c, io, n, stream);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// The four bulk sedimentation velocities a host model precomputes (ClimaAtmos set_sedimentation_precomputed_quantities;
// test/gpu_clima_core_test.jl:36-45, KA kernel test/gpu_tests.jl:608-630)
template <typename FT, typename MP, typename ST, typename CH, typename CI>
static int32_t sedimentation_entry(const MP *mp, const ST *stokes, const CH *chen_rain, const CI *chen_ice, int64_t n, const FT *rho,
                                   const FT *q_lcl, const FT *q_icl, const FT *q_rai, const FT *q_sno, FT *w_lcl, FT *w_icl, FT *w_rai,
                                   FT *w_sno, void *stream) {
    if (!mp || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!rho || (w_lcl && (!q_lcl || !stokes)) || (w_icl && (!q_icl || !chen_ice)) || (w_rai && (!q_rai || !chen_rain)) ||
        (w_sno && (!q_sno || !chen_ice)))
        return CMX_ERR_BAD_ARG;
    bool general = false;
    Vel1mConsts<FT> c = make_vel1m_consts<FT>(*mp, chen_rain, &general);
    add_sedimentation_consts<FT>(c, *mp, stokes, chen_ice);
    Vel1mIO<FT> io{rho, q_rai, q_sno, nullptr, nullptr, w_rai, q_lcl, q_icl, w_lcl, w_icl, w_sno};
    launch_velocity<FT>(general, c, io, n, stream);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_sedimentation_velocities_f32(const cmx_microphysics_1m_f32 *mp, const cmx_stokes_vel_f32 *stokes,
                                         const cmx_chen2022_rain_vel_f32 *chen_rain, const cmx_chen2022_ice_vel_f32 *chen_ice, int64_t n,
                                         const float *rho, const float *q_lcl, const float *q_icl, const float *q_rai, const float *q_sno,
                                         float *w_lcl, float *w_icl, float *w_rai, float *w_sno, void *stream) {
    return cmx::sedimentation_entry<float>(mp, stokes, chen_rain, chen_ice, n, rho, q_lcl, q_icl, q_rai, q_sno, w_lcl, w_icl, w_rai, w_sno,
                                           stream);
}
int32_t cmx_sedimentation_velocities_f64(const cmx_microphysics_1m_f64 *mp, const cmx_stokes_vel_f64 *stokes,
                                         const cmx_chen2022_rain_vel_f64 *chen_rain, const cmx_chen2022_ice_vel_f64 *chen_ice, int64_t n,
                                         const double *rho, const double *q_lcl, const double *q_icl, const double *q_rai,
                                         const double *q_sno, double *w_lcl, double *w_icl, double *w_rai, double *w_sno, void *stream) {
    return cmx::sedimentation_entry<double>(mp, stokes, chen_rain, chen_ice, n, rho, q_lcl, q_icl, q_rai, q_sno, w_lcl, w_icl, w_rai, w_sno,
                                            stream);
}

int32_t cmx_mp1m_linearized_average_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, float q_min, float dt,
                                        int32_t nsub, int64_t n, const float *rho, const float *T, const float *q_tot, const float *q_lcl,
                                        const float *q_icl, const float *q_rai, const float *q_sno, float *dq_lcl_dt, float *dq_icl_dt,
                                        float *dq_rai_dt, float *dq_sno_dt, void *stream) {
    return cmx::linearized_1m_entry<float>(mp, tps, flags, q_min, dt, nsub, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl_dt,
                                           dq_icl_dt, dq_rai_dt, dq_sno_dt, stream);
}
int32_t cmx_mp1m_linearized_average_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, double q_min,
                                        double dt, int32_t nsub, int64_t n, const double *rho, const double *T, const double *q_tot,
                                        const double *q_lcl, const double *q_icl, const double *q_rai, const double *q_sno,
                                        double *dq_lcl_dt, double *dq_icl_dt, double *dq_rai_dt, double *dq_sno_dt, void *stream) {
    return cmx::linearized_1m_entry<double>(mp, tps, flags, q_min, dt, nsub, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl_dt,
                                            dq_icl_dt, dq_rai_dt, dq_sno_dt, stream);
}

int32_t cmx_mp1m_linearized_average_fields_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, float q_min, float dt,
                                               int32_t nsub, int64_t n_seg, int64_t seg_len, const float *const *in, const int64_t *in_seg_stride,
                                               float *const *out, const int64_t *out_seg_stride, float *out_aos, void *stream) {
    return cmx::linearized_fields_1m_entry<float>(mp, tps, flags, q_min, dt, nsub, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, out_aos, stream);
}
int32_t cmx_mp1m_linearized_average_fields_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, double q_min, double dt,
                                               int32_t nsub, int64_t n_seg, int64_t seg_len, const double *const *in, const int64_t *in_seg_stride,
                                               double *const *out, const int64_t *out_seg_stride, double *out_aos, void *stream) {
    return cmx::linearized_fields_1m_entry<double>(mp, tps, flags, q_min, dt, nsub, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, out_aos, stream);
}

int32_t cmx_mp1m_tendencies_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n,
                                const float *rho, const float *T, const float *q_tot, const float *q_lcl, const float *q_icl,
                                const float *q_rai, const float *q_sno, float *dq_lcl_dt, float *dq_icl_dt, float *dq_rai_dt,
                                float *dq_sno_dt, void *stream) {
    return cmx::tendencies_1m_entry<float>(mp, tps, flags, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl_dt, dq_icl_dt,
                                           dq_rai_dt, dq_sno_dt, stream);
}
int32_t cmx_mp1m_tendencies_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n,
                                const double *rho, const double *T, const double *q_tot, const double *q_lcl,
                                const double *q_icl, const double *q_rai, const double *q_sno, double *dq_lcl_dt,
                                double *dq_icl_dt, double *dq_rai_dt, double *dq_sno_dt, void *stream) {
    return cmx::tendencies_1m_entry<double>(mp, tps, flags, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl_dt, dq_icl_dt,
                                            dq_rai_dt, dq_sno_dt, stream);
}
int32_t cmx_mp1m_tendencies_fields_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n_seg, int64_t seg_len,
                                       const float *const *in, const int64_t *in_seg_stride, float *const *out, const int64_t *out_seg_stride,
                                       float *out_aos, void *stream) {
    return cmx::fields_1m_entry<float>(mp, tps, flags, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, out_aos, stream);
}
int32_t cmx_mp1m_tendencies_fields_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n_seg, int64_t seg_len,
                                       const double *const *in, const int64_t *in_seg_stride, double *const *out, const int64_t *out_seg_stride,
                                       double *out_aos, void *stream) {
    return cmx::fields_1m_entry<double>(mp, tps, flags, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, out_aos, stream);
}
int32_t cmx_mp1m_source_terms_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n,
                                  const float *rho, const float *T, const float *q_tot, const float *q_lcl, const float *q_icl,
                                  const float *q_rai, const float *q_sno, float *const out[CMX_MP1M_NSRC], void *stream) {
    return cmx::sources_1m_entry<float>(mp, tps, flags, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, out, stream);
}
int32_t cmx_mp1m_source_terms_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n,
                                  const double *rho, const double *T, const double *q_tot, const double *q_lcl,
                                  const double *q_icl, const double *q_rai, const double *q_sno,
                                  double *const out[CMX_MP1M_NSRC], void *stream) {
    return cmx::sources_1m_entry<double>(mp, tps, flags, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, out, stream);
}
int32_t cmx_mp1m_terminal_velocity_f32(const cmx_microphysics_1m_f32 *mp, const cmx_chen2022_rain_vel_f32 *chen, int64_t n,
                                       const float *rho, const float *q_rai, const float *q_sno, float *vt_rai_blk1m,
                                       float *vt_sno_blk1m, float *vt_rai_chen, void *stream) {
    return cmx::velocity_1m_entry<float>(mp, chen, n, rho, q_rai, q_sno, vt_rai_blk1m, vt_sno_blk1m, vt_rai_chen, stream);
}
int32_t cmx_mp1m_terminal_velocity_f64(const cmx_microphysics_1m_f64 *mp, const cmx_chen2022_rain_vel_f64 *chen, int64_t n,
                                       const double *rho, const double *q_rai, const double *q_sno, double *vt_rai_blk1m,
                                       double *vt_sno_blk1m, double *vt_rai_chen, void *stream) {
    return cmx::velocity_1m_entry<double>(mp, chen, n, rho, q_rai, q_sno, vt_rai_blk1m, vt_sno_blk1m, vt_rai_chen, stream);
}

}  // extern "C"
