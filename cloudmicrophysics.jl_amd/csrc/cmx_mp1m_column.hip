// cmx_mp1m_column.hip — the OPERATIONAL 1-moment column step in one pass (SURVEY §8f-1 + §8f-4):
//   (1) tend = BMT.bulk_microphysics_tendencies(Instantaneous() | LinearizedAverage(), Microphysics1Moment(), mp, tps, ρ, T, q_tot,
//              q_lcl, q_icl, q_rai, q_sno [, Δt, nsub])                        /root/reference/src/BulkMicrophysicsTendencies.jl:505-514, 572-632
//   (2) the four bulk fall speeds a host model precomputes for sedimentation (ClimaAtmos set_sedimentation_precomputed_quantities;
//       test/gpu_clima_core_test.jl:36-45, KA kernel test/gpu_tests.jl:608-630):
//         w_lcl = CMNonEq.terminal_velocity(cloud_liquid, StokesRegimeVelType, ρ, q_lcl)       src/MicrophysicsNonEq.jl:250-265
//         w_icl = CMNonEq.terminal_velocity(cloud_ice, Chen2022VelTypeSmallIce, ρ, q_icl)      :267-281
//         w_rai = CM1.terminal_velocity(rain, Chen2022VelTypeRain, ρ, q_rai)                   src/Microphysics1M.jl:251-270
//         w_sno = CM1.terminal_velocity(snow, Chen2022VelTypeLargeIce, ρ, q_sno)               :272-297
//   (3) the host model's first-order upwind ("right-biased") flux divergence of the four falling species — NOT part of the reference
//       package (ClimaAtmos precipitation advection), the same operator as in cmx_sb2006_column.hip:
//           F_k = ρ_k χ_k w_k,   ∂χ_k/∂t |sed = (F_{k+1} − F_k)/(ρ_k Δz_k),   F_{n_lev} = 0,   level 0 = lowest level.
// Unfused that is three kernels: 7 in + 4 out, 5 in + 4 out, then 9 in + 4 read-modify-write = 148 B/point (f32); fused 44 B/point.
//
// Layout and tiling as in cmx_sb2006_column.hip: n_col columns of n_lev CONTIGUOUS levels, lanes along the flat index with 16-byte
// accesses, the flux of the cell above from the next point (registers inside a lane's vector, LDS across lanes, and for the point
// after the workgroup's tile an evaluation from the raw columns).  The fluxes of EVERY point — in the tile or after it — come from
// one function of (ρ, q_lcl, q_icl, q_rai, q_sno) only, so results do not depend on where tile boundaries fall; a NaN in ρ or in a
// species' q poisons that species' flux (and the cell below through it), a NaN in any input poisons the point's four tendencies.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "cmx_launch.hpp"
#include "cmx_mp1m.hpp"
#include "cmx_mp1m_vel.hpp"

namespace cmx {

template <typename FT> struct Mp1mColIO {
    const FT *in[7];       // rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno
    FT *out[4];            // dq_lcl_dt, dq_icl_dt, dq_rai_dt, dq_sno_dt
    const FT *inv_dz;      // n_lev values 1/Δz_k
    FT *precip_rai, *precip_sno;   // n_col values each or nullptr: the surface fluxes F_0 of rain and snow
    int64_t n;             // n_col · n_lev
    int32_t n_lev;
    double inv_n_lev;
    FastDivU32 lev_div;    // division by n_lev for flat indices below 2³² (cmx_launch.hpp fastdiv)
};

// All constants (≈ 170 values: tendencies, LinearizedAverage step, fall speeds) travel as ONE by-value kernel argument, the first, and BOTH
// float types read them through the kernel-argument pointer (front_consts<FT, true>): as plain by-value arguments they overflow the
// SGPR file and were parked in VGPR lanes — 34 v_readlane / v_writelane per Float32 point and 143 VGPRs in the first version.
template <typename FT> struct Mp1mColKernArgs { Mp1mLinKernArgs<FT> k; Vel1mConsts<FT> vc; };

#ifndef CMX_1M_COLUMN_PACKED
#define CMX_1M_COLUMN_PACKED 1      // A/B switch: 0 = one point at a time (rounds 3–4)
#endif
template <typename FT, uint32_t FLAGS, bool LIN, bool GENERAL_GAMMA, int VEC, int BS>
__global__ __launch_bounds__(BS) void mp1m_column_kernel(const Mp1mColKernArgs<FT> a0, const Mp1mColIO<FT> io, const int64_t first, const int64_t nvec) {
    using M = Math<FT>;
    const auto &a = front_consts<FT, true>(a0);
    __shared__ __align__(16) FT halo[BS + 1][4];   // fluxes of every lane's FIRST point, + slot BS for the point that follows the tile
    // tiles overlap by one lane, as in sb2006_column_kernel (cmx_sb2006_column.hip): lane BS−1 evaluates the next tile's first vector and only
    // supplies the flux its first point sends down; the flux-only evaluation of that point runs in the last workgroup of a launch only
    const int64_t tile0 = (int64_t)blockIdx.x * (BS - 1);
    const int64_t v = tile0 + threadIdx.x;
    const bool active = v < nvec;
    const int64_t nvalid = nvec - tile0 < BS ? nvec - tile0 : BS;       // active lanes of this tile (≥ 1)
    const bool owner = threadIdx.x < BS - 1;
    const int64_t i0 = first + v * VEC;                                 // flat index of the lane's first point

    FT x[7][VEC];
    if (active) {
#pragma unroll
        for (int j = 0; j < 7; ++j) load_col<FT, VEC, true>(io.in[j] + first, v, x[j]);
    }
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly (every lane of the workgroup reaches the barrier inside); no-op for Float32
    if (nvalid < BS && threadIdx.x == BS - 1) {   // last workgroup of the launch: the point after its last vector, from the raw columns
        const int64_t e = first + (tile0 + nvalid) * VEC;
        SedFlux4<FT> f{{FT(0), FT(0), FT(0), FT(0)}};
        if (e < io.n) f = mp1m_sed_fluxes<FT, GENERAL_GAMMA>(a.vc, io.in[0][e], io.in[3][e], io.in[4][e], io.in[5][e], io.in[6][e]);
#pragma unroll
        for (int s = 0; s < 4; ++s) halo[BS][s] = f.f[s];
    }
    // per point: A = tendency − (own flux)·g (final but for the inflow), g = 1/(ρ Δz)
    FT A[VEC][4], g[VEC];
    SedFlux4<FT> F[VEC];
    int64_t col = 0;
    int32_t lev = 0;
    if (active) {
        if (io.n < ((int64_t)1 << 32)) {                  // i0 = col·n_lev + k; wave-uniform: one v_mul_hi_u32 (cmx_launch.hpp fastdiv)
            const uint32_t c32 = fastdiv((uint32_t)i0, io.lev_div);
            col = c32;
            lev = (int32_t)((uint32_t)i0 - c32 * (uint32_t)io.n_lev);
        } else {                                          // one double multiply + fix-up per lane; exact below 2^53
            col = (int64_t)((double)i0 * io.inv_n_lev);
            int64_t k64 = i0 - col * io.n_lev;
            if (k64 < 0) { --col; k64 += io.n_lev; }
            if (k64 >= io.n_lev) { ++col; k64 -= io.n_lev; }
            lev = (int32_t)k64;
        }
        int32_t lv = lev;
        const auto &k = a.k;
        // two passes over the lane's points — tendencies, then fluxes — so that only one pass's constants are live at a time
        FT t[VEC][4];
        // one value of the value type VT at a time: a point, or — Float32 with four points per lane — a PAIR of points in packed arithmetic (cmx_math.hpp f32x2;
        // not the run-time-Γ instantiation: an OCML call per lane, nothing to pack)
        constexpr int L = (sizeof(FT) == 4 && VEC % 2 == 0 && CMX_1M_COLUMN_PACKED && !GENERAL_GAMMA) ? 2 : 1;
        using VT = std::conditional_t<L == 2, f32x2, FT>;
        using MV = Math<VT>;
        if constexpr (L == 1) {      // (kept verbatim from rounds 3–4: routing the one-point case through the generic form below doubled its registers)
#pragma unroll
            for (int p = 0; p < VEC; ++p) {
                if constexpr (LIN)
                    mp1m_linearized_point<FT, FLAGS>(k.c, [&](FT dep) -> decltype(auto) { return (consts_after(k, dep).a); }, a0.k.a.nsub, x[0][p], x[1][p], x[2][p],
                                                     x[3][p], x[4][p], x[5][p], x[6][p], t[p][0], t[p][1], t[p][2], t[p][3]);
                else
                    mp1m_tendencies_point<FT, FLAGS>(k.c, x[0][p], x[1][p], x[2][p], x[3][p], x[4][p], x[5][p], x[6][p], t[p][0], t[p][1], t[p][2], t[p][3]);
            }
            const auto &vc = consts_after(a, t[VEC - 1][3]).vc;
#pragma unroll
            for (int p = 0; p < VEC; ++p) {
                F[p] = mp1m_sed_fluxes<FT, GENERAL_GAMMA>(vc, x[0][p], x[3][p], x[4][p], x[5][p], x[6][p]);
                g[p] = io.inv_dz[lv] * M::rcp(max0(x[0][p]));                          // 1/(ρ_k Δz_k)
                if (++lv == io.n_lev) lv = 0;
#pragma unroll
                for (int s = 0; s < 4; ++s) A[p][s] = M::fma(-F[p].f[s], g[p], t[p][s]);
            }
        } else {
            auto val = [&x](int j, int p) -> VT {
                if constexpr (L == 2) return VT{x[j][p], x[j][p + 1]};
                else return x[j][p];
            };
#pragma unroll
            for (int p = 0; p < VEC; p += L) {
                VT tt[4];
                if constexpr (LIN)
                    mp1m_linearized_point<VT, FLAGS>(k.c, [&](VT dep) -> decltype(auto) { return (consts_after(k, dep).a); }, a0.k.a.nsub, val(0, p), val(1, p), val(2, p),
                                                     val(3, p), val(4, p), val(5, p), val(6, p), tt[0], tt[1], tt[2], tt[3]);
                else
                    mp1m_tendencies_point<VT, FLAGS>(k.c, val(0, p), val(1, p), val(2, p), val(3, p), val(4, p), val(5, p), val(6, p), tt[0], tt[1], tt[2], tt[3]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if constexpr (L == 2) { t[p][s] = tt[s].x; t[p + 1][s] = tt[s].y; }
                    else t[p][s] = tt[s];
                }
            }
            const auto &vc = consts_after(a, t[VEC - 1][3]).vc;
#pragma unroll
            for (int p = 0; p < VEC; p += L) {
                const SedFlux4<VT> Fp = mp1m_sed_fluxes<VT, GENERAL_GAMMA>(vc, val(0, p), val(3, p), val(4, p), val(5, p), val(6, p));
                VT gp;                                                                 // 1/(ρ_k Δz_k)
                if constexpr (L == 2) {
                    const FT dz0 = io.inv_dz[lv];
                    if (++lv == io.n_lev) lv = 0;
                    const FT dz1 = io.inv_dz[lv];
                    if (++lv == io.n_lev) lv = 0;
                    gp = VT{dz0, dz1} * MV::rcp(max0(val(0, p)));
                } else {
                    gp = io.inv_dz[lv] * MV::rcp(max0(val(0, p)));
                    if (++lv == io.n_lev) lv = 0;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    VT tt;
                    if constexpr (L == 2) tt = VT{t[p][s], t[p + 1][s]};
                    else tt = t[p][s];
                    const VT As = MV::fma(-Fp.f[s], gp, tt);
                    if constexpr (L == 2) { A[p][s] = As.x; A[p + 1][s] = As.y; F[p].f[s] = Fp.f[s].x; F[p + 1].f[s] = Fp.f[s].y; }
                    else { A[p][s] = As; F[p].f[s] = Fp.f[s]; }
                }
                if constexpr (L == 2) { g[p] = gp.x; g[p + 1] = gp.y; }
                else g[p] = gp;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) halo[threadIdx.x][s] = F[0].f[s];
    }
    __syncthreads();
    if (!active || !owner) return;
    const int up = (threadIdx.x + 1 < nvalid) ? threadIdx.x + 1 : BS;
    FT o[4][VEC];
#pragma unroll
    for (int p = 0; p < VEC; ++p) {
        const bool top = lev == io.n_lev - 1;       // nothing enters through the model top
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const FT upf = (p + 1 < VEC) ? F[p + 1 < VEC ? p + 1 : 0].f[s] : halo[up][s];
            o[s][p] = M::fma(top ? FT(0) : upf, g[p], A[p][s]);      // a select, not a branch per species
        }
        if (lev == 0) {                             // surface precipitation fluxes of the column
            if (io.precip_rai) io.precip_rai[col] = F[p].f[2];
            if (io.precip_sno) io.precip_sno[col] = F[p].f[3];
        }
        if (++lev == io.n_lev) { lev = 0; ++col; }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) store_col<FT, VEC, true>(io.out[s] + first, v, o[s]);
}

// lanes per workgroup: 256 (same-box A/B, round 4, ms per 1e8 points: Float32 64 lanes 1.156, 128 lanes 1.140, 256 lanes 1.120; Float64 4.74 / 4.63 / 4.55;
// profiles/r04_ab_sessions.txt, session 18 — as for the SB2006 column kernel, a longer tile halves the share of the overlap lane)
#ifndef CMX_COLUMN1M_BS
#define CMX_COLUMN1M_BS 256
#endif
constexpr int kColBS1m = CMX_COLUMN1M_BS;

template <typename FT, int VEC>
static void launch_column_1m(bool def, bool lin, bool general, const Mp1mColKernArgs<FT> &a, const Mp1mColIO<FT> &io, int64_t first, int64_t nvec,
                             hipStream_t s) {
    if (nvec <= 0) return;
    const dim3 grid((unsigned)((nvec + (kColBS1m - 1) - 1) / (kColBS1m - 1))), block(kColBS1m);      // tiles overlap by one lane
    constexpr uint32_t DEF = CMX_1M_DEFAULT_OPTIONS | kDefExpBit;
#define CMX_L(FL, LN, GG) CMX_LAUNCH_FRONT((mp1m_column_kernel<FT, FL, LN, GG, VEC, kColBS1m>), grid, block, 0, s, a, io, first, nvec)
#define CMX_G(FL, LN) do { if (general) CMX_L(FL, LN, true); else CMX_L(FL, LN, false); } while (0)
    if (def) { if (lin) CMX_G(DEF, true); else CMX_G(DEF, false); }
    else { if (lin) CMX_G(kRuntimeFlags, true); else CMX_G(kRuntimeFlags, false); }
#undef CMX_G
#undef CMX_L
}

template <typename FT, typename MP, typename TH, typename ST, typename CH, typename CI>
static int32_t column_1m_entry(const MP *mp, const TH *tps, const ST *stokes, const CH *chen_rain, const CI *chen_ice, uint32_t flags, FT q_min, FT dt,
                               int32_t nsub, int64_t n_col, int32_t n_lev, const FT *inv_dz, const FT *const *in, FT *const *out, FT *precip_rai,
                               FT *precip_sno, void *stream) {
    if (!mp || !tps || !stokes || !chen_rain || !chen_ice || n_col < 0 || n_lev < 1 || nsub < 0) return CMX_ERR_BAD_ARG;
    if (const int32_t st = check_flags_1m(flags)) return st;
    const bool lin = nsub > 0;
    if (lin && (!(dt > FT(0)) || !(q_min >= FT(0)))) return CMX_ERR_BAD_ARG;
    if (n_col > kMaxPoints / n_lev) return CMX_ERR_UNSUPPORTED;  // n_col·n_lev must fit one launch (cmx_launch.hpp)
    const int64_t n = n_col * (int64_t)n_lev;
    if (n == 0) return CMX_OK;
    if (!inv_dz || !in || !out) return CMX_ERR_BAD_ARG;
    Mp1mColKernArgs<FT> a{};
    Mp1mLinKernArgs<FT> &k = a.k;
    k.c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    if (lin) k.a = make_mp1m_lin_args<FT>(q_min, dt, nsub, (FT)tps->LH_v0, (FT)tps->LH_s0, (FT)tps->cp_d);
    bool general = false;
    a.vc = make_vel1m_consts<FT>(*mp, chen_rain, &general);
    add_sedimentation_consts<FT>(a.vc, *mp, stokes, chen_ice);
    Mp1mColIO<FT> io{};
    const void *ptrs[11];
    for (int j = 0; j < 7; ++j) { if (!in[j]) return CMX_ERR_BAD_ARG; io.in[j] = in[j]; ptrs[j] = in[j]; }
    for (int j = 0; j < 4; ++j) { if (!out[j]) return CMX_ERR_BAD_ARG; io.out[j] = out[j]; ptrs[7 + j] = out[j]; }
    io.inv_dz = inv_dz; io.precip_rai = precip_rai; io.precip_sno = precip_sno; io.n = n; io.n_lev = n_lev; io.inv_n_lev = 1.0 / (double)n_lev; io.lev_div = make_fastdiv((uint32_t)n_lev);
    const bool def = flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(k.c);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int VEC = sizeof(FT) == 8 ? 1 : Math<FT>::VEC;     // Float64: one point per lane, as in the pointwise kernel
    // alignment dispatch as in cmx_sb2006_column.hip: scalar head up to the common 16-byte boundary, vector body, scalar tail; the halo
    // point of each range is read from the full columns, so the three launches compose exactly
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(ptrs[0]) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (const void *p : ptrs) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(p) & 15u) == mis0);
    if (same_mis && VEC > 1) {
        const int64_t head = std::min<int64_t>(n, mis0 ? (int64_t)((16 - mis0) / sizeof(FT)) : 0);
        const int64_t body = ((n - head) / VEC) * VEC;
        launch_column_1m<FT, 1>(def, lin, general, a, io, 0, head, s);
        launch_column_1m<FT, VEC>(def, lin, general, a, io, head, body / VEC, s);
        launch_column_1m<FT, 1>(def, lin, general, a, io, head + body, n - head - body, s);
    } else {
        launch_column_1m<FT, 1>(def, lin, general, a, io, 0, n, s);
    }
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_mp1m_column_tendencies_sedimentation_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, const cmx_stokes_vel_f32 *stokes,
                                                     const cmx_chen2022_rain_vel_f32 *chen_rain, const cmx_chen2022_ice_vel_f32 *chen_ice, uint32_t flags,
                                                     float q_min, float dt, int32_t nsub, int64_t n_col, int32_t n_lev, const float *inv_dz,
                                                     const float *const *in, float *const *out, float *precip_rai, float *precip_sno, void *stream) {
    return cmx::column_1m_entry<float>(mp, tps, stokes, chen_rain, chen_ice, flags, q_min, dt, nsub, n_col, n_lev, inv_dz, in, out, precip_rai, precip_sno, stream);
}
int32_t cmx_mp1m_column_tendencies_sedimentation_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, const cmx_stokes_vel_f64 *stokes,
                                                     const cmx_chen2022_rain_vel_f64 *chen_rain, const cmx_chen2022_ice_vel_f64 *chen_ice, uint32_t flags,
                                                     double q_min, double dt, int32_t nsub, int64_t n_col, int32_t n_lev, const double *inv_dz,
                                                     const double *const *in, double *const *out, double *precip_rai, double *precip_sno, void *stream) {
    return cmx::column_1m_entry<double>(mp, tps, stokes, chen_rain, chen_ice, flags, q_min, dt, nsub, n_col, n_lev, inv_dz, in, out, precip_rai, precip_sno, stream);
}

}  // extern "C"
