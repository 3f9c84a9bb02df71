// cmx_sb2006_column.hip — SURVEY §8f-4: the 2M warm-rain tendencies FUSED with the sedimentation step that follows them in a
// host model, one pass over the columns.
//
// What is fused.  A host model (ClimaAtmos, KinematicDriver) evaluates per time step
//   (1) tend = BMT.bulk_microphysics_tendencies(Microphysics2Moment(), …)                       BMT:820-854 → :707-782
//   (2) w_r  = CM2.rain_terminal_velocity(sb, vel, q_rai, ρ, ρ n_rai)                           CM2:685-719
//       w_c  = CM2.cloud_terminal_velocity(pdf_c, StokesRegimeVelType, q_lcl, ρ, ρ n_lcl)       CM2:647-664   (optional)
//   (3) the vertical flux divergence of the falling species with those speeds — NOT part of the reference package: it lives in
//       the host model.  The scheme built here is the one ClimaAtmos uses for precipitation, first-order upwind with the value
//       taken from the cell above ("right-biased"), on a Cartesian column:
//           F_k = ρ_k χ_k w_k   (downward flux leaving cell k through its lower face; χ = q_rai, n_rai[, q_lcl, n_lcl])
//           ∂χ_k/∂t |sed = (F_{k+1} − F_k) / (ρ_k Δz_k),    F_{n_lev} = 0 (nothing enters through the model top),
//       level 0 = lowest level; F_0 of q_rai is the surface precipitation flux [kg m⁻² s⁻¹] (optional output, one per column).
//   Unfused, (1)+(2) write 6 columns and (3) reads them back together with ρ, q_rai, n_rai and read-modify-writes two tendencies:
//   88 B/point (f32).  Fused: 7 columns in, 4 out = 44 B/point — one full HBM round trip less.
//
// Layout: n_col columns of n_lev CONTIGUOUS levels (flat index i = col·n_lev + k; a ClimaCore VF / VIJFH(Ni=Nj=1) column field).
// Lanes run along the flat index (coalesced 16-byte accesses as in the pointwise kernel); the flux of the cell above is the next
// point: inside a lane's vector it is in registers, across lanes it goes through LDS (one 16-byte row per lane), and the point
// after the workgroup's tile is evaluated by a flux-only function (rain PSD + fall speeds: the same inline code as the point
// function, hence the same bits — results, NaN propagation included, do not depend on where tile boundaries fall).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "cmx_launch.hpp"
#include "cmx_sb2006.hpp"

namespace cmx {

template <typename FT> struct SbColIO {
    const FT *rho, *T, *q_tot, *q_lcl, *n_lcl, *q_rai, *n_rai;
    FT *dq_lcl, *dn_lcl, *dq_rai, *dn_rai;
    const FT *inv_dz;      // n_lev values 1/Δz_k
    FT *precip;            // n_col values or nullptr
    int64_t n;             // n_col · n_lev
    int32_t n_lev;
    double inv_n_lev;
    FastDivU32 lev_div;    // division by n_lev for flat indices below 2³² (cmx_launch.hpp fastdiv)
};

template <typename FT> struct SedFlux { FT q_rai, n_rai, q_lcl, n_lcl; };

// fluxes of one (clamped) point from its fall speeds
template <typename FT, bool CLOUD, typename CV>      // FT: the value type (a point or a packed pair); CV: CloudVelConsts of its scalar type
__device__ __forceinline__ SedFlux<FT> sed_fluxes(const CV &cv, FT r_, FT ql, FT nl, FT qr, FT nr, FT vt_n, FT vt_m) {
    SedFlux<FT> f;
    f.q_rai = (r_ * qr) * vt_m;
    f.n_rai = (r_ * nr) * vt_n;
    f.q_lcl = FT(0);
    f.n_lcl = FT(0);
    if constexpr (CLOUD) {
        FT vc_n, vc_m;
        sb2006_cloud_velocity(cv, ql, r_, r_ * nl, vc_n, vc_m);
        f.q_lcl = (r_ * ql) * vc_m;
        f.n_lcl = (r_ * nl) * vc_n;
    }
    return f;
}

// flux-only evaluation of a raw point (the point after a workgroup's tile): clamps, rain PSD, fall speeds — the inline functions the
// point function itself calls, on the same operands
template <typename FT, bool LIMITED, int VEL, bool CLOUD, typename C>
__device__ __forceinline__ SedFlux<FT> sed_fluxes_of_point(const C &c, const CloudVelConsts<FT> &cv, FT rho, FT q_lcl, FT n_lcl,
                                                          FT q_rai, FT n_rai) {
    using M = Math<FT>;
    const FT eps = M::eps();
    const FT r_ = max0(rho), ql = max0(q_lcl), nl = max0(n_lcl);
    const FT qr = max0(q_rai), nr = max0(n_rai);
    const FT rs_rho = M::rsqrt(r_);
    const FT N_rai = r_ * nr;
    const FT L_rai = r_ * M::max(qr, eps);
    const SbRainPsd<FT> psd = sb2006_rain_psd<FT, LIMITED>(c, L_rai, M::max(N_rai, eps));
    FT vt_n, vt_m;
    sb2006_rain_velocity<FT, LIMITED, VEL>(c, r_, rs_rho, psd.l2_lam, N_rai < eps, qr < eps, vt_n, vt_m);
    SedFlux<FT> f = sed_fluxes<FT, CLOUD>(cv, r_, ql, nl, qr, nr, vt_n, vt_m);
    if (any_nan(rho, q_lcl, n_lcl, q_rai, n_rai)) f.q_rai = f.n_rai = f.q_lcl = f.n_lcl = M::nan();
    return f;
}

#ifndef CMX_F32_PACKED_COLUMN
#define CMX_F32_PACKED_COLUMN 1        // A/B switch: 0 = one point at a time (rounds 2–4)
#endif
// waves per SIMD the Float32 instantiation is compiled for: 5 (96 VGPRs) for the one-point form of rounds 2–4; the packed form holds two points' intermediates in
// register pairs and spills at 96 (2–25 VGPRs to scratch in the SB2006 / Chen instantiations): 4 (128 VGPRs)
#ifndef CMX_COL_F32_WAVES
#define CMX_COL_F32_WAVES (CMX_F32_PACKED_COLUMN ? 4 : 5)
#endif
#ifndef CMX_COL_F64_WAVES
#define CMX_COL_F64_WAVES 2      // A/B switch: waves per SIMD the Float64 instantiation is compiled for
#endif
template <typename FT, bool LIMITED, int VEL, bool CLOUD, int VEC, int BS, bool INTPOW = false>
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(sizeof(FT) == 4 ? CMX_COL_F32_WAVES : CMX_COL_F64_WAVES))) void sb2006_column_kernel(const SbConsts<FT> c, const CloudVelConsts<FT> cv, const SbColIO<FT> io,
                                                           const int64_t first, const int64_t nvec) {
    using M = Math<FT>;
    // fluxes of every lane's FIRST point, + slot BS for the point that follows the tile
    __shared__ __align__(16) FT halo[BS + 1][4];
    // Tiles OVERLAP by one lane (round 4): workgroup b owns the vectors [b(BS−1), (b+1)(BS−1)); its lane BS−1 evaluates the first vector of the
    // next tile like any other lane and only supplies the flux its first point sends down to lane BS−2.  The redundant work is 1/(BS−1) of the
    // launch; evaluating that one point on its own (rounds 2–3) made the last wave of every workgroup issue the whole flux function once
    // more with one lane active — 12 of the Float32 kernel's 302 instructions per point (PMC).  Only the LAST workgroup of a launch, whose
    // last owned vector is followed by another launch's (or no) point, still takes that path.
    const int64_t tile0 = (int64_t)blockIdx.x * (BS - 1);
    const int64_t v = tile0 + threadIdx.x;
    const bool active = v < nvec;
    const int64_t nvalid = nvec - tile0 < BS ? nvec - tile0 : BS;       // active lanes of this tile (≥ 1)
    const bool owner = threadIdx.x < BS - 1;                            // lane BS−1 never stores: its vector belongs to the next workgroup
    const int64_t i0 = first + v * VEC;                                 // flat index of the lane's first point

    // all global loads first: the lane's seven 16-byte vectors, and (last lane only) the five scalars of the point after the tile
    FT rho[VEC], T[VEC], q_tot[VEC], q_lcl[VEC], n_lcl[VEC], q_rai[VEC], n_rai[VEC];
    if (active) {
        load_col<FT, VEC, true>(io.rho + first, v, rho);
        load_col<FT, VEC, true>(io.T + first, v, T);
        load_col<FT, VEC, true>(io.q_tot + first, v, q_tot);
        load_col<FT, VEC, true>(io.q_lcl + first, v, q_lcl);
        load_col<FT, VEC, true>(io.n_lcl + first, v, n_lcl);
        load_col<FT, VEC, true>(io.q_rai + first, v, q_rai);
        load_col<FT, VEC, true>(io.n_rai + first, v, n_rai);
    }
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly (every lane of the workgroup reaches the barrier inside); no-op for Float32
    // the point after the tile (another workgroup's, or another launch's, first point) is evaluated here from the raw columns —
    // before the lane's own points, while nothing else is live in registers
    if (nvalid < BS && threadIdx.x == BS - 1) {        // last workgroup of the launch only (lane BS−1 has no vector there)
        const int64_t e = first + (tile0 + nvalid) * VEC;
        SedFlux<FT> f{FT(0), FT(0), FT(0), FT(0)};
        if (e < io.n) f = sed_fluxes_of_point<FT, LIMITED, VEL, CLOUD>(front_consts<FT>(c), cv, io.rho[e], io.q_lcl[e], io.n_lcl[e], io.q_rai[e], io.n_rai[e]);
        halo[BS][0] = f.q_rai;
        halo[BS][1] = f.n_rai;
        halo[BS][2] = f.q_lcl;
        halo[BS][3] = f.n_lcl;
    }
    // per point: A = tendency − (own flux)·g (final but for the inflow), g = 1/(ρ Δz) and the flux the cell below receives
    FT A[VEC][4], g[VEC];
    SedFlux<FT> F[VEC];
    int64_t col = 0;
    int32_t lev = 0;
    if (active) {
        // level of the lane's first point: i0 = col·n_lev + k
        if (io.n < ((int64_t)1 << 32)) {             // wave-uniform: one v_mul_hi_u32 (≈ 7 instructions against ≈ 40 of the Float64 quotient)
            const uint32_t c32 = fastdiv((uint32_t)i0, io.lev_div);
            col = c32;
            lev = (int32_t)((uint32_t)i0 - c32 * (uint32_t)io.n_lev);
        } else {                                     // one double multiply + fix-up per lane; exact below 2^53
            col = (int64_t)((double)i0 * io.inv_n_lev);
            int64_t k64 = i0 - col * io.n_lev;
            if (k64 < 0) { --col; k64 += io.n_lev; }
            if (k64 >= io.n_lev) { ++col; k64 -= io.n_lev; }
            lev = (int32_t)k64;
        }
        int32_t lv = lev;
        // one value of the value type VT at a time: a point, or — Float32 with four points per lane — a PAIR of points in packed arithmetic (cmx_math.hpp f32x2)
        constexpr int L = (sizeof(FT) == 4 && VEC % 2 == 0 && CMX_F32_PACKED_COLUMN && VEL != VEL_CHEN_GEN) ? 2 : 1;    // (the run-time Γ of the general Chen instantiation is an OCML call per lane: nothing to pack, 100 spilled VGPRs)
        if constexpr (L == 1) {      // one point at a time — rounds 2–4 verbatim (see sb2006_tendencies_kernel)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                // clamp_to_nonneg — BMT:828-837 (T is not clamped)
                const FT r_ = max0(rho[k]), qt = max0(q_tot[k]), ql = max0(q_lcl[k]);
                const FT qr = max0(q_rai[k]), nl = max0(n_lcl[k]), nr = max0(n_rai[k]);
                const bool poisoned = any_nan(rho[k], q_tot[k], q_lcl[k], n_lcl[k], q_rai[k], n_rai[k], T[k]);
                const SbRates<FT> p = sb2006_point<FT, LIMITED, VEL, false, INTPOW>(front_consts<FT>(c), r_, T[k], qt, ql, qr, r_ * nl, r_ * nr, nl, nr);
                F[k] = sed_fluxes<FT, CLOUD>(cv, r_, ql, nl, qr, nr, p.vt_n, p.vt_m);
                g[k] = io.inv_dz[lv] * p.inv_rho;                                  // 1/(ρ_k Δz_k)
                if (++lv == io.n_lev) lv = 0;
                // sums of warm_rain_tendencies_2m — BMT:738-779 (as in sb2006_tendencies_kernel), minus the outflow through the lower face
                A[k][0] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
                A[k][1] = M::fma(p.lsc_plus_au + p.ac_dN_lcl, p.inv_rho, p.na_lcl);
                A[k][2] = M::fma(-F[k].q_rai, g[k], (p.evq + p.au_dq_rai) + p.ac_dq_rai);
                A[k][3] = M::fma(-F[k].n_rai, g[k], M::fma(((p.evN + p.au_dN_rai) + p.rsc) + p.rbr, p.inv_rho, p.na_rai));
                if constexpr (CLOUD) {
                    A[k][0] = M::fma(-F[k].q_lcl, g[k], A[k][0]);
                    A[k][1] = M::fma(-F[k].n_lcl, g[k], A[k][1]);
                }
                // NaN rule (one rule for every path — ADVICE r02): a NaN in ANY input poisons the point's four tendencies; the FLUXES depend on
                // (ρ, q_lcl, n_lcl, q_rai, n_rai) only — exactly the operands of sed_fluxes_of_point, which evaluates the point after the tile —
                // so a NaN in T or q_tot does not reach the cell below, wherever the tile boundary falls
                if (poisoned) A[k][0] = A[k][1] = A[k][2] = A[k][3] = M::nan();
                if (any_nan(rho[k], q_lcl[k], n_lcl[k], q_rai[k], n_rai[k])) F[k].q_rai = F[k].n_rai = F[k].q_lcl = F[k].n_lcl = M::nan();
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
                const VT rho_k = val(rho), T_k = val(T), qt_k = val(q_tot), ql_k = val(q_lcl), nl_k = val(n_lcl), qr_k = val(q_rai), nr_k = val(n_rai);
                // clamp_to_nonneg — BMT:828-837 (T is not clamped)
                const VT r_ = max0(rho_k), qt = max0(qt_k), ql = max0(ql_k);
                const VT qr = max0(qr_k), nl = max0(nl_k), nr = max0(nr_k);
                const typename MV::Mask poisoned = nan_mask(rho_k, qt_k, ql_k, nl_k, qr_k, nr_k, T_k);
                const SbRates<VT> p = sb2006_point<VT, LIMITED, VEL, false, INTPOW>(front_consts<FT, (L == 2 && CMX_F32_PACKED_PHASE_CONSTS)>(c), r_, T_k, qt, ql, qr, r_ * nl, r_ * nr, nl, nr);
                SedFlux<VT> Fk = sed_fluxes<VT, CLOUD>(cv, r_, ql, nl, qr, nr, p.vt_n, p.vt_m);
                VT gk;                                                             // 1/(ρ_k Δz_k)
                if constexpr (L == 2) {
                    const FT dz0 = io.inv_dz[lv];
                    if (++lv == io.n_lev) lv = 0;
                    const FT dz1 = io.inv_dz[lv];
                    if (++lv == io.n_lev) lv = 0;
                    gk = VT{dz0, dz1} * p.inv_rho;
                } else {
                    gk = io.inv_dz[lv] * p.inv_rho;
                    if (++lv == io.n_lev) lv = 0;
                }
                // sums of warm_rain_tendencies_2m — BMT:738-779 (as in sb2006_tendencies_kernel), minus the outflow through the lower face
                VT Ak[4];
                Ak[0] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
                Ak[1] = MV::fma(p.lsc_plus_au + p.ac_dN_lcl, p.inv_rho, p.na_lcl);
                Ak[2] = MV::fma(-Fk.q_rai, gk, (p.evq + p.au_dq_rai) + p.ac_dq_rai);
                Ak[3] = MV::fma(-Fk.n_rai, gk, MV::fma(((p.evN + p.au_dN_rai) + p.rsc) + p.rbr, p.inv_rho, p.na_rai));
                if constexpr (CLOUD) {
                    Ak[0] = MV::fma(-Fk.q_lcl, gk, Ak[0]);
                    Ak[1] = MV::fma(-Fk.n_lcl, gk, Ak[1]);
                }
                // NaN rule (one rule for every path — ADVICE r02): a NaN in ANY input poisons the point's four tendencies; the FLUXES depend on
                // (ρ, q_lcl, n_lcl, q_rai, n_rai) only — exactly the operands of sed_fluxes_of_point, which evaluates the point after the tile —
                // so a NaN in T or q_tot does not reach the cell below, wherever the tile boundary falls
#pragma unroll
                for (int q = 0; q < 4; ++q) Ak[q] = poisoned ? MV::nan() : Ak[q];
                const typename MV::Mask fpoison = nan_mask(rho_k, ql_k, nl_k, qr_k, nr_k);
                Fk.q_rai = fpoison ? MV::nan() : Fk.q_rai; Fk.n_rai = fpoison ? MV::nan() : Fk.n_rai;
                Fk.q_lcl = fpoison ? MV::nan() : Fk.q_lcl; Fk.n_lcl = fpoison ? MV::nan() : Fk.n_lcl;
                if constexpr (L == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { A[k][q] = Ak[q].x; A[k + 1][q] = Ak[q].y; }
                    g[k] = gk.x; g[k + 1] = gk.y;
                    F[k] = SedFlux<FT>{Fk.q_rai.x, Fk.n_rai.x, Fk.q_lcl.x, Fk.n_lcl.x};
                    F[k + 1] = SedFlux<FT>{Fk.q_rai.y, Fk.n_rai.y, Fk.q_lcl.y, Fk.n_lcl.y};
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) A[k][q] = Ak[q];
                    g[k] = gk;
                    F[k] = Fk;
                }
            }
        }
        halo[threadIdx.x][0] = F[0].q_rai;
        halo[threadIdx.x][1] = F[0].n_rai;
        halo[threadIdx.x][2] = F[0].q_lcl;
        halo[threadIdx.x][3] = F[0].n_lcl;
    }
    __syncthreads();
    if (!active || !owner) return;
    const int up = (threadIdx.x + 1 < nvalid) ? threadIdx.x + 1 : BS;
    const SedFlux<FT> above{halo[up][0], halo[up][1], halo[up][2], halo[up][3]};
    FT o[4][VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const SedFlux<FT> upf = (k + 1 < VEC) ? F[k + 1 < VEC ? k + 1 : 0] : above;
        const bool top = lev == io.n_lev - 1;
        // inflow through the upper face: the flux leaving the cell above (nothing enters through the model top)
        o[0][k] = A[k][0];
        o[1][k] = A[k][1];
        o[2][k] = top ? A[k][2] : M::fma(upf.q_rai, g[k], A[k][2]);
        o[3][k] = top ? A[k][3] : M::fma(upf.n_rai, g[k], A[k][3]);
        if constexpr (CLOUD) {
            o[0][k] = top ? A[k][0] : M::fma(upf.q_lcl, g[k], A[k][0]);
            o[1][k] = top ? A[k][1] : M::fma(upf.n_lcl, g[k], A[k][1]);
        }
        if (lev == 0 && io.precip) io.precip[col] = F[k].q_rai;             // surface precipitation flux of the column
        if (++lev == io.n_lev) { lev = 0; ++col; }
    }
    store_col<FT, VEC, true>(io.dq_lcl + first, v, o[0]);
    store_col<FT, VEC, true>(io.dn_lcl + first, v, o[1]);
    store_col<FT, VEC, true>(io.dq_rai + first, v, o[2]);
    store_col<FT, VEC, true>(io.dn_rai + first, v, o[3]);
}

// lanes per workgroup: 256 (same-box A/B, round 4, ms per 1e8 points: Float32 64 lanes 0.884, 128 lanes 0.852, 256 lanes 0.834; Float64 2.97 / 2.72 / 2.65 —
// a longer tile halves the share of the overlap lane and of the halo traffic; profiles/r04_ab_sessions.txt, session 17)
#ifndef CMX_COLUMN_BS
#define CMX_COLUMN_BS 256
#endif
constexpr int kColBS = CMX_COLUMN_BS;

template <typename FT, int VEC>
static void launch_column(bool limited, int vel, bool cloud, bool intpow, const SbConsts<FT> &c, const CloudVelConsts<FT> &cv, const SbColIO<FT> &io,
                          int64_t first, int64_t nvec, hipStream_t s) {
    if (nvec <= 0) return;
    const int64_t grid = (nvec + (kColBS - 1) - 1) / (kColBS - 1);      // tiles overlap by one lane (sb2006_column_kernel)
    // intpow: the integer-exponent instantiation of the point function (cmx_sb2006.hpp INTPOW)
#define CMX_LAUNCH(L, V, C)                                                                                                                          \
    do {                                                                                                                                             \
        if (intpow) CMX_LAUNCH_FRONT((sb2006_column_kernel<FT, L, V, C, VEC, kColBS, true>), dim3((unsigned)grid), dim3(kColBS), 0, s, c, cv, io, first, nvec); \
        else CMX_LAUNCH_FRONT((sb2006_column_kernel<FT, L, V, C, VEC, kColBS>), dim3((unsigned)grid), dim3(kColBS), 0, s, c, cv, io, first, nvec);   \
    } while (0)
#define CMX_PICK(L, V) do { if (cloud) CMX_LAUNCH(L, V, true); else CMX_LAUNCH(L, V, false); } while (0)
    if (limited) {
        if (vel == VEL_SB) CMX_PICK(true, VEL_SB);
        else if (vel == VEL_CHEN) CMX_PICK(true, VEL_CHEN);
        else CMX_PICK(true, VEL_CHEN_GEN);
    } else {
        if (vel == VEL_SB) CMX_PICK(false, VEL_SB);
        else if (vel == VEL_CHEN) CMX_PICK(false, VEL_CHEN);
        else CMX_PICK(false, VEL_CHEN_GEN);
    }
#undef CMX_PICK
#undef CMX_LAUNCH
}

template <typename FT, typename WR, typename TH, typename VL, typename ST>
static int32_t column_entry(const WR *wr, const TH *tps, const VL *vel, const ST *cloud_vel, uint32_t flags, int64_t n_col, int32_t n_lev,
                            const FT *inv_dz, const FT *rho, const FT *T, const FT *q_tot, const FT *q_lcl, const FT *n_lcl, const FT *q_rai,
                            const FT *n_rai, FT *dq_lcl, FT *dn_lcl, FT *dq_rai, FT *dn_rai, FT *precip, void *stream) {
    if (!wr || !tps || !vel || n_col < 0 || n_lev < 1) return CMX_ERR_BAD_ARG;
    if (flags & ~(uint32_t)(CMX_SB2006_LIMITED | CMX_VEL_SB2006 | CMX_VEL_CHEN2022)) return CMX_ERR_BAD_ARG;
    const bool sbv = flags & CMX_VEL_SB2006, chv = flags & CMX_VEL_CHEN2022;
    if (sbv == chv) return CMX_ERR_BAD_ARG;                      // exactly one rain fall-speed scheme
    if (n_col > kMaxPoints / n_lev) return CMX_ERR_UNSUPPORTED;  // n_col·n_lev must fit one launch (cmx_launch.hpp)
    const int64_t n = n_col * (int64_t)n_lev;
    if (n == 0) return CMX_OK;
    if (!inv_dz || !rho || !T || !q_tot || !q_lcl || !n_lcl || !q_rai || !n_rai || !dq_lcl || !dn_lcl || !dq_rai || !dn_rai)
        return CMX_ERR_BAD_ARG;
    if ((flags & CMX_SB2006_LIMITED) && !sb_limiters_ok(*wr)) return CMX_ERR_BAD_ARG;      // clamp_ordered needs ordered limiter pairs
    const SbConsts<FT> c = make_sb_consts<FT>(*wr, *tps, vel, (double)Math<FT>::eps_1m());
    const bool intpow = sb_integer_exponents(*wr);
    CloudVelConsts<FT> cv{};
    if (cloud_vel) cv = make_cloud_vel_consts<FT>(wr->seifert_beheng.pdf_c, *cloud_vel);
    const SbColIO<FT> io{rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, dq_lcl, dn_lcl, dq_rai, dn_rai, inv_dz, precip, n, n_lev, 1.0 / (double)n_lev, make_fastdiv((uint32_t)n_lev)};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool limited = flags & CMX_SB2006_LIMITED;
    const int velk = sbv ? VEL_SB : chen_vel_kind<FT>(vel->chen2022);   // fitted Γ(b(ρ)+1) or the general instantiation (cmx_math.hpp ChenGamma)
    constexpr int VEC = sizeof(FT) == 8 ? 1 : Math<FT>::VEC;     // Float64: one point per lane, as in the pointwise kernel
    // alignment dispatch as in cmx_sb2006_warm_rain_tendencies_*: a scalar head up to the common 16-byte boundary, the vector body,
    // a scalar tail; the halo point of each range is read from the full columns, so the three launches compose exactly
    const void *ptrs[] = {rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, dq_lcl, dn_lcl, dq_rai, dn_rai};
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(rho) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (const void *p : ptrs) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(p) & 15u) == mis0);
    const bool cloud = cloud_vel != nullptr;
    if (same_mis && VEC > 1) {
        const int64_t head = std::min<int64_t>(n, mis0 ? (int64_t)((16 - mis0) / sizeof(FT)) : 0);
        const int64_t body = ((n - head) / VEC) * VEC;
        launch_column<FT, 1>(limited, velk, cloud, intpow, c, cv, io, 0, head, s);
        launch_column<FT, VEC>(limited, velk, cloud, intpow, c, cv, io, head, body / VEC, s);
        launch_column<FT, 1>(limited, velk, cloud, intpow, c, cv, io, head + body, n - head - body, s);
    } else {
        launch_column<FT, 1>(limited, velk, cloud, intpow, c, cv, io, 0, n, s);
    }
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_sb2006_column_tendencies_sedimentation_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps, const cmx_rain_vel_f32 *vel,
                                                       const cmx_stokes_vel_f32 *cloud_vel, uint32_t flags, int64_t n_col, int32_t n_lev,
                                                       const float *inv_dz, const float *rho, const float *T, const float *q_tot,
                                                       const float *q_lcl, const float *n_lcl, const float *q_rai, const float *n_rai,
                                                       float *dq_lcl_dt, float *dn_lcl_dt, float *dq_rai_dt, float *dn_rai_dt,
                                                       float *precip_flux, void *stream) {
    return cmx::column_entry<float>(warm_rain, tps, vel, cloud_vel, flags, n_col, n_lev, inv_dz, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai,
                                    dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt, precip_flux, stream);
}
int32_t cmx_sb2006_column_tendencies_sedimentation_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps, const cmx_rain_vel_f64 *vel,
                                                       const cmx_stokes_vel_f64 *cloud_vel, uint32_t flags, int64_t n_col, int32_t n_lev,
                                                       const double *inv_dz, const double *rho, const double *T, const double *q_tot,
                                                       const double *q_lcl, const double *n_lcl, const double *q_rai, const double *n_rai,
                                                       double *dq_lcl_dt, double *dn_lcl_dt, double *dq_rai_dt, double *dn_rai_dt,
                                                       double *precip_flux, void *stream) {
    return cmx::column_entry<double>(warm_rain, tps, vel, cloud_vel, flags, n_col, n_lev, inv_dz, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai,
                                     dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt, precip_flux, stream);
}

}  // extern "C"
