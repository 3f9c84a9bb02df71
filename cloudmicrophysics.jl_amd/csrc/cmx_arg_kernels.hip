// cmx_arg_kernels.hip — Abdul-Razzak & Ghan (2000) aerosol activation, fused per thermodynamic state, for
// gfx950; C-ABI entry points of include/cmx.h §(6).
//
// Reference (src = /root/reference/src): AerosolActivation.jl — coeff_of_curvature :35-40,
// critical_supersaturation :107-118, max_supersaturation :138-200, N_activated_per_mode :235-259,
// M_activated_per_mode :294-321 (which evaluates critical_supersaturation and the thermodynamics twice per call).
//
// HBM-bound pointwise map: 4 (up to 8) state columns in, n_modes (+n_modes +1) columns out — 36 B/state for the
// BASELINE config (4 in, 5 modes out, f32).  The aerosol distribution is shared by all states, so everything that
// depends only on a mode (f_i, g_i, ln σ_i terms, N_i, hygroscopicity, r_dry) is folded on the host into per-mode
// constants and lives in SGPRs; per state the mode loop costs one log2 + one exp2 for the S_max sum and one erfc per
// requested output (Float32: t·P₆(t)·e^{−x²}, RELATIVE error ≤ 6e-7 + 1.2e-7·x² — tools/gen_erfc_f32.py; Float64:
// table-driven, cmx_lean_f64.hpp).  Float32 with four states per lane evaluates two PAIRS of states in packed
// arithmetic (cmx_math.hpp f32x2): the point functions are templates on the value type.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "cmx_arg.hpp"
#include "cmx_launch.hpp"
#include "cmx_math.hpp"

namespace cmx {

template <typename FT> struct ArgIO {
    const FT *T, *p, *w, *q_tot, *q_liq, *q_ice, *N_liq, *N_ice;
    FT *N_act[CMX_ARG_MAX_MODES], *M_act[CMX_ARG_MAX_MODES], *S_max;
    FT *N_tot, *M_tot;   // total_N_activated / total_M_activated: Σ over the modes, left to right like Julia's sum of the tuple (AA:355-433)
    bool want_N, want_M;
};

// N_ONLY: the number-activation-only request (N_act columns, no M_act — the BASELINE configuration) as a compile-time fact; with
// the two wants as run-time flags the compiler keeps both erfc chains and their selects alive (1212 → 729 VALU per 4 points).
#ifndef CMX_ARG_BS
#define CMX_ARG_BS 128
#endif
#ifndef CMX_ARG_F32_PACKED
#define CMX_ARG_F32_PACKED 1     // 0: one state at a time (A/B switch)
#endif
constexpr int kArgBS = CMX_ARG_BS;   // lanes per workgroup of the activation kernel: 128 (same-box A/B, f32: 256 → 0.650 ms, 128 → 0.638, 512 → 0.637–0.655; f64 indifferent)
template <typename FT, int NM, bool SINKS, int VEC, bool N_ONLY = false>
__global__ __launch_bounds__(kArgBS) void arg_activation_kernel(const ArgConsts<FT> c, const ArgIO<FT> io, const int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * kArgBS + threadIdx.x;
    FT T[VEC], p[VEC], w[VEC], qt[VEC], ql[VEC] = {}, qi[VEC] = {}, Nl[VEC] = {}, Ni[VEC] = {};
    if (i < nvec) {
        load_col<FT, VEC>(io.T, i, T); load_col<FT, VEC>(io.p, i, p); load_col<FT, VEC>(io.w, i, w); load_col<FT, VEC>(io.q_tot, i, qt);
        if (io.q_liq) load_col<FT, VEC>(io.q_liq, i, ql);
        if (io.q_ice) load_col<FT, VEC>(io.q_ice, i, qi);
        if constexpr (SINKS) {
            if (io.N_liq) load_col<FT, VEC>(io.N_liq, i, Nl);
            if (io.N_ice) load_col<FT, VEC>(io.N_ice, i, Ni);
        }
    }
    if constexpr (sizeof(FT) == 8 && CMX_ARG_LEAN_ERFC) lean::erfc_tab_fill();   // published by the barrier inside prepare()
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly (every lane of the workgroup reaches the barrier inside); no-op for Float32
    if (i >= nvec) return;
    FT sm[VEC], na[NM][VEC], ma[NM][VEC];
    // one value of the value type VT at a time: a state, or — Float32 with four states per lane — a PAIR of states in packed arithmetic (cmx_math.hpp f32x2):
    // the same IEEE operations in the same order, so the one-state launches of an unaligned head / tail give the same bits
    constexpr int L = (sizeof(FT) == 4 && VEC % 2 == 0 && CMX_HAVE_PACKED && CMX_ARG_F32_PACKED) ? 2 : 1;
    using VT = std::conditional_t<L == 2, f32x2, FT>;
    const bool want_smax = io.S_max != nullptr;
#pragma unroll
    for (int k = 0; k < VEC; k += L) {
        auto val = [k](const FT (&a)[VEC]) -> VT {
            if constexpr (L == 2) return VT{a[k], a[k + 1]};
            else return a[k];
        };
        auto put = [k](FT (&a)[VEC], VT x) {
            if constexpr (L == 2) { a[k] = x.x; a[k + 1] = x.y; }
            else a[k] = x;
        };
        // (number AND mass, Float32: 7 constants per mode + the packed literals overflow the SGPR file — 30–54 spilled to VGPR lanes; that instantiation reads
        // its constants through the kernel-argument pointer, phase by phase)
        const ArgOut<VT, NM> o = arg_point<VT, NM, SINKS>(front_consts<float, (sizeof(FT) == 4 && !N_ONLY)>(c), val(T), val(p), val(w), val(qt), val(ql), val(qi),
                                                         val(Nl), val(Ni), N_ONLY ? true : io.want_N, N_ONLY ? false : io.want_M, want_smax);
        put(sm, o.smax);
#pragma unroll
        for (int j = 0; j < NM; ++j) { put(na[j], o.n[j]); put(ma[j], o.m[j]); }
    }
    if (io.S_max) store_col<FT, VEC>(io.S_max, i, sm);
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        if ((N_ONLY || io.want_N) && io.N_act[j]) store_col<FT, VEC>(io.N_act[j], i, na[j]);
        if constexpr (!N_ONLY) {
            if (io.want_M && io.M_act[j]) store_col<FT, VEC>(io.M_act[j], i, ma[j]);
        }
    }
    if (io.N_tot) {
        FT t[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            t[k] = na[0][k];
#pragma unroll
            for (int j = 1; j < NM; ++j) t[k] += na[j][k];
        }
        store_col<FT, VEC>(io.N_tot, i, t);
    }
    if constexpr (!N_ONLY) {
        if (io.M_tot) {
            FT t[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                t[k] = ma[0][k];
#pragma unroll
                for (int j = 1; j < NM; ++j) t[k] += ma[j][k];
            }
            store_col<FT, VEC>(io.M_tot, i, t);
        }
    }
}

template <typename FT, int NM, bool SINKS>
static void launch_arg(const ArgConsts<FT> &c, const ArgIO<FT> &io0, int64_t n, const void *const *ptrs, int nptrs, hipStream_t s) {
    constexpr int VEC = Math<FT>::VEC;
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(io0.T) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (int k = 0; k < nptrs; ++k)
        if (ptrs[k]) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(ptrs[k]) & 15u) == mis0);
    auto off = [](auto *p, int64_t lo) { return p ? p + lo : p; };
    auto launch_range = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        ArgIO<FT> io = io0;
        io.T += lo; io.p += lo; io.w += lo; io.q_tot += lo;
        io.q_liq = off(io.q_liq, lo); io.q_ice = off(io.q_ice, lo); io.N_liq = off(io.N_liq, lo); io.N_ice = off(io.N_ice, lo);
        io.S_max = off(io.S_max, lo); io.N_tot = off(io.N_tot, lo); io.M_tot = off(io.M_tot, lo);
        for (int j = 0; j < CMX_ARG_MAX_MODES; ++j) { io.N_act[j] = off(io.N_act[j], lo); io.M_act[j] = off(io.M_act[j], lo); }
        const int64_t nv = count / V;
        const dim3 grid((unsigned)((nv + kArgBS - 1) / kArgBS));
        if (io.want_N && !io.want_M) hipLaunchKernelGGL((arg_activation_kernel<FT, NM, SINKS, V, true>), grid, dim3(kArgBS), 0, s, c, io, nv);
        else hipLaunchKernelGGL((arg_activation_kernel<FT, NM, SINKS, V, false>), grid, dim3(kArgBS), 0, s, c, io, nv);
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
}

template <typename FT, typename AP, typename AD, typename AI, typename TH>
static int32_t arg_entry(const AP *ap, const AD *ad, const AI *aip, const TH *tps, int64_t n, const FT *T, const FT *p,
                         const FT *w, const FT *q_tot, const FT *q_liq, const FT *q_ice, const FT *N_liq, const FT *N_ice,
                         FT *const *N_act, FT *const *M_act, FT *S_max, void *stream, FT *N_tot = nullptr, FT *M_tot = nullptr) {
    if (!ap || !ad || !aip || !tps || n < 0 || ad->n_modes < 1 || ad->n_modes > CMX_ARG_MAX_MODES) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!T || !p || !w || !q_tot) return CMX_ERR_BAD_ARG;
    const ArgConsts<FT> c = make_arg_consts<FT>(*ap, *ad, *aip, *tps);
    ArgIO<FT> io{};
    io.T = T; io.p = p; io.w = w; io.q_tot = q_tot; io.q_liq = q_liq; io.q_ice = q_ice; io.N_liq = N_liq; io.N_ice = N_ice;
    io.S_max = S_max; io.want_N = N_act != nullptr || N_tot != nullptr; io.want_M = M_act != nullptr || M_tot != nullptr;
    io.N_tot = N_tot; io.M_tot = M_tot;
    const void *ptrs[8 + 2 * CMX_ARG_MAX_MODES + 3] = {T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, S_max, N_tot, M_tot};
    int np = 11;
    for (int j = 0; j < ad->n_modes; ++j) {
        io.N_act[j] = N_act ? N_act[j] : nullptr;
        io.M_act[j] = M_act ? M_act[j] : nullptr;
        ptrs[np++] = io.N_act[j];
        ptrs[np++] = io.M_act[j];
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool sinks = N_liq || N_ice;
#define CMX_ARG_CASE(NM)                                                             \
    case NM:                                                                         \
        if (sinks) launch_arg<FT, NM, true>(c, io, n, ptrs, np, s);                  \
        else launch_arg<FT, NM, false>(c, io, n, ptrs, np, s);                       \
        break;
    switch (ad->n_modes) {
        CMX_ARG_CASE(1) CMX_ARG_CASE(2) CMX_ARG_CASE(3) CMX_ARG_CASE(4)
        CMX_ARG_CASE(5) CMX_ARG_CASE(6) CMX_ARG_CASE(7) CMX_ARG_CASE(8)
        default: return CMX_ERR_BAD_ARG;
    }
#undef CMX_ARG_CASE
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// ---- aerosol that varies in space: mode descriptors are columns (test/gpu_tests.jl:45-79) --------------------------------
// The reference's own GPU test builds one AerosolDistribution per element.  Per state the kernel streams the modes TWICE through
// registers instead of holding every mode's constants (round 3: 17 values per mode alive across the whole point function — 414
// VGPRs and 322 spilled SGPRs for 8 Float64 modes, one wave per SIMD):
//   pass 1  per mode: load (r_dry, σ, N, hygroscopicity), form the mode's two terms of the S_max sum (AA:170-183) in the log2 domain —
//           c1 = Sm_c⁻² f N^p1 as ONE exponential, c2 = Sm_c⁻² g Sm_c^(2 p2) as one — and accumulate; keep (ln σ, log2 Sm_c, N [, ΣM w]);
//   S_max   arg_smax (the same function as the shared-distribution kernel);
//   pass 2  per mode: u, ½ N erfc(u) [, ½ M erfc(u − fac)] → store.
// 3 (4) kept values per mode and point; a lane owns VEC consecutive points of every column (16-byte loads / stores in Float32).
template <typename FT> struct ArgColIO {
    const FT *r_dry[CMX_ARG_MAX_MODES], *stdev[CMX_ARG_MAX_MODES], *N[CMX_ARG_MAX_MODES], *hyg[CMX_ARG_MAX_MODES], *mmix[CMX_ARG_MAX_MODES];
};
template <typename FT> struct ArgColPar { FT l2_f1, f2_l2e, g1, g2; };   // log2 f1, f2·log2 e, g1, g2 of the ARG fit (f_i = f1 exp(f2 ln²σ), g_i = g1 + g2 ln σ)

// lanes per workgroup, per float type (same-box A/B, round 4, ms per 1e8 states of 5 modes: Float32 64 lanes 1.967, 128 lanes 2.11, 256 lanes 2.085;
// Float64 5.715 / 5.43 / 5.36 — profiles/r04_ab_sessions.txt, session 17).  -DCMX_ARGCOL_BS=n forces one size for both (A/B switch).
#ifdef CMX_ARGCOL_BS
template <typename FT> constexpr int kArgColBS = CMX_ARGCOL_BS;
#else
template <typename FT> constexpr int kArgColBS = sizeof(FT) == 4 ? 64 : 256;
#endif
template <typename FT, int NM, bool SINKS, int VEC, bool N_ONLY>
__global__ __launch_bounds__(kArgColBS<FT>) void arg_activation_columns_kernel(const ArgConsts<FT> c, const ArgColPar<FT> par, const ArgIO<FT> io,
                                                                           const ArgColIO<FT> mc, const int64_t nvec) {
    using M = Math<FT>;
    // workgroup-base addressing (cmx_launch.hpp load_col_wg): `wg0` vectors into every column is uniform, the lane adds its own 32-bit offset
    const int64_t wg0 = (int64_t)blockIdx.x * kArgColBS<FT>;
    const uint32_t lane = threadIdx.x;
    const int64_t i = wg0 + lane;
    auto at = [wg0](auto *q) { return q + wg0 * VEC; };
    FT T[VEC], p[VEC], w[VEC], qt[VEC], ql[VEC] = {}, qi[VEC] = {}, Nl[VEC] = {}, Ni[VEC] = {};
    if (i < nvec) {
        load_col_wg<FT, VEC>(at(io.T), lane, T); load_col_wg<FT, VEC>(at(io.p), lane, p); load_col_wg<FT, VEC>(at(io.w), lane, w); load_col_wg<FT, VEC>(at(io.q_tot), lane, qt);
        if (io.q_liq) load_col_wg<FT, VEC>(at(io.q_liq), lane, ql);
        if (io.q_ice) load_col_wg<FT, VEC>(at(io.q_ice), lane, qi);
        if constexpr (SINKS) {
            if (io.N_liq) load_col_wg<FT, VEC>(at(io.N_liq), lane, Nl);
            if (io.N_ice) load_col_wg<FT, VEC>(at(io.N_ice), lane, Ni);
        }
    }
    if constexpr (sizeof(FT) == 8 && CMX_ARG_LEAN_ERFC) lean::erfc_tab_fill();   // published by the barrier inside prepare()
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly; no-op for Float32
    if (i >= nvec) return;
    const FT ln2 = FT(0.6931471805599453);
    const bool p2_34 = sizeof(FT) == 8 && CMX_ARG_P2_ROOTS && c.p2 == FT(0.75);
    ArgPre<FT> pre[VEC];
    FT sum1[VEC], sum2[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        pre[v] = arg_pre<FT, SINKS>(c, T[v], p[v], w[v], qt[v], ql[v], qi[v], Nl[v], Ni[v]);
        sum1[v] = FT(0); sum2[v] = FT(0);
    }
    FT k_ls[NM][VEC], k_l2sm[NM][VEC], k_N[NM][VEC], k_mm[N_ONLY ? 1 : NM][VEC];
    // ---- pass 1: the S_max sum
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        FT r[VEC], sd[VEC], Nk[VEC], hy[VEC];
        load_col_wg<FT, VEC>(at(mc.r_dry[k]), lane, r); load_col_wg<FT, VEC>(at(mc.stdev[k]), lane, sd); load_col_wg<FT, VEC>(at(mc.N[k]), lane, Nk); load_col_wg<FT, VEC>(at(mc.hyg[k]), lane, hy);
        if constexpr (!N_ONLY) {
            if (mc.mmix[k]) load_col_wg<FT, VEC>(at(mc.mmix[k]), lane, k_mm[k]);
            else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) k_mm[k][v] = FT(0);
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const FT ls = M::log2(sd[v]) * ln2;                                           // ln σ
            // log2 Sm_c = log2(2/√B) − 1.5 log2(3 r_dry) (Sm = Sm_c A^1.5, AA:107-118)
            FT l2sm;
            const FT t3 = FT(3) * r[v];
            if constexpr (sizeof(FT) == 8) l2sm = FT(1) - FT(0.5) * M::log2(hy[v] * (t3 * t3 * t3));      // one table-driven log2; (3r)³ B ≥ 1e-40 is far inside the Float64 range
            else l2sm = M::fma(FT(-1.5), M::log2(t3), M::fma(FT(-0.5), M::log2(hy[v]), FT(1)));          // Float32: (3r)³ would underflow for r < 1e-13
            const FT l2N = M::log2(Nk[v]);
            // c1 = Sm_c⁻² f1 exp(f2 ln²σ) N^p1,  c2 = Sm_c⁻² (g1 + g2 ln σ) Sm_c^(2 p2)
            const FT c1 = M::exp2(M::fma(c.p1, l2N, M::fma(par.f2_l2e * ls, ls, M::fma(FT(-2), l2sm, par.l2_f1))));
            const FT c2 = M::fma(par.g2, ls, par.g1) * M::exp2((FT(2) * c.p2 - FT(2)) * l2sm);
            sum1[v] += c1;
            const FT y = M::fma(pre[v].Q, M::rcp(Nk[v]), FT(1));                            // (η_k + 3ζ)/(3ζ) = 1 + Q/N_k (arg_pre)
            sum2[v] = sum2[v] + c2 * arg_pow_p2<FT>(c, y, p2_34);                           // (mul + add: independent of the order of two modes, cmx_arg.hpp arg_point)
            k_ls[k][v] = ls; k_l2sm[k][v] = l2sm; k_N[k][v] = Nk[v];
        }
        // one mode at a time (Float64: keeps the next mode's table reads and loads behind this mode's arithmetic, like arg_point's erfc loop).  BOTH sums are
        // pinned: with sum2 alone (rounds 3–5) the compiler left every mode's c1 exponential half-evaluated until the sums are used — argument, rounded
        // exponent, table entry: 9 VGPRs per mode alive through pass 1, 157–213 VGPRs and 68 B of scratch at 8 modes
        if constexpr (sizeof(FT) == 8) asm volatile("" : "+v"(sum2[0]), "+v"(sum1[0]) : : "memory");
    }
    // ---- S_max
    FT dl0[VEC], sm[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const FT l2_smax = arg_smax<FT, SINKS>(pre[v], sum1[v], sum2[v], io.S_max != nullptr, sm[v]);
        dl0[v] = pre[v].l2_A15 - l2_smax;                                                // log2(Sm_k / S_max) = log2 Sm_c + dl0
    }
    if (io.S_max) store_col_wg<FT, VEC>(at(io.S_max), lane, sm);
    // ---- pass 2: activated number (and mass) per mode
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        FT na[VEC], ma[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const FT ls = k_ls[k][v];
            const FT u = FT(0.3267527144895157) * M::rcp(ls) * (k_l2sm[k][v] + dl0[v]);   // 2 ln2/(3√2 ln σ) · log2(Sm_k/S_max)   AA:255
            na[v] = FT(0.5) * k_N[k][v] * erfc_dev<FT>(u);                                  // N ½ (1 − erf u)   AA:257
            if constexpr (sizeof(FT) == 8 && CMX_ARG_LEAN_ERFC) asm volatile("" : "+v"(na[v]) : : "memory");
            if constexpr (!N_ONLY) ma[v] = FT(0.5) * k_mm[k][v] * erfc_rel_dev<FT>(u - FT(2.121320343559643) * ls);   // M/2 erfc(u − 3 ln σ √2/2)   AA:319
        }
        if (io.N_act[k]) store_col_wg<FT, VEC>(at(io.N_act[k]), lane, na);
        if constexpr (!N_ONLY) {
            if (io.M_act[k]) store_col_wg<FT, VEC>(at(io.M_act[k]), lane, ma);
        }
    }
}

template <typename FT, int NM, bool SINKS>
static void launch_arg_columns(const ArgConsts<FT> &c, const ArgColPar<FT> &par, const ArgIO<FT> &io0, const ArgColIO<FT> &mc0, int64_t n,
                               const void *const *ptrs, int nptrs, hipStream_t s) {
    // Float64 runs one point per lane: the kernel is VALU-bound there and the second point's kept values would cost a wave per SIMD
    constexpr int VEC = sizeof(FT) == 4 ? Math<FT>::VEC : 1;
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(io0.T) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (int k = 0; k < nptrs; ++k)
        if (ptrs[k]) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(ptrs[k]) & 15u) == mis0);
    auto off = [](auto *p, int64_t lo) { return p ? p + lo : p; };
    auto launch_range = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        ArgIO<FT> io = io0;
        ArgColIO<FT> mc = mc0;
        io.T += lo; io.p += lo; io.w += lo; io.q_tot += lo;
        io.q_liq = off(io.q_liq, lo); io.q_ice = off(io.q_ice, lo); io.N_liq = off(io.N_liq, lo); io.N_ice = off(io.N_ice, lo);
        io.S_max = off(io.S_max, lo);
        for (int j = 0; j < NM; ++j) {
            io.N_act[j] = off(io.N_act[j], lo); io.M_act[j] = off(io.M_act[j], lo);
            mc.r_dry[j] += lo; mc.stdev[j] += lo; mc.N[j] += lo; mc.hyg[j] += lo; mc.mmix[j] = off(mc.mmix[j], lo);
        }
        const int64_t nv = count / V;
        const dim3 grid((unsigned)((nv + kArgColBS<FT> - 1) / kArgColBS<FT>));
        if (!io.want_M) hipLaunchKernelGGL((arg_activation_columns_kernel<FT, NM, SINKS, V, true>), grid, dim3(kArgColBS<FT>), 0, s, c, par, io, mc, nv);
        else hipLaunchKernelGGL((arg_activation_columns_kernel<FT, NM, SINKS, V, false>), grid, dim3(kArgColBS<FT>), 0, s, c, par, io, mc, nv);
    };
    if (same_mis && VEC > 1) {
        const int64_t head = std::min<int64_t>(n, mis0 ? (int64_t)((16 - mis0) / sizeof(FT)) : 0);
        const int64_t body = ((n - head) / VEC) * VEC;
        launch_range(std::integral_constant<int, 1>{}, 0, head);
        launch_range(std::integral_constant<int, VEC>{}, head, body);
        launch_range(std::integral_constant<int, 1>{}, head + body, n - head - body);
    } else {
        launch_range(std::integral_constant<int, 1>{}, 0, n);
    }
}

template <typename FT, typename AP, typename AI, typename TH>
static int32_t arg_columns_entry(const AP *ap, const AI *aip, const TH *tps, int32_t n_modes, int64_t n, const FT *T, const FT *p,
                                 const FT *w, const FT *q_tot, const FT *q_liq, const FT *q_ice, const FT *N_liq, const FT *N_ice,
                                 const FT *const *r_dry, const FT *const *stdev, const FT *const *N_mode, const FT *const *hyg,
                                 const FT *const *mmix, FT *const *N_act, FT *const *M_act, FT *S_max, void *stream) {
    if (!ap || !aip || !tps || n < 0 || n_modes < 1 || n_modes > CMX_ARG_MAX_MODES) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!T || !p || !w || !q_tot || !r_dry || !stdev || !N_mode || !hyg) return CMX_ERR_BAD_ARG;
    if (M_act && !mmix) return CMX_ERR_BAD_ARG;
    // thermodynamic / parameter constants from the shared builder with an empty distribution
    struct EmptyDist { int32_t n_modes = 0; struct Mode { FT r_dry, stdev, N, hygroscopicity, molar_mass_mix; } modes[1]; } none;
    const ArgConsts<FT> c = make_arg_consts<FT>(*ap, none, *aip, *tps);
    ArgIO<FT> io{};
    io.T = T; io.p = p; io.w = w; io.q_tot = q_tot; io.q_liq = q_liq; io.q_ice = q_ice; io.N_liq = N_liq; io.N_ice = N_ice;
    io.S_max = S_max; io.want_N = N_act != nullptr; io.want_M = M_act != nullptr;
    ArgColIO<FT> mc{};
    const void *ptrs[9 + 7 * CMX_ARG_MAX_MODES] = {T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, S_max};
    int np = 9;
    for (int j = 0; j < n_modes; ++j) {
        if (!r_dry[j] || !stdev[j] || !N_mode[j] || !hyg[j]) return CMX_ERR_BAD_ARG;
        mc.r_dry[j] = r_dry[j]; mc.stdev[j] = stdev[j]; mc.N[j] = N_mode[j]; mc.hyg[j] = hyg[j]; mc.mmix[j] = mmix ? mmix[j] : nullptr;
        io.N_act[j] = N_act ? N_act[j] : nullptr;
        io.M_act[j] = M_act ? M_act[j] : nullptr;
        const void *q[7] = {mc.r_dry[j], mc.stdev[j], mc.N[j], mc.hyg[j], mc.mmix[j], io.N_act[j], io.M_act[j]};
        for (const void *x : q) ptrs[np++] = x;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool sinks = N_liq || N_ice;
    const double l2e = 1.4426950408889634074;
    const ArgColPar<FT> par{(FT)std::log2((double)ap->f1), (FT)((double)ap->f2 * l2e), (FT)ap->g1, (FT)ap->g2};
#define CMX_ARGC_CASE(NM)                                                                     \
    case NM:                                                                                  \
        if (sinks) launch_arg_columns<FT, NM, true>(c, par, io, mc, n, ptrs, np, s);          \
        else launch_arg_columns<FT, NM, false>(c, par, io, mc, n, ptrs, np, s);               \
        break;
    switch (n_modes) {
        CMX_ARGC_CASE(1) CMX_ARGC_CASE(2) CMX_ARGC_CASE(3) CMX_ARGC_CASE(4)
        CMX_ARGC_CASE(5) CMX_ARGC_CASE(6) CMX_ARGC_CASE(7) CMX_ARGC_CASE(8)
        default: return CMX_ERR_BAD_ARG;
    }
#undef CMX_ARGC_CASE
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_arg2000_activation_f32(const cmx_aerosol_activation_params_f32 *ap, const cmx_aerosol_distribution_f32 *ad,
                                   const cmx_air_properties_f32 *aip, const cmx_thermo_f32 *tps, int64_t n, const float *T,
                                   const float *p, const float *w, const float *q_tot, const float *q_liq, const float *q_ice,
                                   const float *N_liq, const float *N_ice, float *const *N_act, float *const *M_act,
                                   float *S_max, void *stream) {
    return cmx::arg_entry<float>(ap, ad, aip, tps, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, N_act, M_act, S_max, stream);
}
int32_t cmx_arg2000_activation_f64(const cmx_aerosol_activation_params_f64 *ap, const cmx_aerosol_distribution_f64 *ad,
                                   const cmx_air_properties_f64 *aip, const cmx_thermo_f64 *tps, int64_t n, const double *T,
                                   const double *p, const double *w, const double *q_tot, const double *q_liq,
                                   const double *q_ice, const double *N_liq, const double *N_ice, double *const *N_act,
                                   double *const *M_act, double *S_max, void *stream) {
    return cmx::arg_entry<double>(ap, ad, aip, tps, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, N_act, M_act, S_max, stream);
}

int32_t cmx_arg2000_activation_columns_f32(const cmx_aerosol_activation_params_f32 *ap, const cmx_air_properties_f32 *aip,
                                           const cmx_thermo_f32 *tps, int32_t n_modes, int64_t n, const float *T, const float *p,
                                           const float *w, const float *q_tot, const float *q_liq, const float *q_ice, const float *N_liq,
                                           const float *N_ice, const float *const *r_dry, const float *const *stdev,
                                           const float *const *N_mode, const float *const *hygroscopicity, const float *const *molar_mass,
                                           float *const *N_act, float *const *M_act, float *S_max, void *stream) {
    return cmx::arg_columns_entry<float>(ap, aip, tps, n_modes, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, r_dry, stdev, N_mode,
                                         hygroscopicity, molar_mass, N_act, M_act, S_max, stream);
}
int32_t cmx_arg2000_activation_columns_f64(const cmx_aerosol_activation_params_f64 *ap, const cmx_air_properties_f64 *aip,
                                           const cmx_thermo_f64 *tps, int32_t n_modes, int64_t n, const double *T, const double *p,
                                           const double *w, const double *q_tot, const double *q_liq, const double *q_ice,
                                           const double *N_liq, const double *N_ice, const double *const *r_dry,
                                           const double *const *stdev, const double *const *N_mode, const double *const *hygroscopicity,
                                           const double *const *molar_mass, double *const *N_act, double *const *M_act, double *S_max,
                                           void *stream) {
    return cmx::arg_columns_entry<double>(ap, aip, tps, n_modes, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, r_dry, stdev, N_mode,
                                          hygroscopicity, molar_mass, N_act, M_act, S_max, stream);
}

// AA.total_N_activated / total_M_activated — src/AerosolActivation.jl:355-433: Σ over the modes, formed in the activation kernel itself
int32_t cmx_arg2000_total_activated_f32(const cmx_aerosol_activation_params_f32 *ap, const cmx_aerosol_distribution_f32 *ad,
                                        const cmx_air_properties_f32 *aip, const cmx_thermo_f32 *tps, int64_t n, const float *T, const float *p,
                                        const float *w, const float *q_tot, const float *q_liq, const float *q_ice, const float *N_liq,
                                        const float *N_ice, float *N_total, float *M_total, void *stream) {
    if (!N_total && !M_total) return CMX_ERR_BAD_ARG;
    return cmx::arg_entry<float>(ap, ad, aip, tps, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, nullptr, nullptr, nullptr, stream, N_total, M_total);
}
int32_t cmx_arg2000_total_activated_f64(const cmx_aerosol_activation_params_f64 *ap, const cmx_aerosol_distribution_f64 *ad,
                                        const cmx_air_properties_f64 *aip, const cmx_thermo_f64 *tps, int64_t n, const double *T, const double *p,
                                        const double *w, const double *q_tot, const double *q_liq, const double *q_ice, const double *N_liq,
                                        const double *N_ice, double *N_total, double *M_total, void *stream) {
    if (!N_total && !M_total) return CMX_ERR_BAD_ARG;
    return cmx::arg_entry<double>(ap, ad, aip, tps, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, nullptr, nullptr, nullptr, stream, N_total, M_total);
}

}  // extern "C"
