// f64_census.cpp — TOOL (not part of libcmx.so): the operation census of the Float64 point functions, for the instruction floor of DESIGN §4.2.
// The device point functions (csrc/cmx_sb2006.hpp, cmx_mp1m.hpp) are templates on the value type; here they are instantiated on a COUNTING type that takes
// the Float64 code paths (sizeof = 8, Math<Cnt>::IS_F64) and tallies every operation the source asks for: transcendental calls by kind, fused multiply-adds,
// multiplies, adds, compares / selects, min / max.  The gates of the point functions are selects, so the tally does not depend on the state.
//   g++ -std=c++17 -O1 -DCMX_HOST_BUILD=1 -I../../include -o f64_census f64_census.cpp && ./f64_census        (tools/f64_floor.py drives it)
#define CMX_HOST_BUILD 1
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

static std::map<std::string, long> g_tally;
static inline void tick(const char *k) { ++g_tally[k]; }

struct Cnt {
    double v;
    Cnt() : v(0) {}
    Cnt(double x) : v(x) {}
    explicit operator double() const { return v; }
};
static_assert(sizeof(Cnt) == 8, "the census takes the Float64 paths");
#define CNT_BIN(op, name)                                                            \
    inline Cnt operator op(Cnt a, Cnt b) { tick(name); return Cnt(a.v op b.v); }     \
    inline Cnt operator op(double a, Cnt b) { tick(name); return Cnt(a op b.v); }    \
    inline Cnt operator op(Cnt a, double b) { tick(name); return Cnt(a.v op b); }
CNT_BIN(+, "add") CNT_BIN(-, "add") CNT_BIN(*, "mul") CNT_BIN(/, "div")
inline Cnt operator-(Cnt a) { return Cnt(-a.v); }                                    // a source modifier on the device: free
inline Cnt &operator+=(Cnt &a, Cnt b) { tick("add"); a.v += b.v; return a; }
inline Cnt &operator-=(Cnt &a, Cnt b) { tick("add"); a.v -= b.v; return a; }
inline Cnt &operator*=(Cnt &a, Cnt b) { tick("mul"); a.v *= b.v; return a; }
#define CNT_CMP(op)                                                                 \
    inline bool operator op(Cnt a, Cnt b) { tick("cmp"); return a.v op b.v; }       \
    inline bool operator op(double a, Cnt b) { tick("cmp"); return a op b.v; }      \
    inline bool operator op(Cnt a, double b) { tick("cmp"); return a.v op b; }
CNT_CMP(<) CNT_CMP(<=) CNT_CMP(>) CNT_CMP(>=) CNT_CMP(==) CNT_CMP(!=)
inline bool isfinite(Cnt a) { tick("cmp"); return std::isfinite(a.v); }

namespace cmx {
inline bool any_nan(Cnt) { tick("cmp"); return false; }                  // one unordered compare per pair of inputs on the device
inline bool any_nan(Cnt, Cnt) { tick("cmp"); return false; }
template <typename... R> inline bool any_nan(Cnt a, Cnt b, R... r) { return (bool)((int)any_nan(a, b) | (int)any_nan(r...)); }
}  // namespace cmx
#include "../../cloudmicrophysics.jl_amd/csrc/cmx_math.hpp"

namespace cmx {
template <> struct Math<Cnt> {
    using Scalar = double;
    using Mask = bool;
    static constexpr bool IS_F64 = true;
    static constexpr int VEC = 2;
    static constexpr double eps() { return Math<double>::eps(); }
    static constexpr double eps_1m() { return Math<double>::eps_1m(); }
    static void prepare() {}
#define CNT_FN1(name, expr) static Cnt name(Cnt x) { tick(#name); return Cnt(expr); }
    CNT_FN1(exp2, std::exp2(x.v)) CNT_FN1(log2, std::log2(x.v)) CNT_FN1(rcp, 1.0 / x.v) CNT_FN1(exp2_fin, std::exp2(x.v)) CNT_FN1(rcp_nz, 1.0 / x.v)
    static Cnt log2_pn(Cnt x) { tick("log2"); return Cnt(std::log2(x.v)); }   // the main path of log2: the cost the lean-cost run measures for it
    CNT_FN1(rcp_nz1, 1.0 / x.v) CNT_FN1(sqrt, std::sqrt(x.v)) CNT_FN1(rsqrt, 1.0 / std::sqrt(x.v)) CNT_FN1(sqrt_pos, std::sqrt(x.v))
    CNT_FN1(rsqrt_pos, 1.0 / std::sqrt(x.v)) CNT_FN1(log1p, std::log1p(x.v)) CNT_FN1(expm1, std::expm1(x.v))
    template <typename A, typename B> static Cnt div(A a, B b) { tick("rcp"); tick("mul"); return Cnt((double)Cnt(a) / (double)Cnt(b)); }
    template <typename A, typename B, typename C> static Cnt fma(A a, B b, C c) { tick("fma"); return Cnt(std::fma((double)Cnt(a), (double)Cnt(b), (double)Cnt(c))); }
    template <typename A, typename B> static Cnt max(A a, B b) { tick("minmax"); return Cnt(std::fmax((double)Cnt(a), (double)Cnt(b))); }
    template <typename A, typename B> static Cnt min(A a, B b) { tick("minmax"); return Cnt(std::fmin((double)Cnt(a), (double)Cnt(b))); }
    static Cnt nan() { return Cnt(__builtin_nan("")); }
};
inline Cnt max0(Cnt x) { tick("minmax"); return Cnt(std::fmax(0.0, x.v)); }
inline Cnt clamp_ordered(Cnt x, Cnt lo, Cnt hi) { tick("minmax"); tick("minmax"); return Cnt(std::fmin(std::fmax(x.v, lo.v), hi.v)); }
inline Cnt clamp_ordered(Cnt x, double lo, double hi) { return clamp_ordered(x, Cnt(lo), Cnt(hi)); }
template <> inline Cnt tgamma_general<Cnt>(Cnt z) { tick("tgamma"); return Cnt(std::tgamma(z.v)); }
}  // namespace cmx
// a select: `cond ? a : b` on the counting type is a plain C++ conditional — count the compares (each gate ends in one select; v_cndmask per 32-bit half)

namespace cmx { namespace lean {
inline Cnt erfc(Cnt x) { tick("erfc"); return Cnt(std::erfc(x.v)); }
inline Cnt pow_m34_pos(Cnt y) { tick("pow_m34_pos"); return Cnt(std::pow(y.v, -0.75)); }
inline Cnt rsqrt(Cnt y) { tick("rsqrt"); return Cnt(1.0 / std::sqrt(y.v)); }
inline Cnt sqrt(Cnt y) { tick("sqrt"); return Cnt(std::sqrt(y.v)); }
} }
inline Cnt erfc(Cnt x) { tick("erfc_ocml"); return Cnt(std::erfc(x.v)); }
namespace cmx { inline Cnt arg_pow_m34(Cnt y) { return lean::pow_m34_pos(y); } }
#include "../../cloudmicrophysics.jl_amd/csrc/cmx_arg.hpp"
#include "../../cloudmicrophysics.jl_amd/csrc/cmx_mp1m.hpp"
#include "../../cloudmicrophysics.jl_amd/csrc/cmx_sb2006.hpp"
#include "../../tests/native/abi_caller_params.h"

using namespace cmx;

static void report(const char *what) {
    std::printf("%s", what);
    for (auto &kv : g_tally) std::printf(" %s=%ld", kv.first.c_str(), kv.second);
    std::printf("\n");
    g_tally.clear();
}

int main() {
    // SB2006 warm rain, limited PSD, SB2006 velocities, integer exponents: the instantiation of the bench line (sb2006 f64)
    {
        const cmx_warm_rain_2m_f64 &wr = WARM_RAIN_2M;
        const cmx_thermo_f64 &tps = THERMO;
        const cmx_rain_vel_f64 &vel = RAIN_VEL;
        const SbConsts<double> c = make_sb_consts<double>(wr, tps, &vel, (double)Math<double>::eps_1m());
        g_tally.clear();
        const Cnt rho(1.0), T(285.0), qt(0.012), ql(1e-3), qr(5e-4), nl(1e8), nr(1e4);
        const Cnt r_ = max0(rho), qt_ = max0(qt), ql_ = max0(ql), nl_ = max0(nl), qr_ = max0(qr), nr_ = max0(nr);
        const SbRates<Cnt> p = sb2006_point<Cnt, true, VEL_SB, false, true>(c, r_, T, qt_, ql_, qr_, r_ * nl_, r_ * nr_, nl_, nr_);
        Cnt y[6];
        y[0] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
        y[1] = Math<Cnt>::fma(p.lsc_plus_au + p.ac_dN_lcl, p.inv_rho, p.na_lcl);
        y[2] = (p.evq + p.au_dq_rai) + p.ac_dq_rai;
        y[3] = Math<Cnt>::fma(((p.evN + p.au_dN_rai) + p.rsc) + p.rbr, p.inv_rho, p.na_rai);
        y[4] = p.vt_n; y[5] = p.vt_m;
        for (int q = 0; q < 6; ++q) y[q] += Cnt(0);      // the NaN poison
        report("sb2006_f64");
    }
    // 1-moment tendencies, default options and exponents: the instantiation of the bench line (mp1m f64)
    {
        const cmx_microphysics_1m_f64 &mp = MICROPHYSICS_1M;
        const cmx_thermo_f64 &tps = THERMO;
        const Mp1mConsts<double> c = make_mp1m_consts<double>(mp, tps, CMX_1M_DEFAULT_OPTIONS, (double)Math<double>::eps_1m());
        g_tally.clear();
        Cnt o[4];
        mp1m_tendencies_point<Cnt, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>(c, Cnt(1.0), Cnt(268.0), Cnt(0.008), Cnt(1e-3), Cnt(2e-4), Cnt(5e-4), Cnt(3e-4), o[0], o[1], o[2], o[3]);
        report("mp1m_f64");
    }
    // ARG-2000, 5 shared modes, activated number only, no sinks: the instantiation of the bench line (arg2000 f64)
    {
        cmx_aerosol_activation_params_f64 ap{};
        ap.M_w = 0.01801528; ap.R = 8.314462618; ap.rho_w = 1000.0; ap.rho_i = 916.7; ap.sigma = 0.072; ap.g = 9.81;
        ap.f1 = 0.5; ap.f2 = 2.5; ap.g1 = 1.0; ap.g2 = 0.25; ap.p1 = 1.5; ap.p2 = 0.75;
        cmx_aerosol_distribution_f64 ad{};
        ad.n_modes = 5;
        for (int k = 0; k < 5; ++k) { ad.modes[k].r_dry = 2e-8 * (k + 1); ad.modes[k].stdev = 1.6 + 0.1 * k; ad.modes[k].N = 1e8 / (k + 1); ad.modes[k].hygroscopicity = 0.5; ad.modes[k].molar_mass_mix = 0.1; }
        cmx_air_properties_f64 aip = WARM_RAIN_2M.air_properties;
        const ArgConsts<double> c = make_arg_consts<double>(ap, ad, aip, THERMO);
        g_tally.clear();
        const ArgOut<Cnt, 5> o = arg_point<Cnt, 5, false>(c, Cnt(285.0), Cnt(9e4), Cnt(0.5), Cnt(0.008), Cnt(0.0), Cnt(0.0), Cnt(0.0), Cnt(0.0), true, false, false);
        (void)o;
        report("arg2000_f64");
    }
    return 0;
}
