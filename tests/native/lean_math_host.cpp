// Host build of csrc/cmx_lean_f64.hpp for tests/test_lean_math.py: evaluates each lean function on an array.
#include <cstdint>
#include "../../cloudmicrophysics.jl_amd/csrc/cmx_lean_f64.hpp"
extern "C" void lean_eval(int which, int64_t n, const double *x, double *y) {
    namespace L = cmx::lean;
    for (int64_t i = 0; i < n; ++i) {
        switch (which) {
            case 0: y[i] = L::exp2(x[i]); break;
            case 1: y[i] = L::log2(x[i]); break;
            case 2: y[i] = L::exp(x[i]); break;
            case 3: y[i] = L::log(x[i]); break;
            case 4: y[i] = L::rcp(x[i]); break;
            case 5: y[i] = L::sqrt(x[i]); break;
            case 6: y[i] = L::rsqrt(x[i]); break;
            case 7: y[i] = L::expm1(x[i]); break;
            case 8: y[i] = L::log1p(x[i]); break;
            case 9: y[i] = L::erfc(x[i]); break;
            case 10: y[i] = L::lgamma_pos(x[i]); break;
            case 11: y[i] = L::exp2_fin(x[i]); break;
            case 12: y[i] = L::exp_fin(x[i]); break;
            case 13: y[i] = L::rcp_finite(x[i]); break;
            case 14: y[i] = L::rcp_nz(x[i]); break;
            case 15: y[i] = L::sqrt_pos(x[i]); break;
            case 16: y[i] = L::rsqrt_pos(x[i]); break;
            case 17: y[i] = L::pow_m34_pos(x[i]); break;
            case 18: y[i] = L::log_pos(x[i]); break;
        }
    }
}
// the register-pinned table-driven forms the P3 quadrature loops use
extern "C" void lean_eval_pinned(int which, int64_t n, const double *x, double *y) {
    namespace L = cmx::lean;
    const L::TabCoefs k = L::tab_coefs();
    for (int64_t i = 0; i < n; ++i) y[i] = which == 2 ? L::exp(x[i], k) : (which == 18 ? L::log_pos(x[i], k) : L::log(x[i], k));
}
