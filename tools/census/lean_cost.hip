// lean_cost.hip — TOOL: one kernel per Float64 elementary function of csrc/cmx_lean_f64.hpp, so that its VALU instruction count can be read off the
// listing (tools/f64_floor.py: count of the kernel minus the count of the empty kernel W = -1).  Built like the production Float64 translation units
// (-DCMX_LEAN_COEFS_LIT_TU=1: polynomial coefficients as SGPR literals).
#include <hip/hip_runtime.h>

#include "../../cloudmicrophysics.jl_amd/csrc/cmx_math.hpp"

namespace cmx {
template <int W> __global__ __launch_bounds__(256) void lean_cost_kernel(const double *__restrict__ x, double *__restrict__ y) {
    lean::erfc_tab_fill();
    Math<double>::prepare();
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double v = x[i];
    double r = v;
    if constexpr (W == 0) r = lean::exp2(v);
    if constexpr (W == 1) r = lean::log2(v);
    if constexpr (W == 4) r = lean::rcp(v);
    if constexpr (W == 5) r = lean::sqrt(v);
    if constexpr (W == 6) r = lean::rsqrt(v);
    if constexpr (W == 7) r = lean::expm1(v);
    if constexpr (W == 8) r = lean::log1p(v);
    if constexpr (W == 9) r = lean::erfc(v);
    if constexpr (W == 11) r = lean::exp2_fin(v);
    if constexpr (W == 13) r = lean::rcp_finite(v);
    if constexpr (W == 14) r = lean::rcp_nz(v);
    if constexpr (W == 15) r = lean::sqrt_pos(v);
    if constexpr (W == 16) r = lean::rsqrt_pos(v);
    if constexpr (W == 17) r = lean::pow_m34_pos(v);
    y[i] = r;
}
#define INST(W) template __global__ void lean_cost_kernel<W>(const double *, double *);
INST(-1) INST(0) INST(1) INST(4) INST(5) INST(6) INST(7) INST(8) INST(9) INST(11) INST(13) INST(14) INST(15) INST(16) INST(17)
}  // namespace cmx
