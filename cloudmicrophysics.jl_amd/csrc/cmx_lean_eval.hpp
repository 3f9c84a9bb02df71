// cmx_lean_eval.hpp — diagnostic kernel behind cmx_lean_eval_f64 / cmx_lean_eval_literal_f64: the Float64 elementary functions of
// cmx_lean_f64.hpp evaluated on the device (tests/test_lean_math.py compares them with libm in ulps).  The header is included by TWO
// translation units so that both builds of the polynomial coefficients are measured: TAG 0 in cmx_common.hip (coefficients in LDS) and
// TAG 1 in cmx_icenuc_kernels.hip (a CMX_LEAN_COEFS_LIT_TU unit: SGPR literals — the variant the production Float64 kernels run).
#pragma once
#include <hip/hip_runtime.h>

#include "cmx_launch.hpp"
#include "cmx_math.hpp"

namespace cmx {

template <int TAG>
__global__ __launch_bounds__(kBlock) void lean_eval_kernel(const int which, const int64_t n, const double *__restrict__ x, double *__restrict__ y) {
    lean::erfc_tab_fill();
    Math<double>::prepare();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = v;
    switch (which) {
        case 0: r = lean::exp2(v); break;
        case 1: r = lean::log2(v); break;
        case 2: r = lean::exp(v); break;
        case 3: r = lean::log(v); break;
        case 4: r = lean::rcp(v); break;
        case 5: r = lean::sqrt(v); break;
        case 6: r = lean::rsqrt(v); break;
        case 7: r = lean::expm1(v); break;
        case 8: r = lean::log1p(v); break;
        case 9: r = lean::erfc(v); break;
        case 10: r = lean::lgamma_pos(v); break;
            case 11: r = lean::exp2_fin(v); break;
            case 12: r = lean::exp_fin(v); break;
            case 13: r = lean::rcp_finite(v); break;
            case 14: r = lean::rcp_nz(v); break;
            case 15: r = lean::sqrt_pos(v); break;
            case 16: r = lean::rsqrt_pos(v); break;
            case 17: r = lean::pow_m34_pos(v); break;
            case 18: r = lean::log_pos(v); break;
            case 19: break;      // identity: the kernel's own instructions (tools/f64_floor.py subtracts them)
        default: break;
    }
    y[i] = r;
}

template <int TAG> static int32_t lean_eval_entry(int32_t which, int64_t n, const double *x, double *y, void *stream) {
    if (which < 0 || which > 19 || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;
    if (n == 0) return CMX_OK;
    if (!x || !y) return CMX_ERR_BAD_ARG;
    hipLaunchKernelGGL((lean_eval_kernel<TAG>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), which, n,
                       x, y);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx
