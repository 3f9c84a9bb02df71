// cmx_mp0m_kernels.hip — the 0-moment entry of bulk_microphysics_tendencies (src/BulkMicrophysicsTendencies.jl:658-680):
// precipitation removal CM0.remove_precipitation and its q_tot derivative (src/Microphysics0M.jl:35-75).
// A pure stream: 12 B/point (f32, qc_0 form), 16 B with the q_vap_sat column; +4 B for the derivative column.  HBM-bound.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "../../include/cmx.h"
#include "cmx_launch.hpp"
#include "cmx_math.hpp"

namespace cmx {

template <typename FT> struct Mp0mIO { const FT *q_lcl, *q_icl, *q_vap_sat; FT *dq_tot_dt, *ddq_dq_tot; };

// The reference's arithmetic is kept operation for operation (clamp, sum, threshold product, subtraction, max, true division) so the
// result is bit-identical to the scalar formula its own tests compare with `==` (test/gpu_tests.jl:115-138).
template <typename FT, int VEC, bool SAT, bool DERIV>
__global__ __launch_bounds__(kBlock) void mp0m_tendencies_kernel(const FT tau_precip, const FT qc_0, const FT S_0, const Mp0mIO<FT> io, const int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nvec) return;
    FT ql[VEC], qi[VEC], qs[VEC], out[VEC], der[VEC];
    load_col<FT, VEC>(io.q_lcl, i, ql);
    load_col<FT, VEC>(io.q_icl, i, qi);
    if constexpr (SAT) load_col<FT, VEC>(io.q_vap_sat, i, qs);
    const FT neg_inv_tau = FT(-1) / tau_precip;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const FT qc = Math<FT>::max(ql[k], FT(0)) + Math<FT>::max(qi[k], FT(0));      // UT.clamp_to_nonneg, BMT:662-663
        const FT thr = SAT ? S_0 * qs[k] : qc_0;
        out[k] = -Math<FT>::max(FT(0), qc - thr) / tau_precip;                         // CM0:35-46
        der[k] = qc > thr ? neg_inv_tau : FT(0);                                        // CM0:64-75
        if (SAT ? any_nan(ql[k], qi[k], qs[k]) : any_nan(ql[k], qi[k])) out[k] = Math<FT>::nan();   // max(0, NaN) = NaN in the reference
    }
    store_col<FT, VEC>(io.dq_tot_dt, i, out);
    if constexpr (DERIV) store_col<FT, VEC>(io.ddq_dq_tot, i, der);
}

template <typename FT, typename PR>
static int32_t mp0m_entry(const PR *p, int64_t n, const FT *q_lcl, const FT *q_icl, const FT *q_vap_sat, FT *dq_tot_dt, FT *ddq_dq_tot,
                          void *stream) {
    if (!p || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!q_lcl || !q_icl || !dq_tot_dt) return CMX_ERR_BAD_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int VEC = Math<FT>::VEC;
    const void *ptrs[] = {q_lcl, q_icl, q_vap_sat, dq_tot_dt, ddq_dq_tot};
    bool vec_ok = true;
    for (const void *q : ptrs) vec_ok = vec_ok && (!q || aligned16(q));
    auto launch = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        const Mp0mIO<FT> io{q_lcl + lo, q_icl + lo, q_vap_sat ? q_vap_sat + lo : nullptr, dq_tot_dt + lo, ddq_dq_tot ? ddq_dq_tot + lo : nullptr};
        const int64_t nvec = count / V;
        const dim3 grid((unsigned)((nvec + kBlock - 1) / kBlock));
#define CMX_LAUNCH(S, D) hipLaunchKernelGGL((mp0m_tendencies_kernel<FT, V, S, D>), grid, dim3(kBlock), 0, s, p->tau_precip, p->qc_0, p->S_0, io, nvec)
        if (q_vap_sat) { if (ddq_dq_tot) CMX_LAUNCH(true, true); else CMX_LAUNCH(true, false); }
        else           { if (ddq_dq_tot) CMX_LAUNCH(false, true); else CMX_LAUNCH(false, false); }
#undef CMX_LAUNCH
    };
    const int64_t body = vec_ok ? (n / VEC) * VEC : 0;
    launch(std::integral_constant<int, VEC>{}, 0, body);
    launch(std::integral_constant<int, 1>{}, body, n - body);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {
int32_t cmx_mp0m_tendencies_f32(const cmx_parameters_0m_f32 *p, int64_t n, const float *q_lcl, const float *q_icl, const float *q_vap_sat,
                                float *dq_tot_dt, float *ddq_dq_tot, void *stream) {
    return cmx::mp0m_entry<float>(p, n, q_lcl, q_icl, q_vap_sat, dq_tot_dt, ddq_dq_tot, stream);
}
int32_t cmx_mp0m_tendencies_f64(const cmx_parameters_0m_f64 *p, int64_t n, const double *q_lcl, const double *q_icl, const double *q_vap_sat,
                                double *dq_tot_dt, double *ddq_dq_tot, void *stream) {
    return cmx::mp0m_entry<double>(p, n, q_lcl, q_icl, q_vap_sat, dq_tot_dt, ddq_dq_tot, stream);
}
}
