// cmx_common.hip — library queries, error plumbing, launch geometry cache, diagnostic column sums.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "cmx_launch.hpp"
#include "cmx_math.hpp"
#include "cmx_lean_eval.hpp"

namespace cmx {

static thread_local char g_err[256] = "";

void set_hip_error(hipError_t e, const char *where) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}

DeviceInfo device_info() {
    // one entry per device ordinal, each filled exactly once under its own std::once_flag: ctypes / ccall release the host
    // runtime's lock around every cmx_* call, so two host threads (one per stream or device) may arrive here together.
    // CMX_BLOCKS_PER_CU overrides the resident-workgroup factor.
    static DeviceInfo cache[64];
    static std::once_flag once[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DeviceInfo{256, 8};
    std::call_once(once[dev], [dev] {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        int bpc = 8;
        if (const char *e = std::getenv("CMX_BLOCKS_PER_CU")) {
            const int v = std::atoi(e);
            if (v > 0 && v <= 64) bpc = v;
        }
        cache[dev] = DeviceInfo{cus, bpc};
    });
    return cache[dev];
}

// Σ over one column: wave shuffle-reduce (64 lanes) → LDS across the 4 waves → one atomic per block.
template <typename FT>
__global__ __launch_bounds__(kBlock) void column_sum_kernel(const FT *__restrict__ x, const int64_t n, double *sum) {
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc += (double)x[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) t += part[w];
        atomicAdd(sum, t);
    }
}

template <typename FT>
static int32_t column_sums(int32_t ncols, const FT *const *cols, int64_t n, double *sums, void *stream) {
    if (ncols < 0 || n < 0 || (ncols > 0 && (!cols || !sums))) return CMX_ERR_BAD_ARG;
    if (ncols == 0) return CMX_OK;
    for (int32_t k = 0; k < ncols; ++k)
        if (!cols[k]) return CMX_ERR_BAD_ARG;          // validate everything before the first enqueue
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    CMX_HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * (size_t)ncols, s));
    if (n == 0) return CMX_OK;
    const int grid = grid_for(n, kBlock * 8);
    for (int32_t k = 0; k < ncols; ++k) {
        hipLaunchKernelGGL((column_sum_kernel<FT>), dim3(grid), dim3(kBlock), 0, s, cols[k], n, sums + k);
    }
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_lean_eval_f64(int32_t which, int64_t n, const double *x, double *y, void *stream) { return cmx::lean_eval_entry<0>(which, n, x, y, stream); }

int32_t cmx_version(void) { return (CMX_VERSION_MAJOR << 16) | CMX_VERSION_MINOR; }

const char *cmx_last_hip_error(void) { return cmx::g_err; }

int32_t cmx_column_sums_f32(int32_t ncols, const float *const *cols, int64_t n, double *sums, void *stream) {
    return cmx::column_sums<float>(ncols, cols, n, sums, stream);
}
int32_t cmx_column_sums_f64(int32_t ncols, const double *const *cols, int64_t n, double *sums, void *stream) {
    return cmx::column_sums<double>(ncols, cols, n, sums, stream);
}

}  // extern "C"
