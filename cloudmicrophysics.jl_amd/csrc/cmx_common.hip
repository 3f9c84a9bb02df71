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

// Σ of up to CMX_COLUMN_SUMS_MAX_COLS columns — ONE launch for all columns, two stages, NO floating-point atomics: the value is a pure
// function of (column, n), bit-identical from run to run and independent of the device (the decomposition below does not look at the
// CU count).
//   stage 1  grid (CMX_COLUMN_SUMS_PARTIALS, ncols): workgroup b of column k owns the contiguous chunk [b·chunk, (b+1)·chunk) (chunk = n
//            split 1024 ways, rounded up to a multiple of 1024 elements); lane t adds its elements t, t+256, … in index order into four
//            interleaved double accumulators (summed 0+1, 2+3, then the two), the 64 lanes of a wave combine in a fixed shuffle tree, the
//            four waves left to right → workspace[k][b];
//   stage 2  one workgroup per column: lane t holds workspace[k][4t … 4t+3] (added left to right), then the same fixed tree → sums[k].
// Every rounding is the same in every run; only the ORDER OF ADDITION differs from a serial sum (a pairwise-like tree: error growth
// O(log n) instead of O(n)).
template <typename FT> struct ColumnSumArgs {
    const FT *col[CMX_COLUMN_SUMS_MAX_COLS];
};
__device__ __forceinline__ double block_tree_sum(double acc) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) t += part[w];
    }
    return t;     // valid in thread 0
}
template <typename FT>
__global__ __launch_bounds__(kBlock) void column_partials_kernel(const ColumnSumArgs<FT> a, const int64_t n, const int64_t chunk, double *__restrict__ ws) {
    const FT *__restrict__ x = a.col[blockIdx.y];
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    const int64_t hi = lo + chunk < n ? lo + chunk : n;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t i = lo + threadIdx.x;
    for (; i + 3 * kBlock < hi; i += 4 * kBlock) {
        a0 += (double)x[i];
        a1 += (double)x[i + kBlock];
        a2 += (double)x[i + 2 * kBlock];
        a3 += (double)x[i + 3 * kBlock];
    }
    if (i < hi) a0 += (double)x[i];
    if (i + kBlock < hi) a1 += (double)x[i + kBlock];
    if (i + 2 * kBlock < hi) a2 += (double)x[i + 2 * kBlock];
    const double t = block_tree_sum((a0 + a1) + (a2 + a3));
    if (threadIdx.x == 0) ws[(int64_t)blockIdx.y * CMX_COLUMN_SUMS_PARTIALS + blockIdx.x] = t;
}
__global__ __launch_bounds__(kBlock) void column_finish_kernel(const double *__restrict__ ws, double *__restrict__ sums) {
    static_assert(CMX_COLUMN_SUMS_PARTIALS == 4 * kBlock, "stage 2 gives each lane four partials");
    const double *p = ws + (int64_t)blockIdx.x * CMX_COLUMN_SUMS_PARTIALS + 4 * threadIdx.x;
    const double t = block_tree_sum(((p[0] + p[1]) + p[2]) + p[3]);
    if (threadIdx.x == 0) sums[blockIdx.x] = t;
}

template <typename FT>
static int32_t column_sums(int32_t ncols, const FT *const *cols, int64_t n, double *sums, double *workspace, void *stream) {
    if (ncols < 0 || ncols > CMX_COLUMN_SUMS_MAX_COLS || n < 0 || (ncols > 0 && (!cols || !sums))) return CMX_ERR_BAD_ARG;
    if (ncols == 0) return CMX_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n == 0) {                                       // empty columns (their pointers may be NULL): the sums are 0
        CMX_HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * (size_t)ncols, s));
        return CMX_OK;
    }
    ColumnSumArgs<FT> a{};
    for (int32_t k = 0; k < ncols; ++k) {
        if (!cols[k]) return CMX_ERR_BAD_ARG;          // validate everything before the first enqueue
        a.col[k] = cols[k];
    }
    if (!workspace) return CMX_ERR_BAD_ARG;
    int64_t chunk = (n + CMX_COLUMN_SUMS_PARTIALS - 1) / CMX_COLUMN_SUMS_PARTIALS;
    chunk = (chunk + 1023) / 1024 * 1024;
    hipLaunchKernelGGL((column_partials_kernel<FT>), dim3(CMX_COLUMN_SUMS_PARTIALS, (unsigned)ncols), dim3(kBlock), 0, s, a, n, chunk, workspace);
    hipLaunchKernelGGL(column_finish_kernel, dim3((unsigned)ncols), dim3(kBlock), 0, s, workspace, sums);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_lean_eval_f64(int32_t which, int64_t n, const double *x, double *y, void *stream) { return cmx::lean_eval_entry<0>(which, n, x, y, stream); }

int32_t cmx_version(void) { return (CMX_VERSION_MAJOR << 16) | CMX_VERSION_MINOR; }

const char *cmx_last_hip_error(void) { return cmx::g_err; }

int32_t cmx_column_sums_f32(int32_t ncols, const float *const *cols, int64_t n, double *sums, double *workspace, void *stream) {
    return cmx::column_sums<float>(ncols, cols, n, sums, workspace, stream);
}
int32_t cmx_column_sums_f64(int32_t ncols, const double *const *cols, int64_t n, double *sums, double *workspace, void *stream) {
    return cmx::column_sums<double>(ncols, cols, n, sums, workspace, stream);
}

}  // extern "C"
