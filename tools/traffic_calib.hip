// traffic_calib.hip — what do FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count in the access patterns this library uses?
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/traffic_calib.hip -o tools/traffic_calib
//   rocprofv3 --pmc FETCH_SIZE -- tools/traffic_calib      (and WRITE_SIZE, and the raw TCC_EA0_* request counters, one pass each:
//   tools/traffic_calib.sh runs the passes and tools/traffic_calib_summary.py tabulates counter ÷ known bytes per kernel)
//
// MI355X_MICROARCH.md §HBM calibrates ONE pattern — a wide coalesced streaming read (16 B per lane) reports exactly half its bytes — and says
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  The streaming kernels of
// this library are that pattern; the P3 collision kernels are not: 8 lanes share one state, so a wave's load of one column touches 8 consecutive
// elements (64 B in Float64, 32 B in Float32), and the one-launch 2M + P3 form writes 32 consecutive elements from half of one wave per
// workgroup.  Each kernel below moves `bytes` exactly once in one of those patterns; every kernel has its own name so that the per-dispatch
// counter rows can be grouped.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// ---- reads: the sum is stored only if it equals a value it cannot have, so the loads stay and nothing is written --------------------------------
template <typename V> __device__ __forceinline__ float hsum(V v);
template <> __device__ __forceinline__ float hsum<float>(float v) { return v; }
template <> __device__ __forceinline__ float hsum<f2>(f2 v) { return v.x + v.y; }
template <> __device__ __forceinline__ float hsum<f4>(f4 v) { return (v.x + v.y) + (v.z + v.w); }

template <typename V> __device__ __forceinline__ void read_stream(const V *in, float *sink, int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    const float s = hsum<V>(in[i]);
    if (s == -1.2345e30f) sink[0] = s;
}
__global__ __launch_bounds__(256) void calib_read_16B_per_lane(const f4 *in, float *sink, int64_t nvec) { read_stream<f4>(in, sink, nvec); }
__global__ __launch_bounds__(256) void calib_read_8B_per_lane(const f2 *in, float *sink, int64_t nvec) { read_stream<f2>(in, sink, nvec); }
__global__ __launch_bounds__(256) void calib_read_4B_per_lane(const float *in, float *sink, int64_t nvec) { read_stream<float>(in, sink, nvec); }

// the collision kernels' input pattern: lane l of a 256-lane workgroup reads element 32·block + l/8 of each of NCOL columns (8 lanes one address)
template <typename T, int NCOL> __device__ __forceinline__ void read_group8(const T *in, float *sink, int64_t n, int64_t col_stride) {
    const int64_t i = (int64_t)blockIdx.x * 32 + threadIdx.x / 8;
    if (i >= n) return;
    float s = 0;
#pragma unroll
    for (int k = 0; k < NCOL; ++k) s += (float)in[(int64_t)k * col_stride + i];
    if (s == -1.2345e30f) sink[0] = s;
}
__global__ __launch_bounds__(256) void calib_read_group8_f64_12col(const double *in, float *sink, int64_t n, int64_t cs) { read_group8<double, 12>(in, sink, n, cs); }
__global__ __launch_bounds__(256) void calib_read_group8_f32_12col(const float *in, float *sink, int64_t n, int64_t cs) { read_group8<float, 12>(in, sink, n, cs); }
__global__ __launch_bounds__(256) void calib_read_group8_f64_1col(const double *in, float *sink, int64_t n, int64_t cs) { read_group8<double, 1>(in, sink, n, cs); }

// ---- writes ---------------------------------------------------------------------------------------------------------------------------------------
template <typename V> __device__ __forceinline__ void write_stream(V *out, int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    V v;
    __builtin_memset(&v, 0, sizeof v);
    reinterpret_cast<float *>(&v)[0] = (float)i;
    out[i] = v;
}
__global__ __launch_bounds__(256) void calib_write_16B_per_lane(f4 *out, int64_t nvec) { write_stream<f4>(out, nvec); }
__global__ __launch_bounds__(256) void calib_write_8B_per_lane(f2 *out, int64_t nvec) { write_stream<f2>(out, nvec); }
__global__ __launch_bounds__(256) void calib_write_4B_per_lane(float *out, int64_t nvec) { write_stream<float>(out, nvec); }
// the one-launch 2M + P3 form's output pattern: the first 32 lanes of a 256-lane workgroup write elements 32·block … 32·block + 31 of NCOL columns
template <typename T, int NCOL> __device__ __forceinline__ void write_first32(T *out, int64_t n, int64_t col_stride) {
    const int64_t i = (int64_t)blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x >= 32 || i >= n) return;
#pragma unroll
    for (int k = 0; k < NCOL; ++k) out[(int64_t)k * col_stride + i] = (T)(i + k);
}
__global__ __launch_bounds__(256) void calib_write_first32_f64_8col(double *out, int64_t n, int64_t cs) { write_first32<double, 8>(out, n, cs); }
__global__ __launch_bounds__(256) void calib_write_first32_f32_8col(float *out, int64_t n, int64_t cs) { write_first32<float, 8>(out, n, cs); }
// one lane per 8-lane group writes (the two-launch collision entries: lane 0 of each group stores its state's sums)
template <typename T, int NCOL> __device__ __forceinline__ void write_group8(T *out, int64_t n, int64_t col_stride) {
    const int64_t i = (int64_t)blockIdx.x * 32 + threadIdx.x / 8;
    if (threadIdx.x % 8 != 0 || i >= n) return;
#pragma unroll
    for (int k = 0; k < NCOL; ++k) out[(int64_t)k * col_stride + i] = (T)(i + k);
}
__global__ __launch_bounds__(256) void calib_write_group8_f64_8col(double *out, int64_t n, int64_t cs) { write_group8<double, 8>(out, n, cs); }
__global__ __launch_bounds__(256) void calib_write_group8_f32_8col(float *out, int64_t n, int64_t cs) { write_group8<float, 8>(out, n, cs); }

__global__ void calib_fill(float *p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023) * 1e-3f;
}

int main(int argc, char **argv) {
    // bytes moved by every kernel: 768 MiB by default — three times the 256-MiB Infinity Cache, so that a launch cannot find its data on the die
    const int64_t bytes = (argc > 1 ? std::atoll(argv[1]) : 768ll) << 20;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 3;
    char *buf = nullptr;
    float *sink = nullptr;
    CK(hipMalloc(&buf, (size_t)bytes)); CK(hipMalloc(&sink, 256));
    hipLaunchKernelGGL(calib_fill, dim3(8192), dim3(256), 0, 0, reinterpret_cast<float *>(buf), bytes / 4);
    CK(hipDeviceSynchronize());
    auto grid = [](int64_t n, int per) { return dim3((unsigned)((n + per - 1) / per)); };
    std::printf("bytes per launch: %lld; %d launches of each kernel\n", (long long)bytes, reps);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(calib_read_16B_per_lane, grid(bytes / 16, 256), dim3(256), 0, 0, reinterpret_cast<const f4 *>(buf), sink, bytes / 16);
        hipLaunchKernelGGL(calib_read_8B_per_lane, grid(bytes / 8, 256), dim3(256), 0, 0, reinterpret_cast<const f2 *>(buf), sink, bytes / 8);
        hipLaunchKernelGGL(calib_read_4B_per_lane, grid(bytes / 4, 256), dim3(256), 0, 0, reinterpret_cast<const float *>(buf), sink, bytes / 4);
        { const int64_t n = bytes / 8 / 12; hipLaunchKernelGGL(calib_read_group8_f64_12col, grid(n, 32), dim3(256), 0, 0, reinterpret_cast<const double *>(buf), sink, n, n); }
        { const int64_t n = bytes / 4 / 12; hipLaunchKernelGGL(calib_read_group8_f32_12col, grid(n, 32), dim3(256), 0, 0, reinterpret_cast<const float *>(buf), sink, n, n); }
        { const int64_t n = bytes / 8; hipLaunchKernelGGL(calib_read_group8_f64_1col, grid(n, 32), dim3(256), 0, 0, reinterpret_cast<const double *>(buf), sink, n, n); }
        hipLaunchKernelGGL(calib_write_16B_per_lane, grid(bytes / 16, 256), dim3(256), 0, 0, reinterpret_cast<f4 *>(buf), bytes / 16);
        hipLaunchKernelGGL(calib_write_8B_per_lane, grid(bytes / 8, 256), dim3(256), 0, 0, reinterpret_cast<f2 *>(buf), bytes / 8);
        hipLaunchKernelGGL(calib_write_4B_per_lane, grid(bytes / 4, 256), dim3(256), 0, 0, reinterpret_cast<float *>(buf), bytes / 4);
        { const int64_t n = bytes / 8 / 8; hipLaunchKernelGGL(calib_write_first32_f64_8col, grid(n, 32), dim3(256), 0, 0, reinterpret_cast<double *>(buf), n, n); }
        { const int64_t n = bytes / 4 / 8; hipLaunchKernelGGL(calib_write_first32_f32_8col, grid(n, 32), dim3(256), 0, 0, reinterpret_cast<float *>(buf), n, n); }
        { const int64_t n = bytes / 8 / 8; hipLaunchKernelGGL(calib_write_group8_f64_8col, grid(n, 32), dim3(256), 0, 0, reinterpret_cast<double *>(buf), n, n); }
        { const int64_t n = bytes / 4 / 8; hipLaunchKernelGGL(calib_write_group8_f32_8col, grid(n, 32), dim3(256), 0, 0, reinterpret_cast<float *>(buf), n, n); }
        CK(hipDeviceSynchronize());
    }
    CK(hipGetLastError());
    CK(hipFree(buf)); CK(hipFree(sink));
    return 0;
}
