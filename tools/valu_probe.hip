// valu_probe.hip — issue-rate probe for the f32 VALU paths the pointwise kernels use (gfx950):
// cycles per wave64 instruction for v_fma_f32, v_pk_fma_f32, v_mul_f32, v_exp_f32, v_rcp_f32, v_cndmask, v_fma_f64.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe && tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kUnroll = 8;

template <int MODE> __global__ __launch_bounds__(256) void probe(float *out, float a, float b) {
    float x[kUnroll]; f32x2 v[kUnroll]; double d[kUnroll];
    for (int k = 0; k < kUnroll; ++k) { x[k] = threadIdx.x * 1e-3f + k; v[k] = f32x2{x[k], x[k] + 1.0f}; d[k] = x[k]; }
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {
            if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
            if constexpr (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(f32x2{a, a}), "v"(f32x2{b, b}));
            if constexpr (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
            if constexpr (MODE == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]));
            if constexpr (MODE == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k]));
            if constexpr (MODE == 5) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[k]) : "v"(b), "v"(a) : "vcc");
            if constexpr (MODE == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"((double)a), "v"((double)b));
            if constexpr (MODE == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(f32x2{a, a}));
            if constexpr (MODE == 8) asm volatile("v_mul_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %2" : "+v"(x[k]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
    for (int k = 0; k < kUnroll; ++k) s += x[k] + v[k].x + v[k].y + (float)d[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int instr_per_iter, float *out) {
    const int blocks = 256 * 8;   // 8 workgroups (32 waves) per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double instr_per_simd = waves_per_simd * (double)kIters * kUnroll * instr_per_iter;
    printf("%-14s %8.3f ms  -> %.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", 1, out); run<1>("v_pk_fma_f32", 1, out); run<2>("v_mul_f32", 1, out); run<7>("v_pk_mul_f32", 1, out);
    run<3>("v_exp_f32", 1, out); run<4>("v_rcp_f32", 1, out); run<5>("cmp+cndmask", 2, out); run<8>("mul+max", 2, out);
    run<6>("v_fma_f64", 1, out);
    return 0;
}
