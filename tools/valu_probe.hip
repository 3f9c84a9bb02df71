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
            // Float64 paths (round 2): what the table-driven exp2 / log2 and the Newton reciprocals are made of
            if constexpr (MODE == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"((double)a));
            if constexpr (MODE == 11) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"((double)b));
            if constexpr (MODE == 12) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
            if constexpr (MODE == 13) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[k]));
            if constexpr (MODE == 14) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[k]));
            if constexpr (MODE == 15) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[k]) : "v"(1));
            if constexpr (MODE == 16) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[k]));
            if constexpr (MODE == 17) asm volatile("v_cvt_i32_f64 %1, %0\n\tv_cvt_f64_i32 %0, %1" : "+v"(d[k]), "+v"(x[k]));
            if constexpr (MODE == 18) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(d[k]), "v"(0x3) : "vcc");
            if constexpr (MODE == 19) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[k]) : "v"((double)b));
            if constexpr (MODE == 20) asm volatile("v_cmp_gt_f64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc" : "+v"(d[k]) : "v"((double)b), "v"(x[k]), "v"(a) : "vcc");
            if constexpr (MODE == 21) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[k]) : "v"((double)a), "v"((double)b));
            if constexpr (MODE == 22) asm volatile("v_log_f32 %0, %0" : "+v"(x[k]));
            if constexpr (MODE == 23) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[k]));
            if constexpr (MODE == 24) asm volatile("v_cvt_f32_f64 %1, %0\n\tv_cvt_f64_f32 %0, %1" : "+v"(d[k]), "+v"(x[k]));
            if constexpr (MODE == 25) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(x[k]) : "s20");
            if constexpr (MODE == 26) asm volatile("v_mov_b32 %0, %1" : "+v"(x[k]) : "s"(a));
            // round 3: encoding / operand-source variants (which of VOP3, an SGPR operand, a literal costs issue time?)
            if constexpr (MODE == 30) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[k]) : "s"(a));
            if constexpr (MODE == 31) asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0" : "+v"(x[k]));
            if constexpr (MODE == 32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(b));
            if constexpr (MODE == 33) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
            if constexpr (MODE == 34) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[k]) : "s"(a), "v"(b));
            if constexpr (MODE == 35) asm volatile("v_fmamk_f32 %0, %0, 0x3f000000, %1" : "+v"(x[k]) : "v"(b));
            if constexpr (MODE == 36) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(b) : );
            if constexpr (MODE == 37) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[k]) : "v"(b));
            if constexpr (MODE == 38) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(x[k]), "v"(b) : "vcc");
            if constexpr (MODE == 39) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(x[k]), "v"(b) : "s20", "s21");
            if constexpr (MODE == 40) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
            if constexpr (MODE == 41) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[k]) : "v"(b));
            if constexpr (MODE == 42) asm volatile("v_mov_b32 %0, %1" : "+v"(x[k]) : "v"(b));
            if constexpr (MODE == 43) asm volatile("v_max_i32 %0, 0, %0" : "+v"(x[k]));
            if constexpr (MODE == 44) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x[k]) : "v"(a));
            if constexpr (MODE == 45) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(x[k]) : "v"(a));
            if constexpr (MODE == 46) asm volatile("v_add_f32_e64 %0, %0, -%1" : "+v"(x[k]) : "v"(b));
            if constexpr (MODE == 47) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "s"((double)a));
            if constexpr (MODE == 48) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "s"((double)a), "v"((double)b));
            if constexpr (MODE == 49) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[k]) : "v"(1));
            if constexpr (MODE == 50) asm volatile("v_mul_f32 %0, %0, %1\n\ts_mul_i32 s22, s22, s23" : "+v"(x[k]) : "v"(a) : "s22");
            if constexpr (MODE == 51) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[k]));
            if constexpr (MODE == 52) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(x[k]) : "v"(b));
            // round 5: packed Float32 with a scalar operand, legacy multiply, 64-bit move
            if constexpr (MODE == 60) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[k]) : "s"(f32x2{a, a}));
            if constexpr (MODE == 61) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(v[k]) : "s"(f32x2{a, b}));
            if constexpr (MODE == 62) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "s"(f32x2{a, a}), "v"(f32x2{b, b}));
            if constexpr (MODE == 63) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(f32x2{b, b}));
            if constexpr (MODE == 64) asm volatile("v_pk_fma_f32 %0, %0, %0, %1" : "+v"(v[k]) : "v"(f32x2{b, b}));
            if constexpr (MODE == 65) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
            if constexpr (MODE == 66) asm volatile("v_mov_b64 %0, %1" : "+v"(d[k]) : "v"((double)b));
            if constexpr (MODE == 67) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[k]) : "s"(f32x2{b, b}));
            if constexpr (MODE == 68) asm volatile("v_fma_f64 %0, %0, %1, 1.0" : "+v"(d[k]) : "s"((double)a));
            if constexpr (MODE == 69) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(v[k]) : "s"(f32x2{a, a}), "s"(f32x2{a, a}));   // same SGPR pair twice
            if constexpr (MODE == 53) { float t = v[k].x; asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %2\n\tv_exp_f32 %1, %1" : "+v"(x[k]), "+v"(t) : "v"(a)); v[k].x = t; }
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
// dependent-issue latency: CH independent v_fma_f64 (or v_fma_f32) chains per wave, W waves per SIMD
template <int CH, bool F64> __global__ __launch_bounds__(64) void chain_probe(float *out, float a, float b, int iters) {
    double d[CH]; float x[CH];
    for (int k = 0; k < CH; ++k) { d[k] = threadIdx.x * 1e-3 + k; x[k] = (float)d[k]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if constexpr (F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"((double)a), "v"((double)b));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
            }
    }
    float s = 0;
    for (int k = 0; k < CH; ++k) s += (float)d[k] + x[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH, bool F64> void run_chain(int waves_per_simd, float *out) {
    const int iters = 2048;
    const int blocks = 256 * 4 * waves_per_simd;   // one-wave workgroups: W per SIMD when they spread evenly
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chain_probe<CH, F64>), dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((chain_probe<CH, F64>), dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double instr_per_wave = (double)iters * 8 * CH;
    printf("%s chains/wave %d waves/SIMD %d: %.3f ms -> %.2f cycles per instruction per wave, %.2f per SIMD (at 2.4 GHz)\n", F64 ? "v_fma_f64" : "v_fma_f32", CH,
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_wave, ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd));
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", 1, out); run<1>("v_pk_fma_f32", 1, out); run<2>("v_mul_f32", 1, out); run<7>("v_pk_mul_f32", 1, out);
    run<3>("v_exp_f32", 1, out); run<4>("v_rcp_f32", 1, out); run<5>("cmp+cndmask", 2, out); run<8>("mul+max", 2, out);
    run<6>("v_fma_f64", 1, out);
    run<21>("v_fmac_f64", 1, out); run<10>("v_mul_f64", 1, out); run<11>("v_add_f64", 1, out); run<19>("v_max_f64", 1, out);
    run<12>("v_rcp_f64", 1, out); run<13>("v_rsq_f64", 1, out); run<14>("v_sqrt_f64", 1, out); run<15>("v_ldexp_f64", 1, out);
    run<16>("v_rndne_f64", 1, out); run<17>("cvt i32<->f64", 2, out); run<24>("cvt f32<->f64", 2, out); run<18>("v_cmp_class_f64", 1, out);
    run<20>("cmp_f64+cndmask", 2, out); run<22>("v_log_f32", 1, out); run<23>("v_sqrt_f32", 1, out); run<25>("v_readlane_b32", 1, out);
    run<26>("v_mov_b32 v,s", 1, out);
    run<30>("v_mul_f32 s,v", 1, out); run<31>("v_mul_f32 lit,v", 1, out); run<45>("v_mul_f32_e64", 1, out); run<32>("v_add_f32", 1, out);
    run<46>("v_add_f32 neg", 1, out); run<33>("v_fmac_f32", 1, out); run<34>("v_fma_f32 s,v,v", 1, out); run<44>("v_fma_f32 ..,1.0", 1, out);
    run<52>("v_fma_f32 x,x,b", 1, out); run<35>("v_fmamk_f32", 1, out); run<36>("v_cndmask e32", 1, out); run<37>("v_cndmask e64", 1, out);
    run<38>("v_cmp e32", 1, out); run<39>("v_cmp e64", 1, out); run<40>("v_med3_f32", 1, out); run<41>("v_max_f32", 1, out);
    run<42>("v_mov_b32 v,v", 1, out); run<43>("v_max_i32", 1, out); run<49>("v_ldexp_f32", 1, out); run<51>("v_rsq_f32", 1, out);
    run<47>("v_mul_f64 s,v", 1, out); run<48>("v_fma_f64 s,v,v", 1, out); run<50>("v_mul+s_mul", 1, out);
    run<60>("v_pk_mul_f32 v,s", 1, out); run<61>("v_pk_mul op_sel s", 1, out); run<62>("v_pk_fma_f32 v,s,v", 1, out); run<63>("v_pk_add_f32", 1, out);
    run<64>("v_pk_fma x,x,b", 1, out); run<65>("v_mul_legacy_f32", 1, out); run<66>("v_mov_b64", 1, out); run<67>("v_pk_add_f32 v,s", 1, out);
    run<68>("v_fma_f64 s,..1.0", 1, out); run<69>("v_pk_fma s,v,s", 1, out);
    run<53>("3 v_mul + v_exp", 4, out);   // does a transcendental overlap with ordinary VALU issue?
    for (int w : {1, 2, 3, 4, 8}) { run_chain<1, true>(w, out); run_chain<2, true>(w, out); run_chain<4, true>(w, out); }
    for (int w : {1, 2, 4}) { run_chain<1, false>(w, out); run_chain<2, false>(w, out); }
    return 0;
}
