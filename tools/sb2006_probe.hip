// sb2006_probe.hip — roofline probe for the fused SB2006 kernel (standalone, no torch).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/sb2006_probe.hip \
//         cloudmicrophysics.jl_amd/csrc/cmx_common.hip -o tools/sb2006_probe && tools/sb2006_probe [n] [reps]
//
// Times, with hipEvents on one stream over `reps` back-to-back launches at n points (default 1e8 f32):
//   stream13      7 dwordx4 loads + 6 dwordx4 stores per lane, trivial math — the practical HBM ceiling of
//                 this access pattern (13 concurrent streams), with and without the non-temporal hint;
//   sb2006        the product kernel at several workgroup sizes / tiles per workgroup and hint settings.
// Prints ms, GB/s at the ALGORITHMIC 52 B/point and the fraction of the 8 TB/s spec peak.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../cloudmicrophysics.jl_amd/csrc/cmx_sb2006_kernels.hpp"

using namespace cmx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

__device__ inline float u01(uint64_t i, uint32_t salt) {   // splitmix64 → [0,1)
    uint64_t z = (i + 0x9E3779B97F4A7C15ull * (salt + 1));
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void fill_state(SbIn<float> in, int64_t n) {
    float *rho = const_cast<float *>(in.rho), *T = const_cast<float *>(in.T), *qt = const_cast<float *>(in.q_tot),
          *ql = const_cast<float *>(in.q_lcl), *nl = const_cast<float *>(in.n_lcl), *qr = const_cast<float *>(in.q_rai),
          *nr = const_cast<float *>(in.n_rai);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float z = 15000.f * u01(i, 0);
        const float Tv = fmaxf(215.f, 300.f - 6.5e-3f * z) + 4.f * u01(i, 1) - 2.f;
        const float p = 1e5f * __expf(-z / 8000.f);
        const float r = p / (287.f * Tv);
        const float psat = 611.657f * __powf(Tv / 273.16f, -5.0314f) * __expf(6793.f * (1.f / 273.16f - 1.f / Tv));
        const float qv = fminf(0.04f, (0.05f + u01(i, 2)) * 0.622f * psat / p);
        const float qsat = psat / (r * 461.5f * Tv);
        const float qlv = fmaxf(0.f, qv - qsat) + (u01(i, 3) < 0.7f ? 1e-4f * u01(i, 4) : 0.f);
        float qrv = u01(i, 5) < 0.5f ? 1e-4f * u01(i, 6) : 0.f;
        if (u01(i, 7) < 0.1f) qrv = __expf(__logf(1e-7f) + u01(i, 8) * __logf(5e4f));
        rho[i] = r; T[i] = Tv; qt[i] = qv + qlv + qrv; ql[i] = qlv; qr[i] = qrv;
        nl[i] = u01(i, 9) < 0.2f ? 1e8f : __expf(__logf(1e6f) + u01(i, 10) * __logf(1e3f));
        nr[i] = __expf(__logf(1e1f) + u01(i, 11) * __logf(1e6f));
    }
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void stream13(const SbIn<float> in, const SbOut<float> out, const int64_t nvec) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
        float a[4], b[4], c[4], d[4], e[4], f[4], g[4], o[4];
        load_col<float, 4, NT>(in.rho, i, a); load_col<float, 4, NT>(in.T, i, b); load_col<float, 4, NT>(in.q_tot, i, c);
        load_col<float, 4, NT>(in.q_lcl, i, d); load_col<float, 4, NT>(in.n_lcl, i, e); load_col<float, 4, NT>(in.q_rai, i, f);
        load_col<float, 4, NT>(in.n_rai, i, g);
        for (int k = 0; k < 4; ++k) o[k] = a[k] + b[k] + c[k] + d[k] + e[k] + f[k] + g[k];
        store_col<float, 4, NT>(out.dq_lcl, i, o); store_col<float, 4, NT>(out.dn_lcl, i, a); store_col<float, 4, NT>(out.dq_rai, i, b);
        store_col<float, 4, NT>(out.dn_rai, i, c); store_col<float, 4, NT>(out.vt_n, i, d); store_col<float, 4, NT>(out.vt_m, i, e);
    }
}

// MODE 0: tile = blockIdx (one 16-B vector per lane, non-persistent)
// MODE 1: XCD-contiguous: workgroup b runs on XCD b%8 → give each XCD one contiguous eighth of every column
// MODE 2: each workgroup owns C consecutive tiles, all 7·C loads issued before the first store
template <int BS, int MODE, int C, bool RD, bool WR>
__global__ __launch_bounds__(BS) void stream13_np(const SbIn<float> in, const SbOut<float> out, const int64_t nvec) {
    int64_t tile = blockIdx.x;
    if constexpr (MODE == 1) {
        const int64_t per = (gridDim.x + 7) / 8;
        tile = (int64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
    }
    float a[C][4], b[C][4], c[C][4], d[C][4], e[C][4], f[C][4], g[C][4];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int64_t i = (tile * C + k) * BS + threadIdx.x;
        if (i < nvec) {
            if constexpr (RD) {
                load_col<float, 4, true>(in.rho, i, a[k]); load_col<float, 4, true>(in.T, i, b[k]);
                load_col<float, 4, true>(in.q_tot, i, c[k]); load_col<float, 4, true>(in.q_lcl, i, d[k]);
                load_col<float, 4, true>(in.n_lcl, i, e[k]); load_col<float, 4, true>(in.q_rai, i, f[k]);
                load_col<float, 4, true>(in.n_rai, i, g[k]);
            } else {
                for (int j = 0; j < 4; ++j) a[k][j] = b[k][j] = c[k][j] = d[k][j] = e[k][j] = f[k][j] = g[k][j] = (float)i;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int64_t i = (tile * C + k) * BS + threadIdx.x;
        if (i < nvec) {
            float o[4];
            for (int j = 0; j < 4; ++j) o[j] = a[k][j] + b[k][j] + c[k][j] + d[k][j] + e[k][j] + f[k][j] + g[k][j];
            if constexpr (WR) {
                store_col<float, 4, true>(out.dq_lcl, i, o); store_col<float, 4, true>(out.dn_lcl, i, a[k]);
                store_col<float, 4, true>(out.dq_rai, i, b[k]); store_col<float, 4, true>(out.dn_rai, i, c[k]);
                store_col<float, 4, true>(out.vt_n, i, d[k]); store_col<float, 4, true>(out.vt_m, i, e[k]);
            } else {
                if (o[0] == 1.2345e-30f) store_col<float, 4, true>(out.dq_lcl, i, o);
            }
        }
    }
}

__global__ __launch_bounds__(256) void copy1(const float *__restrict__ src, float *__restrict__ dst, const int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) {
        float v[4];
        load_col<float, 4, true>(src, i, v);
        store_col<float, 4, true>(dst, i, v);
    }
}

template <typename F> static float time_ms(F &&launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 100000000ll;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 20;
    float *buf[13];
    for (auto &p : buf) CK(hipMalloc(&p, sizeof(float) * n));
    SbIn<float> in{buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6]};
    SbOut<float> out{buf[7], buf[8], buf[9], buf[10], buf[11], buf[12]};
    hipLaunchKernelGGL(fill_state, dim3(4096), dim3(256), 0, 0, in, n);
    CK(hipDeviceSynchronize());

    cmx_warm_rain_2m_f32 wr{};   // ClimaParams defaults (cmx/parameters.py)
    auto &sb = wr.seifert_beheng;
    sb.pdf_c = {1.f, 1.f, 4.2e-15f, 2.6e-10f, 1000.f, 0.f, 0.f};
    sb.pdf_r = {-2.f / 3.f, 1.f / 3.f, 2.6e-10f, 5e-6f, 2.5e5f, 2e7f, 1e3f, 1e4f, 1000.f, 1.225f};
    sb.acnv = {4.44e9f, 2.6e-10f, 1.225f, 400.f, 0.7f, 3.f};
    sb.accr = {5.25f, 5e-5f, 1.225f, 4.f};
    sb.self = {7.12f, 60.7f, -5.f};
    sb.brek = {0.9e-3f, 0.35e-3f, 1000.f, 2300.f};
    sb.evap = {0.78f, 0.308f, 159.f, 0.266f, 1.225f, 0.42925f, 0.1786f, 2.5755f, 0.5955f, -0.101f};
    sb.numadj = {100.f};
    wr.air_properties = {0.024f, 2.26e-5f, 1.6e-5f};
    wr.condevap_tau_relax = 10.f;
    wr.subdep_tau_relax = 10.f;
    cmx_thermo_f32 tp{461.5f, 287.f, 1004.5f, 1859.f, 4181.f, 2070.f, 2.5008e6f, 2.8344e6f, 273.16f, 273.16f, 611.657f, 273.15f};
    cmx_rain_vel_f32 vel{};
    vel.sb2006 = {1.225f, 9.65f, 10.3f, 600.f, 1000.f, 1.6e-5f, 9.81f};
    const SbConsts<float> c = make_sb_consts<float>(wr, tp, &vel, (double)Math<float>::eps_1m());

    const int64_t nvec = n / 4;
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const double bytes = 52.0 * (double)n;
    auto report = [&](const char *name, int grid, float ms) {
        std::printf("%-34s grid %7d  %8.4f ms  %8.1f GB/s  %5.1f %% of 8 TB/s\n", name, grid, ms, bytes / ms * 1e-6,
                    bytes / ms * 1e-6 / 8000.0 * 100.0);
        std::fflush(stdout);
    };
    std::printf("n = %lld f32 points, %d CUs, %d reps\n", (long long)n, cus, reps);
    for (int bpc : {4, 8, 16}) {
        const int grid = cus * bpc;
        report("stream13 nt", grid, time_ms([&] { hipLaunchKernelGGL(stream13<true>, dim3(grid), dim3(kBlock), 0, 0, in, out, nvec); }, reps));
        report("stream13 plain", grid, time_ms([&] { hipLaunchKernelGGL(stream13<false>, dim3(grid), dim3(kBlock), 0, 0, in, out, nvec); }, reps));
    }
    {
        const int grid = (int)((nvec + kBlock - 1) / kBlock);
        report("stream13 nt, one vec/lane", grid, time_ms([&] { hipLaunchKernelGGL(stream13<true>, dim3(grid), dim3(kBlock), 0, 0, in, out, nvec); }, reps));
    }
    {
        auto rep2 = [&](const char *name, int grid, float ms, double b) {
            std::printf("%-34s grid %7d  %8.4f ms  %8.1f GB/s  %5.1f %% of 8 TB/s (actual bytes)\n", name, grid, ms, b / ms * 1e-6,
                        b / ms * 1e-6 / 8000.0 * 100.0);
        };
        const int gc = (int)((nvec + 255) / 256);
        rep2("copy 1r/1w float4", gc, time_ms([&] { hipLaunchKernelGGL(copy1, dim3(gc), dim3(256), 0, 0, buf[0], buf[7], nvec); }, reps), 8.0 * n);
#define NP(BS, MODE, C, RD, WR, label, bytes)                                                                        \
    {                                                                                                                \
        const int g = (int)((nvec + (int64_t)BS * C - 1) / ((int64_t)BS * C));                                        \
        rep2(label, g, time_ms([&] { hipLaunchKernelGGL((stream13_np<BS, MODE, C, RD, WR>), dim3(g), dim3(BS), 0, 0, in, out, nvec); }, reps), bytes); \
    }
        NP(256, 0, 1, true, false, "read-only 7 streams", 28.0 * n)
        NP(256, 0, 1, false, true, "write-only 6 streams", 24.0 * n)
        NP(64, 0, 1, true, true, "np13 BS=64", 52.0 * n)
        NP(128, 0, 1, true, true, "np13 BS=128", 52.0 * n)
        NP(256, 0, 1, true, true, "np13 BS=256", 52.0 * n)
        NP(512, 0, 1, true, true, "np13 BS=512", 52.0 * n)
        NP(1024, 0, 1, true, true, "np13 BS=1024", 52.0 * n)
        NP(256, 1, 1, true, true, "np13 BS=256 XCD-contiguous", 52.0 * n)
        NP(512, 1, 1, true, true, "np13 BS=512 XCD-contiguous", 52.0 * n)
        NP(256, 2, 2, true, true, "np13 BS=256 C=2", 52.0 * n)
        NP(256, 2, 4, true, true, "np13 BS=256 C=4", 52.0 * n)
        NP(128, 2, 2, true, true, "np13 BS=128 C=2", 52.0 * n)
        NP(64, 2, 4, true, true, "np13 BS=64 C=4", 52.0 * n)
#undef NP
    }
#define SB(L, V, VECN, BS, C, NT, label)                                                                               \
    {                                                                                                                  \
        const int64_t nv = (VECN == 4) ? nvec : n;                                                                     \
        const int g = (int)((nv + (int64_t)BS * C - 1) / ((int64_t)BS * C));                                            \
        report(label, g, time_ms([&] {                                                                                 \
                   hipLaunchKernelGGL((sb2006_tendencies_kernel<float, L, V, VECN, BS, C, NT>), dim3(g), dim3(BS), 0, 0, c, in, \
                                      out, nv); }, reps));                                                             \
    }
    SB(true, VEL_SB, 4, 64, 1, true, "sb2006 lim+vel BS=64")
    SB(true, VEL_SB, 4, 128, 1, true, "sb2006 lim+vel BS=128")
    SB(true, VEL_SB, 4, 256, 1, true, "sb2006 lim+vel BS=256")
    SB(true, VEL_SB, 4, 512, 1, true, "sb2006 lim+vel BS=512")
    SB(true, VEL_SB, 4, 1024, 1, true, "sb2006 lim+vel BS=1024")
    SB(true, VEL_SB, 4, 64, 2, true, "sb2006 lim+vel BS=64 C=2")
    SB(true, VEL_SB, 4, 128, 2, true, "sb2006 lim+vel BS=128 C=2")
    SB(true, VEL_SB, 4, 256, 2, true, "sb2006 lim+vel BS=256 C=2")
    SB(true, VEL_SB, 4, 256, 1, false, "sb2006 lim+vel BS=256 plain ld/st")
    SB(true, VEL_NONE, 4, 256, 1, true, "sb2006 lim, no vel BS=256")
    SB(false, VEL_SB, 4, 256, 1, true, "sb2006 notlim+vel BS=256")
    SB(true, VEL_CHEN, 4, 256, 1, true, "sb2006 lim+chen BS=256")
    SB(true, VEL_SB, 1, 256, 1, true, "sb2006 lim+vel 1 pt/lane")
#undef SB
    for (auto &p : buf) CK(hipFree(p));
    return 0;
}
