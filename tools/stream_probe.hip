// stream_probe.hip — what bounds a 7-load / 6-store SoA sweep on MI355X?  (round 2; standalone, no torch)
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/stream_probe.hip -o tools/stream_probe && tools/stream_probe [n] [reps]
//
// The north-star kernel runs at the ceiling of its own access pattern (13 concurrent HBM streams: a same-shape copy kernel
// reaches 70-73 % of the 8 TB/s spec peak, pure 7-stream reads 87 %, pure 6-stream writes 76-78 %; profiles/r01_probe_*.txt).
// This probe varies what a kernel CAN change about that pattern, with trivial arithmetic:
//   A  cache-policy bits on the loads and on the stores (plain / nt / sc1 / sc0 sc1 / sc1 nt / sc0 sc1 nt), by inline asm;
//   B  relative placement of the 13 columns in memory (skew between column bases: all columns of a torch allocation start on
//      2-MiB boundaries, i.e. the 13 streams walk the HBM channels / banks in lock step);
//   C  workgroup shape: lanes per workgroup, tiles per workgroup, persistent software-pipelined loop (next tile's loads in
//      flight while the current tile is stored);
//   D  a synthetic VALU load per point (dependent FMA chain) to see when issue pressure starts to cost bandwidth;
//   E  s_setprio around the store burst.
// Prints ms, GB/s over the 52 B/point actually moved and the fraction of 8 TB/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
struct In { const float *p[7]; };
struct Out { float *p[6]; };

// ---- policy-tagged 16-byte accesses -----------------------------------------------------------------------------------------
enum { P_PLAIN = 0, P_NT, P_SC1, P_SC0SC1, P_SC1NT, P_SC0SC1NT, P_SC0, NPOL };
static const char *const POLNAME[NPOL] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0"};

template <int P> __device__ __forceinline__ f4 ld16(const float *p) {
    f4 v;
    if constexpr (P == P_PLAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == P_NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == P_SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == P_SC0SC1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == P_SC1NT) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == P_SC0SC1NT) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int P> __device__ __forceinline__ void st16(float *p, f4 v) {
    if constexpr (P == P_PLAIN) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == P_NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == P_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == P_SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == P_SC1NT) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == P_SC0SC1NT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
// all outstanding loads have landed; the operands tie the wait to the registers so nothing is moved across it
__device__ __forceinline__ void wait7(f4 &a, f4 &b, f4 &c, f4 &d, f4 &e, f4 &f, f4 &g) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g)::"memory");
}

// synthetic VALU work: WORK dependent FMAs per component (4 per lane-vector) — ≈ WORK VALU instructions per point
template <int WORK> __device__ __forceinline__ f4 burn(f4 x) {
#pragma unroll
    for (int k = 0; k < WORK; ++k) x = x * 1.0000001f + 1e-9f;
    return x;
}

// ---- A/B/D/E: one 16-byte vector per lane, one short-lived workgroup per tile -------------------------------------------------
template <int BS, int LP, int SP, int WORK, bool PRIO, int NR = 7, int NW = 6>
__global__ __launch_bounds__(BS) void np13(const In in, const Out out, const int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    if (i >= nvec) return;
    f4 v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = (f4)(float)(i + k);
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = ld16<LP>(in.p[k] + 4 * i);
    if constexpr (NR > 0) wait7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
    f4 o = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + v[6]);
    if constexpr (WORK > 0) o = burn<WORK>(o);
    if constexpr (NW == 0) {
        if (o.x == 1.2345e-30f) st16<SP>(out.p[0] + 4 * i, o);
        return;
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
    st16<SP>(out.p[0] + 4 * i, o);
#pragma unroll
    for (int k = 1; k < NW; ++k) st16<SP>(out.p[k] + 4 * i, v[k - 1] + o);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
}

// ---- C: persistent, software-pipelined: tile t+G's loads are in flight while tile t is reduced and stored -----------------------
template <int BS, bool NT, int WORK>
__global__ __launch_bounds__(BS) void pipe13(const In in, const Out out, const int64_t nvec) {
    const int64_t stride = (int64_t)gridDim.x * BS;
    int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    f4 cur[7], nxt[7];
    auto load = [&](f4(&r)[7], int64_t j) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const f4 *p = reinterpret_cast<const f4 *>(in.p[k]) + j;
            r[k] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    };
    if (i < nvec) load(cur, i);
    for (; i < nvec; i += stride) {
        const int64_t j = i + stride;
        if (j < nvec) load(nxt, j);
        f4 o = ((cur[0] + cur[1]) + (cur[2] + cur[3])) + ((cur[4] + cur[5]) + cur[6]);
        if constexpr (WORK > 0) o = burn<WORK>(o);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            f4 *p = reinterpret_cast<f4 *>(out.p[k]) + i;
            const f4 val = k == 0 ? o : cur[k - 1] + o;
            if (NT) __builtin_nontemporal_store(val, p);
            else *p = val;
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) cur[k] = nxt[k];
    }
}

// ---- C': C consecutive tiles per workgroup, all loads first (the round-1 shape, for reference) ----------------------------------
template <int BS, int C, int WORK>
__global__ __launch_bounds__(BS) void tiles13(const In in, const Out out, const int64_t nvec) {
    f4 v[C][7];
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * C + t) * BS + threadIdx.x;
        if (i < nvec) {
#pragma unroll
            for (int k = 0; k < 7; ++k) v[t][k] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(in.p[k]) + i);
        }
    }
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * C + t) * BS + threadIdx.x;
        if (i < nvec) {
            f4 o = ((v[t][0] + v[t][1]) + (v[t][2] + v[t][3])) + ((v[t][4] + v[t][5]) + v[t][6]);
            if constexpr (WORK > 0) o = burn<WORK>(o);
#pragma unroll
            for (int k = 0; k < 6; ++k)
                __builtin_nontemporal_store(k == 0 ? o : v[t][k - 1] + o, reinterpret_cast<f4 *>(out.p[k]) + i);
        }
    }
}

__global__ void fill(float *p, int64_t n, int random) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!random) { p[i] = (float)(i & 1023) * 1e-3f; continue; }
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;      // splitmix64: every mantissa bit toggles
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (float)(z >> 40) * (1.0f / 16777216.0f) + 1e-3f;
    }
}

template <typename F> static float time_ms(F &&launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}

static int64_t g_n;
static void report(const char *name, float ms, double bytes_per_point = 52.0) {
    const double gbs = bytes_per_point * (double)g_n / ms * 1e-6;
    std::printf("%-64s %8.4f ms  %8.1f GB/s  %5.1f %%\n", name, ms, gbs, gbs / 80.0);
    std::fflush(stdout);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 100000000ll;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 20;
    const bool separate = argc > 3 && std::string(argv[3]) == "separate";      // 13 hipMallocs (what a caching allocator does) instead of one slab
    const bool random = argc > 4 && std::string(argv[4]) == "random";          // full-entropy data instead of a 1024-periodic ramp
    const bool quick = argc > 5 && std::string(argv[5]) == "quick";            // only the headline rows
    const bool writes = argc > 5 && std::string(argv[5]) == "writes";          // the write-path matrix (box calibration)
    g_n = n;
    const int64_t nvec = n / 4;
    const size_t colbytes = ((size_t)n * 4 + (2u << 20) - 1) / (2u << 20) * (2u << 20);   // 2-MiB granules, like a caching allocator
    const size_t maxskew = 1u << 20;
    char *slab = nullptr;
    CK(hipMalloc(&slab, 13 * (colbytes + maxskew) + (2u << 20)));
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, reinterpret_cast<float *>(slab), (int64_t)(13 * (colbytes + maxskew) / 4), (int)random);
    CK(hipDeviceSynchronize());
    float *sep[13] = {};
    if (separate)
        for (auto &q : sep) {
            CK(hipMalloc(&q, (size_t)n * 4));
            hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, q, n, (int)random);
        }
    CK(hipDeviceSynchronize());
    auto place = [&](size_t skew, In &in, Out &out) {   // column k starts at k·(colbytes + skew)
        for (int k = 0; k < 7; ++k) in.p[k] = separate ? sep[k] : reinterpret_cast<const float *>(slab + (size_t)k * (colbytes + skew));
        for (int k = 0; k < 6; ++k) out.p[k] = separate ? sep[7 + k] : reinterpret_cast<float *>(slab + (size_t)(7 + k) * (colbytes + skew));
    };
    In in; Out out;
    place(0, in, out);
    std::printf("n = %lld f32 points, %d reps; %s; %s data; columns on 2-MiB boundaries unless a skew is given\n", (long long)n, reps,
                separate ? "13 separate hipMallocs" : "one slab", random ? "random" : "1024-periodic ramp");
    if (separate) for (int k = 0; k < 13; ++k) std::printf("  column %d at %p\n", k, (void *)sep[k]);
    char label[200];

#define RUN_NP(BS, LP, SP, WORK, PRIO, NR, NW, bpp, text)                                                                     \
    {                                                                                                                         \
        const int g = (int)((nvec + BS - 1) / BS);                                                                            \
        std::snprintf(label, sizeof label, "np BS=%d ld[%s] st[%s] work=%d%s %s", BS, POLNAME[LP], POLNAME[SP], WORK, PRIO ? " prio" : "", text); \
        report(label, time_ms([&] { hipLaunchKernelGGL((np13<BS, LP, SP, WORK, PRIO, NR, NW>), dim3(g), dim3(BS), 0, 0, in, out, nvec); }, reps), bpp); \
    }
    if (writes) {   // what does THIS box's write path deliver, and does any shape or policy change it?
        RUN_NP(128, P_NT, P_NT, 0, false, 7, 0, 28.0, "read-only 7")
        RUN_NP(128, P_NT, P_NT, 0, false, 0, 1, 4.0, "write-only 1") RUN_NP(128, P_NT, P_NT, 0, false, 0, 2, 8.0, "write-only 2")
        RUN_NP(128, P_NT, P_NT, 0, false, 0, 3, 12.0, "write-only 3") RUN_NP(128, P_NT, P_NT, 0, false, 0, 6, 24.0, "write-only 6")
        RUN_NP(128, P_NT, P_PLAIN, 0, false, 0, 6, 24.0, "write-only 6") RUN_NP(128, P_NT, P_SC1, 0, false, 0, 6, 24.0, "write-only 6")
        RUN_NP(128, P_NT, P_SC0SC1, 0, false, 0, 6, 24.0, "write-only 6") RUN_NP(128, P_NT, P_SC1NT, 0, false, 0, 6, 24.0, "write-only 6")
        RUN_NP(64, P_NT, P_NT, 0, false, 0, 6, 24.0, "write-only 6") RUN_NP(256, P_NT, P_NT, 0, false, 0, 6, 24.0, "write-only 6")
        RUN_NP(1024, P_NT, P_NT, 0, false, 0, 6, 24.0, "write-only 6")
        RUN_NP(128, P_NT, P_NT, 0, false, 1, 1, 8.0, "copy 1r/1w") RUN_NP(128, P_NT, P_NT, 0, false, 7, 6, 52.0, "7r/6w")
        RUN_NP(128, P_NT, P_NT, 256, false, 7, 6, 52.0, "7r/6w") RUN_NP(128, P_NT, P_SC1NT, 256, false, 7, 6, 52.0, "7r/6w")
        return 0;
    }
    const bool shapes = argc > 5 && std::string(argv[5]) == "shapes";          // the column counts of the other HBM-bound kernels (bare copies)
    if (shapes) {
        for (int rep = 0; rep < 2; ++rep) {
            RUN_NP(128, P_NT, P_NT, 0, false, 3, 2, 20.0, "3r/2w  (ice nucleation)") RUN_NP(256, P_NT, P_NT, 0, false, 3, 2, 20.0, "3r/2w  (ice nucleation)")
            RUN_NP(128, P_NT, P_NT, 0, false, 2, 1, 12.0, "2r/1w  (0-moment)") RUN_NP(128, P_NT, P_NT, 0, false, 7, 4, 44.0, "7r/4w  (1-moment, fields)")
            RUN_NP(128, P_NT, P_NT, 0, false, 4, 5, 36.0, "4r/5w  (ARG2000, 5 modes)") RUN_NP(128, P_NT, P_NT, 0, false, 7, 6, 52.0, "7r/6w  (north star)")
            RUN_NP(128, P_NT, P_NT, 64, false, 3, 2, 20.0, "3r/2w") RUN_NP(128, P_NT, P_NT, 174, false, 4, 5, 36.0, "4r/5w") RUN_NP(128, P_NT, P_NT, 253, false, 7, 4, 44.0, "7r/4w")
        }
        return 0;
    }
    if (quick) {
        for (int rep = 0; rep < 3; ++rep) {
            RUN_NP(128, P_NT, P_NT, 0, false, 7, 6, 52.0, "") RUN_NP(128, P_NT, P_SC1NT, 0, false, 7, 6, 52.0, "") RUN_NP(128, P_NT, P_NT, 256, false, 7, 6, 52.0, "")
            RUN_NP(128, P_NT, P_NT, 320, false, 7, 6, 52.0, "") RUN_NP(128, P_NT, P_NT, 0, false, 7, 0, 28.0, "read-only") RUN_NP(128, P_NT, P_NT, 0, false, 0, 6, 24.0, "write-only")
        }
        return 0;
    }
    // ---- A: cache policies (BS = 128) --------------------------------------------------------------------------------------
    std::printf("-- A: load policy x store policy\n");
#define ROW(LP) RUN_NP(128, LP, P_PLAIN, 0, false, 7, 6, 52.0, "") RUN_NP(128, LP, P_NT, 0, false, 7, 6, 52.0, "") RUN_NP(128, LP, P_SC1, 0, false, 7, 6, 52.0, "") \
    RUN_NP(128, LP, P_SC0SC1, 0, false, 7, 6, 52.0, "") RUN_NP(128, LP, P_SC1NT, 0, false, 7, 6, 52.0, "") RUN_NP(128, LP, P_SC0SC1NT, 0, false, 7, 6, 52.0, "")
    ROW(P_PLAIN) ROW(P_NT) ROW(P_SC1) ROW(P_SC1NT) ROW(P_SC0SC1NT)
#undef ROW
    std::printf("-- A': pure reads (7 streams) and pure writes (6 streams) per policy\n");
    RUN_NP(128, P_PLAIN, P_NT, 0, false, 7, 0, 28.0, "read-only") RUN_NP(128, P_NT, P_NT, 0, false, 7, 0, 28.0, "read-only")
    RUN_NP(128, P_SC1, P_NT, 0, false, 7, 0, 28.0, "read-only") RUN_NP(128, P_SC1NT, P_NT, 0, false, 7, 0, 28.0, "read-only")
    RUN_NP(128, P_NT, P_PLAIN, 0, false, 0, 6, 24.0, "write-only") RUN_NP(128, P_NT, P_NT, 0, false, 0, 6, 24.0, "write-only")
    RUN_NP(128, P_NT, P_SC1, 0, false, 0, 6, 24.0, "write-only") RUN_NP(128, P_NT, P_SC0SC1, 0, false, 0, 6, 24.0, "write-only")
    RUN_NP(128, P_NT, P_SC1NT, 0, false, 0, 6, 24.0, "write-only") RUN_NP(128, P_NT, P_SC0SC1NT, 0, false, 0, 6, 24.0, "write-only")
    RUN_NP(128, P_NT, P_NT, 0, false, 1, 1, 8.0, "copy 1r/1w") RUN_NP(128, P_NT, P_NT, 0, false, 2, 2, 16.0, "copy 2r/2w")
    RUN_NP(128, P_NT, P_NT, 0, false, 4, 4, 32.0, "copy 4r/4w") RUN_NP(128, P_NT, P_NT, 0, false, 7, 1, 32.0, "7r/1w")
    RUN_NP(128, P_NT, P_NT, 0, false, 7, 3, 40.0, "7r/3w") RUN_NP(128, P_NT, P_NT, 0, false, 1, 6, 28.0, "1r/6w")

    // ---- B: skew between column bases ------------------------------------------------------------------------------------
    std::printf("-- B: column k at k*(2MiB-rounded size + skew)\n");
    for (size_t skew : {(size_t)0, (size_t)256, (size_t)512, (size_t)1024, (size_t)2048, (size_t)4096, (size_t)8192, (size_t)16384, (size_t)65536,
                        (size_t)(65536 + 4096 + 256), (size_t)262144, (size_t)(1u << 20)}) {
        place(skew, in, out);
        char sk[48];
        std::snprintf(sk, sizeof sk, "skew %zu B", skew);
        RUN_NP(128, P_NT, P_NT, 0, false, 7, 6, 52.0, sk)
    }
    place(0, in, out);

    // ---- C: workgroup shape ------------------------------------------------------------------------------------------------
    std::printf("-- C: shape\n");
    RUN_NP(64, P_NT, P_NT, 0, false, 7, 6, 52.0, "") RUN_NP(128, P_NT, P_NT, 0, false, 7, 6, 52.0, "") RUN_NP(256, P_NT, P_NT, 0, false, 7, 6, 52.0, "")
    RUN_NP(512, P_NT, P_NT, 0, false, 7, 6, 52.0, "") RUN_NP(1024, P_NT, P_NT, 0, false, 7, 6, 52.0, "")
#define RUN_PIPE(BS, NT, WORK, G)                                                                                             \
    {                                                                                                                         \
        std::snprintf(label, sizeof label, "pipelined persistent BS=%d grid=%d %s work=%d", BS, G, NT ? "nt" : "plain", WORK); \
        report(label, time_ms([&] { hipLaunchKernelGGL((pipe13<BS, NT, WORK>), dim3(G), dim3(BS), 0, 0, in, out, nvec); }, reps)); \
    }
    for (int g : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32}) { RUN_PIPE(128, true, 0, g) }
    for (int g : {256 * 2, 256 * 4, 256 * 8, 256 * 16}) { RUN_PIPE(256, true, 0, g) }
    for (int g : {256 * 8, 256 * 16, 256 * 32}) { RUN_PIPE(64, true, 0, g) }
    for (int g : {256 * 4, 256 * 8}) { RUN_PIPE(128, true, 256, g) }
#define RUN_TILES(BS, C, WORK)                                                                                                \
    {                                                                                                                         \
        const int g = (int)((nvec + (int64_t)BS * C - 1) / ((int64_t)BS * C));                                                 \
        std::snprintf(label, sizeof label, "tiles BS=%d C=%d work=%d", BS, C, WORK);                                           \
        report(label, time_ms([&] { hipLaunchKernelGGL((tiles13<BS, C, WORK>), dim3(g), dim3(BS), 0, 0, in, out, nvec); }, reps)); \
    }
    RUN_TILES(128, 1, 0) RUN_TILES(128, 2, 0) RUN_TILES(128, 4, 0) RUN_TILES(64, 4, 0) RUN_TILES(64, 8, 0) RUN_TILES(256, 2, 0)

    // ---- D: VALU pressure -----------------------------------------------------------------------------------------------
    std::printf("-- D: synthetic VALU work per point (dependent FMA chain)\n");
    RUN_NP(128, P_NT, P_NT, 64, false, 7, 6, 52.0, "") RUN_NP(128, P_NT, P_NT, 128, false, 7, 6, 52.0, "") RUN_NP(128, P_NT, P_NT, 192, false, 7, 6, 52.0, "")
    RUN_NP(128, P_NT, P_NT, 256, false, 7, 6, 52.0, "") RUN_NP(128, P_NT, P_NT, 320, false, 7, 6, 52.0, "") RUN_NP(128, P_NT, P_NT, 384, false, 7, 6, 52.0, "")
    RUN_NP(256, P_NT, P_NT, 256, false, 7, 6, 52.0, "") RUN_NP(64, P_NT, P_NT, 256, false, 7, 6, 52.0, "")
    RUN_TILES(128, 2, 256) RUN_TILES(64, 4, 256)

    // ---- E: priority around the store burst ---------------------------------------------------------------------------------
    std::printf("-- E: s_setprio 3 around the stores\n");
    RUN_NP(128, P_NT, P_NT, 0, true, 7, 6, 52.0, "") RUN_NP(128, P_NT, P_NT, 256, true, 7, 6, 52.0, "") RUN_NP(256, P_NT, P_NT, 256, true, 7, 6, 52.0, "")
    CK(hipFree(slab));
    return 0;
}
