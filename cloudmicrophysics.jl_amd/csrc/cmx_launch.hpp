// cmx_launch.hpp — launch geometry, vector column access and error plumbing shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <tuple>
#include <type_traits>

#include <cstdint>
#include <cstdlib>

#include "../../include/cmx.h"

namespace cmx {

// --- error plumbing ------------------------------------------------------------------------
void set_hip_error(hipError_t e, const char *where);   // cmx_common.hip
#define CMX_HIP_TRY(expr)                                 \
    do {                                                  \
        hipError_t e_ = (expr);                           \
        if (e_ != hipSuccess) {                           \
            ::cmx::set_hip_error(e_, #expr);              \
            return CMX_ERR_HIP;                           \
        }                                                 \
    } while (0)

// --- geometry ------------------------------------------------------------------------------
// MI355X: 256 CUs in 8 XCDs.  The streaming kernels are NON-persistent: one short-lived workgroup per tile
// (256 lanes; 128 for the SB2006 sweep), one lane owns 16 bytes of every column, so the grid is ≫ 256 workgroups and
// workgroup b lands on XCD b % 8 — consecutive workgroups stream consecutive tiles and each XCD L2 sees a disjoint,
// dense address set.  (A grid-stride loop over CUs × k resident workgroups was measured slower with 13 concurrent
// streams, DESIGN.md §4: every wave of the chip sits in the same load → compute → store phase.)  `grid_for` (capped,
// grid-stride) is kept for the reduction kernel only; `tile_grid` sizes and range-checks the one-workgroup-per-tile grids.
constexpr int kBlock = 256;

struct DeviceInfo { int cus; int blocks_per_cu; };
DeviceInfo device_info();                                // cmx_common.hip (cached per device)

inline int grid_for(int64_t work_items, int items_per_block = kBlock) {
    const DeviceInfo di = device_info();
    const int64_t need = (work_items + items_per_block - 1) / items_per_block;
    const int64_t cap = (int64_t)di.cus * di.blocks_per_cu;
    const int64_t g = need < cap ? need : cap;
    return (int)(g < 1 ? 1 : g);
}

// Number of workgroups for `work_items` items at `items_per_block` each, or -1 if that exceeds what one launch can
// express (HIP: 2^31 − 1 workgroups in x).  Callers return CMX_ERR_UNSUPPORTED instead of silently truncating the grid.
inline int64_t tile_grid(int64_t work_items, int64_t items_per_block) {
    const int64_t g = (work_items + items_per_block - 1) / items_per_block;
    return g > (int64_t)0x7fffffff ? -1 : (g < 1 ? 1 : g);
}

// Upper bound on the points / states of ONE call, checked by every entry point (CMX_ERR_UNSUPPORTED above it): the streaming kernels give a
// workgroup at least 16 points, so n ≤ 16·(2^31 − 1) ≈ 3.4e10 keeps their grids inside HIP's 2^31 − 1 workgroups; at ≥ 12 B per point that is
// beyond the 288 GB of HBM anyway.  The P3 collision kernels have SMALLER tiles (Float32: 64-lane workgroups = 8 states at 8 lanes per state, 4 at
// 16) and check their own rounded-up tile count (collision_geometry, cmx_p3_collisions.hip): CMX_ERR_UNSUPPORTED above 8·(2^31 − 8) resp. 4·(…) states.
constexpr int64_t kMaxPoints = 16ll * 0x7fffffffll;

// The two heaviest Float64 kernels (SB2006, 1-moment) run one point per lane (8-byte loads) instead of two (16-byte): they are
// VALU-bound, not HBM-bound, and the two-points-per-lane variants spend VALU instructions on SGPR-spill traffic for the 2-SGPR
// Float64 constants (SB2006: 1307 vs 1048 instructions per point in round 1; 1-moment 4.60 → 4.36 ms per 1e8 points in a same-box
// A/B, round 2).  The lighter ones keep two points per lane (same A/B: ARG 3.25 vs 3.50 ms, ice nucleation 0.73 vs 0.88 with one).
// A/B switch for the 1-moment kernel: -DCMX_F64_ONE_POINT_PER_LANE=0.
#ifndef CMX_F64_ONE_POINT_PER_LANE
#define CMX_F64_ONE_POINT_PER_LANE 1
#endif

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// --- vector column access --------------------------------------------------------------------
template <typename FT, int VEC> struct VecT;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));
template <> struct VecT<float, 4> { using type = f32x4_t; };    // global_load_dwordx4: 16 B / lane
template <> struct VecT<float, 1> { using type = float; };
template <> struct VecT<double, 2> { using type = f64x2_t; };   // 16 B / lane as well
template <> struct VecT<double, 1> { using type = double; };

// NT = non-temporal hint: every byte of a state column is touched exactly once per sweep.
template <typename FT, int VEC, bool NT = true>
__device__ __forceinline__ void load_col(const FT *__restrict__ p, int64_t i, FT (&x)[VEC]) {
    using V = typename VecT<FT, VEC>::type;
    V v;
    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const V *>(p) + i);
    else v = reinterpret_cast<const V *>(p)[i];
    const FT *e = reinterpret_cast<const FT *>(&v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) x[k] = e[k];
}
template <typename FT, int VEC, bool NT = true>
__device__ __forceinline__ void store_col(FT *__restrict__ p, int64_t i, const FT (&x)[VEC]) {
    using V = typename VecT<FT, VEC>::type;
    V v;
    FT *e = reinterpret_cast<FT *>(&v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) e[k] = x[k];
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V *>(p) + i);
    else reinterpret_cast<V *>(p)[i] = v;
}

// Workgroup-base addressing (round 6): `wg` = the column's address at the workgroup's first vector — a UNIFORM pointer the compiler keeps in an SGPR pair —
// plus the lane's own vector index as a 32-bit offset: one `global_load … v_off, s[base:base+1]` per access with ONE offset VGPR shared by every column,
// where `p + i` with a 64-bit per-lane index costs an address pair per column (the per-element ARG kernel streams 4 + 5·NM columns: 16 VGPRs per mode).
template <typename FT, int VEC, bool NT = true>
__device__ __forceinline__ void load_col_wg(const FT *__restrict__ wg, uint32_t lane, FT (&x)[VEC]) {
    using V = typename VecT<FT, VEC>::type;
    const char *a = reinterpret_cast<const char *>(wg) + (uint32_t)(lane * (uint32_t)sizeof(V));
    V v;
    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const V *>(a));
    else v = *reinterpret_cast<const V *>(a);
    const FT *e = reinterpret_cast<const FT *>(&v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) x[k] = e[k];
}
template <typename FT, int VEC, bool NT = true>
__device__ __forceinline__ void store_col_wg(FT *__restrict__ wg, uint32_t lane, const FT (&x)[VEC]) {
    using V = typename VecT<FT, VEC>::type;
    char *a = reinterpret_cast<char *>(wg) + (uint32_t)(lane * (uint32_t)sizeof(V));
    V v;
    FT *e = reinterpret_cast<FT *>(&v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) e[k] = x[k];
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V *>(a));
    else *reinterpret_cast<V *>(a) = v;
}

// --- division of a 32-bit index by a run-time constant (Granlund & Montgomery 1994; Hacker's Delight §10-9) ---------------------
// The layout and column kernels turn a flat element index into (run, offset) or (column, level).  As a 64-bit quotient through
// doubles that is ≈ 40 VALU instructions per lane (conversions, a Float64 multiply, a 64-bit multiply-subtract, two fix-ups) — in the
// Float32 kernels, which are VALU-bound, 3–8 % of the lane's work (round 4: PMC counts).  For indices below 2³² (every launch up
// to 4.29e9 elements; the kernels keep the 64-bit path for larger ones) the quotient is exact with one v_mul_hi_u32:
//     t = mulhi(n, magic);   q = (t + ((n − t) >> sh1)) >> sh2;      magic = ⌊2³²(2^s − d)/d⌋ + 1,  s = ⌈log2 d⌉, sh1 = min(s, 1), sh2 = max(s − 1, 0)
struct FastDivU32 { uint32_t magic, sh1, sh2, d; };
inline FastDivU32 make_fastdiv(uint32_t d) {          // d ≥ 1
    uint32_t s = 0;
    while (s < 32 && ((uint64_t)1 << s) < d) ++s;      // s = ⌈log2 d⌉
    FastDivU32 f;
    f.magic = (uint32_t)((((uint64_t)1 << 32) * ((((uint64_t)1) << s) - d)) / d + 1);
    f.sh1 = s < 1 ? s : 1;
    f.sh2 = s > 0 ? s - 1 : 0;
    f.d = d;
    return f;
}
__host__ __device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDivU32 &f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(n, f.magic);
#else
    const uint32_t t = (uint32_t)(((uint64_t)n * f.magic) >> 32);
#endif
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// Launch of a kernel whose Float64 instantiation reads its constants through front_consts() (cmx_math.hpp), i.e. straight from the start of the
// kernel-argument segment: the constants struct MUST be the first kernel parameter.  The macro checks that at compile time — the first
// parameter type of the kernel is the type of the first argument passed — so reordering or prepending a parameter no longer compiles
// (nothing enforced the assumption).  K is the parenthesised kernel name, as for hipLaunchKernelGGL.
template <typename K> struct first_kernel_param;
template <typename A0, typename... A> struct first_kernel_param<void (*)(A0, A...)> { using type = std::remove_cv_t<A0>; };
#define CMX_LAUNCH_FRONT(K, grid, block, lds, stream, c, ...)                                                                                     \
    do {                                                                                                                                         \
        static_assert(std::is_same_v<typename ::cmx::first_kernel_param<decltype(&K)>::type, std::remove_cv_t<std::remove_reference_t<decltype(c)>>>, \
                      "front_consts() reads the constants struct as the FIRST kernel argument");                                                 \
        hipLaunchKernelGGL(K, grid, block, lds, stream, c, __VA_ARGS__);                                                                         \
    } while (0)

// Byte offset of the LAST parameter of a kernel inside its kernel-argument segment, derived from the kernel's own signature (the
// one-launch 2M + P3 kernel found its EXTRA argument with offsetof() on a hand-copied mirror struct that nothing tied to the parameter list).
// The AMDGPU kernarg ABI lays explicit arguments out in declaration order, each at the next multiple of its ABI alignment — the same rule as a
// C struct of those members — so the fold below follows a reordered, added or removed parameter by itself; `Last` must be the type the caller
// expects there, or it does not compile.
template <typename Last, typename F> struct kernarg_last;
template <typename Last, typename... A> struct kernarg_last<Last, void (*)(A...)> {
    static_assert(sizeof...(A) > 0, "a kernel without parameters has no last argument");
    static_assert(std::is_same_v<std::remove_cv_t<std::tuple_element_t<sizeof...(A) - 1, std::tuple<A...>>>, std::remove_cv_t<Last>>,
                  "the kernel's last parameter is not the type read through the kernel-argument segment");
    static constexpr size_t offset() {
        constexpr size_t sz[] = {sizeof(A)...}, al[] = {alignof(A)...};
        size_t off = 0;
        for (size_t i = 0; i < sizeof...(A); ++i) {
            off = (off + al[i] - 1) / al[i] * al[i];
            if (i + 1 < sizeof...(A)) off += sz[i];
        }
        return off;
    }
};
// use: kernarg_offset_of_last<EXTRA, decltype(&kernel<...>)>() — the kernel is named in an unevaluated operand only
template <typename Last, typename F> constexpr size_t kernarg_offset_of_last() { return kernarg_last<Last, F>::offset(); }

}  // namespace cmx
