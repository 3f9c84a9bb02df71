// cmx_layout.hpp — host-model layout adapters shared by the bulk-tendency kernels.
//
// The per-point arithmetic of a fused tendency kernel behind two layouts:
//   * SEGMENTED columns: column k is n_seg runs of seg_len contiguous elements, run s starting at base_k + s·stride_k.
//     That is a ClimaCore field in its storage: a DataLayouts.VIJFH array (Nv, Ni, Nj, Nf, Nh) holds component f of
//     element h as the contiguous run [v + Nv (i + Ni j)] of length Nv·Ni·Nj at offset Nv·Ni·Nj·(f + Nf h); VF columns
//     and VIJHF are the single-run case.  Every column has its own stride (state and tendency fields differ in Nf).
//   * AoS output: the reference's result type, an array of NamedTuples (2M warm rain: 8 fields, the last four identically
//     zero — BMT:852-853, test/gpu_performance.jl:212-216; 1M: the 4 tendencies — BMT:246-251, test/gpu_performance.jl:178-182).
//     A lane owns VEC consecutive points = VEC·NAOS contiguous values of the output; written directly that is a handful of
//     16-byte stores at a large lane stride (partial cache lines).  The workgroup's tile goes through LDS instead (rows
//     padded by 16 B) and leaves as fully coalesced 1-KiB-per-wave non-temporal stores.
// POLICY supplies: NIN, NOUT, NAOS, Consts<FT>, and  static void point(const Consts&, const FT (&x)[NIN], FT (&y)[NOUT])
// (clamps included); AoS rows are (y[0..NOUT), 0 …).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cmx_launch.hpp"
#include "cmx_math.hpp"

namespace cmx {

template <typename FT, int NIN, int NOUT> struct LayoutIO {
    const FT *in[NIN];
    int64_t in_stride[NIN];
    FT *out[NOUT];            // SoA mode
    int64_t out_stride[NOUT];
    FT *aos;                  // n × NAOS (AoS mode)
    int64_t seg_len;
    double inv_seg_len;
    FastDivU32 seg_div;       // division by seg_len for element indices below 2³² (idx32)
    bool idx32;               // every element index and every (run · stride + offset) product of this call fits the 32-bit forms
};

// does POLICY::point accept the packed pair type?  (policies declare `static constexpr bool PACKABLE = true`)
template <typename P, typename = void> struct layout_packable : std::false_type {};
template <typename P> struct layout_packable<P, std::enable_if_t<P::PACKABLE>> : std::true_type {};
// a policy whose Float32 constants overflow the SGPR file (the 1-moment kernels with run-time option flags: P::PHASE_CONSTS) reads them through the
// kernel-argument pointer like the Float64 kernels do (cmx_math.hpp front_consts ALSO) — as by-value arguments 32 of them were spilled to VGPR lanes
template <typename P, typename = void> struct layout_phase_consts : std::false_type {};
template <typename P> struct layout_phase_consts<P, std::enable_if_t<P::PHASE_CONSTS>> : std::true_type {};
#ifndef CMX_F32_PACKED_LAYOUT
#define CMX_F32_PACKED_LAYOUT 1          // A/B switch: 0 = one point at a time
#endif
#ifndef CMX_F32_PACKED_LAYOUT_PHASE
#define CMX_F32_PACKED_LAYOUT_PHASE 1    // the packed instantiations read their constants phase by phase (cmx_sb2006_kernels.hpp CMX_F32_PACKED_PHASE_CONSTS)
#endif
#ifndef CMX_LAYOUT_F64_VEC
#define CMX_LAYOUT_F64_VEC 1     // A/B switch (2: 16-byte accesses for Float64 as well)
#endif
#ifndef CMX_LAYOUT_BS
#define CMX_LAYOUT_BS 128
#endif
constexpr int kLayoutBS = CMX_LAYOUT_BS;   // lanes per workgroup of the adapter kernel (A/B switch)

template <typename FT, typename POLICY, int VEC, bool SEG, bool AOS, int BS = kLayoutBS>
__global__ __launch_bounds__(BS) void tendencies_layout_kernel(const typename POLICY::Consts c,
                                                               const LayoutIO<FT, POLICY::NIN, POLICY::NOUT> io, const int64_t nvec) {
    constexpr int NIN = POLICY::NIN, NOUT = POLICY::NOUT, NAOS = POLICY::NAOS;
    constexpr int CH = 16 / (int)sizeof(FT);                  // elements per 16-byte chunk
    constexpr int ROW = VEC * NAOS + CH;                      // LDS row of one lane (+1 chunk of padding against bank conflicts)
    static_assert((VEC * NAOS) % CH == 0, "a lane's AoS rows must be whole 16-byte chunks");
    const int64_t tile0 = (int64_t)blockIdx.x * BS;
    const int64_t i = tile0 + threadIdx.x;
    const bool active = i < nvec;
    FT y[VEC][NOUT];
    int64_t seg = 0, off = i * VEC;                           // element index → (run, offset in run)
    uint32_t seg32 = 0, off32 = 0;
    if constexpr (SEG) {
        const int64_t e = i * VEC;
        if (io.idx32) {                                        // wave-uniform: one v_mul_hi_u32 instead of the Float64 quotient (cmx_launch.hpp fastdiv)
            seg32 = fastdiv((uint32_t)e, io.seg_div);
            off32 = (uint32_t)e - seg32 * (uint32_t)io.seg_len;
            seg = seg32; off = off32;
        } else {
            seg = (int64_t)((double)e * io.inv_seg_len);
            off = e - seg * io.seg_len;
            if (off < 0) { --seg; off += io.seg_len; }
            if (off >= io.seg_len) { ++seg; off -= io.seg_len; }
        }
    }
    // element offset of the lane's first point in column k: run · stride + offset — with 32-bit operands one v_mad_u64_u32
    auto col_off = [&](int64_t stride) -> int64_t {
        if constexpr (!SEG) return off;
        else return io.idx32 ? (int64_t)((uint64_t)seg32 * (uint32_t)stride + off32) : seg * stride + off;
    };
    FT x[NIN][VEC];
    if (active) {
#pragma unroll
        for (int k = 0; k < NIN; ++k) load_col<FT, VEC, true>(io.in[k] + (col_off(io.in_stride[k]) - off), off / VEC, x[k]);
    }
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly (every lane of the workgroup reaches the barrier inside); no-op for Float32
    if (active) {
        // Float32 with four points per lane and a policy whose point function takes the packed pair type: two PAIRS of points (cmx_math.hpp f32x2)
        if constexpr (layout_packable<POLICY>::value && sizeof(FT) == 4 && VEC % 2 == 0 && CMX_F32_PACKED_LAYOUT) {
#pragma unroll
            for (int k = 0; k < VEC; k += 2) {
                f32x2 xi[NIN], yi[NOUT];
#pragma unroll
                for (int q = 0; q < NIN; ++q) xi[q] = f32x2{x[q][k], x[q][k + 1]};
                POLICY::point(front_consts<FT, (bool)CMX_F32_PACKED_LAYOUT_PHASE>(c), xi, yi);
#pragma unroll
                for (int q = 0; q < NOUT; ++q) { y[k][q] = yi[q].x; y[k + 1][q] = yi[q].y; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                FT xi[NIN];
#pragma unroll
                for (int q = 0; q < NIN; ++q) xi[q] = x[q][k];
                POLICY::point(front_consts<FT, layout_phase_consts<POLICY>::value>(c), xi, y[k]);   // Float64 (and P::PHASE_CONSTS): phase-local constants (cmx_math.hpp); c is the first argument
            }
        }
    }
    if constexpr (!AOS) {
        if (!active) return;
#pragma unroll
        for (int q = 0; q < NOUT; ++q) {
            FT col[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) col[k] = y[k][q];
            store_col<FT, VEC, true>(io.out[q] + (col_off(io.out_stride[q]) - off), off / VEC, col);
        }
    } else {
        extern __shared__ __align__(16) unsigned char lds_raw[];
        FT *lds = reinterpret_cast<FT *>(lds_raw);
        using V16 = typename VecT<FT, CH>::type;
        if (active) {
            FT *row = lds + threadIdx.x * ROW;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
#pragma unroll
                for (int q = 0; q < NAOS; ++q) row[k * NAOS + q] = q < NOUT ? y[k][q < NOUT ? q : 0] : FT(0);
            }
        }
        __syncthreads();
        constexpr int CPL = VEC * NAOS / CH;                  // 16-byte chunks per lane
        const int64_t nvalid = nvec - tile0 < BS ? nvec - tile0 : BS;
        V16 *dst = reinterpret_cast<V16 *>(io.aos) + tile0 * CPL;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int t = j * BS + threadIdx.x;                // chunk of the tile, in output order
            const int src = t / CPL, sub = t % CPL;
            if (src < nvalid) __builtin_nontemporal_store(*reinterpret_cast<const V16 *>(lds + src * ROW + sub * CH), dst + t);
        }
    }
}

// host side: validation + launch shared by the `_fields` entry points
template <typename FT, typename POLICY>
static int32_t launch_layout(const typename POLICY::Consts &c, int64_t n_seg, int64_t seg_len, const FT *const *in, const int64_t *in_stride,
                             FT *const *out, const int64_t *out_stride, FT *aos, hipStream_t s) {
    constexpr int NIN = POLICY::NIN, NOUT = POLICY::NOUT, NAOS = POLICY::NAOS;
    if (n_seg < 0 || seg_len < 0 || !in) return CMX_ERR_BAD_ARG;
    if (seg_len > 0 && n_seg > kMaxPoints / seg_len) return CMX_ERR_UNSUPPORTED;      // n_seg·seg_len must fit one launch (cmx_launch.hpp)
    if ((out != nullptr) == (aos != nullptr)) return CMX_ERR_BAD_ARG;                // exactly one output form
    const int64_t n = n_seg * seg_len;
    if (n == 0) return CMX_OK;
    if (n_seg > 1 && (!in_stride || (out && !out_stride))) return CMX_ERR_BAD_ARG;
    // Float64: one point per lane, like the column kernels (VALU-bound; two points per lane are 290–300 VGPRs — 1 wave per SIMD)
    constexpr int VEC = sizeof(FT) == 8 ? CMX_LAYOUT_F64_VEC : Math<FT>::VEC;
    bool vec_ok = seg_len % VEC == 0;
    LayoutIO<FT, NIN, NOUT> io{};
    for (int k = 0; k < NIN; ++k) {
        if (!in[k]) return CMX_ERR_BAD_ARG;
        io.in[k] = in[k]; io.in_stride[k] = n_seg > 1 ? in_stride[k] : 0;
        if (n_seg > 1 && io.in_stride[k] < seg_len) return CMX_ERR_BAD_ARG;
        vec_ok = vec_ok && aligned16(in[k]) && io.in_stride[k] % VEC == 0;
    }
    for (int k = 0; k < NOUT && out; ++k) {
        if (!out[k]) return CMX_ERR_BAD_ARG;
        io.out[k] = out[k]; io.out_stride[k] = n_seg > 1 ? out_stride[k] : 0;
        if (n_seg > 1 && io.out_stride[k] < seg_len) return CMX_ERR_BAD_ARG;
        vec_ok = vec_ok && aligned16(out[k]) && io.out_stride[k] % VEC == 0;
    }
    if (aos && !aligned16(aos)) return CMX_ERR_BAD_ARG;
    io.aos = aos; io.seg_len = seg_len; io.inv_seg_len = 1.0 / (double)seg_len;
    // the 32-bit index forms: every element index < 2³², every stride < 2³² (then run · stride + offset fits 64 bits from 32-bit factors)
    io.idx32 = n < ((int64_t)1 << 32) && seg_len < ((int64_t)1 << 32);
    for (int k = 0; k < NIN; ++k) io.idx32 = io.idx32 && io.in_stride[k] < ((int64_t)1 << 32);
    for (int k = 0; k < NOUT && out; ++k) io.idx32 = io.idx32 && io.out_stride[k] < ((int64_t)1 << 32);
    io.seg_div = make_fastdiv((uint32_t)(io.idx32 ? seg_len : 1));
    const bool seg = n_seg > 1;
    auto launch = [&](auto vec_tag) {
        constexpr int V = decltype(vec_tag)::value;
        const int64_t nvec = n / V;
        const dim3 grid((unsigned)((nvec + kLayoutBS - 1) / kLayoutBS)), block(kLayoutBS);
        const size_t lds = aos ? sizeof(FT) * (size_t)kLayoutBS * (V * NAOS + 16 / sizeof(FT)) : 0;
        if (seg) {
            if (aos) CMX_LAUNCH_FRONT((tendencies_layout_kernel<FT, POLICY, V, true, true>), grid, block, lds, s, c, io, nvec);
            else CMX_LAUNCH_FRONT((tendencies_layout_kernel<FT, POLICY, V, true, false>), grid, block, lds, s, c, io, nvec);
        } else {
            if (aos) CMX_LAUNCH_FRONT((tendencies_layout_kernel<FT, POLICY, V, false, true>), grid, block, lds, s, c, io, nvec);
            else CMX_LAUNCH_FRONT((tendencies_layout_kernel<FT, POLICY, V, false, false>), grid, block, lds, s, c, io, nvec);
        }
    };
    if (vec_ok) launch(std::integral_constant<int, VEC>{});
    else launch(std::integral_constant<int, 1>{});
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx
