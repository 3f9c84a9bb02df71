/*
 * cmx_oracle.c — CPU oracle for the cmx hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the scalar arithmetic of CliMA/CloudMicrophysics.jl
 * v0.38.1 for the functions on the hot path; every function in
 * cmx_oracle_impl.h cites the reference file:line it follows.  The reference is
 * Julia and cannot be executed in the build image (no julia, un-vendored
 * ClimaParams/Thermodynamics/SpecialFunctions), so the oracle is PINNED by the
 * reference's own known-answer tests instead — the JSON fixtures under tests/golden/, checked by
 * tests/test_oracle_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libcmx.so) never links, loads or calls it.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "../include/cmx.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- Float64 instantiation ---- */
#define FT double
#define SFX f64
#define M_POW pow
#define M_FMA fma
#define M_EXP exp
#define M_LOG log
#define M_CBRT cbrt
#define M_SQRT sqrt
#define M_ABS fabs
#define M_TGAMMA tgamma
#define M_LGAMMA lgamma
#define M_LOG1P log1p
#define M_EXPM1 expm1
#define M_ERF erf
#define M_ERFC erfc
#define M_TANH tanh
#define M_ATANH atanh
#define M_LOG2 log2
#define M_EPS DBL_EPSILON
#include "cmx_oracle_impl.h"
#undef M_ERF
#undef M_ERFC
#undef M_TANH
#undef M_ATANH
#undef M_LOG2
#undef M_EPS
#undef FT
#undef SFX
#undef M_POW
#undef M_FMA
#undef M_EXP
#undef M_LOG
#undef M_CBRT
#undef M_SQRT
#undef M_ABS
#undef M_TGAMMA
#undef M_LGAMMA
#undef M_LOG1P
#undef M_EXPM1

/* ---- Float32 instantiation (the reference's Float32 path: float arithmetic, float gates) ---- */
#define FT float
#define SFX f32
#define M_POW powf
#define M_FMA fmaf
#define M_EXP expf
#define M_LOG logf
#define M_CBRT cbrtf
#define M_SQRT sqrtf
#define M_ABS fabsf
#define M_TGAMMA tgammaf
#define M_LGAMMA lgammaf
#define M_LOG1P log1pf
#define M_EXPM1 expm1f
#define M_ERF erff
#define M_ERFC erfcf
#define M_TANH tanhf
#define M_ATANH atanhf
#define M_LOG2 log2f
#define M_EPS FLT_EPSILON
#include "cmx_oracle_impl.h"
