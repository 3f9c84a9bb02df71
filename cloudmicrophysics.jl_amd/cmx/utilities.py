"""Special functions over columns — host-side mirror of the array use of `CloudMicrophysics.Utilities` (UT):

    UT.gamma_inc.(a, x)            →  gamma_inc(a, x)            -> (P, Q)        src/Utilities.jl:54-61, 93-144
    UT.gamma_inc_inv.(a, p, q)     →  gamma_inc_inv(a, p, q)     -> x             src/Utilities.jl:162-165, 205-252

(the reference's device test is `test_gamma_inc_kernel!`, test/gpu_tests.jl:456-461,1314-1337; its CPU test test/gamma_inc_tests.jl).
The kernels behind them call the same device routines as the P3 shape solver, the quantile bounds and the collision integrals."""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

import torch

from . import _lib
from .bulk_tendencies import _check_cols, _fam_of, _ptr

GammaInc = namedtuple("GammaInc", ["P", "Q"])


def gamma_inc(a: torch.Tensor, x: torch.Tensor, *, stream=None) -> GammaInc:
    """Regularised lower / upper incomplete gamma functions P(a, x), Q(a, x) (a > 0), the reference's fast approximation."""
    ref = _check_cols([a, x], ["a", "x"])
    fam = _fam_of(ref)
    out = GammaInc(torch.empty_like(ref), torch.empty_like(ref))
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_gamma_inc_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(ref.numel(), _ptr(a), _ptr(x), _ptr(out.P), _ptr(out.Q), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


def gamma_inc_inv(a: torch.Tensor, p: torch.Tensor, q: torch.Tensor, *, stream=None) -> torch.Tensor:
    """x with P(a, x) = p, Q(a, x) = q (Halley's method, ≤ 15 iterations, on whichever of the two residuals does not cancel)."""
    ref = _check_cols([a, p, q], ["a", "p", "q"])
    fam = _fam_of(ref)
    out = torch.empty_like(ref)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_gamma_inc_inv_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(ref.numel(), _ptr(a), _ptr(p), _ptr(q), _ptr(out), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out
