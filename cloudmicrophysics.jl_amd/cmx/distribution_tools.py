"""Size-distribution helpers over columns — host-side mirror of `CloudMicrophysics.DistributionTools` (DT) and of the SB2006 PSD accessors
of `CloudMicrophysics.Microphysics2M`:

    DT.generalized_gamma_quantile.(ν, μ, B, Y), DT.generalized_gamma_cdf.(ν, μ, B, x)          src/DistributionTools.jl:44-82
    DT.exponential_quantile.(D_mean, Y), DT.exponential_cdf.(D_mean, D)                        :124-151
    CM2.size_distribution_value.(Ref(pdf), q, ρₐ, N, D), CM2.get_size_distribution_bounds.(Ref(pdf), q, ρₐ, N, p)     src/Microphysics2M.jl:270-354

Where the scalar functions throw a DomainError the array entries write NaN (include/cmx.h)."""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

import torch

from . import _abi, _lib
from .bulk_tendencies import _check_cols, _fam_of, _ptr

Distribution = namedtuple("Distribution", ["quantile", "cdf"])
SizeDistribution = namedtuple("SizeDistribution", ["n_D", "D_min", "D_max"])


def generalized_gamma(nu: float, mu: float, B: torch.Tensor, Y=None, x=None, *, stream=None) -> Distribution:
    """quantile (needs Y) and / or cdf (needs x) of g(x) = A x^ν exp(−B x^μ), ν and μ shared, B per point."""
    cols = [c for c in (B, Y, x) if c is not None]
    ref = _check_cols(cols, ["B", "Y", "x"][:len(cols)])
    if Y is None and x is None:
        raise ValueError("pass Y (quantile) and / or x (cdf)")
    fam = _fam_of(ref)
    out = Distribution(torch.empty_like(ref) if Y is not None else None, torch.empty_like(ref) if x is not None else None)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_generalized_gamma_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(nu, mu, ref.numel(), _ptr(B), _ptr(Y), _ptr(x), _ptr(out.quantile), _ptr(out.cdf), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


def exponential_distribution(D_mean: torch.Tensor, Y=None, D=None, *, stream=None) -> Distribution:
    """quantile (needs Y) and / or cdf (needs D) of n(D) ∝ exp(−D / D_mean)."""
    cols = [c for c in (D_mean, Y, D) if c is not None]
    ref = _check_cols(cols, ["D_mean", "Y", "D"][:len(cols)])
    if Y is None and D is None:
        raise ValueError("pass Y (quantile) and / or D (cdf)")
    fam = _fam_of(ref)
    out = Distribution(torch.empty_like(ref) if Y is not None else None, torch.empty_like(ref) if D is not None else None)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_exponential_distribution_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(ref.numel(), _ptr(D_mean), _ptr(Y), _ptr(D), _ptr(out.quantile), _ptr(out.cdf), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


def size_distribution(pdf, q, rho, N, D=None, *, p=None, bounds=True, is_limited=None, stream=None) -> SizeDistribution:
    """`pdf` = a cmx_cloud_pdf_sb2006 (SB2006(FT).pdf_c) or cmx_rain_pdf_sb2006 (SB2006(FT, is_limited).pdf_r) struct.  n_D = the size
    distribution at diameter D (if D is given), (D_min, D_max) = get_size_distribution_bounds at probability level p (default eps(FT)).

    The reference picks the rain PSD variant from the struct's TYPE (`RainParticlePDF_SB2006_limited` / `_notlimited`,
    src/parameters/Microphysics2M.jl:314-375); the C layout is one struct for both, so here the variant is read from the struct's CONTENT:
    the not-limited constructor leaves the N0 / λ limiters at zero.  `is_limited` may still be passed explicitly; asking for the limited
    variant with a struct whose limiters are not positive and ordered is an error (the C entry returns CMX_ERR_BAD_ARG as well)."""
    cols = [c for c in (q, rho, N, D) if c is not None]
    ref = _check_cols(cols, ["q", "rho", "N", "D"][:len(cols)])
    fam = _fam_of(ref)
    cloud = isinstance(pdf, fam.cloud_pdf_sb2006)
    if not cloud and not isinstance(pdf, fam.rain_pdf_sb2006):
        raise TypeError("pdf must be the cloud or rain PSD struct of the columns' float type")
    if D is None and not bounds:
        raise ValueError("nothing to compute: pass D and / or bounds=True")
    if p is None:
        p = float(torch.finfo(ref.dtype).eps)
    if not cloud:
        has_limiters = (0 < pdf.N0_min <= pdf.N0_max) and (0 < pdf.lambda_min <= pdf.lambda_max)
        if is_limited is None:
            is_limited = has_limiters
        elif is_limited and not has_limiters:
            raise ValueError("is_limited=True needs a rain PSD struct with positive, ordered N0 / lambda limiters "
                             "(RainParticlePDF_SB2006(FT, is_limited=True)); this one is the not-limited variant")
    out = SizeDistribution(torch.empty_like(ref) if D is not None else None, torch.empty_like(ref) if bounds else None, torch.empty_like(ref) if bounds else None)
    flags = (_abi.CMX_PSD_CLOUD if cloud else 0) | (_abi.CMX_SB2006_LIMITED if (is_limited and not cloud) else 0)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_sb2006_size_distribution_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(pdf) if cloud else None, None if cloud else C.byref(pdf), flags, p, ref.numel(), _ptr(q), _ptr(rho), _ptr(N), _ptr(D), _ptr(out.n_D),
                _ptr(out.D_min), _ptr(out.D_max), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out
