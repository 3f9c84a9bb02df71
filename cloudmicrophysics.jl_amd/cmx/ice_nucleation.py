"""Ice-nucleation rates over columns — host-side mirror of `CloudMicrophysics.HetIceNucleation`
(`ABIFM_J`), `HomIceNucleation` (`homogeneous_J_cubic`, `homogeneous_J_linear`) and the water-activity
helpers of `CloudMicrophysics.Common` (`a_w_ice`, `a_w_eT`), evaluated by one fused gfx950 kernel
(include/cmx.h §4).

Reference broadcasts being replaced (KA wrappers test/gpu_tests.jl:294-362):

    Δa_w  = a_w .- CO.a_w_ice.(Ref(tps), T)
    J_het = CMI_het.ABIFM_J.(Ref(dust), Δa_w)
    J_hom = CMI_hom.homogeneous_J_cubic.(Ref(ip.homogeneous), Δa_w)        # throws DomainError out of range
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple

import torch

from . import _abi, _lib
from .bulk_tendencies import _check_cols, _fam_of, _ptr

IceNucleationRates = namedtuple("IceNucleationRates",
                                ["delta_a_w", "J_het", "J_hom", "rate_het", "rate_hom", "n_domain_errors"])


def ice_nucleation_rates(tps, dust, koop, T, a_w, r=None, *, linear=False, want=("rate_het", "rate_hom"),
                         count_domain_errors=True, out=None, stream=None, h2so4=None) -> IceNucleationRates:
    """ABIFM immersion-freezing and Koop-2000 homogeneous-freezing rates for every (T, a_w, r) point.

    `want` selects the output columns among delta_a_w, J_het [m⁻² s⁻¹], J_hom [m⁻³ s⁻¹], rate_het = J_het·4πr²,
    rate_hom = J_hom·4/3πr³ [s⁻¹].  Where the reference's `homogeneous_J_cubic` would throw (Δa_w outside
    [Δa_w_min, Δa_w_max], src/IceNucleation.jl:558-562) J_hom/rate_hom are NaN and the point is counted in the
    device buffer `n_domain_errors` (CMX_ICENUC_ERR_WORDS int64 slot counters; `domain_error_count(result)` sums
    them — that read synchronises, the call itself does not).
    `linear=True` uses `homogeneous_J_linear` (:581-584), which has no domain restriction.
    `h2so4` (H2SO4SolutionParameters): the second column is the H2SO4 weight fraction x of solution droplets and
    a_w = CO.a_w_xT(h2so4, tps, x, T) is formed in the kernel (`cmx_ice_nucleation_rates_xT_*`; parcel/ParcelTendencies.jl:120-133)."""
    cols = (T, a_w) if r is None else (T, a_w, r)
    ref = _check_cols(cols, ("T", "a_w", "r"))
    fam = _fam_of(ref)
    if not (isinstance(tps, fam.thermo) and isinstance(dust, fam.abifm_dust) and isinstance(koop, fam.koop2000)):
        raise TypeError("parameter float type does not match the state columns")
    names = IceNucleationRates._fields[:5]
    unknown = set(want) - set(names)
    if unknown:
        raise ValueError(f"unknown output column(s) {sorted(unknown)}")
    if r is None and ({"rate_het", "rate_hom"} & set(want)):
        raise ValueError("rate_het / rate_hom need the radius column r")
    if out is not None:   # caller-provided columns (KA-kernel style); the counter ACCUMULATES across calls
        outs = {k: getattr(out, k) for k in names}
        _check_cols([ref] + [o for o in outs.values() if o is not None], ["T"] + ["out"] * 5)
        nerr = out.n_domain_errors
    else:
        outs = {k: (torch.empty_like(ref) if k in want else None) for k in names}
        nerr = (torch.zeros(_abi.CMX_ICENUC_ERR_WORDS, dtype=torch.int64, device=ref.device)
                if (count_domain_errors and not linear) else None)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    if h2so4 is not None and not isinstance(h2so4, fam.h2so4_solution_params):
        raise TypeError("h2so4 must be the H2SO4SolutionParameters of the columns' float type")
    fn = getattr(_lib.lib(), f"cmx_ice_nucleation_rates_{'xT_' if h2so4 is not None else ''}{fam.sfx}")
    head = (C.byref(tps), C.byref(dust), C.byref(koop)) + ((C.byref(h2so4),) if h2so4 is not None else ())
    with torch.cuda.device(ref.device):
        st = fn(*head, _abi.CMX_ICENUC_HOM_LINEAR if linear else 0, ref.numel(),
                _ptr(T), _ptr(a_w), _ptr(r), *[_ptr(outs[k]) for k in names], _ptr(nerr), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return IceNucleationRates(*[outs[k] for k in names], nerr)


def domain_error_count(result: IceNucleationRates) -> int:
    """Number of points where the reference would have thrown DomainError (sum of the slot counters)."""
    return 0 if result.n_domain_errors is None else int(result.n_domain_errors.sum().item())


def a_w_ice(tps, T, stream=None) -> torch.Tensor:
    """CO.a_w_ice.(Ref(tps), T) — src/Common.jl:267-271."""
    return _water_activity(tps, T, None, stream)[0]


def a_w_eT(tps, e, T, stream=None) -> torch.Tensor:
    """CO.a_w_eT.(Ref(tps), e, T) — src/Common.jl:250-253."""
    return _water_activity(tps, T, e, stream)[1]


def _water_activity(tps, T, e, stream):
    cols = (T,) if e is None else (T, e)
    ref = _check_cols(cols, ("T", "e"))
    fam = _fam_of(ref)
    if not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    ice = torch.empty_like(ref) if e is None else None
    eT = torch.empty_like(ref) if e is not None else None
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_water_activity_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(tps), ref.numel(), _ptr(T), _ptr(e), _ptr(ice), _ptr(eT), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return ice, eT


LiquidFreezingRate = namedtuple("LiquidFreezingRate", ["dn_frz", "dq_frz"])


def liquid_freezing_rate(ice_params, tps, q, rho, N, T, *, cloud: bool = False, stream=None):
    """`CMI_het.liquid_freezing_rate(rf, pdf, tps, q, ρ, N, T)` for every point — Bigg (1953) immersion freezing integrated over the
    SB2006 rain PSD (src/IceNucleation.jl:274-311) or, with `cloud=True`, the generalized-gamma cloud PSD (:355-389).  `ice_params`
    is a `P3IceParams` (its rain_freezing, rain_pdf / cloud_pdf members are used).  Returns (∂ₜn_frz [1/kg/s], ∂ₜq_frz [kg/kg/s])."""
    from .parameters import P3IceParams
    if not isinstance(ice_params, P3IceParams):
        raise TypeError("ice_params must be P3IceParams")
    cols = (q, rho, N, T)
    for c in cols:
        if not (c.is_cuda and c.is_contiguous() and c.dim() == 1 and c.dtype == q.dtype and c.numel() == q.numel()):
            raise TypeError("columns must be contiguous 1-D GPU tensors of one dtype and length")
    fam = _abi.family(q.dtype)
    if fam is not ice_params.fam or not isinstance(tps, fam.thermo):
        raise TypeError("parameter float type does not match the state columns")
    out = LiquidFreezingRate(torch.empty_like(q), torch.empty_like(q))
    flags = (_abi.CMX_FREEZE_CLOUD_PSD if cloud else 0) | (_abi.CMX_P3_RAIN_PDF_LIMITED if ice_params.is_limited else 0)
    s = stream if stream is not None else torch.cuda.current_stream(q.device)
    fn = getattr(_lib.lib(), f"cmx_liquid_freezing_rate_{fam.sfx}")
    with torch.cuda.device(q.device):
        st = fn(C.byref(ice_params.c), C.byref(tps), flags, q.numel(), *[C.c_void_p(t.data_ptr()) for t in cols],
                C.c_void_p(out.dn_frz.data_ptr()), C.c_void_p(out.dq_frz.data_ptr()), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return out


# ---- round 3: the remaining public functions of HetIceNucleation / Common (include/cmx.h §4) ---------------------------------------
def _cols1(cols, names):
    ref = _check_cols([c for c in cols if c is not None], names)
    return ref, _fam_of(ref)


def h2so4_solution(prs, tps, x_sulph, T, *, stream=None):
    """(CO.H2SO4_soln_saturation_vapor_pressure.(Ref(prs), x, T) [Pa], CO.a_w_xT.(Ref(prs), Ref(tps), x, T)) — src/Common.jl:188-246."""
    ref, fam = _cols1((x_sulph, T), ("x_sulph", "T"))
    p_sol, a_w = torch.empty_like(ref), torch.empty_like(ref)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_h2so4_solution_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(prs), C.byref(tps), ref.numel(), _ptr(x_sulph), _ptr(T), _ptr(p_sol), _ptr(a_w), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return p_sol, a_w


MohlerDeposition = namedtuple("MohlerDeposition", ["act_frac", "dep_rate", "n_domain_errors"])


def mohler2006_deposition(dust, ip, S_i, T, dSi_dt=None, N_aer=None, *, stream=None) -> MohlerDeposition:
    """CMI_het.dust_activated_number_fraction.(Ref(dust), Ref(ip), S_i, T) and, with dSi_dt and N_aer columns,
    CMI_het.MohlerDepositionRate.(…, S_i, T, dSi_dt, N_aer) — src/IceNucleation.jl:44-79.  Where the reference asserts (S_i ≥ Sᵢ_max) the
    outputs are NaN and the point is counted in the one-element device tensor `n_domain_errors`."""
    ref, fam = _cols1((S_i, T, dSi_dt, N_aer), ("S_i", "T", "dSi_dt", "N_aer"))
    if (dSi_dt is None) != (N_aer is None):
        raise ValueError("MohlerDepositionRate needs both dSi_dt and N_aer")
    frac = torch.empty_like(ref)
    rate = torch.empty_like(ref) if dSi_dt is not None else None
    nerr = torch.zeros(1, dtype=torch.int64, device=ref.device)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_mohler2006_deposition_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(dust), C.byref(ip), ref.numel(), _ptr(S_i), _ptr(T), _ptr(dSi_dt), _ptr(N_aer), _ptr(frac), _ptr(rate), _ptr(nerr),
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return MohlerDeposition(frac, rate, nerr)


def deposition_J(dust, delta_a_w, *, stream=None) -> torch.Tensor:
    """CMI_het.deposition_J.(Ref(dust), Δa_w) [m⁻² s⁻¹] — src/IceNucleation.jl:81-102."""
    ref, fam = _cols1((delta_a_w,), ("delta_a_w",))
    J = torch.empty_like(ref)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_deposition_J_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(dust), ref.numel(), _ptr(delta_a_w), _ptr(J), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return J


def INP_concentration_frequency(ip, INPC, T, *, stream=None) -> torch.Tensor:
    """CMI_het.INP_concentration_frequency.(Ref(ip), INPC, T) — src/IceNucleation.jl:219-226 (Frostenberg et al. 2023)."""
    ref, fam = _cols1((INPC, T), ("INPC", "T"))
    f = torch.empty_like(ref)
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_inp_concentration_frequency_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(ip), ref.numel(), _ptr(INPC), _ptr(T), _ptr(f), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return f
