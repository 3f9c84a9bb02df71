"""Aerosol model + ARG2000 activation over columns — host-side mirror of `CloudMicrophysics.AerosolModel` (AM:
`Mode_B`, `Mode_κ`, `AerosolDistribution`) and `CloudMicrophysics.AerosolActivation` (AA), include/cmx.h §6.

Reference broadcast being replaced (KA wrapper test/gpu_tests.jl:45-79):

    AA.N_activated_per_mode.(Ref(ap), Ref(ad), Ref(aip), Ref(tps), T, p, w, q_tot, q_liq, q_ice)

The aerosol distribution is shared by all states (BASELINE config 3); its per-mode reductions over the chemical
components (`mean_hygroscopicity_parameter`, AA:55-95, and Σ M_j·w_j, AA:313) are parameter-only and evaluated here
on the host, exactly as the reference does per call.
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple
from typing import Optional, Sequence

import torch

from . import _abi, _lib
from .bulk_tendencies import _check_cols, _fam_of, _ptr


def _tup(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x,)


class Mode_B:
    """AM.Mode_B — src/AerosolModel.jl:26-45 (Abdul-Razzak & Ghan 2000 chemistry: per-component tuples)."""

    def __init__(self, r_dry, stdev, N, mass_mix_ratio, soluble_mass_frac, osmotic_coeff, molar_mass, dissoc, aerosol_density):
        self.r_dry, self.stdev, self.N = r_dry, stdev, N
        self.mass_mix_ratio, self.soluble_mass_frac = _tup(mass_mix_ratio), _tup(soluble_mass_frac)
        self.osmotic_coeff, self.molar_mass = _tup(osmotic_coeff), _tup(molar_mass)
        self.dissoc, self.aerosol_density = _tup(dissoc), _tup(aerosol_density)

    def hygroscopicity(self, ap):
        """mean_hygroscopicity_parameter(ap, ::Mode_B) — AA:55-80 (mass-weighted B)."""
        nom = sum(w * nu * phi * eps / M for w, nu, phi, eps, M in zip(self.mass_mix_ratio, self.dissoc, self.osmotic_coeff,
                                                                        self.soluble_mass_frac, self.molar_mass))
        den = sum(w / rho for w, rho in zip(self.mass_mix_ratio, self.aerosol_density))
        return nom / den * ap.M_w / ap.rho_w


class Mode_kappa:
    """AM.Mode_κ — src/AerosolModel.jl:60-76 (Petters & Kreidenweis 2007 chemistry)."""

    def __init__(self, r_dry, stdev, N, vol_mix_ratio, mass_mix_ratio, molar_mass, kappa):
        self.r_dry, self.stdev, self.N = r_dry, stdev, N
        self.vol_mix_ratio, self.mass_mix_ratio = _tup(vol_mix_ratio), _tup(mass_mix_ratio)
        self.molar_mass, self.kappa = _tup(molar_mass), _tup(kappa)

    def hygroscopicity(self, ap):
        """mean_hygroscopicity_parameter(ap, ::Mode_κ) — AA:81-95 (volume-weighted κ)."""
        return sum(v * k for v, k in zip(self.vol_mix_ratio, self.kappa))


class AerosolDistribution:
    """AM.AerosolDistribution(modes) — src/AerosolModel.jl:88-99: all modes Mode_B or all Mode_kappa."""

    def __init__(self, modes: Sequence):
        modes = tuple(modes)
        if not modes or len(modes) > _abi.CMX_ARG_MAX_MODES:
            raise ValueError(f"1 … {_abi.CMX_ARG_MAX_MODES} modes supported")
        if not (all(isinstance(m, Mode_B) for m in modes) or all(isinstance(m, Mode_kappa) for m in modes)):
            raise TypeError("all modes must be Mode_B or all Mode_kappa")
        self.modes = modes

    def c_struct(self, ap, fam):
        ad = fam.aerosol_distribution()
        ad.n_modes = len(self.modes)
        for k, m in enumerate(self.modes):
            ad.modes[k] = fam.aerosol_mode(
                r_dry=m.r_dry, stdev=m.stdev, N=m.N, hygroscopicity=m.hygroscopicity(ap),
                molar_mass_mix=sum(M * w for M, w in zip(m.molar_mass, m.mass_mix_ratio)))
        return ad


ActivationResult = namedtuple("ActivationResult", ["N_act", "M_act", "S_max"])


def aerosol_activation(ap, ad: AerosolDistribution, aip, tps, T, p, w, q_tot, q_liq=None, q_ice=None, N_liq=None,
                       N_ice=None, *, want=("N_act",), out=None, stream=None) -> ActivationResult:
    """ARG2000 activation for every state: `N_act` = AA.N_activated_per_mode (AA:235-259), `M_act` =
    AA.M_activated_per_mode (AA:294-321) — tuples of one column per mode — and `S_max` = AA.max_supersaturation
    (AA:138-200).  `want` selects which are computed.  q_liq, q_ice, N_liq, N_ice default to zero columns (the
    reference's 10-argument methods)."""
    cols = [c for c in (T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice) if c is not None]
    ref = _check_cols(cols, ["T", "p", "w", "q_tot", "q_liq", "q_ice", "N_liq", "N_ice"])
    fam = _fam_of(ref)
    if not (isinstance(ap, fam.aerosol_activation_params) and isinstance(aip, fam.air_properties) and isinstance(tps, fam.thermo)):
        raise TypeError("parameter float type does not match the state columns")
    unknown = set(want) - {"N_act", "M_act", "S_max"}
    if unknown:
        raise ValueError(f"unknown output(s) {sorted(unknown)}")
    adc = ad.c_struct(ap, fam)
    nm = adc.n_modes
    if out is not None:   # caller-provided output columns (KA-kernel style)
        n_act, m_act, s_max = out
        for grp in (n_act, m_act):
            if grp is not None and len(grp) != nm:
                raise ValueError("out: one column per mode expected")
    else:
        n_act = tuple(torch.empty_like(ref) for _ in range(nm)) if "N_act" in want else None
        m_act = tuple(torch.empty_like(ref) for _ in range(nm)) if "M_act" in want else None
        s_max = torch.empty_like(ref) if "S_max" in want else None
    arr = lambda cols_: (C.c_void_p * nm)(*[c.data_ptr() for c in cols_]) if cols_ is not None else None  # noqa: E731
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_arg2000_activation_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(ap), C.byref(adc), C.byref(aip), C.byref(tps), ref.numel(), _ptr(T), _ptr(p), _ptr(w), _ptr(q_tot),
                _ptr(q_liq), _ptr(q_ice), _ptr(N_liq), _ptr(N_ice), arr(n_act), arr(m_act), _ptr(s_max),
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return ActivationResult(n_act, m_act, s_max)


def total_activated(ap, ad: AerosolDistribution, aip, tps, T, p, w, q_tot, q_liq=None, q_ice=None, N_liq=None, N_ice=None, *,
                    want=("N", "M"), stream=None):
    """(AA.total_N_activated.(…), AA.total_M_activated.(…)) — src/AerosolActivation.jl:355-433: the sums over the modes, formed in the
    activation kernel (`cmx_arg2000_total_activated_*`).  `want` ⊆ {"N", "M"}; the other member of the result is None."""
    cols = [c for c in (T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice) if c is not None]
    ref = _check_cols(cols, ["T", "p", "w", "q_tot", "q_liq", "q_ice", "N_liq", "N_ice"])
    fam = _fam_of(ref)
    if not (isinstance(ap, fam.aerosol_activation_params) and isinstance(aip, fam.air_properties) and isinstance(tps, fam.thermo)):
        raise TypeError("parameter float type does not match the state columns")
    adc = ad.c_struct(ap, fam)
    n_tot = torch.empty_like(ref) if "N" in want else None
    m_tot = torch.empty_like(ref) if "M" in want else None
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_arg2000_total_activated_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(ap), C.byref(adc), C.byref(aip), C.byref(tps), ref.numel(), _ptr(T), _ptr(p), _ptr(w), _ptr(q_tot),
                _ptr(q_liq), _ptr(q_ice), _ptr(N_liq), _ptr(N_ice), _ptr(n_tot), _ptr(m_tot), C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return n_tot, m_tot


ModeColumns = namedtuple("ModeColumns", ["r_dry", "stdev", "N", "hygroscopicity", "molar_mass_mix"], defaults=(None,))
ModeColumns.__doc__ = """One aerosol mode whose descriptors vary in space: device columns r_dry [m], stdev, N [1/m³], the mode's mean
hygroscopicity (B̄ of Mode_B or κ̄ of Mode_κ: `Mode_B.hygroscopicity` / `Mode_kappa.hygroscopicity` work elementwise on
tensors too) and, if M_act is wanted, molar_mass_mix = Σ w_j M_j."""


def aerosol_activation_columns(ap, modes: Sequence[ModeColumns], aip, tps, T, p, w, q_tot, q_liq=None, q_ice=None, N_liq=None,
                               N_ice=None, *, want=("N_act",), out: Optional[ActivationResult] = None, stream=None) -> ActivationResult:
    """ARG2000 activation when the aerosol itself varies in space — the reference's own KA kernel builds one
    `AerosolDistribution` per element from columns (aerosol_activation_kernel!, test/gpu_tests.jl:45-79).  Same
    outputs as `aerosol_activation`."""
    modes = tuple(modes)
    nm = len(modes)
    if not 1 <= nm <= _abi.CMX_ARG_MAX_MODES:
        raise ValueError(f"1 … {_abi.CMX_ARG_MAX_MODES} modes supported")
    cols = [c for c in (T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice) if c is not None]
    names = ["T", "p", "w", "q_tot", "q_liq", "q_ice", "N_liq", "N_ice"][:len(cols)]
    for k, m in enumerate(modes):
        for f in ("r_dry", "stdev", "N", "hygroscopicity"):
            cols.append(getattr(m, f)); names.append(f"modes[{k}].{f}")
        if m.molar_mass_mix is not None:
            cols.append(m.molar_mass_mix); names.append(f"modes[{k}].molar_mass_mix")
    ref = _check_cols(cols, names)
    fam = _fam_of(ref)
    if not (isinstance(ap, fam.aerosol_activation_params) and isinstance(aip, fam.air_properties) and isinstance(tps, fam.thermo)):
        raise TypeError("parameter float type does not match the state columns")
    unknown = set(want) - {"N_act", "M_act", "S_max"}
    if unknown:
        raise ValueError(f"unknown output(s) {sorted(unknown)}")
    have_mm = all(m.molar_mass_mix is not None for m in modes)
    if "M_act" in want and not have_mm:
        raise ValueError("M_act needs molar_mass_mix for every mode")
    if out is not None:          # caller-provided output columns (KA-kernel style): no allocation in a time loop
        n_act, m_act, s_max = out.N_act, out.M_act, out.S_max
        given = [c for grp in (n_act, m_act) if grp is not None for c in grp] + ([s_max] if s_max is not None else [])
        if (n_act is not None and len(n_act) != nm) or (m_act is not None and len(m_act) != nm):
            raise ValueError("out: one column per mode")
        if m_act is not None and not have_mm:
            raise ValueError("M_act needs molar_mass_mix for every mode")
        _check_cols([ref] + given, ["T"] + ["out"] * len(given))
    else:
        n_act = tuple(torch.empty_like(ref) for _ in range(nm)) if "N_act" in want else None
        m_act = tuple(torch.empty_like(ref) for _ in range(nm)) if "M_act" in want else None
        s_max = torch.empty_like(ref) if "S_max" in want else None
    arr = lambda cols_: (C.c_void_p * nm)(*[c.data_ptr() for c in cols_]) if cols_ is not None else None  # noqa: E731
    s = stream if stream is not None else torch.cuda.current_stream(ref.device)
    fn = getattr(_lib.lib(), f"cmx_arg2000_activation_columns_{fam.sfx}")
    with torch.cuda.device(ref.device):
        st = fn(C.byref(ap), C.byref(aip), C.byref(tps), nm, ref.numel(), _ptr(T), _ptr(p), _ptr(w), _ptr(q_tot), _ptr(q_liq),
                _ptr(q_ice), _ptr(N_liq), _ptr(N_ice), arr([m.r_dry for m in modes]), arr([m.stdev for m in modes]),
                arr([m.N for m in modes]), arr([m.hygroscopicity for m in modes]),
                arr([m.molar_mass_mix for m in modes]) if have_mm else None, arr(n_act), arr(m_act), _ptr(s_max),
                C.c_void_p(s.cuda_stream))
    _lib.check(fn.__name__, st)
    return ActivationResult(n_act, m_act, s_max)
