"""Loader for libcmx.so (the C-ABI shared library, include/cmx.h).

There is no fallback: if the HIP extension is missing or fails to load, every
product entry point raises `CmxLibraryError`.  Build it with
`python __graft_entry__.py build` (or `make -C cloudmicrophysics.jl_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from . import _abi

CSRC_DIR = Path(__file__).resolve().parent.parent / "csrc"
LIB_PATH = Path(os.environ.get("CMX_LIB", CSRC_DIR / "libcmx.so"))


class CmxLibraryError(RuntimeError):
    pass


class CmxStatusError(RuntimeError):
    def __init__(self, fn, status, detail=""):
        self.status = status
        names = {_abi.CMX_ERR_BAD_ARG: "CMX_ERR_BAD_ARG", _abi.CMX_ERR_HIP: "CMX_ERR_HIP",
                 _abi.CMX_ERR_UNSUPPORTED: "CMX_ERR_UNSUPPORTED"}
        super().__init__(f"{fn} returned {names.get(status, status)}{': ' + detail if detail else ''}")


_lib = None

_FP = {"f32": C.POINTER(C.c_float), "f64": C.POINTER(C.c_double)}


def _declare(lib):
    vp, i32, i64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
    lib.cmx_version.restype = i32
    lib.cmx_version.argtypes = []
    lib.cmx_last_hip_error.restype = C.c_char_p
    lib.cmx_last_hip_error.argtypes = []
    lib.cmx_lean_eval_f64.restype = i32
    lib.cmx_lean_eval_f64.argtypes = [i32, i64, vp, vp, vp]
    lib.cmx_lean_eval_literal_f64.restype = i32
    lib.cmx_lean_eval_literal_f64.argtypes = [i32, i64, vp, vp, vp]
    for fam in (_abi.F32, _abi.F64):
        s = fam.sfx
        f = getattr(lib, f"cmx_sb2006_warm_rain_tendencies_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.warm_rain_2m), C.POINTER(fam.thermo), C.POINTER(fam.rain_vel), u32, i64] + [vp] * 13 + [vp]
        f = getattr(lib, f"cmx_sb2006_process_rates_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.warm_rain_2m), C.POINTER(fam.thermo), C.POINTER(fam.rain_vel), u32, i64] + [vp] * 7 + [
            C.POINTER(vp), vp]
        f = getattr(lib, f"cmx_ice_nucleation_rates_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.thermo), C.POINTER(fam.abifm_dust), C.POINTER(fam.koop2000), u32, i64] + [vp] * 8 + [vp, vp]
        f = getattr(lib, f"cmx_water_activity_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.thermo), i64] + [vp] * 4 + [vp]
        f = getattr(lib, f"cmx_ice_nucleation_rates_xT_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.thermo), C.POINTER(fam.abifm_dust), C.POINTER(fam.koop2000), C.POINTER(fam.h2so4_solution_params), u32, i64] + [vp] * 8 \
            + [vp, vp]
        f = getattr(lib, f"cmx_h2so4_solution_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.h2so4_solution_params), C.POINTER(fam.thermo), i64] + [vp] * 4 + [vp]
        f = getattr(lib, f"cmx_mohler2006_deposition_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.mohler_dust), C.POINTER(fam.mohler2006), i64] + [vp] * 7 + [vp]
        f = getattr(lib, f"cmx_deposition_J_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.deposition_dust), i64, vp, vp, vp]
        f = getattr(lib, f"cmx_inp_concentration_frequency_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.frostenberg2023), i64, vp, vp, vp, vp]
        f = getattr(lib, f"cmx_arg2000_total_activated_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.aerosol_activation_params), C.POINTER(fam.aerosol_distribution), C.POINTER(fam.air_properties),
                      C.POINTER(fam.thermo), i64] + [vp] * 8 + [vp, vp, vp]
        f = getattr(lib, f"cmx_mp0m_tendencies_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.parameters_0m), i64] + [vp] * 5 + [vp]
        f = getattr(lib, f"cmx_mp1m_tendencies_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.microphysics_1m), C.POINTER(fam.thermo), u32, i64] + [vp] * 11 + [vp]
        f = getattr(lib, f"cmx_bulk_2m_cloud_to_rain_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.bulk_2m_schemes), u32, i64] + [vp] * 6 + [vp]
        f = getattr(lib, f"cmx_sedimentation_velocities_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.microphysics_1m), C.POINTER(fam.stokes_vel), C.POINTER(fam.chen2022_rain_vel),
                      C.POINTER(fam.chen2022_ice_vel), i64] + [vp] * 9 + [vp]
        f = getattr(lib, f"cmx_arg2000_activation_columns_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.aerosol_activation_params), C.POINTER(fam.air_properties), C.POINTER(fam.thermo), i32, i64] \
            + [vp] * 8 + [vp] * 5 + [vp] * 3 + [vp]
        f = getattr(lib, f"cmx_sb2006_cloud_terminal_velocity_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.cloud_pdf_sb2006), C.POINTER(fam.stokes_vel), i64] + [vp] * 5 + [vp]
        f = getattr(lib, f"cmx_sb2006_column_tendencies_sedimentation_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.warm_rain_2m), C.POINTER(fam.thermo), C.POINTER(fam.rain_vel), C.POINTER(fam.stokes_vel), u32, i64, i32] \
            + [vp] * 13 + [vp]
        f = getattr(lib, f"cmx_mp1m_linearized_average_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.microphysics_1m), C.POINTER(fam.thermo), u32, fam.ft, fam.ft, i32, i64] + [vp] * 11 + [vp]
        f = getattr(lib, f"cmx_mp1m_source_terms_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.microphysics_1m), C.POINTER(fam.thermo), u32, i64] + [vp] * 7 + [C.POINTER(vp), vp]
        f = getattr(lib, f"cmx_mp1m_terminal_velocity_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.microphysics_1m), C.POINTER(fam.chen2022_rain_vel), i64] + [vp] * 6 + [vp]
        f = getattr(lib, f"cmx_arg2000_activation_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.aerosol_activation_params), C.POINTER(fam.aerosol_distribution),
                      C.POINTER(fam.air_properties), C.POINTER(fam.thermo), i64] + [vp] * 8 + [C.POINTER(vp), C.POINTER(vp), vp, vp]
        f = getattr(lib, f"cmx_p3_shape_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.p3_params), u32, i32, i64] + [vp] * 10 + [vp]
        f = getattr(lib, f"cmx_p3_terminal_velocities_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.p3_params), C.POINTER(fam.chen2022_ice_vel), C.POINTER(fam.quadrature), u32, fam.ft, i64] \
            + [vp] * 8 + [vp]
        f = getattr(lib, f"cmx_p3_shape_terminal_velocities_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.p3_params), C.POINTER(fam.chen2022_ice_vel), C.POINTER(fam.quadrature), u32, i32, fam.ft, i64] \
            + [vp] * 10 + [vp]
        f = getattr(lib, f"cmx_p3_ice_melt_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.p3_params), C.POINTER(fam.chen2022_ice_vel), C.POINTER(fam.air_properties), C.POINTER(fam.thermo),
                      C.POINTER(fam.ventilation), C.POINTER(fam.quadrature), u32, fam.ft, i64] + [vp] * 9 + [vp]
        f = getattr(lib, f"cmx_p3_ice_self_collection_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.p3_params), C.POINTER(fam.chen2022_ice_vel), C.POINTER(fam.quadrature), u32, i64] + [vp] * 7 + [vp]
        f = getattr(lib, f"cmx_liquid_freezing_rate_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.p3_ice_params), C.POINTER(fam.thermo), u32, i64] + [vp] * 6 + [vp]
        f = getattr(lib, f"cmx_p3_het_ice_nucleation_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.abifm_dust), C.POINTER(fam.thermo), i64] + [vp] * 7 + [vp]
        f = getattr(lib, f"cmx_p3_liquid_ice_collisions_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.p3_ice_params), C.POINTER(fam.air_properties), C.POINTER(fam.thermo), C.POINTER(fam.quadrature),
                      u32, i64] + [vp] * 11 + [C.POINTER(vp), C.POINTER(vp), vp]
        f = getattr(lib, f"cmx_sb2006_warm_rain_tendencies_fields_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.warm_rain_2m), C.POINTER(fam.thermo), u32, i64, i64, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp),
                      C.POINTER(i64), vp, vp]
        f = getattr(lib, f"cmx_mp1m_column_tendencies_sedimentation_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.microphysics_1m), C.POINTER(fam.thermo), C.POINTER(fam.stokes_vel), C.POINTER(fam.chen2022_rain_vel),
                      C.POINTER(fam.chen2022_ice_vel), u32, fam.ft, fam.ft, i32, i64, i32, vp, C.POINTER(vp), C.POINTER(vp), vp, vp, vp]
        f = getattr(lib, f"cmx_mp1m_linearized_average_fields_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.microphysics_1m), C.POINTER(fam.thermo), u32, fam.ft, fam.ft, i32, i64, i64, C.POINTER(vp), C.POINTER(i64),
                      C.POINTER(vp), C.POINTER(i64), vp, vp]
        f = getattr(lib, f"cmx_mp1m_tendencies_fields_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.microphysics_1m), C.POINTER(fam.thermo), u32, i64, i64, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp),
                      C.POINTER(i64), vp, vp]
        f = getattr(lib, f"cmx_microphysics_2m_p3_tendencies_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.warm_rain_2m), C.POINTER(fam.p3_ice_params), C.POINTER(fam.thermo), u32, i64] + [vp] * 13 + \
                     [C.POINTER(vp), vp]
        f = getattr(lib, f"cmx_microphysics_2m_p3_tendencies_fields_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.warm_rain_2m), C.POINTER(fam.p3_ice_params), C.POINTER(fam.thermo), u32, i64, i64, C.POINTER(vp), C.POINTER(i64),
                      C.POINTER(vp), C.POINTER(i64), vp]
        f = getattr(lib, f"cmx_gamma_inc_{s}")
        f.restype = i32
        f.argtypes = [i64, vp, vp, vp, vp, vp]
        f = getattr(lib, f"cmx_gamma_inc_inv_{s}")
        f.restype = i32
        f.argtypes = [i64, vp, vp, vp, vp, vp]
        f = getattr(lib, f"cmx_generalized_gamma_{s}")
        f.restype = i32
        f.argtypes = [fam.ft, fam.ft, i64, vp, vp, vp, vp, vp, vp]
        f = getattr(lib, f"cmx_exponential_distribution_{s}")
        f.restype = i32
        f.argtypes = [i64, vp, vp, vp, vp, vp, vp]
        f = getattr(lib, f"cmx_sb2006_size_distribution_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.cloud_pdf_sb2006), C.POINTER(fam.rain_pdf_sb2006), u32, fam.ft, i64] + [vp] * 7 + [vp]
        f = getattr(lib, f"cmx_cloud_diagnostics_{s}")
        f.restype = i32
        f.argtypes = [C.POINTER(fam.rain), C.POINTER(fam.cloud_pdf_sb2006), C.POINTER(fam.rain_pdf_sb2006), fam.ft, u32, i64] + [vp] * 9 + [vp]
        f = getattr(lib, f"cmx_column_sums_{s}")
        f.restype = i32
        f.argtypes = [i32, C.POINTER(vp), i64, vp, vp, vp]


def lib():
    """The loaded library; raises CmxLibraryError (never falls back) if it is unavailable."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise CmxLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            handle = C.CDLL(str(LIB_PATH))
        except OSError as e:  # missing libamdhip64 etc.
            raise CmxLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        _declare(handle)
        _lib = handle
    return _lib


def check(fn_name, status):
    if status < 0:
        detail = lib().cmx_last_hip_error().decode() if status == _abi.CMX_ERR_HIP else ""
        raise CmxStatusError(fn_name, status, detail)
    return status
