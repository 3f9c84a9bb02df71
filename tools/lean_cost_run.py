#!/usr/bin/env python3
"""Launches cmx_lean_eval_literal_f64 (the Float64 elementary functions of csrc/cmx_lean_f64.hpp, built like the production Float64 kernels) once per function
on 2^22 normal positive arguments, in the order of FUNCS.  Run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES`: the per-dispatch counters give the DYNAMIC VALU
instructions per wave of each function (the static listing counts the slow paths for zeros / subnormals / infinities, which no lane takes here);
tools/f64_floor.py reads the CSV.

    cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $REPO/gpurun_out/lean_cost -o p -- python3 $REPO/tools/lean_cost_run.py"""
import ctypes as C
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "cloudmicrophysics.jl_amd"))
from cmx import _lib  # noqa: E402

# (which, name, argument range) — the launch ORDER is what tools/f64_floor.py relies on
FUNCS = [(19, "identity", (0.5, 2.0)), (0, "exp2", (-20.0, 20.0)), (11, "exp2_fin", (-20.0, 20.0)), (1, "log2", (1e-6, 1e6)), (4, "rcp", (1e-6, 1e6)),
         (14, "rcp_nz", (1e-6, 1e6)), (13, "rcp_nz1", (1e-6, 1e6)), (5, "sqrt", (1e-6, 1e6)), (15, "sqrt_pos", (1e-6, 1e6)), (6, "rsqrt", (1e-6, 1e6)),
         (16, "rsqrt_pos", (1e-6, 1e6)), (17, "pow_m34_pos", (1.0, 1e6)), (9, "erfc", (-2.0, 5.0)), (7, "expm1", (-3.0, 3.0)), (8, "log1p", (-0.5, 10.0))]


def main():
    lib = _lib.lib()
    n = 1 << 22
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    for which, name, (lo, hi) in FUNCS:
        x = lo + (hi - lo) * torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        torch.cuda.synchronize()
        st = lib.cmx_lean_eval_literal_f64(which, n, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), None)
        assert st == 0, (name, st)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y).all()), name
    print("launched", len(FUNCS), "functions on", n, "points")


if __name__ == "__main__":
    main()
