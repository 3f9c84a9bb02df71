#!/usr/bin/env python3
"""bench.py — SB2006 two-moment warm-rain fused-tendency sweep (BASELINE.json metric).

One "step" = one pass of the fused kernel (7 state columns in, 4 tendencies + 2 rain fall-speed columns
out) over the rank's resident batch of synthetic grid points (default 1e8 Float32 points per GPU — the
configuration BASELINE.json quotes the metric on).  Inputs are generated on the device before the timed
region; nothing crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points P] [--dtype f32|f64] [--scaling weak|strong]
                    [--rotate K] [--workload sb2006|sb2006_chen|sb2006_column|sb2006_aos|sb2006_fields|icenuc|mp0m|mp1m|mp1m_lin|mp1m_column|mp1m_column_lin|
                                             arg2000|arg2000_columns|p3|p3_split|p3_selfcol|mp2m_p3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: called as a plain process with `--gpus N` (no WORLD_SIZE in the environment) this
script is a launcher — before anything touches torch or the GPU it starts the second form as a child process
(cmx/launcher.py), relays rank 0's JSON line and exits with the child's code.  `--scaling weak` (default) keeps
`--points` per GPU; `--scaling strong` splits `--points` over the ranks with cmx.sharding.shard_bounds.

Two timed regions of exactly K steps each, both bracketed by barrier + synchronize: one re-sweeps ONE buffer set (`same_buffer_ms_per_step`),
one visits `--rotate` disjoint buffer sets round-robin (`rotating_ms_per_step`).  `ms_per_step` / `value` are the same-buffer figures unless the
rotating region is more than 5 % slower — then they are the rotating ones (`value_uses` says which).

Rank 0 prints ONE JSON line (contract: see the task statement; DESIGN.md §7 explains every field).
`--workload` selects one of the other hot-path kernels for roofline measurements (same JSON shape); the default,
and the line the driver records, is the north-star SB2006 sweep.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "cloudmicrophysics.jl_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ≈6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=30,
                    help="untimed launches before the warm-up: after idle the first ≈15 launches run 5–20 %% slower while the clocks settle")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="… and keep launching (untimed) until this much time has passed since the first launch: the clock / power state settles on a "
                         "time scale of ~0.1 s, not on a launch count (one box needed > 70 launches of the 0.85-ms sweep; profiles/r04_round.log)")
    ap.add_argument("--points", type=int, default=100_000_000, help="grid points per GPU (weak scaling) or in total (strong scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --points per GPU; strong: --points in total, sharded over the ranks (cmx.sharding.shard_bounds)")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--workload", choices=["sb2006", "sb2006_chen", "sb2006_column", "icenuc", "mp0m", "mp1m", "mp1m_lin", "mp1m_column", "mp1m_column_lin", "arg2000", "arg2000_columns", "p3", "p3_split", "p3_selfcol", "mp2m_p3", "sb2006_aos", "sb2006_fields", "cloud_diag"], default="sb2006")
    ap.add_argument("--rotate", type=int, default=4,
                    help="number of DISJOINT input/output buffer sets visited round-robin in the rotating timed region, so that no step re-touches the "
                         "pages of the previous one (what a model time loop sees: other arrays are touched between two microphysics calls); 1 = off")
    ap.add_argument("--no-cold-probes", action="store_true", help="skip the idle / fresh-buffer probes that separate clock ramp from TLB effects")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend for N > 1: nccl (= RCCL over xGMI; the measured configuration) or gloo (TEST MODE: ranks may share "
                         "a device — local_rank modulo the device count — so the N > 1 code path can be exercised on a 1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-telemetry", action="store_true",
                    help="skip the engine-clock / package-power samples (amdgpu sysfs reads while the kernel loops, outside every timed region)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the bounded baseline sample")
    ap.add_argument("--dry-run-scaffolding", action="store_true",
                    help="TEST MODE (tests/test_launcher.py, no GPU): run only the multi-rank scaffolding — sharding, barriers, max-over-ranks timing, "
                         "reductions, rank-0 JSON — over gloo with a step that does nothing; the line carries \"dry_run\": true and no throughput")
    ap.add_argument("--diagnostics", action="store_true",
                    help="also reduce Σ of the output columns per step (block reduce + RCCL all-reduce of a few doubles)")
    return ap.parse_args()


def usable_cores() -> int:
    """Host threads this process may actually run on: affinity mask ∩ cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 quota ("max 100000" or "<quota> <period>")
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pmc_traffic(workload: str, dtype: str, n: int):
    """HBM bytes per launch measured with rocprofv3 PMC counters (FETCH_SIZE, WRITE_SIZE; separate passes, gfx950
    FETCH_SIZE ×2 correction of MI355X_MICROARCH.md §HBM) for this exact workload, if a committed profile matches;
    PMC counters cannot be read from inside the benchmark process, so this is null otherwise."""
    tag = "" if workload == "sb2006" else f"_{workload}"
    for p in sorted((REPO / "profiles").glob(f"r*_pmc_traffic{tag}_{dtype}.json"), reverse=True):
        try:
            d = json.loads(p.read_text())
        except (OSError, ValueError):
            continue
        if d.get("points") == n and d.get("dtype") == dtype:
            return d.get("hbm_bytes_per_launch"), str(p.relative_to(REPO))
    return None, None


def source_digest() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.hpp, *.inc, Makefile, include/cmx.h): a PMC instruction count taken from another build of
    the kernels is not a property of THIS code — pmc_valu() only accepts a committed profile whose digest matches."""
    import hashlib
    h = hashlib.sha256()
    src = REPO / "cloudmicrophysics.jl_amd" / "csrc"
    files = sorted(list(src.glob("*.hip")) + list(src.glob("*.hpp")) + list(src.glob("*.inc")) + [src / "Makefile", REPO / "include" / "cmx.h"])
    for f in files:
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


# The VALU ceiling, in wave64 instructions per second: 256 CUs × 4 SIMDs issuing one instruction per 2.4 cycles of the 2.4 GHz clock — the
# FASTEST rate any VALU instruction was measured to issue at on this part (tools/valu_probe.hip → profiles/rNN_probe_valu.txt: v_fmamk_f32,
# v_mul_f32 with a literal 2.37–2.40 cycles; v_add/v_mul 2.5–2.9; v_fma_f32 with three registers, any SGPR operand, compares and selects
# 4.2–4.7; every Float64 instruction 4.2–4.8; transcendentals 8.2 (f32) / 16.3 (f64)).  Round 2's "one per 4 cycles" put Float32 kernels
# above 1.  With this ceiling the fraction is ≤ 1 by construction; a Float64 kernel cannot exceed ≈ 0.55 of it (its instructions take ≥ 4.2
# cycles each) — `issue_slots_4cycle` in the same object is the old figure.
VALU_ISSUE_CYCLES_MIN = 2.4
VALU_PEAK_GINST = 1024 * 2.4e9 / VALU_ISSUE_CYCLES_MIN / 1e9
# … and the guide's own figure (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on the SIMD-32): 1228.8 G/s.  Both fractions
# are printed (`frac` against the measured 2.4 cycles, `frac_vs_guide` against 2 cycles = 0.83 × frac).
VALU_PEAK_GINST_GUIDE = 1024 * 2.4e9 / 2.0 / 1e9


def pmc_valu(workload: str, dtype: str, n: int):
    """Wave64 VALU instructions per launch (SQ_INSTS_VALU, summed over the kernels of one step) from the committed rocprofv3 PMC pass of
    this exact workload (tools/profile.sh … valu → profiles/rNN_pmc_valu_<workload>_<dtype>.json), newest round first; (None, None) if
    no committed profile matches the size AND the kernel sources (source_digest).  The count is a property of the code and the input
    distribution, not of the box."""
    for p in sorted((REPO / "profiles").glob(f"r*_pmc_valu_{workload}_{dtype}.json"), reverse=True):
        try:
            d = json.loads(p.read_text())
        except (OSError, ValueError):
            continue
        if d.get("points") != n or d.get("source_digest") != source_digest():
            continue
        insts = [k.get("counters", {}).get("SQ_INSTS_VALU") for k in d.get("kernels", {}).values()]
        if insts and all(v is not None for v in insts):
            return float(sum(insts)), str(p.relative_to(REPO))
    return None, None


# ------------------------------------------------------------------------------------------------------------
# workloads: each returns (state columns, step(), output columns, description dict, cpu_run(cols_np, m, threads))
# ------------------------------------------------------------------------------------------------------------
def setup_sb2006(args, dev, dtype, rank):
    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    state = synthetic.sb2006_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    mp, tps = P.Microphysics2MParams(args.dtype), P.ThermodynamicsParameters(args.dtype)
    out = cmx.WarmRainTendencies2M(*[__import__("torch").empty_like(state.rho) for _ in range(6)])
    scheme = cmx.Microphysics2Moment()

    chen = args.workload == "sb2006_chen"          # the Chen-2022 rain fall-speed variant of the same fused sweep (CM2:703-719)
    velty = cmx.Chen2022VelTypeRain if chen else cmx.SB2006VelType

    def step():
        cmx.bulk_microphysics_tendencies(scheme, mp, tps, *state, vel=velty, out=out)

    def cpu_run(ob, cols, threads):
        fam = _abi.family(args.dtype)
        wr, t, vel = P.WarmRainParams2M(args.dtype).c, P.ThermodynamicsParameters(args.dtype), P.rain_vel_params(args.dtype)
        flags = _abi.CMX_SB2006_LIMITED | (_abi.CMX_VEL_CHEN2022 if chen else _abi.CMX_VEL_SB2006)
        return lambda: ob.sb2006_warm_rain_tendencies(fam, wr, t, vel, flags, *cols, nthreads=threads, want_scale=False)

    desc = {
        "metric": "grid-points/sec SB2006 2M tendency sweep",
        "bytes_per_point": {"f32": 52, "f64": 104}[args.dtype],     # 7 in + 4 tendencies + 2 velocities
        "kernel": "sb2006_tendencies_kernel",
        "workload": "Microphysics2M SB2006 fused warm-rain tendencies (cond/evap, autoconversion, accretion, "
                    "self-collection, breakup, evaporation, number adjustment) + " + ("Chen-2022" if chen else "SB2006") +
                    " rain terminal velocities, limited rain PSD",
        "columns_in": 7, "columns_out": 6, "diag_cols": list(out[:4]),
    }
    return list(state), step, desc, cpu_run


def setup_sb2006_column(args, dev, dtype, rank):
    """the north-star tendencies fused with the host model's upwind sedimentation step, columns of 74 levels (the RCEMIP
    column of test/gpu_clima_core_test.jl:88-100); 7 columns in, 4 out."""
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    n_lev = 74
    n_col = max(1, args.points // n_lev)
    args.points = n_col * n_lev
    st = synthetic.sb2006_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    cols = [c.reshape(n_col, n_lev) for c in st]
    g = torch.Generator(device="cpu").manual_seed(7)
    inv_dz_cpu = (1.0 / (30.0 + 470.0 * torch.rand(n_lev, generator=g, dtype=torch.float64))).to(dtype)
    inv_dz = inv_dz_cpu.to(dev)
    mp, tps = P.Microphysics2MParams(args.dtype), P.ThermodynamicsParameters(args.dtype)
    out = cmx.ColumnTendencies2M(*[torch.empty_like(cols[0]) for _ in range(4)], None)

    def step():
        cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *cols, vel=cmx.SB2006VelType, out=out)

    def cpu_run(ob, c, threads):
        fam = _abi.family(args.dtype)
        wr, t, vel = P.WarmRainParams2M(args.dtype).c, P.ThermodynamicsParameters(args.dtype), P.rain_vel_params(args.dtype)
        flags = _abi.CMX_SB2006_LIMITED | _abi.CMX_VEL_SB2006
        m_col = c[0].size // n_lev
        c2 = [a[:m_col * n_lev].reshape(m_col, n_lev) for a in c]
        return lambda: ob.sb2006_column_tendencies_sedimentation(fam, wr, t, vel, None, flags, inv_dz_cpu.numpy(), *c2, nthreads=threads)

    desc = {
        "metric": "grid-points/sec SB2006 2M tendency + upwind sedimentation column sweep (74 levels)",
        "bytes_per_point": {"f32": 44, "f64": 88}[args.dtype],     # 7 in + 4 out; the unfused sequence moves 88 / 176 B per point
        "kernel": "sb2006_column_kernel",
        "workload": "Microphysics2M SB2006 fused warm-rain tendencies + SB2006 rain fall speeds + first-order upwind sedimentation flux "
                    "divergence of q_rai, n_rai per column of 74 levels (host-model step, SURVEY 8f-4)",
        "columns_in": 7, "columns_out": 4, "diag_cols": list(out[:4]),
    }
    return list(st), step, desc, cpu_run


def setup_sb2006_layout(args, dev, dtype, rank):
    """The north-star tendencies behind the host model's layouts: `sb2006_aos` writes the reference's own result type —
    an array of 8-field NamedTuples, 4 fields identically zero (test/gpu_performance.jl:212-216); `sb2006_fields` reads and writes
    components of ClimaCore VIJFH fields in place (state field with 9 components, tendency field with 5; runs of Nv·Ni·Nj = 74·16)."""
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    mp, tps = P.Microphysics2MParams(args.dtype), P.ThermodynamicsParameters(args.dtype)
    scheme = cmx.Microphysics2Moment()
    aos = args.workload == "sb2006_aos"
    S = 74 * 16
    n = args.points if aos else (args.points // S) * S
    state = synthetic.sb2006_state(n, dtype=dtype, device=dev, seed=1234 + rank)
    holder = {}
    if aos:
        cols = list(state)

        def step():
            holder["out"] = cmx.bulk_microphysics_tendencies_fields(scheme, mp, tps, *cols, aos=True)
    else:
        Nh = n // S
        Y = torch.empty((Nh, 9, S), dtype=dtype, device=dev)
        for f, c in enumerate(state):
            Y[:, f + 1, :] = c.reshape(Nh, S)
        Yt = torch.empty((Nh, 5, S), dtype=dtype, device=dev)
        cols = [Y[:, f + 1, :] for f in range(7)]
        outs = [Yt[:, k, :] for k in range(4)]

        def step():
            cmx.bulk_microphysics_tendencies_fields(scheme, mp, tps, *cols, out=outs)

    def cpu_run(ob, c, threads):
        fam = _abi.family(args.dtype)
        wr, t = P.WarmRainParams2M(args.dtype).c, P.ThermodynamicsParameters(args.dtype)
        return lambda: ob.sb2006_warm_rain_tendencies(fam, wr, t, None, _abi.CMX_SB2006_LIMITED, *c, nthreads=threads, want_scale=False)

    args.points = n
    desc = {
        "metric": "grid-points/sec SB2006 2M tendency sweep, " + ("array-of-NamedTuples output" if aos else "ClimaCore VIJFH fields in place"),
        "bytes_per_point": ({"f32": 60, "f64": 120} if aos else {"f32": 44, "f64": 88})[args.dtype],     # 7 in + 8 (AoS) / 4 out
        "kernel": "sb2006_tendencies_layout_kernel",
        "workload": "Microphysics2M SB2006 fused warm-rain tendencies, limited rain PSD, " +
                    ("result written as Vector{NamedTuple} (8 fields, 4 zero) through an LDS transpose" if aos else
                     "7 components of a 9-component VIJFH state field → 4 components of a 5-component tendency field"),
        "columns_in": 7, "columns_out": 8 if aos else 4, "diag_cols": [],
    }
    return [c.contiguous() for c in state], step, desc, cpu_run


def setup_icenuc(args, dev, dtype, rank):
    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    state = synthetic.ice_nucleation_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    tps, dust, koop = P.ThermodynamicsParameters(args.dtype), P.Kaolinite(args.dtype), P.Koop2000(args.dtype)
    holder = {}

    holder["out"] = cmx.ice_nucleation_rates(tps, dust, koop, *state, count_domain_errors=True)

    def step():   # reuse the output columns and the (accumulating) domain-error counter: no allocation in the loop
        cmx.ice_nucleation_rates(tps, dust, koop, *state, out=holder["out"])

    def cpu_run(ob, cols, threads):
        fam = _abi.family(args.dtype)
        return lambda: ob.ice_nucleation_rates(fam, tps, dust, koop, 0, *cols)   # scalar port: 1 thread

    step()
    desc = {
        "metric": "grid-points/sec ABIFM + Koop2000 ice-nucleation rate sweep",
        "bytes_per_point": {"f32": 20, "f64": 40}[args.dtype],      # 3 in + 2 out
        "kernel": "ice_nucleation_kernel",
        "workload": "IceNucleation ABIFM immersion (kaolinite) + Koop2000 cubic homogeneous rates over (T, a_w, r)",
        "columns_in": 3, "columns_out": 2, "diag_cols": [holder["out"].rate_het],
        "cpu_threads": 1,
    }
    return list(state), step, desc, cpu_run


def setup_cloud_diag(args, dev, dtype, rank):
    """CloudDiagnostics (src/CloudDiagnostics.jl) over the SB2006 synthetic state: 1M and 2M radar reflectivity, 2M and Liu–Hallett effective radius in ONE pass."""
    import torch

    from cmx import _abi
    from cmx import cloud_diagnostics as CD
    from cmx import parameters as P
    from cmx import synthetic
    st = synthetic.sb2006_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    rho, q_lcl, q_rai = st.rho, st.q_lcl, st.q_rai
    N_lcl, N_rai = (st.rho * st.n_lcl).contiguous(), (st.rho * st.n_rai).contiguous()      # per m³, as the reference's diagnostics take them
    sb, rain = P.SB2006(args.dtype), P.Microphysics1MParams(args.dtype).c.rain
    out = CD.Diagnostics(*[torch.empty_like(rho) for _ in range(4)])

    def step():
        CD.cloud_diagnostics(rain, sb, 1000.0, rho, q_lcl, q_rai, N_lcl, N_rai, out=out)

    def cpu_run(ob, cols, threads):
        fam = _abi.family(args.dtype)
        return lambda: ob.cloud_diagnostics(fam, cols[0], cols[1], cols[2], cols[3], cols[4], rain=rain, pdf_c=sb.pdf_c, pdf_r=sb.pdf_r, rho_w=1000.0, limited=True,
                                            float32_gates=(args.dtype == "f32"))   # scalar port: 1 thread

    desc = {
        "metric": "grid-points/sec cloud-diagnostics sweep (radar reflectivity 1M / 2M, effective radius 2M / Liu-Hallett)",
        "bytes_per_point": {"f32": 36, "f64": 72}[args.dtype],      # 5 in + 4 out
        "kernel": "cloud_diagnostics_kernel",
        "workload": "CloudDiagnostics radar_reflectivity_1M + radar_reflectivity_2M + effective_radius_2M + effective_radius_Liu_Hallet_97 over (rho, q_lcl, q_rai, N_lcl, N_rai), limited rain PSD",
        "columns_in": 5, "columns_out": 4, "diag_cols": [out.reff_2m],
        "cpu_threads": 1,
    }
    return [rho, q_lcl, q_rai, N_lcl, N_rai], step, desc, cpu_run


def setup_mp0m(args, dev, dtype, rank):
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    q_lcl = torch.rand(args.points, generator=g, device=dev, dtype=dtype) * 2e-3
    q_icl = torch.rand(args.points, generator=g, device=dev, dtype=dtype) * 1e-3
    mp = P.Microphysics0MParams(args.dtype)
    out = torch.empty_like(q_lcl)
    scheme = cmx.Microphysics0Moment()

    def step():
        cmx.bulk_microphysics_tendencies_0m(scheme, mp, None, q_lcl, q_lcl, q_icl, out=out)

    def cpu_run(ob, cols, threads):
        fam = _abi.family(args.dtype)
        return lambda: ob.mp0m_tendencies(fam, mp.precip, *cols)   # scalar port: 1 thread

    desc = {
        "metric": "grid-points/sec 0-moment precipitation-removal sweep",
        "bytes_per_point": {"f32": 12, "f64": 24}[args.dtype],      # 2 in + 1 out
        "kernel": "mp0m_tendencies_kernel",
        "workload": "BulkMicrophysicsTendencies Microphysics0Moment (qc_0 threshold) over (q_lcl, q_icl)",
        "columns_in": 2, "columns_out": 1, "diag_cols": [out],
        "cpu_threads": 1,
    }
    return [q_lcl, q_icl], step, desc, cpu_run


def setup_mp1m(args, dev, dtype, rank):
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    state = synthetic.mp1m_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    mp, tps = P.Microphysics1MParams(args.dtype), P.ThermodynamicsParameters(args.dtype)
    out = cmx.Tendencies1M(*[torch.empty_like(state.rho) for _ in range(4)])
    mode, scheme = cmx.Instantaneous(), cmx.Microphysics1Moment()

    def step():
        cmx.bulk_microphysics_tendencies_1m(mode, scheme, mp, tps, *state, out=out)

    def cpu_run(ob, cols, threads):
        fam = _abi.family(args.dtype)
        return lambda: ob.mp1m(fam, mp.c, tps, mp.flags, *cols, nthreads=threads, want_sources=False)

    desc = {
        "metric": "grid-points/sec 1-moment (Marshall-Palmer) tendency sweep",
        "bytes_per_point": {"f32": 44, "f64": 88}[args.dtype],      # 7 in + 4 out
        "kernel": "mp1m_tendencies_kernel",
        "workload": "Microphysics1M Instantaneous bulk tendencies, default Microphysics1MOptions (13 processes)",
        "columns_in": 7, "columns_out": 4, "diag_cols": list(out),
    }
    return list(state), step, desc, cpu_run


def setup_mp1m_lin(args, dev, dtype, rank):
    """1-moment LinearizedAverage mode (the mode ClimaAtmos runs operationally, BMT:112-115): Δt = 30 s, nsub = 2."""
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    state = synthetic.mp1m_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    mp, tps = P.Microphysics1MParams(args.dtype), P.ThermodynamicsParameters(args.dtype)
    out = cmx.Tendencies1M(*[torch.empty_like(state.rho) for _ in range(4)])
    mode, scheme = cmx.LinearizedAverage(), cmx.Microphysics1Moment()
    dt, nsub, q_min = 30.0, 2, P.DEFAULT_PARAMETERS["specific_humidity_minimum"]

    def step():
        cmx.bulk_microphysics_tendencies_1m(mode, scheme, mp, tps, *state, dt, nsub, out=out)

    def cpu_run(ob, cols, threads):
        fam = _abi.family(args.dtype)
        return lambda: ob.mp1m_linearized_average(fam, mp.c, tps, mp.flags, q_min, dt, nsub, *cols, nthreads=threads)

    desc = {
        "metric": "grid-points/sec 1-moment LinearizedAverage tendency sweep (dt = 30 s, nsub = 2)",
        "bytes_per_point": {"f32": 44, "f64": 88}[args.dtype],      # 7 in + 4 out
        "kernel": "mp1m_linearized_pair_kernel (Float32: two points per lane in packed arithmetic)" if args.dtype == "f32" else "mp1m_linearized_kernel", "bound": "valu",
        "workload": "Microphysics1M LinearizedAverage bulk tendencies: 2 linearized implicit substeps (13 processes, 4x4 sparse solve, "
                    "T update) per point",
        "columns_in": 7, "columns_out": 4, "diag_cols": list(out),
    }
    return list(state), step, desc, cpu_run


def setup_mp1m_column(args, dev, dtype, rank):
    """The operational 1-moment column step in one pass (cmx_mp1m_column_tendencies_sedimentation_*): tendencies (Instantaneous, or
    LinearizedAverage Δt = 30 s, nsub = 2 for `mp1m_column_lin`) + the four sedimentation velocities + the host model's upwind flux
    divergence, columns of 74 levels (the RCEMIP column of test/gpu_clima_core_test.jl:88-100); 7 columns in, 4 out."""
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    n_lev = 74
    n_col = max(1, args.points // n_lev)
    args.points = n_col * n_lev
    st = synthetic.mp1m_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    cols = [c.reshape(n_col, n_lev) for c in st]
    g = torch.Generator(device="cpu").manual_seed(7)
    inv_dz_cpu = (1.0 / (30.0 + 470.0 * torch.rand(n_lev, generator=g, dtype=torch.float64))).to(dtype)
    inv_dz = inv_dz_cpu.to(dev)
    mp, tps = P.Microphysics1MParams(args.dtype), P.ThermodynamicsParameters(args.dtype)
    vel = (P.StokesRegimeVelType(args.dtype), P.Chen2022VelTypeRain(args.dtype), P.Chen2022VelTypeIce(args.dtype))
    lin = args.workload == "mp1m_column_lin"
    mode, scheme = (cmx.LinearizedAverage() if lin else cmx.Instantaneous()), cmx.Microphysics1Moment()
    dt, nsub, q_min = (30.0, 2) if lin else (None, 1), None, P.DEFAULT_PARAMETERS["specific_humidity_minimum"]
    dt, nsub = dt
    holder = {}

    def step():
        holder["out"] = cmx.column_tendencies_sedimentation_1m(mode, scheme, mp, tps, *vel, inv_dz, *cols, dt, nsub)

    def cpu_run(ob, c, threads):
        fam = _abi.family(args.dtype)
        m_col = c[0].size // n_lev
        c2 = [a[:m_col * n_lev].reshape(m_col, n_lev) for a in c]
        return lambda: ob.mp1m_column_tendencies_sedimentation(fam, mp.c, tps, *vel, mp.flags, inv_dz_cpu.numpy(), *c2, q_min=q_min, dt=dt or 0.0,
                                                               nsub=nsub if lin else 0, nthreads=threads)

    step()
    desc = {
        "metric": "grid-points/sec 1-moment " + ("LinearizedAverage (dt = 30 s, nsub = 2) " if lin else "") + "tendency + 4-species upwind sedimentation column sweep (74 levels)",
        "bytes_per_point": {"f32": 44, "f64": 88}[args.dtype],     # 7 in + 4 out; the unfused sequence moves 148 / 296 B per point
        "kernel": "mp1m_column_kernel", "bound": "valu",
        "workload": "Microphysics1M " + ("LinearizedAverage" if lin else "Instantaneous") + " bulk tendencies + Stokes / Chen-2022 fall speeds of cloud liquid, "
                    "cloud ice, rain, snow + first-order upwind sedimentation flux divergence per column of 74 levels (host-model step)",
        "columns_in": 7, "columns_out": 4, "diag_cols": [],
    }
    return list(st), step, desc, cpu_run


def setup_arg2000(args, dev, dtype, rank):
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    state = synthetic.arg_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    ap, aip, tps = P.AerosolActivationParameters(args.dtype), P.AirProperties(args.dtype), P.ThermodynamicsParameters(args.dtype)
    ad = synthetic.arg_config3_distribution()
    out = cmx.ActivationResult(tuple(torch.empty_like(state.T) for _ in range(5)), None, None)

    def step():
        cmx.aerosol_activation(ap, ad, aip, tps, *state, out=out)

    def cpu_run(ob, cols, threads):
        fam = _abi.family(args.dtype)
        adc = ad.c_struct(ap, fam)
        return lambda: ob.arg2000_activation(fam, ap, adc, aip, tps, *cols, nthreads=threads)

    desc = {
        "metric": "states/sec ARG2000 aerosol activation sweep (5 modes)",
        "bytes_per_point": {"f32": 36, "f64": 72}[args.dtype],      # 4 in + 5 N_act out
        "kernel": "arg_activation_kernel",
        "workload": "AerosolActivation ARG2000 N_activated_per_mode, 5 shared lognormal kappa-modes x thermodynamic states",
        "columns_in": 4, "columns_out": 5, "diag_cols": list(out.N_act),
    }
    return list(state), step, desc, cpu_run


def setup_arg2000_columns(args, dev, dtype, rank):
    """ARG2000 with per-element modes — the form the reference's own GPU test runs (aerosol_activation_kernel!, test/gpu_tests.jl:45-79): every
    mode's (r_dry, σ, N, hygroscopicity) is a device column.  5 modes: 4 state + 5 × 4 mode columns in, 5 N_act columns out."""
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    from cmx.aerosol import ModeColumns
    state = synthetic.arg_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    ap, aip, tps = P.AerosolActivationParameters(args.dtype), P.AirProperties(args.dtype), P.ThermodynamicsParameters(args.dtype)
    ad = synthetic.arg_config3_distribution()
    adc = ad.c_struct(ap, _abi.family(args.dtype))
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    modes = []
    for k in range(adc.n_modes):
        m = adc.modes[k]
        jitter = lambda v, rel: (v * (1 + rel * (2 * torch.rand(args.points, generator=g, device=dev, dtype=torch.float64) - 1))).to(dtype)  # noqa: E731
        modes.append(ModeColumns(jitter(m.r_dry, 0.2), jitter(m.stdev, 0.05), jitter(m.N, 0.5), jitter(m.hygroscopicity, 0.2)))
    n_act = tuple(torch.empty_like(state.T) for _ in range(adc.n_modes))
    out = cmx.ActivationResult(n_act, None, None)

    def step():
        cmx.aerosol_activation_columns(ap, modes, aip, tps, *state, out=out)

    def cpu_run(ob, cols, threads):
        fam = _abi.family(args.dtype)
        m = cols[0].size
        mode_cols = [tuple(getattr(mc, f)[:m].cpu().numpy() for f in ("r_dry", "stdev", "N", "hygroscopicity")) for mc in modes]
        return lambda: ob.arg2000_activation_columns(fam, ap, aip, tps, *cols, mode_cols, nthreads=threads)

    nm = adc.n_modes
    desc = {
        "metric": "states/sec ARG2000 aerosol activation sweep, per-element modes (5 modes)",
        "bytes_per_point": (4 + 4 * nm + nm) * {"f32": 4, "f64": 8}[args.dtype],      # 4 state + 5 x 4 mode columns in, 5 N_act out
        "kernel": "arg_activation_columns_kernel",
        "workload": "AerosolActivation ARG2000 N_activated_per_mode with per-state mode descriptors (r_dry, stdev, N, hygroscopicity per mode and "
                    "state) — the reference's aerosol_activation_kernel!, test/gpu_tests.jl:45-79",
        "columns_in": 4 + 4 * nm, "columns_out": nm, "diag_cols": list(n_act),
    }
    return list(state), step, desc, cpu_run


def setup_p3(args, dev, dtype, rank):
    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    st = synthetic.p3_state(args.points, dtype=dtype, device=dev, seed=1234 + rank)
    rho_a = synthetic.p3_air_density(args.points, dtype=dtype, device=dev, seed=4321 + rank)
    p, vel = P.ParametersP3(args.dtype), P.Chen2022VelTypeIce(args.dtype)
    quad = P.ChebyshevGauss(args.dtype, 100)          # the reference's default rule (src/P3_terminal_velocity.jl:74)
    holder = {}

    fused = args.workload == "p3"             # config 5 as ONE launch (cmx_p3_shape_terminal_velocities_*); "p3_split" = the two entries back to back

    def step():   # (ρq_ice, ρn_ice, ρq_rim, ρb_rim, ρₐ) → (logλ, D_m, v_n, v_m)
        if fused:
            holder["out"] = cmx.p3_shape_and_terminal_velocities(p, vel, rho_a, *st, quad=quad)
            return
        shp = cmx.p3_shape(p, *st)
        holder["out"] = (shp, cmx.p3_terminal_velocities(p, vel, rho_a, *st, shp.log_lambda, quad=quad))

    def cpu_run(ob, cols, threads):
        fam = _abi.family(args.dtype)

        def run():
            ll = ob.p3_shape(fam, p.c, 0, *cols[:4], nthreads=threads)["log_lambda"]
            ob.p3_terminal_velocities(fam, p.c, vel, quad, 0, *cols, ll, nthreads=threads)
        return run

    step()
    desc = {
        "metric": "grid-points/sec P3 shape solve + integral properties (log-lambda, D_m, v_n, v_m)",
        "bytes_per_point": {"f32": 36, "f64": 72}[args.dtype],      # 5 in + 4 out
        "kernel": "p3_velocity_kernel<SOLVE> (one launch)" if fused else "p3_shape_kernel + p3_velocity_kernel", "bound": "valu",
        "workload": "P3Scheme state_from_prognostic + get_distribution_logλ (Brent root, incomplete-gamma moments) + D_m + "
                    "number/mass-weighted Chen-2022 fall speeds (ChebyshevGauss(100) x 4 segments, gamma_inc_inv bounds)",
        "columns_in": 5, "columns_out": 4, "diag_cols": [],
    }
    return list(st) + [rho_a], step, desc, cpu_run


def setup_p3_selfcol(args, dev, dtype, rank):
    """The reference's own P3 GPU benchmark kernel (benchmark_p3_kernel!, test/gpu_performance.jl:59-67): P3State from
    (L_ice, N_ice, F_rim, ρ_rim), get_distribution_logλ and ice_self_collection per state."""
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    from cmx import synthetic
    st = synthetic.p3_state(args.points, dtype=torch.float64, device=dev, seed=1234 + rank)
    F = torch.where(st.rho_q_ice > 0, st.rho_q_rim / st.rho_q_ice.clamp(min=1e-300), torch.zeros_like(st.rho_q_ice))
    rr = torch.where(st.rho_b_rim > 0, st.rho_q_rim / st.rho_b_rim.clamp(min=1e-300), torch.full_like(F, 400.0))
    cols = [c.to(dtype) for c in (st.rho_q_ice, st.rho_n_ice, F, rr)]
    rho_a = synthetic.p3_air_density(args.points, dtype=dtype, device=dev, seed=4321 + rank)
    p, vel = P.ParametersP3(args.dtype), P.Chen2022VelTypeIce(args.dtype)
    quad = P.GaussLegendre(args.dtype, 40)            # ClimaAtmos production quadrature order (src/Quadrature.jl:216)
    holder = {}

    def step():
        shp = cmx.p3_shape(p, *cols, from_state=True, want=("log_lambda",))
        holder["out"] = (shp.log_lambda, cmx.p3_ice_self_collection(p, vel, rho_a, *cols, shp.log_lambda, from_state=True, quad=quad))

    def cpu_run(ob, c, threads):
        fam = _abi.family(args.dtype)
        S = _abi.CMX_P3_INPUT_IS_STATE

        def run():
            ll = ob.p3_shape(fam, p.c, S, *c[:4], nthreads=threads)["log_lambda"]
            ob.p3_ice_self_collection(fam, p.c, vel, quad, S, *c, ll, nthreads=threads)
        return run

    step()
    desc = {
        "metric": "states/sec P3 benchmark kernel (log-lambda + ice self-collection, GaussLegendre(40))",
        "bytes_per_point": {"f32": 28, "f64": 56}[args.dtype],      # 5 in + 2 out
        "kernel": "p3_shape_kernel + p3_self_collection_kernel", "bound": "valu",
        "workload": "P3State + get_distribution_logλ + ice_self_collection (double quadrature, 8 n² = 12800 integrand evaluations per "
                    "state) — the reference's benchmark_p3_kernel!",
        "columns_in": 5, "columns_out": 2, "diag_cols": [],
    }
    return cols + [rho_a], step, desc, cpu_run


def setup_mp2m_p3(args, dev, dtype, rank):
    """The 2M + P3 fused entry, bulk_microphysics_tendencies(::Microphysics2Moment, mp{WR, P3IceParams}, …) (BMT:898-1083), at the
    default quadrature order 16 (P3IceParams, src/parameters/Microphysics2MParams.jl:73-80): mixed-phase states, log λ cached by
    the host model (an input, as in the reference's signature)."""
    import numpy as np
    import torch

    import cmx
    from cmx import _abi
    from cmx import parameters as P
    n = args.points
    rng = np.random.default_rng(1234 + rank)
    rho = rng.uniform(0.4, 1.3, n); T = rng.uniform(215.0, 295.0, n)
    q_lcl = np.where(rng.random(n) < 0.7, 10 ** rng.uniform(-6, -3, n), 0.0); n_lcl = 10 ** rng.uniform(6, 9, n)
    q_rai = np.where(rng.random(n) < 0.6, 10 ** rng.uniform(-7, -3, n), 0.0); n_rai = 10 ** rng.uniform(1, 6, n)
    q_ice = np.where(rng.random(n) < 0.8, 10 ** rng.uniform(-6, -3, n), 0.0); n_ice = 10 ** rng.uniform(2, 6, n)
    q_rim = np.where(rng.random(n) < 0.3, 0.0, rng.uniform(0.05, 0.9, n)) * q_ice
    b_rim = q_rim / rng.uniform(200, 800, n)
    q_tot = q_lcl + q_rai + q_ice + 10 ** rng.uniform(-5, -2, n)
    cols = [torch.from_numpy(c).to(dtype).to(dev) for c in (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim)]
    mp, tps = P.Microphysics2MParams(args.dtype, with_ice=True), P.ThermodynamicsParameters(args.dtype)
    ll = cmx.p3_shape(P.ParametersP3(args.dtype), cols[7] * cols[0], cols[8] * cols[0], cols[9] * cols[0], cols[10] * cols[0],
                      want=("log_lambda",)).log_lambda
    ll = torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll))
    holder = {}

    def step():
        holder["out"] = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *cols, ll)

    def cpu_run(ob, c, threads):
        fam = _abi.family(args.dtype)
        return lambda: ob.microphysics_2m_p3_tendencies(fam, mp.warm_rain.c, mp.ice.c, tps, mp.ice.flags, *c, nthreads=threads)

    step()
    desc = {
        "metric": "grid-points/sec 2M + P3 fused tendencies (warm rain + collisions + aggregation + melting + nucleation, GaussLegendre(16))",
        "bytes_per_point": {"f32": 80, "f64": 160}[args.dtype],      # 12 in + 8 out, each column once (one launch since round 3)
        "kernel": "p3_collision_kernel<FUSED, PointwiseExtra> (one launch: ice processes, 8 lanes per state, + the pointwise part per lane)", "bound": "valu",
        "workload": "bulk_microphysics_tendencies(Microphysics2Moment(), mp{WarmRain, P3IceParams}, …) — BMT:898-1083, quadrature_order 16",
        "columns_in": 12, "columns_out": 8, "diag_cols": [],
    }
    return cols + [ll], step, desc, cpu_run


def native_oracle_build():
    """Build the oracle for THIS machine (gcc -O3 -march=native, oracle/Makefile target `native`) into a scratch directory and return
    (path, flags text); (None, reason) if there is no compiler — the portable -O2 build that ships with the repo is used then."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("gcc") or not shutil.which("make"):
        return None, "no gcc/make on this machine"
    out = Path(tempfile.mkdtemp(prefix="cmx_oracle_native_"))
    r = subprocess.run(["make", "-C", str(REPO / "oracle"), "native", f"NATIVE_DIR={out}"], capture_output=True, text=True)
    so = out / "libcmx_oracle_native.so"
    if r.returncode != 0 or not so.exists():
        return None, "native build failed: " + r.stderr.strip()[-200:]
    return so, "gcc -O3 -march=native -ffp-contract=off"


def cpu_baseline(args, cols_np, desc, cpu_run):
    """The oracle — a C restatement of the reference's scalar arithmetic (kind 'port'; the Julia reference cannot
    run here) — timed on the host cores over repeated passes of a bounded sample of the same synthetic workload.  It is built
    for the machine it runs on (-O3 -march=native) just before it is timed; it is the checker nowhere."""
    native, flags = native_oracle_build()
    if native is not None:
        os.environ["CMX_ORACLE_LIB"] = str(native)
    else:
        flags = f"gcc -O2 (portable build: {flags})"
    sys.path.insert(0, str(REPO / "oracle"))
    import oracle_binding as ob
    cores = desc.get("cpu_threads") or usable_cores()

    def timed(threads, seconds):
        run = cpu_run(ob, cols_np, threads)
        run()                                          # first pass: page in, spin up the thread team
        passes, t0 = 0, time.perf_counter()
        while True:
            run()
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or passes >= 1000:
                return passes, dt
    # the all-cores figure (the baseline proper) and, beside it, ONE thread on a quarter of the sample (SURVEY 8d / BASELINE.md:60-72)
    passes, dt = timed(cores, args.cpu_seconds * (0.75 if cores > 1 else 1.0))
    m = cols_np[0].size
    one = None
    if cores > 1:
        full = cols_np
        cols_np = [c[:max(1, m // 4)] for c in full]
        p1, dt1 = timed(1, args.cpu_seconds * 0.25)
        one = {"value": p1 * cols_np[0].size / dt1, "sample": f"{p1} passes over {cols_np[0].size} points, {dt1:.1f} s"}
        cols_np = full
    import shutil
    return {"value": passes * m / dt, "unit": "grid-points/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "single_thread": one,
            # the reference itself (Julia) would be the "reference" kind; this image and the GPU boxes have no Julia runtime:
            "julia": shutil.which("julia") or "absent",
            "sample": f"{passes} passes over {m} of the same synthetic points ({passes * m} point evaluations), "
                      f"{args.dtype} arithmetic, oracle C restatement ({flags}, {cores} OpenMP thread(s)), {dt:.1f} s"}


def _amdgpu_sysfs_dir(index):
    """sysfs directory of the AMD GPU that HIP device `index` is: matched by PCI address when torch exposes it, else the index-th amdgpu card in
    PCI order.  None if there is no such directory (not an amdgpu box)."""
    import glob
    import os

    import torch
    cards = []
    for d in glob.glob("/sys/class/drm/card[0-9]*/device"):
        try:
            if open(os.path.join(d, "vendor")).read().strip() != "0x1002" or not os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                continue
            cards.append((os.path.basename(os.path.realpath(d)), d))      # ("0000:05:00.0", path)
        except OSError:
            continue
    cards.sort()
    if not cards:
        return None
    try:
        pr = torch.cuda.get_device_properties(index)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        for addr, d in cards:
            if addr == want:
                return d
    except (AttributeError, RuntimeError):
        pass
    return cards[index][1] if index < len(cards) else None


def _read_sysfs_telemetry(d):
    """(sclk MHz, mclk MHz, package power W) from the amdgpu sysfs files rocm-smi itself reads: the starred level of pp_dpm_sclk / pp_dpm_mclk and
    hwmon power1_average (or power1_input), in microwatts.  Plain file reads: no child process is started from this (GPU-initialised) process."""
    import glob
    import re

    def starred(name):
        try:
            for line in open(f"{d}/{name}"):
                if "*" in line and (m := re.search(r"(\d+)\s*Mhz", line, re.I)):
                    return int(m.group(1))
        except OSError:
            pass
        return None
    power = None
    for name in ("power1_average", "power1_input"):
        for f in glob.glob(f"{d}/hwmon/hwmon*/{name}"):
            try:
                power = int(open(f).read().strip()) * 1e-6
                break
            except (OSError, ValueError):
                continue
        if power is not None:
            break
    return starred("pp_dpm_sclk"), starred("pp_dpm_mclk"), power


def device_telemetry(step, kern_ms, index, seconds=1.8):
    """Engine clock, memory clock and package power WHILE the workload's kernel loops: the amdgpu sysfs files are read three times during
    about `seconds` of queued launches, after and outside every timed region.  The VALU ceilings of this file assume the 2.4 GHz spec clock;
    under the streaming kernels the package sits at its power cap and the firmware lowers the engine clock (profiles/r04_clock_probe.txt:
    1.87-2.13 GHz at 1400 W for the SB2006 / 1-moment / ARG sweeps, 2.39 GHz for the P3 kernels) — this field says which case a line is."""
    import torch
    d = _amdgpu_sysfs_dir(index)
    if d is None:
        return None
    per = max(4, min(20000, int(seconds / 3 * 1e3 / max(kern_ms, 1e-3))))
    sclk, mclk, power, sustained = [], [], [], []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(per):
            step()
        b.record()
        time.sleep(min(0.3, 0.5 * per * kern_ms * 1e-3))      # sample in the middle of the queued work, not at its first launch
        c, m, w = _read_sysfs_telemetry(d)
        if not torch.cuda.current_stream().query() and c is not None:      # the queue still held work when the sample was taken
            sclk.append(c); mclk.append(m); power.append(w)
        torch.cuda.synchronize()
        sustained.append(a.elapsed_time(b) / per)
    if not sclk:
        return None
    return {"sclk_mhz": sclk, "mclk_mhz": mclk, "package_power_w": power,
            # back-to-back launches for ~0.6 s per batch: the package reaches its power cap and the clock settles lower than in a K-step burst
            "sustained_ms_per_step": sustained, "launches_per_batch": per,
            "how": f"amdgpu sysfs (pp_dpm_sclk, pp_dpm_mclk, hwmon power1_average), read while {per} launches of the step were queued, three times "
                   "(after the timed regions; no child process); sustained_ms_per_step = wall time of each batch / launches (HIP events)"}


def cpu_model() -> str:
    """Model string of the host CPU (/proc/cpuinfo) and the number of logical CPUs the machine has."""
    try:
        txt = Path("/proc/cpuinfo").read_text()
        names = [ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("model name")]
        return f"{names[0]} ({len(names)} logical CPUs on the machine)" if names else "unknown"
    except OSError:
        return "unknown"


def load_launcher():
    """cmx/launcher.py by file path: importing the `cmx` package would import torch, and the launcher process must
    stay clear of torch and HIP (it only starts child processes)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cmx_launcher", REPO / "cloudmicrophysics.jl_amd" / "cmx" / "launcher.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    args = parse()
    launcher = load_launcher()
    if launcher.needs_launch(args.gpus):
        # plain `python bench.py --gpus N`: become the launcher of N ranks (child processes; nothing here has touched the GPU)
        sys.exit(launcher.launch(str(Path(__file__).resolve()), sys.argv[1:], args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist

    from cmx import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    dry = args.dry_run_scaffolding
    if dry:
        args.backend, args.no_cpu_baseline, args.settle = "gloo", True, 0
    n_dev = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if not dry and args.backend == "nccl" and n_dev < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        # one process per GPU over RCCL: fewer devices than ranks cannot work — say so and leave non-zero before any rank blocks in a collective
        sys.exit(f"bench.py: --gpus {world} with the nccl (RCCL) backend needs {world} visible GPUs, this node has {n_dev}")
    if not dry and n_dev == 0:
        sys.exit("bench.py: no GPU visible (the product has no CPU path)")
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, n_dev)
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    # CMX_BENCH_FORCE_DIST=1 (tests/test_bench_gpu.py, under torchrun with one rank): take the process-group path at world size 1 too,
    # so that the RCCL initialisation, barrier and reductions of the N > 1 job run on a 1-GPU box
    use_dist = world > 1 or os.environ.get("CMX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")                  # test mode: host-side barrier / reductions, ranks may share a device
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    points_arg = args.points
    if args.scaling == "strong":                             # fixed total work: rank r owns shard r of [0, points)
        lo, hi = sharding.shard_bounds(points_arg, rank, world)
        args.points = hi - lo
    # weak scaling: fixed work per GPU; rank r owns shard r of the global [0, world·n) index space.
    # Either way: disjoint seeds, no exchange
    setup = {"sb2006": setup_sb2006, "sb2006_chen": setup_sb2006, "sb2006_column": setup_sb2006_column, "icenuc": setup_icenuc, "mp0m": setup_mp0m, "cloud_diag": setup_cloud_diag, "mp1m": setup_mp1m, "mp1m_lin": setup_mp1m_lin, "mp1m_column": setup_mp1m_column, "mp1m_column_lin": setup_mp1m_column, "arg2000": setup_arg2000, "arg2000_columns": setup_arg2000_columns,
             "p3": setup_p3, "p3_split": setup_p3, "p3_selfcol": setup_p3_selfcol, "mp2m_p3": setup_mp2m_p3, "sb2006_aos": setup_sb2006_layout, "sb2006_fields": setup_sb2006_layout}[args.workload]
    if dry:
        state, kernel_step, cpu_run = [], (lambda: None), None
        desc = {"metric": "dry run of the multi-rank scaffolding (no kernel)", "bytes_per_point": 0, "kernel": "none", "workload": "none",
                "columns_in": 0, "columns_out": 0, "diag_cols": []}
    else:
        state, kernel_step, desc, cpu_run = setup(args, dev, dtype, rank)
    n = args.points                                          # a layout workload may round the size to whole field runs
    # --rotate K: K − 1 further DISJOINT buffer sets (inputs from other seeds, their own outputs) for the rotating timed region
    rotate = 1 if dry else max(1, args.rotate)
    sets = [(kernel_step, desc)]
    for k in range(1, rotate):
        _, ks, dk, _ = setup(args, dev, dtype, rank + 1000 * k)
        sets.append((ks, dk))

    def make_step(ks, dk):
        def f():
            ks()
            if args.diagnostics:
                sharding.global_diagnostics(dk["diag_cols"])
        return f
    steps = [make_step(ks, dk) for ks, dk in sets]
    step = steps[0]

    def fence():
        if use_dist:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    # the first five launches after the inputs are generated, timed one by one (clocks not settled: the cold figure next to settle_steps)
    if dry:      # no GPU: the same control flow without HIP events
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = max(time.perf_counter() - t0, 1e-9)
        t = torch.tensor([elapsed], dtype=torch.float64)
        tot = torch.tensor([float(n)], dtype=torch.float64)
        lo_hi = torch.zeros(2 * world, dtype=torch.float64)
        if args.scaling == "strong":
            lo_hi[2 * rank], lo_hi[2 * rank + 1] = sharding.shard_bounds(points_arg, rank, world)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(lo_hi, op=dist.ReduceOp.SUM)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": desc["metric"], "value": None, "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "scaling": args.scaling, "points_per_rank0": n, "points_total": int(tot.item()),
                              "shard_bounds": [[int(lo_hi[2 * r]), int(lo_hi[2 * r + 1])] for r in range(world)] if args.scaling == "strong" else None}),
                  flush=True)
        return
    def timed_each(fn_of_i, count):
        """`count` launches timed one by one with HIP events on the launch stream; returns the list of ms."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
        for i, (a, b) in enumerate(evs):
            a.record()
            fn_of_i(i)()
            b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    def timed_region(fn_of_i):
        """EXACTLY args.steps steps bracketed by barrier + synchronize on both sides: (wall seconds, mean HIP-event ms per step)."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        fence()
        t0 = time.perf_counter()
        for i, (a, b) in enumerate(evs):
            a.record()
            fn_of_i(i)()
            b.record()
        fence()
        wall = time.perf_counter() - t0
        return wall, sum(a.elapsed_time(b) for a, b in evs) / max(1, args.steps)

    # the first five launches of the process on set 0, timed one by one (clocks not settled, code object and TLBs cold)
    cold_first5 = timed_each(lambda i: step, 5)
    cold_ms = sum(cold_first5) / len(cold_first5)
    # first visit of every other buffer set, clocks still ramping
    first_visit_ms = [timed_each(lambda i, k=k: steps[k], 1)[0] for k in range(1, rotate)]
    t_settle = time.perf_counter()
    for i in range(max(0, args.settle - 5)):
        steps[i % rotate]()
    settle_extra = 0
    if args.settle > 0:
        torch.cuda.synchronize()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms and settle_extra < 100000:      # time-based part of the settle phase
            for _ in range(8):
                steps[settle_extra % rotate]()
                settle_extra += 1
            torch.cuda.synchronize()
    for i in range(args.warmup):                       # the W warm-up steps of the contract, immediately before the timed region
        steps[i % rotate]()
    fence()
    # region A: the same buffer set re-swept (rounds 1-3's figure)
    same_wall, same_kern_ms = timed_region(lambda i: step)
    # region B: round-robin over the disjoint buffer sets — no step touches the pages of the previous one
    if rotate > 1:
        rot_wall, rot_kern_ms = timed_region(lambda i: steps[i % rotate])
    else:
        rot_wall, rot_kern_ms = same_wall, same_kern_ms

    # cold probes: what makes the first launches slow — the clock ramp or the address translation of untouched pages?
    probes = None
    if not args.no_cold_probes and world == 1:
        probes = {}
        time.sleep(1.0)                                                 # (1) SAME buffers after one second of idle: clocks dropped, TLB reach unchanged
        probes["after_1s_idle_same_buffers_ms"] = timed_each(lambda i: step, 5)
        for _ in range(20):                                             # clocks back up
            step()
        torch.cuda.synchronize()
        _, fresh_step, fresh_desc, _ = setup(args, dev, dtype, rank + 7777)     # (2) FRESH buffers right after a busy period (the generator kernels
        probes["fresh_buffers_warm_clocks_ms"] = timed_each(lambda i: fresh_step, 5)   #     keep the clocks up); first touch of their output pages
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        probes["steady_same_buffers_ms"] = timed_each(lambda i: step, 5)       # (3) control: the same five-launch probe in the steady state
        del fresh_step, fresh_desc
        torch.cuda.empty_cache()

    telemetry = None
    if rank == 0 and not args.no_telemetry and dev.type == "cuda":
        telemetry = device_telemetry(step, same_kern_ms, dev.index or 0)

    use_rot = rotate > 1 and rot_wall > 1.05 * same_wall
    elapsed = rot_wall if use_rot else same_wall
    kern_ms = rot_kern_ms if use_rot else same_kern_ms
    # max over ranks of BOTH regions; the decision which one `value` uses is taken on the maxima, identically on every rank
    t = torch.tensor([same_wall, rot_wall], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(n)], dtype=torch.float64, device=red_dev)
    per_rank = torch.zeros(2 * world, dtype=torch.float64, device=red_dev)     # every rank's kernel ms (same-buffer, rotating): a slow rank is visible
    per_rank[2 * rank], per_rank[2 * rank + 1] = same_kern_ms, rot_kern_ms
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)     # the points all ranks processed per step (layout workloads round per rank)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    same_wall, rot_wall = float(t[0].item()), float(t[1].item())
    use_rot = rotate > 1 and rot_wall > 1.05 * same_wall
    elapsed = rot_wall if use_rot else same_wall
    kern_ms = rot_kern_ms if use_rot else same_kern_ms
    ranks_kernel_ms = [[float(per_rank[2 * r]), float(per_rank[2 * r + 1])] for r in range(world)]

    if rank == 0:
        total_points = int(tot.item())
        bpp = desc["bytes_per_point"]
        achieved = n * bpp / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(args.workload, args.dtype, n)
        line = {
            "metric": desc["metric"],
            "value": total_points * args.steps / elapsed,
            "unit": "grid-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": args.settle + settle_extra,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": desc["workload"], "points_per_gpu": n, "points_total": total_points, "columns_in": desc["columns_in"],
                       "columns_out": desc["columns_out"],
                       "parallelism": f"shard{world}" + ("+rccl-diag" if args.diagnostics else "") +
                                      ("" if args.backend == "nccl" or world == 1 else " (gloo test mode: ranks may share a device)")},
            "roofline": None,
        }
        line["cold_ms_first5"] = cold_ms      # mean kernel time of the first five launches (HIP events), before the settle launches
        line["rotate"] = rotate
        line["same_buffer_ms_per_step"] = same_wall / args.steps * 1e3
        line["rotating_ms_per_step"] = rot_wall / args.steps * 1e3 if rotate > 1 else None
        line["value_uses"] = "rotating" if use_rot else "same_buffer"        # rotating iff it is more than 5 % slower than the same-buffer region
        line["ranks_kernel_ms"] = {"same_buffer": [r[0] for r in ranks_kernel_ms], "rotating": [r[1] for r in ranks_kernel_ms] if rotate > 1 else None}
        line["timed_region_ms"] = elapsed * 1e3
        if elapsed < 20e-3:
            line["short_timed_region"] = f"the timed region is {elapsed * 1e3:.2f} ms (< 20 ms): one barrier skew of 50-100 µs moves `value` by several per mille to per cent"
        line["cold"] = {"first5_ms": cold_first5, "first_visit_other_sets_ms": first_visit_ms, "probes": probes}
        line["telemetry"] = telemetry
        hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
               "traffic_source": (traffic_source + " (rocprofv3 PMC pass of this command on an earlier run; not measured in this run)")
                                 if traffic_source else None,
               "kernel": desc["kernel"], "kernel_ms": kern_ms, "bytes_per_point": bpp}
        insts, insts_source = pmc_valu(args.workload, args.dtype, n)
        valu = None
        if insts is not None:
            rate = insts / (kern_ms * 1e-3) / 1e9
            valu = {"bound": "valu", "achieved": rate, "peak": VALU_PEAK_GINST, "unit": "G wave64-VALU-instructions/s", "frac": rate / VALU_PEAK_GINST,
                    "frac_vs_guide": rate / VALU_PEAK_GINST_GUIDE, "peak_guide": VALU_PEAK_GINST_GUIDE,
                    "traffic": None, "insts_per_point": insts * 64 / n, "insts_source": insts_source + " (SQ_INSTS_VALU of a rocprofv3 PMC pass of this "
                    "command; a property of the code and the inputs, not of the box)", "kernel": desc["kernel"], "kernel_ms": kern_ms,
                    "issue_slots_4cycle": rate / (1024 * 2.4 / 4),
                    "frac_at_measured_clock": (rate / (VALU_PEAK_GINST * (sum(telemetry["sclk_mhz"]) / len(telemetry["sclk_mhz"])) / 2400.0)) if telemetry else None,
                    "peak_formula": "256 CUs x 4 SIMDs x 2.4 GHz / 2.4 cycles: the fastest measured issue rate of a wave64 VALU instruction "
                                    "(profiles/rNN_probe_valu.txt); frac_vs_guide = the same count against the guide's 2-cycle issue of a wave64 instruction "
                                    "on the SIMD-32 (MI355X_MICROARCH.md: 1228.8 G/s); Float64 instructions take >= 4.2 cycles, so a Float64 kernel tops out near 0.55; "
                                    "issue_slots_4cycle = the same count against one instruction per 4 cycles (round 2's definition, > 1 for Float32)"}
        if desc.get("bound") == "valu" and valu is not None:
            # compute-bound line: the VALU-issue fraction is the roofline, the HBM fraction a secondary field
            line["roofline"] = dict(valu, hbm={k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_source", "bytes_per_point")})
        else:
            line["roofline"] = dict(hbm, valu=({k: valu[k] for k in ("achieved", "peak", "unit", "frac", "frac_vs_guide", "peak_guide", "issue_slots_4cycle", "frac_at_measured_clock", "insts_per_point", "insts_source")} if valu else None))
            if desc.get("bound") == "valu":
                line["roofline"]["note"] = "compute-bound workload, but no committed PMC instruction count matches this size: HBM fraction shown"
        if "note" in desc:
            line["roofline"]["note"] = desc["note"]
        if not args.no_cpu_baseline and world == 1:
            m = min(n, {"sb2006": 20_000_000, "arg2000_columns": 2_000_000, "sb2006_chen": 20_000_000, "sb2006_column": 74 * 270_000, "mp1m_column": 74 * 54_000, "mp1m_column_lin": 74 * 27_000, "sb2006_aos": 20_000_000, "sb2006_fields": 20_000_000, "p3": 100_000, "p3_split": 100_000, "p3_selfcol": 2_000, "mp2m_p3": 20_000}.get(args.workload, 4_000_000))
            cols_np = [np.ascontiguousarray(c[:m].cpu().numpy()) for c in state]
            line["cpu_baseline"] = cpu_baseline(args, cols_np, desc, cpu_run)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
