#!/usr/bin/env python3
"""bench.py — SB2006 two-moment warm-rain fused-tendency sweep (BASELINE.json metric).

One "step" = one pass of the fused kernel (7 state columns in, 4 tendencies + 2 rain fall-speed columns
out) over the rank's resident batch of synthetic grid points (default 1e8 Float32 points per GPU — the
configuration BASELINE.json quotes the metric on).  Inputs are generated on the device before the timed
region; nothing crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points P] [--dtype f32|f64]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract: see the task statement; DESIGN.md §5 explains every field).
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
BYTES_PER_POINT = {"f32": 52, "f64": 104}   # 7 in + 4 tendencies + 2 velocities


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=100_000_000, help="grid points per GPU")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the bounded baseline sample")
    ap.add_argument("--diagnostics", action="store_true",
                    help="also reduce Σ of the 4 tendencies per step (block reduce + RCCL all-reduce of 4 doubles)")
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


def pmc_traffic(dtype: str, n: int):
    """HBM bytes per launch measured with rocprofv3 PMC counters (FETCH_SIZE, WRITE_SIZE; separate passes, gfx950
    FETCH_SIZE ×2 correction of MI355X_MICROARCH.md §HBM) for this exact workload, if a committed profile matches;
    PMC counters cannot be read from inside the benchmark process, so this is null otherwise."""
    for p in sorted((REPO / "profiles").glob(f"r*_pmc_traffic_{dtype}.json"), reverse=True):
        try:
            d = json.loads(p.read_text())
        except (OSError, ValueError):
            continue
        if d.get("points") == n and d.get("dtype") == dtype:
            return d.get("hbm_bytes_per_launch")
    return None


def cpu_baseline(args, state_cpu_sample):
    """The oracle — a C restatement of the reference's scalar arithmetic (kind 'port'; the Julia reference cannot
    run here) — timed on the host cores over a bounded sample of the same synthetic workload."""
    sys.path.insert(0, str(REPO / "oracle"))
    import numpy as np
    import oracle_binding as ob
    from cmx import _abi
    from cmx import parameters as P
    fam = _abi.family(args.dtype)
    cores = usable_cores()
    wr, tps, vel = P.WarmRainParams2M(args.dtype).c, P.ThermodynamicsParameters(args.dtype), P.rain_vel_params(args.dtype)
    flags = _abi.CMX_SB2006_LIMITED | _abi.CMX_VEL_SB2006
    cols = [np.ascontiguousarray(c) for c in state_cpu_sample]
    run = lambda m: ob.sb2006_warm_rain_tendencies(fam, wr, tps, vel, flags, *[c[:m] for c in cols],  # noqa: E731
                                                   nthreads=cores, want_scale=False)
    probe = min(200_000, cols[0].size)
    run(probe)
    t0 = time.perf_counter()
    run(probe)
    rate = probe / (time.perf_counter() - t0)
    m = cols[0].size
    run(m)                                         # first full pass: page in, spin up the thread team
    passes, t0 = 0, time.perf_counter()
    while True:
        run(m)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds or passes >= 1000:
            break
    return {"value": passes * m / dt, "unit": "grid-points/s", "cores": cores, "kind": "port",
            "sample": f"{passes} passes over {m} of the same synthetic points ({passes * m} point evaluations), "
                      f"{args.dtype} arithmetic, oracle/libcmx_oracle.so (gcc -O2, OpenMP {cores} threads), {dt:.1f} s; "
                      f"probe rate {rate:.3g}/s"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import cmx
    from cmx import parameters as P
    from cmx import sharding, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1 (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    n = args.points                                          # weak scaling: fixed work per GPU
    # rank r owns shard r of the global [0, world·n) index space: disjoint seeds, no exchange
    state = synthetic.sb2006_state(n, dtype=dtype, device=dev, seed=1234 + rank)
    mp, tps = P.Microphysics2MParams(args.dtype), P.ThermodynamicsParameters(args.dtype)
    out = cmx.WarmRainTendencies2M(*[torch.empty_like(state.rho) for _ in range(6)])
    scheme = cmx.Microphysics2Moment()

    def step():
        cmx.bulk_microphysics_tendencies(scheme, mp, tps, *state, vel=cmx.SB2006VelType, out=out)
        if args.diagnostics:
            sharding.global_diagnostics(list(out[:4]))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # per-launch kernel duration from HIP events recorded on the stream the kernel is launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        total_points = n * world
        bpp = BYTES_PER_POINT[args.dtype]
        achieved = n * bpp / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "grid-points/sec SB2006 2M tendency sweep",
            "value": total_points * args.steps / elapsed,
            "unit": "grid-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "Microphysics2M SB2006 fused warm-rain tendencies (cond/evap, autoconversion, accretion, "
                                   "self-collection, breakup, evaporation, number adjustment) + SB2006 rain terminal "
                                   "velocities, limited rain PSD", "points_per_gpu": n, "columns_in": 7, "columns_out": 6,
                       "parallelism": f"shard{world}" + ("+rccl-diag" if args.diagnostics else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.dtype, n),
                         "kernel": "sb2006_tendencies_kernel", "kernel_ms": kern_ms, "bytes_per_point": bpp},
        }
        if not args.no_cpu_baseline and world == 1:
            m = min(n, 20_000_000)
            line["cpu_baseline"] = cpu_baseline(args, [c[:m].cpu().numpy() for c in state])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
