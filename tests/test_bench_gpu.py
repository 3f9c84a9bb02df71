"""bench.py contract on the GPU box: the N = 1 line (weak and strong), and — where the box has ≥ 2 GPUs — the self-launched N = 2 job."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

REPO = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def _run(*extra, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "3", "--warmup", "1", "--settle", "2", "--points", "1000003",
                        *extra], capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_single_gpu_line(scaling):
    d = _run("--scaling", scaling, "--cpu-seconds", "0.5")
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["scaling"] == scaling and d["steps"] == 3
    assert d["config"]["points_per_gpu"] == 1000003 and d["config"]["points_total"] == 1000003
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and "traffic_source" in rf
    assert abs(rf["achieved"] - 1000003 * 52 / (rf["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert d["value"] == pytest.approx(1000003 * 3 / (d["ms_per_step"] * 3e-3), rel=1e-9)
    # round 4: the rotating region, the cold probes, per-rank kernel times, telemetry, the baseline's provenance
    assert d["rotate"] == 4 and d["value_uses"] in ("same_buffer", "rotating")
    assert d["same_buffer_ms_per_step"] > 0 and d["rotating_ms_per_step"] > 0
    assert d["ms_per_step"] == pytest.approx(d["rotating_ms_per_step"] if d["value_uses"] == "rotating" else d["same_buffer_ms_per_step"], rel=1e-9)
    assert (d["value_uses"] == "rotating") == (d["rotating_ms_per_step"] > 1.05 * d["same_buffer_ms_per_step"])
    assert len(d["ranks_kernel_ms"]["same_buffer"]) == 1 and len(d["ranks_kernel_ms"]["rotating"]) == 1
    assert d["timed_region_ms"] < 20 and "short_timed_region" in d                         # 3 steps of a 1e6-point sweep: flagged
    cold = d["cold"]
    assert len(cold["first5_ms"]) == 5 and len(cold["first_visit_other_sets_ms"]) == 3
    assert set(cold["probes"]) == {"after_1s_idle_same_buffers_ms", "fresh_buffers_warm_clocks_ms", "steady_same_buffers_ms"}
    tel = d["telemetry"]
    assert tel is None or (len(tel["sclk_mhz"]) >= 1 and all(100 < v < 3000 for v in tel["sclk_mhz"]))
    assert cb["cpu_model"] and (cb["cores"] == 1 or cb["single_thread"]["value"] > 0) and (cb["julia"] == "absent" or str(cb["julia"]).startswith("/"))
    assert "frac_vs_guide" in (rf.get("valu") or {"frac_vs_guide": None})


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_self_launched_two_rank_job(scaling):
    d = _run("--gpus", "2", "--scaling", scaling, "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["scaling"] == scaling
    assert d["config"]["points_total"] == (2 * 1000003 if scaling == "weak" else 1000003)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_sharing_this_gpu_gloo_test_mode(scaling):
    """The N = 2 code path of bench.py (launcher → torch.distributed.run → two ranks, shard arithmetic, max-over-ranks timing, the
    summed point count, rank-0 JSON) on whatever box this is: with `--backend gloo` the two ranks may share GPU 0.  Not a scaling
    measurement — the JSON says so in config.parallelism."""
    d = _run("--gpus", "2", "--scaling", scaling, "--backend", "gloo", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and "gloo test mode" in d["config"]["parallelism"]
    assert d["config"]["points_total"] == (2 * 1000003 if scaling == "weak" else 1000003)
    per = d["config"]["points_per_gpu"]
    assert per == (1000003 if scaling == "weak" else 500224)        # shard_bounds(1000003, 0, 2): 256-point-aligned halves
    assert len(d["ranks_kernel_ms"]["same_buffer"]) == 2 and all(v > 0 for v in d["ranks_kernel_ms"]["same_buffer"])      # every rank's kernel time reaches rank 0
    assert d["value"] == pytest.approx(d["config"]["points_total"] * 3 / (d["ms_per_step"] * 3e-3), rel=1e-9)


def test_rccl_process_group_path_with_one_rank():
    """The process-group calls of the N > 1 job — `init_process_group("nccl", device_id=…)` (RCCL), the barrier, the MAX / SUM
    all-reduces on device tensors, `destroy_process_group` — on a 1-GPU box: bench.py under torchrun with ONE rank and
    CMX_BENCH_FORCE_DIST=1 (two ranks cannot share a GPU under RCCL; the two-rank arithmetic is the gloo test above)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, CMX_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(REPO / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--settle", "2",
                        "--points", "1000003", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["points_total"] == 1000003 and d["value"] > 0
