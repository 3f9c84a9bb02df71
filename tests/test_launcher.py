"""CPU tests of the single-node launcher behind `bench.py --gpus N` (cmx/launcher.py): the child command line, the
JSON/exit-code relay, and a real world-size-2 job of a trivial child script started through torch.distributed.run
(gloo rendezvous on 127.0.0.1).  Nothing here computes on a GPU and no bench code path is stubbed."""
import importlib.util
import io
import json
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
LAUNCHER = REPO / "cloudmicrophysics.jl_amd" / "cmx" / "launcher.py"


def _load():
    spec = importlib.util.spec_from_file_location("cmx_launcher_under_test", LAUNCHER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


launcher = _load()


def test_launcher_module_stays_clear_of_torch_and_hip():
    src = LAUNCHER.read_text()
    assert "import torch" not in src and "ctypes" not in src and "_lib" not in src
    r = subprocess.run([sys.executable, "-c",
                        "import importlib.util, sys\n"
                        f"spec = importlib.util.spec_from_file_location('l', r'{LAUNCHER}')\n"
                        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
                        "assert 'torch' not in sys.modules, 'launcher pulled torch in'\n"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_needs_launch():
    assert not launcher.needs_launch(1, {})
    assert launcher.needs_launch(2, {})
    assert not launcher.needs_launch(8, {"WORLD_SIZE": "8"})          # already a rank of a torchrun job


def test_child_command_is_the_drivers_command():
    cmd = launcher.child_command("bench.py", ["--gpus", "4", "--steps", "20", "--warmup", "5"], 4, 29511, python="python")
    assert cmd == ["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                   "--master-port", "29511", "bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"]
    with pytest.raises(ValueError):
        launcher.child_command("bench.py", [], 0, 1)


def test_child_env_keeps_dmabuf_ipc():
    env = launcher.child_env({"PATH": "/bin"})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
    assert launcher.child_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # caller's choice wins


def test_split_output():
    line, rest = launcher.split_output('warning: x\n{"a": 1}\nnoise {not json}\n{"metric": "m", "value": 2}\n\n')
    assert json.loads(line) == {"metric": "m", "value": 2}
    assert sorted(rest) == sorted(["warning: x", '{"a": 1}', "noise {not json}"])     # an earlier JSON line is demoted to chatter
    assert launcher.split_output("nothing here\n") == (None, ["nothing here"])


CHILD = textwrap.dedent("""
    import json, os, sys
    import torch.distributed as dist
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    dist.init_process_group("gloo")
    import torch
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    if mode == "fail" and rank == world - 1:
        sys.exit(3)
    if mode == "silent":
        sys.exit(0)
    print(f"rank {rank} chatter")
    if rank == 0:
        print(json.dumps({"metric": "trivial", "value": float(t.item()), "n_gpus": world, "argv": sys.argv[1:]}), flush=True)
""")


@pytest.fixture()
def child_script(tmp_path):
    p = tmp_path / "trivial_child.py"
    p.write_text(CHILD)
    return str(p)


def test_world2_job_relays_rank0_json(child_script):
    out, err = io.StringIO(), io.StringIO()
    rc = launcher.launch(child_script, ["ok", "--gpus", "2"], 2, timeout=300, out=out, err=err)
    assert rc == 0, err.getvalue()
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1                                             # exactly ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d == {"metric": "trivial", "value": 3.0, "n_gpus": 2, "argv": ["ok", "--gpus", "2"]}
    assert "chatter" in err.getvalue() and "torch.distributed.run" in err.getvalue()


def test_failed_rank_makes_the_launcher_fail(child_script):
    out, err = io.StringIO(), io.StringIO()
    rc = launcher.launch(child_script, ["fail"], 2, timeout=300, out=out, err=err)
    assert rc != 0 and out.getvalue().strip() == ""


def test_job_without_json_line_fails(child_script):
    out, err = io.StringIO(), io.StringIO()
    rc = launcher.launch(child_script, ["silent"], 2, timeout=300, out=out, err=err)
    assert rc == 1 and "no JSON line" in err.getvalue()


def test_bench_becomes_a_launcher_before_importing_torch():
    """`python bench.py --gpus 2` outside torchrun must start the child command, not exit with a usage error; here (no GPU)
    the ranks fail, and that failure — not an argparse/usage exit — is what comes back."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--settle", "0",
                        "--points", "1024", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert "[launcher]" in r.stderr and "--nproc-per-node=2" in r.stderr and "--master-addr 127.0.0.1" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and r.stdout.strip() == ""


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_scaffolding_at_world_size_8(scaling):
    """The N = 8 job of bench.py itself — launcher → torch.distributed.run → 8 ranks → sharding, barriers, max-over-ranks timing,
    reductions, ONE rank-0 JSON line — with the kernel step replaced by nothing (`--dry-run-scaffolding`, gloo, no GPU): what the
    driver's 8-GPU run exercises around the kernel.  Shard sizes must tile the requested range (strong) or add up to 8 × it (weak)."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    n = 100_000_000
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--points", str(n),
                        "--scaling", scaling, "--dry-run-scaffolding"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["scaling"] == scaling and d["value"] is None
    if scaling == "strong":
        assert d["points_total"] == n
        b = d["shard_bounds"]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[k][1] == b[k + 1][0] for k in range(7))
        assert all(lo % 256 == 0 for lo, _ in b) and max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 512
    else:
        assert d["points_total"] == 8 * n and d["points_per_rank0"] == n


def test_nccl_job_with_too_few_gpus_exits_nonzero_with_a_message():
    """`--gpus 2` over RCCL on a node with fewer than 2 GPUs: every rank leaves with a clear message before any collective (here: 0 GPUs)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a node with fewer than 2 GPUs")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--points", "1024", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs 2 visible GPUs" in r.stderr
