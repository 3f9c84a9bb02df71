"""Single-node launcher: `script --gpus N` called as ONE process starts its own N ranks.

The sharded path runs one process per GPU.  A caller that is already inside such a job
(`WORLD_SIZE` set, e.g. by `python -m torch.distributed.run`) just runs its rank; a caller that is a plain
`python bench.py --gpus N` becomes a *launcher*: it starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> script <argv>` as a CHILD process, waits,
relays the JSON line rank 0 printed and exits with the child's code.

Rules this module keeps (the GPU pool takes a machine down when a process that has initialised the GPU is
replaced by another program):
  * it imports neither torch nor the HIP library, and must be called before the caller does;
  * the ranks are child processes — nothing is exec'ed over the current process;
  * a failed rank makes the launcher return non-zero; a job that printed no JSON line returns non-zero too.

The reference has no counterpart (no parallel layer at all).
"""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
from typing import Dict, List, Optional, Sequence, Tuple


def needs_launch(n_ranks: int, environ: Optional[Dict[str, str]] = None) -> bool:
    """True when this process was asked for n_ranks > 1 but is not itself a rank of a running job."""
    env = os.environ if environ is None else environ
    return n_ranks > 1 and "WORLD_SIZE" not in env


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_command(script: str, argv: Sequence[str], n_ranks: int, port: int, python: Optional[str] = None) -> List[str]:
    """The exact command line of the rank-spawning child (the same one the driver uses for N > 1)."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]


def child_env(environ: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if environ is None else environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")              # ranks share the host cores; silences torchrun's notice
    return env


def split_output(stdout: str) -> Tuple[Optional[str], List[str]]:
    """(last line of `stdout` that parses as a JSON object, every other non-empty line)."""
    result, rest = None, []
    for ln in stdout.splitlines():
        s = ln.strip()
        if not s:
            continue
        is_obj = False
        if s.startswith("{") and s.endswith("}"):
            try:
                is_obj = isinstance(json.loads(s), dict)
            except ValueError:
                is_obj = False
        if is_obj:
            if result is not None:
                rest.append(result)
            result = s
        else:
            rest.append(ln)
    return result, rest


def launch(script: str, argv: Sequence[str], n_ranks: int, *, timeout: Optional[float] = None,
           python: Optional[str] = None, out=None, err=None) -> int:
    """Run the N-rank job as a child, relay rank 0's JSON line to `out` (stdout), everything else to `err`
    (stderr); return the exit code to leave with (0 only if every rank succeeded AND a JSON line arrived)."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    cmd = child_command(script, argv, n_ranks, free_port(), python)
    print("[launcher] " + " ".join(cmd), file=err, flush=True)
    try:
        r = subprocess.run(cmd, env=child_env(), stdout=subprocess.PIPE, stderr=None if err is sys.stderr else subprocess.PIPE,
                           text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        print(f"[launcher] timed out after {timeout} s", file=err, flush=True)
        if e.stdout:
            print(e.stdout if isinstance(e.stdout, str) else e.stdout.decode(errors="replace"), file=err)
        return 124
    if r.stderr:
        print(r.stderr, file=err, end="")
    line, rest = split_output(r.stdout or "")
    for ln in rest:
        print(ln, file=err)
    if r.returncode != 0:
        print(f"[launcher] child exited with code {r.returncode}", file=err, flush=True)
        return r.returncode if 0 < r.returncode < 256 else 1
    if line is None:
        print("[launcher] the job printed no JSON line", file=err, flush=True)
        return 1
    print(line, file=out, flush=True)
    return 0
