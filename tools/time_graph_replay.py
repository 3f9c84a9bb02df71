"""Per-step latency of a five-entry microphysics step (2M warm rain + velocities, 1M, 0M, ice nucleation, ARG2000) at host-model
column-block sizes: eager calls through the Python host mirror vs one replay of the captured HIP graph.
    python tools/time_graph_replay.py [f32|f64]"""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "cloudmicrophysics.jl_amd"), str(REPO / "tests")]
from test_graphs_gpu import _step_factory  # noqa: E402

sfx = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda", 0)
print(f"{'points':>10} {'branches':>9} {'eager us/step':>14} {'graph us/step':>14} {'GPU-busy us/step':>17}")
for n, forked in ((4096, False), (4096, True), (65_536, False), (65_536, True), (1_048_576, False), (1_048_576, True), (16_777_216, False)):
    step, _, _ = _step_factory(dev, n, sfx, forked)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / K * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / K * 1e6
    print(f"{n:>10} {'forked' if forked else 'chain':>9} {eager:>14.1f} {graph:>14.1f} {e0.elapsed_time(e1) / K * 1e3:>17.1f}")
