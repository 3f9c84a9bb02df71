"""Times the parts of BASELINE config 5 (shape solve, D_m, fall speeds) separately on the bench's states: where the 21 ms per 1e7 Float64 states go."""
import sys, time, torch
sys.path.insert(0, "cloudmicrophysics.jl_amd")
import cmx
from cmx import parameters as P, synthetic
ft = sys.argv[1] if len(sys.argv) > 1 else "f64"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dt = torch.float64 if ft == "f64" else torch.float32; dev = torch.device("cuda:0")
st = synthetic.p3_state(n, dtype=dt, device=dev, seed=1234); rho_a = synthetic.p3_air_density(n, dtype=dt, device=dev)
p, vel = P.ParametersP3(ft), P.Chen2022VelTypeIce(ft)
def timeit(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
ll = cmx.p3_shape(p, *st, want=("log_lambda",)).log_lambda
print(ft, n,
      "shape(logλ) %.2f ms" % timeit(lambda: cmx.p3_shape(p, *st, want=("log_lambda",))),
      "| shape(logλ, 1 Brent iteration) %.2f ms" % timeit(lambda: cmx.p3_shape(p, *st, want=("log_lambda",), brent_iters=1)),
      "| shape(logλ + D_m) %.2f ms" % timeit(lambda: cmx.p3_shape(p, *st, want=("log_lambda", "D_m"))),
      "| velocities %.2f ms" % timeit(lambda: cmx.p3_terminal_velocities(p, vel, rho_a, *st, ll)),
      "| fused %.2f ms" % timeit(lambda: cmx.p3_shape_and_terminal_velocities(p, vel, rho_a, *st)))
