"""Times the P3 process kernels separately (collisions, self-collection, melt) on the bench's mixed-phase states."""
import sys, time, numpy as np, torch
sys.path.insert(0, "cloudmicrophysics.jl_amd")
import cmx
from cmx import parameters as P
ft = sys.argv[1] if len(sys.argv) > 1 else "f64"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dt = torch.float64 if ft == "f64" else torch.float32; dev = torch.device("cuda:0")
rng = np.random.default_rng(1234)
rho = rng.uniform(0.4, 1.3, n); T = rng.uniform(215.0, 295.0, n)
q_lcl = np.where(rng.random(n) < 0.7, 10 ** rng.uniform(-6, -3, n), 0.0); n_lcl = 10 ** rng.uniform(6, 9, n)
q_rai = np.where(rng.random(n) < 0.6, 10 ** rng.uniform(-7, -3, n), 0.0); n_rai = 10 ** rng.uniform(1, 6, n)
q_ice = np.where(rng.random(n) < 0.8, 10 ** rng.uniform(-6, -3, n), 0.0); n_ice = 10 ** rng.uniform(2, 6, n)
q_rim = np.where(rng.random(n) < 0.3, 0.0, rng.uniform(0.05, 0.9, n)) * q_ice; b_rim = q_rim / rng.uniform(200, 800, n)
c = [torch.from_numpy(x).to(dt).to(dev) for x in (rho, T, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim)]
rho, T, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim = c
ip = P.P3IceParams(ft); aps, tps = P.AirProperties(ft), P.ThermodynamicsParameters(ft); p3 = P.ParametersP3(ft); vel = P.Chen2022VelTypeIce(ft)
st = (q_ice * rho, n_ice * rho, q_rim * rho, b_rim * rho)
ll = cmx.p3_shape(p3, *st, want=("log_lambda",)).log_lambda; ll = torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll))
def timeit(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print(ft, n, "collisions %.1f ms" % timeit(lambda: cmx.p3_liquid_ice_collisions(ip, aps, tps, rho, T, *st, ll, q_lcl * rho, n_lcl * rho, q_rai * rho, n_rai * rho)),
      "selfcol(GL16, lane/pt) %.1f ms" % timeit(lambda: cmx.p3_ice_self_collection(p3, vel, rho, *st, ll, quad=ip.c.quad)),
      "melt %.1f ms" % timeit(lambda: cmx.p3_ice_melt(p3, vel, aps, tps, P.VentilationFactorP3(ft), T, rho, *st, ll, quad=ip.c.quad)),
      "shape %.1f ms" % timeit(lambda: cmx.p3_shape(p3, *st, want=("log_lambda",))))
