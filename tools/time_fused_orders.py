"""2M + P3 fused entry at several quadrature orders (16 = P3IceParams default, 40 = ClimaAtmos production order, src/Quadrature.jl:268)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "cloudmicrophysics.jl_amd")
import cmx
from cmx import parameters as P
n = 1_000_000; dev = torch.device("cuda:0")
rng = np.random.default_rng(1234)
rho = rng.uniform(0.4, 1.3, n); T = rng.uniform(215.0, 295.0, n)
q_lcl = np.where(rng.random(n) < 0.7, 10 ** rng.uniform(-6, -3, n), 0.0); n_lcl = 10 ** rng.uniform(6, 9, n)
q_rai = np.where(rng.random(n) < 0.6, 10 ** rng.uniform(-7, -3, n), 0.0); n_rai = 10 ** rng.uniform(1, 6, n)
q_ice = np.where(rng.random(n) < 0.8, 10 ** rng.uniform(-6, -3, n), 0.0); n_ice = 10 ** rng.uniform(2, 6, n)
q_rim = np.where(rng.random(n) < 0.3, 0.0, rng.uniform(0.05, 0.9, n)) * q_ice; b_rim = q_rim / rng.uniform(200, 800, n)
q_tot = q_lcl + q_rai + q_ice + 10 ** rng.uniform(-5, -2, n)
for ft, dt in (("f64", torch.float64), ("f32", torch.float32)):
    cols = [torch.from_numpy(c).to(dt).to(dev) for c in (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim)]
    tps = P.ThermodynamicsParameters(ft)
    ll = cmx.p3_shape(P.ParametersP3(ft), cols[7] * cols[0], cols[8] * cols[0], cols[9] * cols[0], cols[10] * cols[0], want=("log_lambda",)).log_lambda
    ll = torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll))
    for order in (16, 32, 40, 64):
        mp = P.Microphysics2MParams(ft, with_ice=True, quadrature_order=order)
        f = lambda: cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *cols, ll)
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"{ft} GaussLegendre({order}): {ms:7.1f} ms per 1e6 states = {n / ms * 1e3:.3g} states/s")
