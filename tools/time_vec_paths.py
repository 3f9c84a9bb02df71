"""Times the vector (16-byte) and the one-point-per-lane paths of the streaming kernels: columns sliced by one element are
misaligned and take the scalar path."""
import sys, time, torch
sys.path.insert(0, "cloudmicrophysics.jl_amd")
import cmx
from cmx import parameters as P, synthetic
dev = torch.device("cuda:0"); n = 100_000_000; ft = "f32"; dt = torch.float32
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
tps = P.ThermodynamicsParameters(ft)
# ARG
st = synthetic.arg_state(n + 4, dtype=dt, device=dev)
ap, aip = P.AerosolActivationParameters(ft), P.AirProperties(ft)
ad = synthetic.arg_config3_distribution()
for off in (0, 1):
    cols = [c[off:n + off] for c in st]
    outs = cmx.ActivationResult(tuple(torch.empty(n + 4, dtype=dt, device=dev)[off:n + off] for _ in range(5)), None, None)
    print("arg2000 offset", off, "%.3f ms" % timeit(lambda: cmx.aerosol_activation(ap, ad, aip, tps, *cols, out=outs)))
del st
# 1M
st = synthetic.mp1m_state(n + 4, dtype=dt, device=dev)
mp = P.Microphysics1MParams(ft)
for off in (0, 1):
    cols = [c[off:n + off] for c in st]
    o1 = cmx.Tendencies1M(*[torch.empty(n + 4, dtype=dt, device=dev)[off:n + off] for _ in range(4)])
    print("mp1m offset", off, "%.3f ms" % timeit(lambda: cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *cols, out=o1)))
del st
st = synthetic.sb2006_state(n + 4, dtype=dt, device=dev)
mp2 = P.Microphysics2MParams(ft)
for off in (0, 1):
    cols = [c[off:n + off] for c in st]
    o2 = cmx.WarmRainTendencies2M(*[torch.empty(n + 4, dtype=dt, device=dev)[off:n + off] for _ in range(6)])
    print("sb2006 offset", off, "%.3f ms" % timeit(lambda: cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp2, tps, *cols, vel=cmx.SB2006VelType, out=o2)))
