"""Per-launch duration series of the SB2006 sweep (HIP events): shows the power/clock transient after the first launches."""
import sys, torch
sys.path.insert(0, "cloudmicrophysics.jl_amd")
import cmx
from cmx import parameters as P, synthetic
n = 100_000_000; dev = torch.device("cuda:0")
st = synthetic.sb2006_state(n, dtype=torch.float32, device=dev)
mp, tps = P.Microphysics2MParams("f32"), P.ThermodynamicsParameters("f32")
out = cmx.WarmRainTendencies2M(*[torch.empty_like(st.rho) for _ in range(6)])
K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
torch.cuda.synchronize()
for a, b in ev:
    a.record(); cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *st, vel=cmx.SB2006VelType, out=out); b.record()
torch.cuda.synchronize()
d = [a.elapsed_time(b) * 1e3 for a, b in ev]
for k in range(0, K, 20):
    print("%4d: " % k + " ".join("%.0f" % x for x in d[k:k + 20]))
import statistics
print("mean first 50 after 5: %.1f  mean 50..: %.1f  mean 200..: %.1f  median all: %.1f" % (statistics.mean(d[5:55]), statistics.mean(d[50:]), statistics.mean(d[200:]), statistics.median(d)))
