"""HIP graphs: a host model's microphysics step (several entries back to back on one stream) captured once and replayed.

Every entry of libcmx.so is a pure sequence of kernel launches on the caller's stream (no allocation, no synchronisation, no host
read-back), so a stream capture records it; parameters travel by value in the kernel arguments and are frozen at capture time, state
and output columns are the captured device addresses.  Replays must be bit-identical to eager calls on the same memory."""
import pytest
import torch

from cmx import parameters as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _step_factory(dev, n, sfx, forked=False):
    import cmx
    from cmx import synthetic
    dt = torch.float32 if sfx == "f32" else torch.float64
    st2 = list(synthetic.sb2006_state(n, dtype=dt, device=dev, seed=5))
    st1 = list(synthetic.mp1m_state(n, dtype=dt, device=dev, seed=6))
    sti = list(synthetic.ice_nucleation_state(n, dtype=dt, device=dev, seed=7))
    sta = list(synthetic.arg_state(n, dtype=dt, device=dev, seed=8))
    mp2, mp1, mp0 = P.Microphysics2MParams(sfx), P.Microphysics1MParams(sfx), P.Microphysics0MParams(sfx)
    tps, dust, koop = P.ThermodynamicsParameters(sfx), P.Kaolinite(sfx), P.Koop2000(sfx)
    ap, aip, ad = P.AerosolActivationParameters(sfx), P.AirProperties(sfx), synthetic.arg_config3_distribution()
    mk = lambda: torch.empty(n, dtype=dt, device=dev)  # noqa: E731
    out2 = cmx.WarmRainTendencies2M(*[mk() for _ in range(6)])
    out1 = cmx.Tendencies1M(*[mk() for _ in range(4)])
    out0, vt = mk(), (mk(), mk())
    outa = cmx.ActivationResult(tuple(mk() for _ in range(5)), None, None)
    outi = cmx.ice_nucleation_rates(tps, dust, koop, *sti, count_domain_errors=True)
    sed = [mk() for _ in range(4)]

    calls = [
        lambda: cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp2, tps, *st2, vel=cmx.SB2006VelType, out=out2),
        lambda: cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp1, tps, *st1, out=out1),
        lambda: cmx.bulk_microphysics_tendencies_0m(cmx.Microphysics0Moment(), mp0, tps, st1[1], st1[3], st1[4], out=out0),
        lambda: cmx.ice_nucleation_rates(tps, dust, koop, *sti, out=outi),
        lambda: cmx.aerosol_activation(ap, ad, aip, tps, *sta, out=outa),
    ]
    side = [torch.cuda.Stream(dev) for _ in calls[1:]] if forked else []

    def step():
        # forked: the five entries touch disjoint columns, so each goes to its own stream (fork / join on the current stream) —
        # captured, they become parallel branches of the graph instead of a chain
        main = torch.cuda.current_stream(dev)
        for s in side:
            s.wait_stream(main)
        calls[0]()
        for s, f in zip(side, calls[1:]):
            with torch.cuda.stream(s):
                f()
        if not side:
            for f in calls[1:]:
                f()
        for s in side:
            main.wait_stream(s)

    outs = list(out2) + list(out1) + [out0, outi.rate_het, outi.rate_hom] + list(outa.N_act)
    return step, [st2, st1, sti, sta], outs


@pytest.mark.parametrize("forked", [False, True])
@pytest.mark.parametrize("sfx", ["f32", "f64"])
def test_captured_step_replays_bit_identically(dev, sfx, forked):
    n = 65_536 + 3
    step, states, outs = _step_factory(dev, n, sfx, forked)
    step()
    torch.cuda.synchronize()
    eager = [o.clone() for o in outs]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for o in outs:
        o.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(outs, eager)):
        assert torch.equal(a, b) or torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), k
    # new state written into the captured columns: the replay follows it, and equals an eager step on the same memory
    for st in states:
        for c in st:
            c.copy_(c.flip(0))
    g.replay()
    torch.cuda.synchronize()
    replayed = [o.clone() for o in outs]
    assert any(not torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) for a, b in zip(replayed, eager))
    step()
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(outs, replayed)):
        assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), k


def test_captured_2m_p3_entry(dev):
    """The two-launch 2M + P3 entry (point-wise kernel, then the group-cooperative collision kernel with > 48 KB of LDS) inside a capture."""
    import numpy as np
    import cmx
    from cmx import synthetic
    n = 4096
    mp, tps = P.Microphysics2MParams("f32", with_ice=True), P.ThermodynamicsParameters("f32")
    st = list(synthetic.sb2006_state(n, dtype=torch.float32, device=dev, seed=11))
    st[1].clamp_(max=272.0)
    p3 = synthetic.p3_state(n, dtype=torch.float32, device=dev, seed=12)
    rho = st[0]
    q_ice, n_ice = p3.rho_q_ice / rho, p3.rho_n_ice / rho
    q_rim, b_rim = p3.rho_q_rim / rho, p3.rho_b_rim / rho
    shape = cmx.p3_shape(P.ParametersP3("f32"), p3.rho_q_ice, p3.rho_n_ice, p3.rho_q_rim, p3.rho_b_rim, want=("log_lambda",))
    call = lambda: cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *st, q_ice, n_ice, q_rim, b_rim, shape.log_lambda)  # noqa: E731
    eager = call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = call()
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(cap, eager):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)
