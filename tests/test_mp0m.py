"""0-moment scheme: the oracle against the reference's own 0M tests (CPU), and the device entry against the oracle bit for bit (GPU).

Reference tests mirrored: test/microphysics0M_tests.jl:10-50 (formula on the struct's fields for liquid fractions 0, 0.5, 1 and
q_c = 3e-3…5e-3), test/bulk_tendencies_tests.jl:23-116 (removal above / zero below the threshold, both threshold forms),
test/gpu_tests.jl:105-141, 364-383, 636-665 (the KA kernels compare with the scalar formula with `==`)."""
import numpy as np
import pytest

from cmx import _abi
from cmx import parameters as P

FRAC = [0.0, 0.5, 1.0]
QC = [3e-3, 4e-3, 5e-3]


def _formula(p, ql, qi, qvs=None):
    f = type(ql[0])
    thr = f(p.qc_0) if qvs is None else f(p.S_0) * qvs
    ex = np.maximum(f(0), np.maximum(ql, f(0)) + np.maximum(qi, f(0)) - thr)
    return -ex / f(p.tau_precip), np.where(np.maximum(ql, f(0)) + np.maximum(qi, f(0)) > thr, f(-1) / f(p.tau_precip), f(0))


@pytest.mark.parametrize("sfx", ["f64", "f32"])
def test_oracle_matches_the_reference_formula(oracle, sfx):
    fam = _abi.family(sfx)
    f = np.float32 if sfx == "f32" else np.float64
    p = P.Parameters0M(sfx)
    assert p.tau_precip > 0 and 1e-6 < p.qc_0 < 2e-3 and p.S_0 > 0        # the thresholds the bulk tests bracket (:29-55)
    ql = np.array([fr * qc for fr in FRAC for qc in QC], dtype=f)
    qi = np.array([(1 - fr) * qc for fr in FRAC for qc in QC], dtype=f)
    qvs = np.full_like(ql, 1e-3)
    for sat in (None, qvs):
        out, der = oracle.mp0m_tendencies(fam, p, ql, qi, sat)
        ref, dref = _formula(p, ql, qi, sat)
        assert np.array_equal(out, ref) and np.array_equal(der, dref)
        assert np.all(out < 0) and np.all(der == f(-1) / f(p.tau_precip))
    # below threshold: exactly zero, for both forms (bulk_tendencies_tests.jl:43-55, 85-99); negative inputs are clamped (BMT:662-663)
    lo = np.array([1e-6, 0.0, -1e-3], dtype=f)
    out, der = oracle.mp0m_tendencies(fam, p, lo, np.zeros_like(lo))
    assert np.all(out == 0) and np.all(der == 0)
    out, _ = oracle.mp0m_tendencies(fam, p, np.array([1e-8], dtype=f), np.zeros(1, dtype=f), np.array([1.2e-2], dtype=f))
    assert out[0] == 0
    out, _ = oracle.mp0m_tendencies(fam, p, np.array([2e-3, -1.0], dtype=f), np.array([-5.0, 2e-3], dtype=f))
    assert out[0] == out[1] == -(f(2e-3) - f(p.qc_0)) / f(p.tau_precip)


def test_host_mirror_validates_before_touching_the_gpu():
    import torch
    import cmx
    mp = P.Microphysics0MParams("f32")
    x = torch.zeros(4)
    with pytest.raises(TypeError):
        cmx.bulk_microphysics_tendencies_0m(cmx.Microphysics2Moment(), mp, None, x, x, x)
    with pytest.raises(ValueError):
        cmx.bulk_microphysics_tendencies_0m(cmx.Microphysics0Moment(), mp, None, x, x, x)      # CPU tensors: no CPU path


@pytest.mark.gpu
@pytest.mark.parametrize("sfx", ["f64", "f32"])
@pytest.mark.parametrize("n", [0, 1, 9, 1000, 100_003])
def test_device_is_bit_identical_to_the_oracle(oracle, sfx, n):
    import torch
    import cmx
    fam = _abi.family(sfx)
    dt = torch.float32 if sfx == "f32" else torch.float64
    mp = P.Microphysics0MParams(sfx)
    g = torch.Generator().manual_seed(7 + n)
    ql = (torch.rand(n, generator=g, dtype=torch.float64) * 4e-3 - 5e-4).to(dt)       # some negative, some below threshold
    qi = (torch.rand(n, generator=g, dtype=torch.float64) * 2e-3 - 2e-4).to(dt)
    ql[::7] = 0
    qi[::5] = 0
    qvs = (torch.rand(n, generator=g, dtype=torch.float64) * 0.2).to(dt)
    d = lambda t: t.cuda()
    for sat in (None, qvs):
        ref, dref = (oracle.mp0m_tendencies(fam, mp.precip, ql.numpy(), qi.numpy(), None if sat is None else sat.numpy())
                     if n else (np.empty(0), np.empty(0)))
        satd = None if sat is None else d(sat)
        out = cmx.bulk_microphysics_tendencies_0m(cmx.Microphysics0Moment(), mp, None, d(ql), d(ql), d(qi), satd)
        rp = cmx.remove_precipitation(mp.precip, d(ql), d(qi), satd)
        der = cmx.d_remove_precipitation_d_q_tot(mp.precip, d(ql), d(qi), satd)
        assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(rp.cpu().numpy(), ref)
        assert np.array_equal(der.cpu().numpy(), dref)
        # misaligned slices take the one-point-per-lane path with the same bits
        if n > 8:
            o2 = cmx.remove_precipitation(mp.precip, d(ql)[1:], d(qi)[1:], None if satd is None else satd[1:])
            assert np.array_equal(o2.cpu().numpy(), ref[1:])


@pytest.mark.gpu
def test_reference_gpu_kernel_cases():
    """test/gpu_tests.jl:105-141 + 636-665: S_pr and both derivative forms equal the scalar formula exactly."""
    import torch
    import cmx
    for sfx, dt, f in (("f32", torch.float32, np.float32), ("f64", torch.float64, np.float64)):
        p = P.Parameters0M(sfx)
        fr, qc = torch.tensor(FRAC, dtype=dt), torch.tensor(QC, dtype=dt)
        ql, qi = (fr * qc).cuda(), ((1 - fr) * qc).cuda()
        qvs = torch.full((3,), 1e-3, dtype=dt).cuda()
        for sat in (None, qvs):
            ref, dref = _formula(p, ql.cpu().numpy(), qi.cpu().numpy(), None if sat is None else sat.cpu().numpy())
            assert np.array_equal(cmx.remove_precipitation(p, ql, qi, sat).cpu().numpy(), ref)
            assert np.array_equal(cmx.d_remove_precipitation_d_q_tot(p, ql, qi, sat).cpu().numpy(), dref)
