"""GPU parity of the producer-level arithmetic: the
kernels of csrc/glue.hip behind core/protools.py, spectra/metrics.py and
experimental/coupling/transforms.py against the reference's golden outputs
(G12, G14) and NumPy / SciPy on seeded inputs -- for host chunks (uploaded,
ndarray back) and for device-resident chunks (CUDA tensors end to end).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def producer(*a, **k):
    from openseize_amd import producer as p
    return p(*a, **k)


def host(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)


def close(got, want, tol=1e-12):
    got, want = host(got), np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    return np.allclose(got, want, rtol=tol, atol=tol, equal_nan=True)


@pytest.fixture(scope="module")
def protools():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from openseize_amd import _lib
    _lib.load()
    from openseize_amd.core import protools as pt
    return pt


@pytest.mark.parametrize("resident", [False, True])
def test_protools_arithmetic_golden(protools, golden, resident):
    """add / multiply / multiply_along_axis / mean / std / standardize against
    the reference's outputs (G12 holds a NaN-bearing signal), host-fed and
    device-resident; results keep the memory kind of the chunks."""
    import torch
    g = golden("g12_protools.npz")
    x = g["x"]
    data = torch.from_numpy(x).cuda() if resident else x
    pro = producer(data, 900, axis=-1)

    def kind_ok(a):
        return (torch.is_tensor(a) and a.is_cuda) if resident else isinstance(a, (np.ndarray, np.floating))

    out = protools.add(pro, g["other"]).to_array()
    assert kind_ok(out) and close(out, g["add_arr"])
    two_x = torch.from_numpy(2 * x).cuda() if resident else 2 * x
    assert close(protools.multiply(pro, producer(two_x, 500, axis=-1)).to_array(), g["mul_pro"])
    assert close(protools.multiply(pro, 3.5).to_array(), 3.5 * x)
    with pytest.raises(ValueError):
        list(protools.add(pro, producer(data[:2], 900, axis=-1)))
    assert close(protools.multiply_along_axis(pro, g["w"], -1).to_array(), g["mul_along_prod"])
    assert close(protools.multiply_along_axis(pro, np.array([1.0, 2.0, 3.0]), 0).to_array(),
                 g["mul_along_other"])
    # 3-D, the reference's own outputs: a middle axis is the plain broadcast, axis 0 the quirk
    x3 = torch.from_numpy(g["x3"]).cuda() if resident else g["x3"]
    assert close(protools.multiply_along_axis(producer(x3, 100, axis=-1), g["w4"], 1).to_array(), g["mul3_middle"])
    assert close(protools.multiply_along_axis(producer(x3, 100, axis=-1), g["w2"], 0).to_array(), g["mul3_first"])
    assert close(protools.multiply_along_axis(producer(x3, 1, axis=0), g["w1250"], 2).to_array(),
                 g["mul3_last_prod0"])
    for ignore in (True, False):
        m = protools.mean(pro, -1, ignore, keepdims=True)
        assert kind_ok(m) and close(m, g[f"mean_prod_{int(ignore)}"])
        assert close(protools.std(pro, -1, ignore, keepdims=True), g[f"std_prod_{int(ignore)}"])
    assert close(protools.mean(pro, 0), g["mean_other"])
    assert close(protools.std(pro, 0), g["std_other"])
    assert close(protools.mean(pro, -1), g["mean_prod_1"][..., 0])
    st = protools.standardize(pro, -1)
    chunks = list(st)
    assert all(kind_ok(c) for c in chunks)
    assert close(np.concatenate([host(c) for c in chunks], -1), g["standardize_prod"], 1e-11)
    assert close(protools.standardize(pro, 0).to_array(), g["standardize_other"], 1e-11)


def test_protools_random_shapes(protools):
    """1-D ... 4-D producers with the sample axis anywhere, as the reference's
    tests/test_protools.py draws them: mean / std / standardize / add / multiply
    against whole-array NumPy."""
    rng = np.random.default_rng(3)
    for ndim in (1, 2, 3, 4):
        for axis in range(ndim):
            shape = [int(rng.integers(2, 5)) for _ in range(ndim)]
            shape[axis] = int(rng.integers(3000, 9000))
            x = rng.standard_normal(shape) * 3 + 1
            pro = producer(x, 1777, axis=axis)
            assert close(protools.mean(pro, axis), x.mean(axis))
            assert close(protools.std(pro, axis, keepdims=True), x.std(axis, keepdims=True), 1e-11)
            z = (x - x.mean(axis, keepdims=True)) / x.std(axis, keepdims=True)
            assert close(protools.standardize(pro, axis).to_array(), z, 1e-10)
            other = rng.standard_normal([1 if i == axis else s for i, s in enumerate(shape)])
            assert close(protools.add(pro, other).to_array(), x + other)
            assert close(protools.multiply(pro, -2.0).to_array(), -2.0 * x)
            if ndim > 1:
                ax2 = (axis + 1) % ndim
                assert close(protools.mean(pro, ax2, keepdims=True), x.mean(ax2, keepdims=True))
                assert close(protools.std(pro, ax2), x.std(ax2), 1e-11)
                w = rng.standard_normal(shape[ax2])
                wshape = [1] * ndim
                wshape[ax2] = -1
                # reference quirk Q14: when the multiplied axis is axis 0, chunk k < len(w) is
                # scaled by the single value w[k]; along any other axis it is the plain product
                want = x * w.reshape(wshape)
                if ax2 == 0:
                    for k in range(min(len(w), -(-shape[axis] // 1777))):
                        sl = [slice(None)] * ndim
                        sl[axis] = slice(k * 1777, (k + 1) * 1777)
                        want[tuple(sl)] = x[tuple(sl)] * w[k]
                assert close(protools.multiply_along_axis(pro, w, ax2).to_array(), want)
                # an operand varying along BOTH the sample axis and another axis
                full = rng.standard_normal(shape)
                assert close(protools.add(pro, producer(full, 1777, axis=axis)).to_array(), x + full)


def test_moments_fullsize_256ch(protools):
    """BASELINE chunk shape (256 ch x 2^20, four chunks): streaming moments of
    a device-resident producer against float64 NumPy on three channels, and the
    standardised stream has zero mean / unit std."""
    import torch
    from openseize_amd import _device as dev
    C, cs = 256, 1 << 20
    x = torch.cat([dev.synth_normal(C, cs, seed=9, n0=k * cs) for k in range(4)], 1)
    x = x * 2.5 + 0.75
    pro = producer(x, cs, -1)
    mu, sd = protools.mean(pro, -1), protools.std(pro, -1)
    pick = [0, 100, 255]
    xh = x[pick].cpu().numpy()
    assert np.max(np.abs(host(mu)[pick] - xh.mean(-1))) < 1e-12
    assert np.max(np.abs(host(sd)[pick] - xh.std(-1))) < 1e-11
    z = protools.standardize(pro, -1)
    zpro = producer(torch.cat(list(z), -1), cs, -1)
    assert float(protools.mean(zpro, -1).abs().max()) < 1e-12
    assert float((protools.std(zpro, -1) - 1).abs().max()) < 1e-12


def test_band_metrics_golden(golden):
    """spectra.metrics.power / power_norm: the device Simpson kernel against the
    reference's outputs (odd and even bin counts, either axis) and
    scipy.integrate.simpson on random bands; CUDA estimates stay on the device."""
    import torch
    from scipy.integrate import simpson
    from openseize_amd.spectra import metrics
    g = golden("g14_metrics_analytic.npz")
    psd, freqs = g["psd"], g["freqs"]
    for est in (psd, torch.from_numpy(psd).cuda()):
        assert close(metrics.power(est, freqs), g["power_all"])
        assert close(metrics.power(est, freqs, start=0, stop=40), g["power_0_40"])
        assert close(metrics.power(est, freqs, start=7.3, stop=33.1), g["power_7p3_33p1"])
        assert close(metrics.power(est.T, freqs, start=2, stop=100, axis=0), g["power_axis0"])
        pn = metrics.power_norm(est, freqs, start=4, stop=30)
        assert close(pn, g["power_norm_4_30"])
        assert torch.is_tensor(pn) == torch.is_tensor(est)
    rng = np.random.default_rng(1)
    p = rng.random((5, 2049))
    f = np.fft.rfftfreq(4096, 1 / 4096)
    for _ in range(20):
        a, b = sorted(rng.integers(0, 2049, 2))
        want = simpson(p[:, a:b + 1], dx=1.0, axis=-1) if b > a else np.zeros(5)
        assert close(metrics.power(p, f, start=f[a], stop=f[b]), want)


def test_analytic_amplitude_phase_kernels(golden):
    """Analytic transform: x + i H(x) joined, |z| and the phase taken by the
    device kernels, for host-fed and device-resident data (reference
    experimental/coupling/transforms.py:153-192 via G14)."""
    import torch
    from openseize_amd.experimental.coupling.transforms import Analytic
    g = golden("g14_metrics_analytic.npz")
    for data in (g["x"], torch.from_numpy(g["x"]).cuda()):
        tr = Analytic(data, fs=500, chunksize=2500, axis=-1, width=12.5)
        sig = host(tr.signal.to_array(dtype=complex))
        assert np.max(np.abs(sig - g["signal"])) < 1e-9 * np.max(np.abs(g["signal"]))
        amp = tr.amplitudes.to_array()
        assert torch.is_tensor(amp) == torch.is_tensor(data)
        assert close(amp, g["amplitudes"], 1e-9)
        ph = host(tr.phases.to_array())
        assert ph.min() >= 0 and ph.max() < 2 * np.pi
        assert np.max(np.abs(np.exp(1j * ph) - np.exp(1j * g["phases"]))) < 1e-8
    # the kernels alone against NumPy
    from openseize_amd import _device as dev
    rng = np.random.default_rng(2)
    z = rng.standard_normal((7, 5000)) + 1j * rng.standard_normal((7, 5000))
    mag, ph = dev.magphase(torch.from_numpy(z).cuda())
    ang = np.angle(z)
    ang[ang < 0] += 2 * np.pi
    assert close(mag, np.abs(z), 1e-14) and close(ph, ang, 1e-13)


def test_lfilter_user_zi_any_order(golden):
    """lfilter with a user zi for orders above 2 (reference
    core/numerical.py:437-446) against scipy.signal.lfilter."""
    import scipy.signal as sps
    from openseize_amd.core import numerical as nm
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 20000))
    for b, a in (sps.butter(4, 0.2), sps.cheby1(5, 1, 0.3), sps.ellip(7, 1, 40, 0.25)):
        order = max(len(a), len(b)) - 1
        zi = rng.standard_normal((3, order))
        want, _ = sps.lfilter(b, a, x, axis=-1, zi=zi)
        got = np.concatenate(list(nm.lfilter(producer(x, 3000, -1), (b, a), -1, zi=zi)), -1)
        assert np.max(np.abs(got - want)) < 1e-9 * np.max(np.abs(want))
    # sample axis first, class API
    from openseize_amd.filtering.iir import Butter
    filt = Butter(fpass=100, fstop=200, fs=1000, fmt="ba")
    b, a = filt.coeffs
    order = max(len(a), len(b)) - 1
    zi = rng.standard_normal((order, 3))
    want, _ = sps.lfilter(b, a, x.T, axis=0, zi=zi)
    got = filt(x.T, chunksize=4096, axis=0, dephase=False, zi=zi)
    assert np.max(np.abs(got - want)) < 1e-9 * np.max(np.abs(want))
