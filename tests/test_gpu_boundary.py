"""The C-ABI boundary on the GPU without PyTorch in the loop, and the
checkpoint / resume entry points.

* ``tests/host/abi_driver.c`` is compiled with gcc against include/osz_hip.h and
  run as its own process: osz_malloc -> osz_memcpy_h2d -> osz_sos_forward /
  osz_fir_push -> osz_memcpy_d2h against the reference's golden vector G3 and
  numpy.convolve with G2's taps
  (exactly the DeviceChunk stub of INTEGRATION.md, in C).
* get_state / set_state of every iterator handle: a stream cut at a chunk
  boundary and resumed on a fresh handle continues bit-identically.
"""

import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_drives_the_abi(golden, tmp_path):
    g3, g2 = golden("g3_sosfilt.npz"), golden("g2_fir.npz")
    x = np.ascontiguousarray(g3["x"], dtype=np.float64)            # (3, N)
    nch, n = x.shape
    sos = np.ascontiguousarray(g3["sos_butter_bp6"], dtype=np.float64)
    want_sos = np.ascontiguousarray(g3["y_butter_bp6_cs1000"])
    taps = np.ascontiguousarray(g2["h256"], dtype=np.float64)      # the taps of G2, on G3's signal
    want_fir = np.stack([np.convolve(row, taps)[:n] for row in x])
    case = tmp_path / "case.bin"
    with open(case, "wb") as fh:
        np.array([nch, n, sos.shape[0], len(taps), 1000], dtype=np.int64).tofile(fh)
        for a in (sos, taps, x, want_sos, want_fir):
            a.tofile(fh)
    exe = tmp_path / "abi_driver"
    lib_dir = os.path.join(ROOT, "openseize_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "host", "abi_driver.c"), "-o", str(exe),
                    "-L", lib_dir, "-losz_hip", f"-Wl,-rpath,{lib_dir}", "-lm"], check=True)
    res = subprocess.run([str(exe), str(case)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "max_rel_err_sos" in res.stdout and "bad_create_rc=-1" in res.stdout


def _split_resume(make, push, x, cut):
    """One handle over the whole stream vs. a handle stopped at `cut`, its state
    saved, and a NEW handle resumed from it."""
    import torch
    a = make()
    whole = [push(a, x[:, :cut], False), push(a, x[:, cut:], True)]
    a.close()
    b = make()
    first = push(b, x[:, :cut], False)
    saved = b.get_state()
    b.close()
    c = make()
    c.set_state(saved)
    second = push(c, x[:, cut:], True)
    c.close()
    for u, v in zip(whole, (first, second)):
        assert u.shape == v.shape and torch.equal(u, v)
    return saved


def test_checkpoint_resume_all_iterators():
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev, _lib
    C, n, cut = 5, 60_000, 23_457
    x = dev.synth_normal(C, n, seed=41)
    # FIR (single part and partitioned)
    for ntaps in (255, 5000):
        h = sps.firwin(ntaps, 0.2)
        st = _split_resume(lambda: dev.FirStream(h, C), lambda s, a, last: s.push(a.contiguous()), x, cut)
        assert st.size == (C * (ntaps - 1) if ntaps <= 2049 else st.size)
    # polyphase 3/2
    hp = sps.firwin(91, 1 / 3)
    _split_resume(lambda: dev.PolyStream(hp, 3, 2, C),
                  lambda s, a, last: s.push(a.contiguous(), final=last), x, cut)
    # spectra: STFT segments and the PSD accumulator
    w = sps.get_window("hann", 1000)
    for mode in (_lib.SPEC_DFT_SEGMENTS, _lib.SPEC_PSD_SEGMENTS):
        _split_resume(lambda: dev.SpecStream(1000, 1000, 400, w, 0.01, "constant", mode, C),
                      lambda s, a, last: s.push(a.contiguous()), x, cut)
    w4 = sps.get_window("hann", 4096)

    def run(resume):
        s = dev.SpecStream(4096, 4096, 2048, w4, 0.01, "linear", _lib.SPEC_PSD_MEAN, C)
        s.push(x[:, :cut].contiguous())
        if resume:
            saved = s.get_state()
            s.close()
            s = dev.SpecStream(4096, 4096, 2048, w4, 0.01, "linear", _lib.SPEC_PSD_MEAN, C)
            s.set_state(saved)
        s.push(x[:, cut:].contiguous())
        total, cnt = s.export_sum()
        s.close()
        return total, cnt

    (t0, c0), (t1, c1) = run(False), run(True)
    assert c0 == c1 == (n - 4096) // 2048 + 1 and torch.equal(t0, t1)
    # a state of the wrong size is refused
    f = dev.FirStream(sps.firwin(31, 0.3), C)
    with pytest.raises(ValueError):
        f.set_state(np.zeros(7))
    f.close()


def test_state_snapshots_on_the_device():
    """osz_fir_* / osz_sos_*_state handed a DEVICE array: the same numbers as the host array,
    and a handle restored from the snapshot repeats its output bit for bit; the polyphase
    handle's state starts with host scalars and says so."""
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    C, n = 5, 40_000
    x = dev.synth_normal(C, 2 * n, seed=43)
    f = dev.FirStream(sps.firwin(255, 0.2), C)
    s = dev.SosStream(sps.butter(4, [0.1, 0.3], "bandpass", output="sos"), C)
    for h in (f, s):
        run = (lambda a: h.push(a)) if h is f else (lambda a: h.forward(a))
        run(x[:, :n].contiguous())
        snap = h.snapshot()
        assert snap.is_cuda and np.array_equal(snap.cpu().numpy().ravel(), np.asarray(h.get_state()).ravel())
        first = run(x[:, n:].contiguous()).clone()
        h.restore(snap)
        assert torch.equal(run(x[:, n:].contiguous()), first)
        with pytest.raises(ValueError):
            h.restore(snap[:-1])
        h.close()
    p = dev.PolyStream(sps.firwin(91, 1 / 3), 3, 2, C)
    with pytest.raises(TypeError):
        p.snapshot()
    p.close()


def test_psd_device_input_stays_on_device():
    """psd() of a CUDA tensor returns a CUDA tensor averaged on the device
    (osz_spec_mean_device) and equals the host-input result."""
    import torch
    from openseize_amd import _device as dev
    from openseize_amd.spectra.estimators import psd
    x = dev.synth_normal(6, 50_000, seed=5)
    cd, fd, pd = psd(x, fs=1000, axis=-1, resolution=0.5)
    ch, fh, ph = psd(x.cpu().numpy(), fs=1000, axis=-1, resolution=0.5)
    assert isinstance(pd, torch.Tensor) and pd.is_cuda and isinstance(ph, np.ndarray)
    assert cd == ch and np.array_equal(fd, fh)
    assert np.array_equal(pd.cpu().numpy(), ph)
    # sample axis first
    c2, _, p2 = psd(x.T.contiguous(), fs=1000, axis=0, resolution=0.5)
    assert tuple(p2.shape) == (len(fd), 6) and np.array_equal(p2.cpu().numpy(), ph.T)


def test_host_chain_stays_on_the_device_between_stages():
    """ndarray -> sosfilt -> FIR -> downsample -> psd, every stage a producer of this
    library: the stages hand CUDA tensors to each other (one upload, one result
    down), the caller still gets ndarrays -- and the same numbers as stage-by-stage
    arrays and as the all-resident chain.  Iterating an intermediate producer
    directly yields ndarrays as before."""
    import numpy as np
    import scipy.signal as sps
    import torch
    from functools import partial
    from openseize_amd import _device as dev
    from openseize_amd import producer
    from openseize_amd.core import numerical as nm
    from openseize_amd.core import producer as pmod
    from openseize_amd.filtering.fir import Kaiser
    from openseize_amd.spectra.estimators import psd
    rng = np.random.default_rng(8)
    x = rng.standard_normal((6, 400003))
    cs, fs = 50000, 5000.0
    sos = sps.butter(4, 0.4, output="sos")
    taps = sps.firwin(201, 0.3)

    def chain(data):
        a = producer(data, cs, -1)
        b = producer(partial(nm.sosfilt, a, sos, -1), cs, -1, shape=a.shape)
        c = producer(partial(nm.oaconvolve, b, taps, -1, "same"), cs, -1, shape=a.shape)
        n5 = -(-x.shape[1] // 5)
        d = producer(partial(nm.polyphase_resample, c, 1, 5, fs, Kaiser, -1), cs, -1, shape=(6, n5))
        return a, b, c, d

    uploads, plain_upload = [], dev.HostPipe.upload

    def counting(self, arr):
        uploads.append(arr.shape)
        return plain_upload(self, arr)

    dev.HostPipe.upload = counting
    try:
        a, b, c, d = chain(x)
        pieces = list(d)
    finally:
        dev.HostPipe.upload = plain_upload
    assert all(isinstance(p, np.ndarray) for p in pieces)
    assert len(uploads) == -(-x.shape[1] // cs)          # the source's chunks, once; nothing else went up
    got = np.concatenate(pieces, -1)
    # stage by stage through arrays
    y1 = np.concatenate(list(nm.sosfilt(producer(x, cs, -1), sos, -1)), -1)
    y2 = np.concatenate(list(nm.oaconvolve(producer(y1, cs, -1), taps, -1, "same")), -1)
    y3 = np.concatenate(list(nm.polyphase_resample(producer(y2, cs, -1), 1, 5, fs, Kaiser, -1)), -1)
    assert got.shape == y3.shape and np.max(np.abs(got - y3)) < 1e-12 * np.max(np.abs(y3))
    # OSZ_HOST_CHAIN=0: every stage hands ndarrays on, as the reference's generators do -- one
    # upload per stage and chunk, the same numbers
    import os
    uploads.clear()
    os.environ["OSZ_HOST_CHAIN"] = "0"
    dev.HostPipe.upload = counting
    try:
        apart = np.concatenate(list(chain(x)[3]), -1)
    finally:
        dev.HostPipe.upload = plain_upload
        del os.environ["OSZ_HOST_CHAIN"]
    assert len(uploads) > 2 * -(-x.shape[1] // cs)
    assert apart.shape == got.shape and np.max(np.abs(apart - got)) < 1e-12 * np.max(np.abs(got))
    # all resident
    xd = torch.from_numpy(x).cuda()
    res = torch.cat(list(chain(xd)[3]), -1)
    assert res.is_cuda and float((res.cpu() - torch.from_numpy(y3)).abs().max()) < 1e-12 * np.max(np.abs(y3))
    # an intermediate producer iterated by the caller: ndarrays
    a, b, c, d = chain(x)
    first = next(iter(c))
    assert isinstance(first, np.ndarray) and np.allclose(first, y2[:, :cs], rtol=0, atol=1e-12 * np.max(np.abs(y2)))
    # a stage written by the user in between takes and hands on ndarrays, whatever runs around it
    def rectify(pro):
        for chunk in pro:
            assert isinstance(chunk, np.ndarray)
            yield np.abs(chunk)

    a = producer(x, cs, -1)
    b = producer(partial(nm.sosfilt, a, sos, -1), cs, -1, shape=a.shape)
    u = producer(rectify, cs, -1, shape=a.shape, pro=b)
    c = producer(partial(nm.oaconvolve, u, taps, -1, "same"), cs, -1, shape=a.shape)
    got_u = np.concatenate(list(c), -1)
    want_u = np.concatenate(list(nm.oaconvolve(producer(np.abs(y1), cs, -1), taps, -1, "same")), -1)
    assert np.max(np.abs(got_u - want_u)) < 1e-12 * np.max(np.abs(want_u))
    # producer-level arithmetic inside a host chain: tensors through, ndarrays out
    from openseize_amd.core import protools
    a = producer(x, cs, -1)
    b = producer(partial(nm.sosfilt, a, sos, -1), cs, -1, shape=a.shape)
    z = protools.standardize(b, axis=-1)
    zc = producer(partial(nm.oaconvolve, z, taps, -1, "same"), cs, -1, shape=a.shape)
    got_z = np.concatenate(list(zc), -1)
    y1z = (y1 - y1.mean(-1, keepdims=True)) / y1.std(-1, keepdims=True)
    want_z = np.concatenate(list(nm.oaconvolve(producer(y1z, cs, -1), taps, -1, "same")), -1)
    assert np.max(np.abs(got_z - want_z)) < 1e-10 * np.max(np.abs(want_z))
    m = protools.mean(b, axis=-1)
    assert isinstance(m, np.ndarray) and np.allclose(m, y1.mean(-1), rtol=0, atol=1e-12)
    assert all(isinstance(c_, np.ndarray) for c_ in protools.multiply(b, 2.0))
    # a masked producer between two stages passes the grant on: tensors in, tensors out
    mask = rng.random(x.shape[1]) < 0.7
    b = producer(partial(nm.sosfilt, producer(x, cs, -1), sos, -1), cs, -1, shape=x.shape)
    mk = producer(b, cs, -1, mask=mask)
    kinds = []
    plain_take = pmod._take_device
    pmod._take_device = lambda arr, keep_, axis_: (kinds.append(1), plain_take(arr, keep_, axis_))[1]
    try:
        mc = producer(partial(nm.oaconvolve, mk, taps, -1, "same"), cs, -1, shape=mk.shape)
        got_m = np.concatenate(list(mc), -1)
    finally:
        pmod._take_device = plain_take
    assert kinds, "the masked producer saw host chunks inside a chain of this library"
    want_m = np.concatenate(list(nm.oaconvolve(producer(y1[:, mask], cs, -1), taps, -1, "same")), -1)
    assert got_m.shape == want_m.shape and np.max(np.abs(got_m - want_m)) < 1e-12 * np.max(np.abs(want_m))
    assert all(isinstance(c_, np.ndarray) for c_ in producer(b, cs, -1, mask=mask))
    # the estimator at the end of a host chain: pulls resident, returns a host estimate
    cnt, f, p = psd(chain(x)[2], fs, axis=-1, resolution=1.0)
    cnt2, f2, p2 = psd(y2, fs, axis=-1, resolution=1.0)
    assert cnt == cnt2 and isinstance(p, np.ndarray) and np.max(np.abs(p - p2)) < 1e-10 * np.max(p2)



def _pool_worker(blob):
    """Runs in a fresh process: unpickle a pipeline of producers, pull it, return the stream."""
    import pickle

    import numpy as np
    pro = pickle.loads(blob)
    return np.concatenate(list(pro), axis=-1)


def test_pickled_producers_run_in_other_processes():
    """Why the reference keeps producers picklable (tests/test_concurrency.py:85-167): pipelines
    are handed to worker processes.  Two spawned workers (fresh interpreters, each initialises
    the GPU itself: no handle or stream travels in the pickle) pull FIR -> zero-phase IIR and
    IIR -> downsample chains built here; results equal the same chains pulled in this process."""
    import multiprocessing as mp
    import pickle
    from functools import partial

    import scipy.signal as sps
    from openseize_amd import producer
    from openseize_amd.core import numerical as nm
    from openseize_amd.filtering.fir import Kaiser
    rng = np.random.default_rng(21)
    x = rng.standard_normal((4, 150000))
    taps = sps.firwin(301, 0.2)
    sos = sps.butter(4, [0.05, 0.3], "bandpass", output="sos")
    cs = 40000
    src = producer(x, cs, -1)
    fir = producer(partial(nm.oaconvolve, src, taps, -1, "same"), cs, -1, shape=x.shape)
    chain1 = producer(partial(nm.sosfiltfilt, fir, sos, -1), cs, -1, shape=x.shape)
    iir = producer(partial(nm.sosfilt, src, sos, -1), cs, -1, shape=x.shape)
    chain2 = producer(partial(nm.polyphase_resample, iir, 1, 5, 5000, Kaiser, -1), cs, -1,
                      shape=(4, 30000))
    blobs = [pickle.dumps(chain1), pickle.dumps(chain2)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(2) as pool:
        got = pool.map(_pool_worker, blobs)
    for g, chain in zip(got, (chain1, chain2)):
        want = np.concatenate(list(chain), axis=-1)
        assert g.shape == want.shape and np.array_equal(g, want)
