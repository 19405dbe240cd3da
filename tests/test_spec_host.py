"""The host tables of the spectral FIR -> cascade kernel (openseize_amd/csrc/spec_tables.h,
used by chain_spec.hip), built by g++ from the same header and (1) held against NumPy,
(2) driven through a NumPy restatement of the kernel's dataflow -- whole pairs with the
overlap add and the mode bursts, the opening pair's carry, the generic closing pair, runs
with a one-pair pre-roll -- against scipy's sosfilt(convolve(x, h)): the block algorithm
and its tables are pinned without a GPU."""

import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest
import scipy.signal as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 4096


@pytest.fixture(scope="module")
def exe():
    src = os.path.join(ROOT, "tests", "host", "spec_host_check.cpp")
    inc = os.path.join(ROOT, "openseize_amd", "csrc")
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "spec_host_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", inc, src, "-o", path])
    return path


def tables(exe, taps, sos, forgets=True):
    taps, sos = np.asarray(taps, np.float64), np.atleast_2d(np.asarray(sos, np.float64))
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<iii", len(taps), len(sos), int(forgets)))
            f.write(taps.tobytes())
            f.write(sos.tobytes())
        subprocess.check_call([exe, fin, fout])
        raw = open(fout, "rb").read()
    elig, NR, NM, nm, R = struct.unpack_from("<iiiii", raw, 0)
    ratio, = struct.unpack_from("<d", raw, 20)
    pos, arrs = 28, []
    for _ in range(4):
        n, = struct.unpack_from("<q", raw, pos)
        arrs.append(np.frombuffer(raw, np.float64, n, pos + 8).copy())
        pos += 8 + 8 * n
    return dict(eligible=bool(elig), NR=NR, NM=NM, nm=nm, R=R, ratio=ratio, H=arrs[0], M=arrs[1],
                P=arrs[2], L=arrs[3])


class Model:
    """The kernel's dataflow (chain_spec.hip) on one channel, with the tables of the C++ build."""

    def __init__(self, T, wlen):
        self.NR, self.NM, self.R = T["NR"], T["NM"], T["R"]
        self.S, self.D = 256 * self.NR, 16 - self.NR
        H = T["H"].reshape(N, 2)
        self.Hc = (H[:, 0] + 1j * H[:, 1]) * N            # the tables carry the 1/4096
        M = T["M"].reshape(2 * self.NM, 64)
        self.M = M[:self.NM] + 1j * M[self.NM:]
        P = T["P"].reshape(32, self.NM, 2)
        P = P[..., 0] + 1j * P[..., 1]
        t = np.arange(256)
        self.P = P[t >> 4] * P[16 + (t & 15)]             # lambda^t as the kernel forms it
        L = T["L"].reshape(5, self.NM, 2)
        self.L = L[..., 0] + 1j * L[..., 1]
        self.CL = N + 256 * self.R

    def window(self, x):
        buf = np.zeros(N)
        buf[:len(x)] = x
        return np.real(np.fft.ifft(np.fft.fft(buf) * self.Hc))

    def fit(self, win):
        return self.M @ win[3840:3904]

    def burst(self, mu, e):
        ok = (e >= 0) & (e < 256 * self.R)
        ee = np.where(ok, e, 0)
        return np.where(ok, np.real((self.L[ee >> 8] * self.P[ee & 255]) @ mu), 0.0)

    def chunk(self, x, carry_in, nruns):
        n, S, NR, D, R = len(x), self.S, self.NR, self.D, self.R
        pair = 2 * S
        npw = n // pair
        rem = n - npw * pair
        W = npw - 1 if rem == 0 else npw
        assert W >= 1
        f = np.full(n, np.nan)
        nruns = max(1, min(nruns, W))
        t = np.arange(256)
        carry_out = None
        for run in range(nruns):
            p0, p1 = run * W // nruns, (run + 1) * W // nruns
            cr, mu_prev = np.zeros((D, 256)), np.zeros(self.NM, complex)
            for p in range(p0 if run == 0 else p0 - 1, p1):
                o = p * pair
                wa, wb = self.window(x[o:o + S]), self.window(x[o + S:o + pair])
                mu_a, mu_b = self.fit(wa), self.fit(wb)
                Ya, Yb = wa.reshape(16, 256), wb.reshape(16, 256)
                A, B = Ya[:NR].copy(), Yb[:NR].copy()
                A[:D] += cr
                B[:D] += Ya[NR:]
                cr = Yb[NR:].copy()
                for r in range(R):
                    A[r] += self.burst(-mu_a, 256 * r + t)
                    A[D + r] += self.burst(mu_prev, 256 * r + t)
                    B[r] += self.burst(-mu_b, 256 * r + t)
                    B[D + r] += self.burst(mu_a, 256 * r + t)
                mu_prev = mu_b
                if p == 0:
                    ci = np.zeros(pair)
                    ci[:self.CL] = carry_in[:self.CL]
                    A += ci[:S].reshape(NR, 256)
                    B += ci[S:].reshape(NR, 256)
                if p >= p0:
                    f[o:o + S], f[o + S:o + pair] = A.ravel(), B.ravel()
            if run == nruns - 1:
                o = W * pair
                la = min(n - o, S)
                lb = n - o - la
                wa, wb = self.window(x[o:o + la]), self.window(x[o + la:o + la + lb])
                mu_a, mu_b = self.fit(wa), self.fit(wb)
                acc, i = np.zeros(8192), np.arange(8192)
                acc[:256 * D] += cr.ravel()
                acc[:N] += wa
                acc[la:la + N] += wb
                for mu, off in ((mu_prev, 256 * D), (-mu_a, 0), (-mu_b, la), (mu_a, N), (mu_b, la + N)):
                    acc += self.burst(mu, i - off)
                f[o:n] = acc[:la + lb]
                carry_out = np.zeros(7680)
                seg = acc[la + lb:]
                carry_out[:min(len(seg), 7680)] = seg[:7680]
        return f, carry_out


CASES = [
    ("butter6 band-pass, 1024 taps (cfg-3)", 1024, sps.butter(6, [0.05, 0.3], "bandpass", output="sos")),
    ("butter6 band-pass, 300 taps", 300, sps.butter(6, [0.05, 0.3], "bandpass", output="sos")),
    ("cheby1 band-pass, 777 taps", 777, sps.cheby1(3, 1, [0.16, 0.48], "bandpass", output="sos")),
    ("butter5 low-pass (a real pole), 513 taps", 513, sps.butter(5, 0.3, output="sos")),
    ("elliptic high-pass, 64 taps", 64, sps.ellip(4, 0.5, 50, 0.25, "highpass", output="sos")),
]


@pytest.mark.parametrize("name,ntaps,sos", CASES, ids=[c[0] for c in CASES])
def test_tables_and_block_algorithm(exe, name, ntaps, sos):
    taps = sps.firwin(ntaps, 0.2)
    T = tables(exe, taps, sos)
    assert T["eligible"], name
    NR, NM, R = T["NR"], T["NM"], T["R"]
    assert NR == min((3841 - ntaps) // 256, 15) and 1 <= R <= min(16 - NR, 2 * NR - 16, 5)
    # (1) against NumPy: the composite spectrum and the mode powers
    w, h = sps.sosfreqz(sos, worN=N, whole=True)
    Hc = np.fft.fft(taps, N) * h / N
    H = T["H"].reshape(N, 2)
    assert np.max(np.abs(H[:, 0] + 1j * H[:, 1] - Hc)) < 1e-15 * np.max(np.abs(Hc)) * 50
    P = T["P"].reshape(32, NM, 2)
    lam = P[17, :, 0] + 1j * P[17, :, 1]
    assert np.all(np.abs(lam[:T["nm"]]) < 1) and np.all(lam[T["nm"]:] == 0)
    poles = np.concatenate([np.roots(s[3:]) for s in sos if s[4] or s[5]])
    for q in range(T["nm"]):
        assert np.min(np.abs(poles - lam[q])) < 1e-12
    assert np.allclose(P[3, :T["nm"], 0] + 1j * P[3, :T["nm"], 1], lam[:T["nm"]] ** 48, rtol=1e-13, atol=0)
    L = T["L"].reshape(5, NM, 2)
    assert np.allclose(L[2, :T["nm"], 0] + 1j * L[2, :T["nm"], 1], lam[:T["nm"]] ** 512, rtol=1e-12, atol=1e-300)
    # (2) the block algorithm with these tables against scipy
    rng = np.random.default_rng(len(taps))
    m = Model(T, ntaps)
    S = m.S
    lens = [2 * S * 5 + 1024, 2 * S * 4, 2 * S * 3 + S + 17, 2 * S + 5, 2 * S * 3 + 2 * S - 1]
    x = rng.standard_normal(sum(lens))
    u = np.convolve(x, taps)
    zi0 = sps.sosfilt_zi(sos) * u[0]
    ref, _ = sps.sosfilt(sos, u, zi=zi0)
    # import: the pending FIR tail (none yet) through the cascade from its state
    carry = np.zeros(7680)
    carry[:m.CL] = sps.sosfilt(sos, np.zeros(m.CL), zi=zi0)[0]
    o, scale = 0, np.max(np.abs(ref))
    for k, n in enumerate(lens):
        f, carry = m.chunk(x[o:o + n], carry, nruns=[1, 2, 3, 4, 2][k])
        assert np.max(np.abs(f - ref[o:o + n])) < 1e-12 * scale, (name, k)
        o += n
    # the carry IS the stream's future if the input stops: the flush of the chain
    assert np.max(np.abs(carry[:ntaps - 1] - ref[o:o + ntaps - 1])) < 1e-12 * scale


def test_what_the_scheme_does_not_take(exe):
    """Cascades outside the scheme are refused (chain_kernel serves them): a double pole,
    ringing that outlasts the guard rows, a filter too long for row 15 to be free, a
    cascade that does not forget."""
    bp = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    assert tables(exe, sps.firwin(1024, 0.2), bp)["eligible"]
    twice = np.vstack([sps.butter(2, 0.2, output="sos")] * 2)          # every pole twice
    assert not tables(exe, sps.firwin(256, 0.2), twice)["eligible"]
    narrow = sps.butter(4, [0.0002, 0.0016], "bandpass", output="sos")  # 0.5-4 Hz at 5 kHz
    assert not tables(exe, sps.firwin(256, 0.2), narrow)["eligible"]
    assert not tables(exe, sps.firwin(1900, 0.2), bp)["eligible"]
    assert not tables(exe, sps.firwin(1024, 0.2), bp, forgets=False)["eligible"]
    assert not tables(exe, sps.firwin(64, 0.2), sps.butter(14, 0.3, output="sos"))["eligible"]   # 7 modes


# ------------------------------------------------------------------ the two-sided scheme
def tables_zp(exe, taps, sos, forgets=True, mode="zp"):
    taps, sos = np.asarray(taps, np.float64), np.atleast_2d(np.asarray(sos, np.float64))
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<iii", len(taps), len(sos), int(forgets)))
            f.write(taps.tobytes())
            f.write(sos.tobytes())
        subprocess.check_call([exe, fin, fout, mode])
        raw = open(fout, "rb").read()
    elig, NR, NM, nm, R, nh, Rf, NS = struct.unpack_from("<iiiiiiii", raw, 0)
    ratio, = struct.unpack_from("<d", raw, 32)
    pos, arrs = 40, []
    for _ in range(4):
        n, = struct.unpack_from("<q", raw, pos)
        arrs.append(np.frombuffer(raw, np.float64, n, pos + 8).copy())
        pos += 8 + 8 * n
    return dict(eligible=bool(elig), NR=NR, NM=NM, nm=nm, R=R, Rf=Rf, nh=nh, NS=NS, ratio=ratio, H=arrs[0],
                M=arrs[1], P=arrs[2], L=arrs[3])


class ModelZp:
    """The dataflow of chain_zp_kernel on one channel with the tables of the C++ build:
    whole pairs (overlap add, four forward and four backward bursts, the last R rows of
    block b held back until the next pair's block a has been fitted), runs that start one
    pair early and store that pair's held rows for the run before, the opening pair (carry, held samples of the previous chunk), the
    generic closing pair, outputs delayed by L = 256 R samples."""

    def __init__(self, T):
        self.NR, self.NM, self.R, self.Rf, self.nh = T["NR"], T["NM"], T["R"], T["Rf"], T["nh"]
        self.S, self.D, self.L = 256 * self.NR, 16 - self.NR, 256 * T["R"]
        H = T["H"].reshape(N, 2)
        self.Hc = (H[:, 0] + 1j * H[:, 1]) * N
        M = T["M"].reshape(4 * self.NM, 2 * self.nh)
        self.Mmu = M[:self.NM] + 1j * M[self.NM:2 * self.NM]
        self.Mnu = M[2 * self.NM:3 * self.NM] + 1j * M[3 * self.NM:]
        P = T["P"].reshape(20, self.NM, 2)
        P = P[..., 0] + 1j * P[..., 1]
        t = np.arange(256)
        self.P = P[t >> 5] * P[8 + ((t >> 2) & 7)] * P[16 + (t & 3)]     # lambda^t as the kernel forms it
        Lr = T["L"].reshape(5, self.NM, 2)
        self.Lr = Lr[..., 0] + 1j * Lr[..., 1]
        self.lsel = np.concatenate([np.arange(self.nh), np.arange(256 - self.nh, 256)])

    def window(self, x):
        buf = np.zeros(N)
        buf[:len(x)] = x
        return np.real(np.fft.ifft(np.fft.fft(buf) * self.Hc))

    def fit(self, win):
        y = win[3840 + self.lsel]
        return self.Mmu @ y, self.Mnu @ y

    def burst(self, amp, e, rows):
        ok = (e >= 0) & (e < 256 * rows)
        ee = np.where(ok, e, 0)
        return np.where(ok, np.real((self.Lr[ee >> 8] * self.P[ee & 255]) @ amp), 0.0)

    def chunk(self, x, carry_in, held_in, nruns):
        n, S, NR, D, R, L, Rf = len(x), self.S, self.NR, self.D, self.R, self.L, self.Rf
        pair = 2 * S
        npw = n // pair
        rem = n - npw * pair
        W = npw - 1 if rem == 0 else npw
        assert W >= 1
        lc = n - W * pair
        y, held_out = np.full(n, np.nan), np.full(L, np.nan)
        t = np.arange(256)

        def put(i, v):
            q = i + L
            m = q < n
            y[q[m]] = v[m]
            m2 = (~m) & (i < n)
            held_out[q[m2] - n] = v[m2]

        F = lambda amp, r: self.burst(amp, 256 * r + t, Rf)            # forward burst, row r (Rf rows)
        Bk = lambda amp, r: self.burst(amp, 256 * r + 255 - t, R)      # backward burst, r-th row down (R rows)
        nruns = max(1, min(nruns, W))
        carry_out = None
        for run in range(nruns):
            p0, p1 = run * W // nruns, (run + 1) * W // nruns
            first = p0 if run == 0 else p0 - 1
            lastf = p1 - 1
            cr = np.zeros((D, 256))
            mu_pb = nu_pb = np.zeros(self.NM, complex)
            held = None
            for p in range(first, lastf + 1):
                o = p * pair
                wa, wb = self.window(x[o:o + S]), self.window(x[o + S:o + pair])
                (mu_a, nu_a), (mu_b, nu_b) = self.fit(wa), self.fit(wb)
                Ya, Yb = wa.reshape(16, 256), wb.reshape(16, 256)
                A, B = Ya[:NR].copy(), Yb[:NR].copy()
                A[:D] += cr
                B[:D] += Ya[NR:]
                cr = Yb[NR:].copy()
                for r in range(R):
                    A[r] += F(-mu_a, r)
                    A[D + r] += F(mu_pb, r)
                    A[D - 1 - r] += Bk(-nu_pb, r)
                    A[NR - 1 - r] += Bk(nu_b, r)
                    B[r] += F(-mu_b, r)
                    B[D + r] += F(mu_a, r)
                    B[D - 1 - r] += Bk(-nu_a, r)
                if p == 0:
                    ci = np.zeros(pair)
                    ci[:len(carry_in)] = carry_in[:pair]
                    A += ci[:S].reshape(NR, 256)
                    B += ci[S:].reshape(NR, 256)
                if held is not None:
                    # (also the rows of the pair a run starts early with: its block b depends
                    # on nothing before it, and the run before leaves them to this one)
                    for r in range(R):
                        held[r] += Bk(nu_a, r)
                        put((p - 1) * pair + S + 256 * (NR - 1 - r) + t, held[r])
                elif p == 0:
                    for rr in range(R):
                        y[256 * rr + t] = held_in[256 * rr + t] + Bk(nu_a, R - 1 - rr)
                if p0 <= p < p1:
                    for j in range(NR):
                        put(o + 256 * j + t, A[j])
                    for j in range(NR - R):
                        put(o + S + 256 * j + t, B[j])
                held = [B[NR - 1 - r].copy() for r in range(R)]
                mu_pb, nu_pb = mu_b, nu_b
            if run == nruns - 1:
                o = W * pair
                la = min(lc, S)
                lb = lc - la
                wa, wb = self.window(x[o:o + la]), self.window(x[o + la:o + la + lb])
                (mu_a, nu_a), (mu_b, nu_b) = self.fit(wa), self.fit(wb)
                for r in range(R):
                    held[r] += Bk(nu_a, r)
                    put((W - 1) * pair + S + 256 * (NR - 1 - r) + t, held[r])
                acc, i = np.zeros(8192), np.arange(8192)
                acc[:256 * D] += cr.ravel()
                acc[:N] += wa
                acc[la:la + N] += wb
                for amp, off in ((mu_pb, 256 * D), (-mu_a, 0), (-mu_b, la), (mu_a, N), (mu_b, la + N)):
                    acc += self.burst(amp, i - off, Rf)
                for amp, e0 in ((-nu_pb, 256 * D - 1), (-nu_a, N - 1), (-nu_b, la + N - 1), (nu_b, la - 1)):
                    acc += self.burst(amp, e0 - i, R)
                put(o + i[:lc], acc[:lc])
                carry_out = np.zeros(7680)
                seg = acc[lc:]
                carry_out[:min(len(seg), 7680)] = seg[:7680]
        return y, carry_out, held_out


ZP_CASES = [
    ("butter6 band-pass, 1024 taps (cfg-3)", 1024, sps.butter(6, [0.05, 0.3], "bandpass", output="sos")),
    ("butter6 band-pass, 300 taps", 300, sps.butter(6, [0.05, 0.3], "bandpass", output="sos")),
    ("cheby1 band-pass, 777 taps", 777, sps.cheby1(3, 1, [0.16, 0.48], "bandpass", output="sos")),
    ("butter5 low-pass (a real pole), 513 taps", 513, sps.butter(5, 0.3, output="sos")),
]


@pytest.mark.parametrize("name,ntaps,sos", ZP_CASES, ids=[c[0] for c in ZP_CASES])
def test_zero_phase_tables_and_block_algorithm(exe, name, ntaps, sos):
    """FIR -> forward cascade -> backward cascade as one multiplication per bin: the C++
    tables drive the NumPy restatement of chain_zp_kernel against scipy (forward pass from
    sosfilt_zi * u[0], backward pass over the whole stream), every kind of chunk."""
    taps = sps.firwin(ntaps, 0.2)
    T = tables_zp(exe, taps, sos)
    assert T["eligible"], name
    NR, R, Rf = T["NR"], T["R"], T["Rf"]
    assert 8 <= NR <= min((3841 - ntaps) // 256, 15) and 1 <= Rf <= R <= min(16 - NR, 5) and 16 - NR + Rf <= NR
    w, h = sps.sosfreqz(sos, worN=N, whole=True)
    Hc = np.fft.fft(taps, N) * np.abs(h) ** 2 / N
    H = T["H"].reshape(N, 2)
    assert np.max(np.abs(H[:, 0] + 1j * H[:, 1] - Hc)) < 1e-15 * np.max(np.abs(Hc)) * 50
    m = ModelZp(T)
    S, L = m.S, m.L
    rng = np.random.default_rng(ntaps)
    lens = [2 * S * 5 + 1024, 2 * S * 4, 2 * S * 3 + S + 17, 2 * S + 5, 2 * S * 3 + 2 * S - 1, 2 * S * 3 + 300]
    x = rng.standard_normal(sum(lens))
    u = np.convolve(x, taps)
    zi0 = sps.sosfilt_zi(sos) * u[0]
    f, _ = sps.sosfilt(sos, u, zi=zi0)
    ref = sps.sosfilt(sos, np.concatenate([f, np.zeros(8192)])[::-1])[::-1][:len(f)]
    # the stream opens: what the forward start state rings, filtered backwards as well
    zir = sps.sosfilt(sos, np.zeros(7680), zi=zi0)[0]
    carry, held = sps.sosfilt(sos, zir[::-1])[::-1], np.zeros(L)
    out, o = [], 0
    for k, n in enumerate(lens):
        y, carry, held = m.chunk(x[o:o + n], carry, held, nruns=[1, 2, 3, 1, 2, 2][k])
        out.append(y)
        o += n
    got = np.concatenate(out)                     # got[q] is stream sample q - L
    assert np.isfinite(got).all()
    assert np.max(np.abs(got[L:] - ref[:len(got) - L])) < 1e-12 * np.max(np.abs(ref)), name


# ------------------------------------------------- one real block per transform (chain_zpn.hip)
MW = 8192


class ModelZpn:
    """The dataflow of chain_zpn_kernel on one channel with the tables of the C++ build: a window
    of 8192 samples per block through the 4096-point transform at the odd frequencies (fft::nega:
    negacyclic wrap), the fit on row 31, the in-window corrections ADDED (the wrap changes the
    sign), the right tail's continuation from the previous block's amplitudes, the last R rows
    of a block held back until the next block has been fitted, runs that start one block early,
    the opening block (carry, held samples of the previous chunk), the generic closing block
    (window in an accumulator of 8192, the burst behind it straight into the carry), outputs
    delayed by L = 256 R samples."""

    def __init__(self, T):
        self.NB, self.NM, self.R, self.Rf, self.nh = T["NR"], T["NM"], T["R"], T["Rf"], T["nh"]
        self.NS = NS = T["NS"]
        self.S, self.D, self.L = 256 * self.NB, 32 - self.NB, 256 * T["R"]
        H = T["H"].reshape(N, 2)
        self.Hq = (H[:, 0] + 1j * H[:, 1]) * N
        M = T["M"].reshape(2 * NS + 2 * self.NM, 2 * self.nh)
        self.Mmu = M[:NS] + 1j * M[NS:2 * NS]                        # the slow modes' right tails only
        self.Mnu = M[2 * NS:2 * NS + self.NM] + 1j * M[2 * NS + self.NM:]
        P = T["P"].reshape(20, self.NM, 2)
        P = P[..., 0] + 1j * P[..., 1]
        t = np.arange(256)
        self.P = P[t >> 5] * P[8 + ((t >> 2) & 7)] * P[16 + (t & 3)]
        Lr = T["L"].reshape(-1, self.NM, 2)                            # (eight rows: spec::kRMaxN)
        self.Lr = Lr[..., 0] + 1j * Lr[..., 1]
        self.lsel = np.concatenate([np.arange(self.nh), np.arange(256 - self.nh, 256)])
        self.tw = np.exp(-1j * np.pi * np.arange(N) / MW)

    def window(self, x):
        buf = np.zeros(MW)
        buf[:len(x)] = x
        z = (buf[:N] - 1j * buf[N:]) * self.tw
        w = np.fft.ifft(np.fft.fft(z) * self.Hq) * np.conj(self.tw)
        return np.concatenate([w.real, -w.imag])

    def fit(self, win):
        y = win[MW - 256 + self.lsel]
        return self.Mmu @ y, self.Mnu @ y

    def burst(self, amp, e, rows):
        """Re sum_q amp_q lambda_q^e over the modes `amp` names -- mu: the NS slow ones; nu: all NM
        in its first row of 256 samples, the slow ones behind it"""
        ok = (e >= 0) & (e < 256 * rows)
        ee = np.where(ok, e, 0)
        nq = len(amp)
        terms = (self.Lr[ee >> 8, :nq] * self.P[ee & 255, :nq]) * amp
        if nq > self.NS:
            terms[ee >= 256, self.NS:] = 0.0
        return np.where(ok, np.real(terms.sum(-1)), 0.0)

    def chunk(self, x, carry_in, held_in, nruns):
        n, S, NB, D, R, L, Rf = len(x), self.S, self.NB, self.D, self.R, self.L, self.Rf
        W = (n - 1) // S                       # whole blocks; the closing block has 1 .. S samples
        assert W >= 1
        lc = n - W * S
        y, held_out = np.full(n, np.nan), np.full(L, np.nan)
        t = np.arange(256)

        def put(i, v):
            q = i + L
            m = q < n
            y[q[m]] = v[m]
            m2 = (~m) & (i < n)
            held_out[q[m2] - n] = v[m2]

        F = lambda amp, r: self.burst(amp, 256 * r + t, Rf)
        Bk = lambda amp, r: self.burst(amp, 256 * r + 255 - t, R)
        nruns = max(1, min(nruns, W))
        carry_out = None
        for run in range(nruns):
            p0, p1 = run * W // nruns, (run + 1) * W // nruns
            first, lastf = (p0 if run == 0 else p0 - 1), p1 - 1
            cr = np.zeros((D, 256))
            mu_p = np.zeros(self.NS, complex)
            held = None
            for p in range(first, lastf + 1):
                o = p * S
                win = self.window(x[o:o + S])
                mu, nu = self.fit(win)
                Y = win.reshape(32, 256).copy()
                for r in range(Rf):
                    Y[r] += F(mu, r)                # the wrapped right tail (sign changed) leaves the window
                for r in range(R):
                    Y[31 - r] += Bk(nu, r)          # and the wrapped left tail
                A = Y[:NB].copy()
                A[:D] += cr
                cr = Y[NB:].copy()
                for r in range(Rf):
                    A[D + r] += F(mu_p, r)          # the previous block's right tail continues here
                if p == 0:
                    ci = np.zeros(S)
                    m = min(len(carry_in), S)
                    ci[:m] = carry_in[:m]
                    A += ci.reshape(NB, 256)
                if held is not None:
                    for r in range(R):
                        held[r] += Bk(nu, r)
                        put((p - 1) * S + 256 * (NB - 1 - r) + t, held[r])
                elif p == 0:
                    for rr in range(R):
                        y[256 * rr + t] = held_in[256 * rr + t] + Bk(nu, R - 1 - rr)
                if p0 <= p < p1:
                    for j in range(NB - R):
                        put(o + 256 * j + t, A[j])
                held = [A[NB - 1 - r].copy() for r in range(R)]
                mu_p = mu
            if run == nruns - 1:
                o, la = W * S, lc
                win = self.window(x[o:o + la])
                mu, nu = self.fit(win)
                for r in range(R):
                    held[r] += Bk(nu, r)
                    put((W - 1) * S + 256 * (NB - 1 - r) + t, held[r])
                i = np.arange(MW)
                acc = win.copy()
                acc[:256 * D] += cr.ravel()
                acc += self.burst(mu_p, i - 256 * D, Rf)
                acc += self.burst(mu, i, Rf)
                acc += self.burst(nu, MW - 1 - i, R)
                put(o + i[:lc], acc[:lc])
                k = np.arange(7680)
                src = lc + k
                carry_out = np.where(src < MW, acc[np.minimum(src, MW - 1)], self.burst(mu, src - MW, Rf))
        return y, carry_out, held_out


ZPN_CASES = ZP_CASES + [
    ("the identity as the FIR (plain sosfiltfilt)", 2, sps.butter(6, [0.05, 0.3], "bandpass", output="sos")),
    ("cheby1 low-pass, 57 taps", 57, sps.cheby1(5, 1, 0.2, output="sos")),
    ("eight sections (eight modes), 1024 taps", 1024, sps.butter(8, [0.05, 0.3], "bandpass", output="sos")),
    ("Butter [8, 30] / [3, 60] Hz at 500 Hz (SURVEY 8d's class-API cfg-3), 1024 taps", 1024,
     sps.butter(6, [8 / 250, 30 / 250], "bandpass", output="sos")),
    ("a cascade alone whose left tail takes eight rows (blocks of 24)", 2, sps.cheby1(5, 1, 0.2, output="sos")),
]


@pytest.mark.parametrize("name,ntaps,sos", ZPN_CASES, ids=[c[0] for c in ZPN_CASES])
def test_single_block_tables_and_block_algorithm(exe, name, ntaps, sos):
    """The same chain with ONE real block of 8192 samples per transform (odd frequencies,
    negacyclic wrap): tables of spec::build_zpn through the NumPy restatement of
    chain_zpn_kernel against scipy, every kind of chunk."""
    taps = sps.firwin(ntaps, 0.2) if ntaps > 2 else np.array([1.0, 0.0])
    T = tables_zp(exe, taps, sos, mode="zpn")
    assert T["eligible"], name
    NB, R, Rf = T["NR"], T["R"], T["Rf"]
    assert 24 <= NB <= min((7937 - ntaps) // 256, 30) and 1 <= Rf <= R <= min(32 - NB, 8) and 32 - NB + Rf <= NB
    assert T["NS"] in (2, 4, 6) and T["NS"] <= T["NM"] <= 8
    wq = 2 * np.pi * (np.arange(N) + 0.25) / N
    _, h = sps.sosfreqz(sos, worN=wq)
    Hq = np.polyval(taps[::-1], np.exp(-1j * wq)) * np.abs(h) ** 2 / N
    H = T["H"].reshape(N, 2)
    assert np.max(np.abs(H[:, 0] + 1j * H[:, 1] - Hq)) < 1e-12 * np.max(np.abs(Hq))   # (NumPy's polyval is the weak side)
    m = ModelZpn(T)
    S, L = m.S, m.L
    rng = np.random.default_rng(ntaps)
    lens = [S * 5 + 1024, S * 4, S * 3 + S - 17, S + 5, S * 3 + S, S * 3 + 300, 2 * S]
    x = rng.standard_normal(sum(lens))
    u = np.convolve(x, taps)
    zi0 = sps.sosfilt_zi(sos) * u[0]
    f, _ = sps.sosfilt(sos, u, zi=zi0)
    ref = sps.sosfilt(sos, np.concatenate([f, np.zeros(32768)])[::-1])[::-1][:len(f)]
    zir = sps.sosfilt(sos, np.zeros(7680), zi=zi0)[0]
    carry, held = sps.sosfilt(sos, zir[::-1])[::-1], np.zeros(L)
    out, o = [], 0
    for k, n in enumerate(lens):
        y, carry, held = m.chunk(x[o:o + n], carry, held, nruns=[1, 2, 3, 1, 2, 2, 1][k])
        out.append(y)
        o += n
    got = np.concatenate(out)
    assert np.isfinite(got).all()
    # the fit's conditioning sets the floor: the transform's rounding reaches the amplitudes
    # multiplied by about 1 / ratio (spec::build_zpn admits ratio > 3e-7)
    tol = max(3e-12, 1e-16 / T["ratio"])
    assert tol < 4e-10 and np.max(np.abs(got[L:] - ref[:len(got) - L])) < tol * np.max(np.abs(ref)), name


# ------------------------------------- the forward chain on one real block per transform
class ModelSpecN:
    """FIR -> sosfilt (no backward pass) with the tables of spec::build_specn: the causal half of
    ModelZpn -- the right tail wraps with its sign changed and is added back in the window,
    continues into the next block from the previous block's amplitudes; no lag, nothing held."""

    def __init__(self, T):
        self.NB, self.NM, self.NS, self.Rf, self.nh = T["NR"], T["NM"], T["NS"], T["Rf"], T["nh"]
        self.S, self.D = 256 * self.NB, 32 - self.NB
        H = T["H"].reshape(N, 2)
        self.Hq = (H[:, 0] + 1j * H[:, 1]) * N
        M = T["M"].reshape(2 * self.NS, 2 * self.nh)
        self.Mmu = M[:self.NS] + 1j * M[self.NS:]
        P = T["P"].reshape(20, self.NM, 2)
        P = P[..., 0] + 1j * P[..., 1]
        t = np.arange(256)
        self.P = P[t >> 5] * P[8 + ((t >> 2) & 7)] * P[16 + (t & 3)]
        Lr = T["L"].reshape(-1, self.NM, 2)
        self.Lr = Lr[..., 0] + 1j * Lr[..., 1]
        self.lsel = np.concatenate([np.arange(self.nh), np.arange(256 - self.nh, 256)])
        self.tw = np.exp(-1j * np.pi * np.arange(N) / MW)

    def window(self, x):
        buf = np.zeros(MW)
        buf[:len(x)] = x
        z = (buf[:N] - 1j * buf[N:]) * self.tw
        w = np.fft.ifft(np.fft.fft(z) * self.Hq) * np.conj(self.tw)
        return np.concatenate([w.real, -w.imag])

    def burst(self, amp, e):
        ok = (e >= 0) & (e < 256 * self.Rf)
        ee = np.where(ok, e, 0)
        nq = len(amp)
        return np.where(ok, np.real(((self.Lr[ee >> 8, :nq] * self.P[ee & 255, :nq]) * amp).sum(-1)), 0.0)

    def chunk(self, x, carry_in, nruns):
        n, S, NB, D, Rf = len(x), self.S, self.NB, self.D, self.Rf
        W = (n - 1) // S
        assert W >= 1
        lc = n - W * S
        f = np.full(n, np.nan)
        t = np.arange(256)
        nruns = max(1, min(nruns, W))
        carry_out = None
        for run in range(nruns):
            p0, p1 = run * W // nruns, (run + 1) * W // nruns
            cr, mu_p = np.zeros((D, 256)), np.zeros(self.NS, complex)
            blocks = list(range(p0 if run == 0 else p0 - 1, p1)) + ([W] if run == nruns - 1 else [])
            for p in blocks:
                o = p * S
                la = S if p < W else lc
                win = self.window(x[o:o + la])
                mu = self.Mmu @ win[MW - 256 + self.lsel]
                Y = win.reshape(32, 256).copy()
                Y[:D] += cr
                for r in range(Rf):
                    Y[r] += self.burst(mu, 256 * r + t)
                    Y[D + r] += self.burst(mu_p, 256 * r + t)
                if p == 0:
                    ci = np.zeros(MW)
                    m = min(len(carry_in), MW)
                    ci[:m] = carry_in[:m]
                    Y += ci.reshape(32, 256)
                if p < W:
                    if p >= p0:
                        f[o:o + S] = Y[:NB].ravel()
                    cr, mu_p = Y[NB:].copy(), mu
                else:
                    flat = Y.ravel()
                    f[o:n] = flat[:lc]
                    k = np.arange(7680)
                    src = lc + k
                    carry_out = np.where(src < MW, flat[np.minimum(src, MW - 1)], self.burst(mu, src - MW))
        return f, carry_out


SPECN_CASES = CASES + [("eight sections, 1024 taps", 1024, sps.butter(8, [0.05, 0.3], "bandpass", output="sos"))]


@pytest.mark.parametrize("name,ntaps,sos", SPECN_CASES, ids=[c[0] for c in SPECN_CASES])
def test_forward_single_block_tables_and_block_algorithm(exe, name, ntaps, sos):
    """FIR -> forward cascade with one real block of 8192 samples per transform: the tables of
    spec::build_specn through the NumPy restatement of the kernel against scipy's
    sosfilt(convolve(x, h)), every kind of chunk, the carry across chunks and as the flush."""
    taps = sps.firwin(ntaps, 0.2)
    T = tables_zp(exe, taps, sos, mode="specn")
    assert T["eligible"], name
    NB, Rf = T["NR"], T["Rf"]
    assert 24 <= NB <= min((7937 - ntaps) // 256, 30) and 1 <= Rf <= 5 and 32 - NB + Rf <= NB
    wq = 2 * np.pi * (np.arange(N) + 0.25) / N
    _, h = sps.sosfreqz(sos, worN=wq)
    Hq = np.polyval(taps[::-1], np.exp(-1j * wq)) * h / N
    H = T["H"].reshape(N, 2)
    assert np.max(np.abs(H[:, 0] + 1j * H[:, 1] - Hq)) < 1e-12 * np.max(np.abs(Hq))
    m = ModelSpecN(T)
    S = m.S
    rng = np.random.default_rng(len(taps))
    lens = [S * 5 + 1024, S * 4, S * 3 + S - 17, S + 5, S * 3 + S, 2 * S]
    x = rng.standard_normal(sum(lens))
    u = np.convolve(x, taps)
    zi0 = sps.sosfilt_zi(sos) * u[0]
    ref, _ = sps.sosfilt(sos, u, zi=zi0)
    carry = np.zeros(7680)
    cl = 4096 + 256 * Rf
    carry[:cl] = sps.sosfilt(sos, np.zeros(cl), zi=zi0)[0]
    o, scale = 0, np.max(np.abs(ref))
    tol = max(1e-12, 1e-16 / T["ratio"])
    for k, n in enumerate(lens):
        f, carry = m.chunk(x[o:o + n], carry, nruns=[1, 2, 3, 1, 2, 1][k])
        assert np.max(np.abs(f - ref[o:o + n])) < tol * scale, (name, k)
        o += n
    assert np.max(np.abs(carry[:ntaps - 1] - ref[o:o + ntaps - 1])) < tol * scale
