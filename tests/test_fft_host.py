"""CPU replay of the 256-thread 4096-point FFT used by the FIR and spectra
kernels (openseize_amd/csrc/fft4096.h): the phase functions are
__host__ __device__, so the index algebra, twiddles and LDS slot maps are
checked here without a GPU (g++ build of tests/host/fft_host_check.cpp)."""

import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft4096_host_replay():
    src = os.path.join(ROOT, "tests", "host", "fft_host_check.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "fft_host_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout


def test_fft4096_cube2_host_replay():
    """fft::cube2, the layout of the spectral chain kernels: bijective, conflict-free views, the
    second exchange inside a 16-lane row, forward / inverse against a direct DFT."""
    src = os.path.join(ROOT, "tests", "host", "fft_cube2_check.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "fft_cube2_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "exchange 2 row-local: yes" in out.stdout and "OK" in out.stdout


def test_fft4096_nega_host_replay():
    """fft::nega: one real block of 8192 samples on the 4096-point transform at the odd
    frequencies (the zero-phase chain kernel of chain_zpn.hip): forward against a direct DFT at
    2 pi (2 j + 1/2) / 8192, a filter applied bin by bin against the negacyclic convolution."""
    src = os.path.join(ROOT, "tests", "host", "fft_nega_check.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "fft_nega_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout


def test_fft8_host_replay():
    """fft8.h (N = 512 ... 8192, 8 points per thread): forward against a naive
    long-double DFT, inverse(forward) = N x, the digit-reversal map, the LDS
    swizzles are bijections and every stage access is bank-conflict free for the
    lane groups of MI355X_MICROARCH.md."""
    src = os.path.join(ROOT, "tests", "host", "fft8_host_check.cpp")
    inc = os.path.join(ROOT, "openseize_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "fft8_host_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", inc, src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("worst_bank_ways=1") == 5
