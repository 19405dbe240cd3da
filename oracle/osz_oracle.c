/*
 * osz_oracle.c -- CPU restatement (plain C) of the scalar inner loops of the
 * reference hot path.  TEST INFRASTRUCTURE ONLY: nothing in the product path
 * (openseize_amd/) may link or call this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * Parity status: PINNED.  Each routine is checked in tests/test_oracle.py
 * against the golden vectors under tests/golden/ that were produced by
 * running the reference itself (tests/golden/make_golden.py).
 *
 * The reference delegates these loops to SciPy's compiled code, which is not
 * under /root/reference; the call sites restated here are cited per function.
 */
#include <stddef.h>
#include <stdint.h>

/*
 * Cascaded second-order sections, direct-form-II transposed, one channel.
 * Restates scipy.signal.sosfilt as the reference calls it at
 * src/openseize/core/numerical.py:334 (forward, carried zi), :399/:402/:410
 * (backward passes on flipped data).  Sections are the inner loop per sample,
 * which is SciPy's order, so results are bit-identical to SciPy's.
 *
 *   sos : nsec x 6  (b0 b1 b2 a0 a1 a2), a0 == 1
 *   zi  : nsec x 2  in/out (final state written back)
 *   x,y : n samples with element strides sx, sy (negative stride = flipped)
 */
void osz_ref_sosfilt(const double *sos, int nsec, const double *x, ptrdiff_t sx,
                     double *y, ptrdiff_t sy, long n, double *zi)
{
    for (long i = 0; i < n; ++i) {
        double v = x[i * sx];
        for (int s = 0; s < nsec; ++s) {
            const double *c = sos + 6 * s;
            double *z = zi + 2 * s;
            double out = c[0] * v + z[0];
            z[0] = c[1] * v - c[4] * out + z[1];
            z[1] = c[2] * v - c[5] * out;
            v = out;
        }
        y[i * sy] = v;
    }
}

/*
 * Full linear convolution of one channel with h (direct form).  This is what
 * oaconvolve (numerical.py:158-298) computes in exact arithmetic; used as the
 * size-independent reference for FFT-based results.  y has n + m - 1 samples.
 */
void osz_ref_convolve_full(const double *x, long n, const double *h, long m,
                           double *y)
{
    for (long i = 0; i < n + m - 1; ++i) {
        long k0 = i - (n - 1) > 0 ? i - (n - 1) : 0;
        long k1 = i < m - 1 ? i : m - 1;
        double acc = 0.0;
        for (long k = k0; k <= k1; ++k)
            acc += h[k] * x[i - k];
        y[i] = acc;
    }
}

/*
 * Rational resampling of one channel, global definition that
 * polyphase_resample (numerical.py:523-632) reproduces chunk by chunk through
 * scipy.signal.resample_poly(window=h):
 *     out[j] = sum_k L*h[k] * xup[j*M + half - k],  half = (len(h)-1)/2,
 * xup = x zero-stuffed by L, zeros outside, nout = ceil(n*L/M)
 * (resampling/resampling.py:91).
 */
void osz_ref_resample(const double *x, long n, const double *h, long m, int L,
                      int M, double *y, long nout)
{
    long half = (m - 1) / 2;
    for (long j = 0; j < nout; ++j) {
        long t = j * (long)M + half; /* index into the upsampled stream */
        double acc = 0.0;
        /* k runs over taps with (t - k) a multiple of L and 0 <= (t-k)/L < n */
        long k = t % L;
        for (; k < m; k += L) {
            long i = (t - k) / L;
            if (t - k < 0)
                break;
            if (i < n)
                acc += (double)L * h[k] * x[i];
        }
        y[j] = acc;
    }
}

/*
 * Direct-form-II-transposed filter of order K for one channel: restates
 * scipy.signal.lfilter as the reference calls it at core/numerical.py:445
 * (forward, carried zi) and :508/:511/:519 (backward passes of filtfilt).
 * b, a: K+1 coefficients each (a[0] divided out by the caller); z: K states.
 */
void osz_ref_lfilter(const double *b, const double *a, int K, const double *x,
                     ptrdiff_t sx, double *y, ptrdiff_t sy, long n, double *z)
{
    for (long i = 0; i < n; ++i) {
        const double v = x[i * sx];
        /* operand order as in SciPy's C loop: z[k] = z[k+1] + b x - a y */
        const double out = K > 0 ? z[0] + b[0] * v : b[0] * v;
        for (int k = 0; k + 1 < K; ++k)
            z[k] = z[k + 1] + b[k + 1] * v - a[k + 1] * out;
        if (K > 0)
            z[K - 1] = b[K] * v - a[K] * out;
        y[i * sy] = out;
    }
}
