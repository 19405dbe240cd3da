/* abi_driver.c -- a host WITHOUT PyTorch or Python driving libosz_hip.so through
 * the C ABI of include/osz_hip.h, exactly as INTEGRATION.md's stub does:
 * osz_malloc -> osz_memcpy_h2d -> osz_sos_forward / osz_fir_push ->
 * osz_memcpy_d2h.  Reads a case file written by tests/test_gpu_boundary.py
 * (inputs + the reference's golden outputs), prints the largest errors.
 *
 *   gcc -std=c99 -O2 -I include tests/host/abi_driver.c -o abi_driver \
 *       -L openseize_amd/lib -losz_hip -Wl,-rpath,$PWD/openseize_amd/lib -lm
 *   ./abi_driver case.bin
 *
 * Case file (native endian): int64 nch, n, nsec, ntaps, chunk; then doubles
 * sos[nsec*6], taps[ntaps], x[nch*n], y_sos[nch*n], y_fir[nch*n] (the first n
 * samples of the full convolution).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "osz_hip.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_ != OSZ_OK) {                                                   \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, osz_last_error());   \
            return 2;                                                          \
        }                                                                      \
    } while (0)

static double *read_doubles(FILE *f, size_t n) {
    double *p = (double *)malloc(n * sizeof(double));
    if (!p || fread(p, sizeof(double), n, f) != n) {
        fprintf(stderr, "short case file\n");
        exit(3);
    }
    return p;
}

static double max_rel(const double *a, const double *b, size_t n) {
    double err = 0.0, scale = 1e-300;
    for (size_t i = 0; i < n; ++i) {
        const double d = fabs(a[i] - b[i]);
        if (d > err) err = d;
        if (fabs(b[i]) > scale) scale = fabs(b[i]);
    }
    return err / scale;
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int64_t hdr[5];
    if (fread(hdr, sizeof(int64_t), 5, f) != 5) return 3;
    const int64_t nch = hdr[0], n = hdr[1], nsec = hdr[2], ntaps = hdr[3], chunk = hdr[4];
    double *sos = read_doubles(f, (size_t)nsec * 6);
    double *taps = read_doubles(f, (size_t)ntaps);
    double *x = read_doubles(f, (size_t)(nch * n));
    double *want_sos = read_doubles(f, (size_t)(nch * n));
    double *want_fir = read_doubles(f, (size_t)(nch * n));
    fclose(f);

    int cus = 0;
    size_t hbm = 0;
    char name[128];
    CHECK(osz_device_info(&cus, &hbm, name, (int)sizeof name));
    printf("device: %s, %d CUs, %.0f GB\n", name, cus, (double)hbm / 1e9);

    const size_t bytes = (size_t)(nch * n) * sizeof(double);
    void *dx = NULL, *dy = NULL;
    CHECK(osz_malloc(&dx, bytes));
    CHECK(osz_malloc(&dy, bytes));
    CHECK(osz_memcpy_h2d(dx, x, bytes, NULL));
    double *got = (double *)malloc(bytes);

    /* sosfilt, chunk by chunk with the state carried in the handle
     * (reference core/numerical.py:332-335); rows have pitch n */
    osz_sos_t iir = NULL;
    CHECK(osz_sos_create(&iir, sos, (int)nsec, (int)nch));
    for (int64_t s = 0; s < n; s += chunk) {
        const int64_t m = n - s < chunk ? n - s : chunk;
        CHECK(osz_sos_forward(iir, (const double *)dx + s, n, (double *)dy + s, n, m, NULL));
    }
    CHECK(osz_memcpy_d2h(got, dy, bytes, NULL));
    CHECK(osz_stream_sync(NULL));
    const double e_sos = max_rel(got, want_sos, (size_t)(nch * n));
    double *zf = (double *)malloc((size_t)nsec * nch * 2 * sizeof(double));
    CHECK(osz_sos_get_state(iir, zf, NULL));
    CHECK(osz_sos_destroy(iir));

    /* overlap-add FIR: the stream of full-convolution samples
     * (reference core/numerical.py:229-298) */
    osz_fir_t fir = NULL;
    CHECK(osz_fir_create(&fir, taps, (int)ntaps, (int)nch));
    CHECK(osz_memset(dy, 0, bytes, NULL));
    for (int64_t s = 0; s < n; s += chunk) {
        const int64_t m = n - s < chunk ? n - s : chunk;
        CHECK(osz_fir_push(fir, (const double *)dx + s, n, m, (double *)dy + s, n, 0, NULL));
    }
    CHECK(osz_memcpy_d2h(got, dy, bytes, NULL));
    CHECK(osz_stream_sync(NULL));
    const double e_fir = max_rel(got, want_fir, (size_t)(nch * n));
    CHECK(osz_fir_destroy(fir));

    /* error path: a bad argument comes back as a status + message, no abort */
    osz_sos_t bad = NULL;
    const int rc = osz_sos_create(&bad, sos, 0, (int)nch);
    printf("max_rel_err_sos=%.3e max_rel_err_fir=%.3e bad_create_rc=%d\n", e_sos, e_fir, rc);
    CHECK(osz_free(dx));
    CHECK(osz_free(dy));
    return (e_sos < 1e-9 && e_fir < 1e-9 && rc == OSZ_ERR_INVALID) ? 0 : 4;
}
