// Builds the host tables of the spectral FIR -> cascade kernel (csrc/spec_tables.h)
// with g++ and writes them out for tests/test_spec_host.py.
//   spec_host_check <in.bin> <out.bin> [zp|zpn|specn]  (specn: the forward chain on one real block per transform)
//   spec_host_check <in.bin> <out.bin> [zp|zpn]  (zp: the two-sided tables of chain_zp.hip;
//                                                  zpn: those of chain_zpn.hip, one real block per transform)
// in:  int32 wlen, int32 nsec, int32 forgets, double taps[wlen], double sos[nsec][6]
// out: int32 eligible, NR, NM, nm, R, double fit_ratio, then H, M, P, L (each: int64 count, doubles)
#include <cstdint>
#include <cstdio>
#include <vector>

#include <cstdlib>
#include <cstring>

#include "spec_tables.h"

int main(int argc, char **argv) {
    if (argc != 3 && argc != 4) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[3];
    if (fread(hdr, sizeof(int32_t), 3, f) != 3) return 2;
    std::vector<double> taps(hdr[0]), sos((size_t)hdr[1] * 6);
    if (fread(taps.data(), sizeof(double), taps.size(), f) != taps.size()) return 2;
    if (fread(sos.data(), sizeof(double), sos.size(), f) != sos.size()) return 2;
    fclose(f);
    if (argc == 4) {
        const osz::spec::TablesZp T = !strcmp(argv[3], "specn")
                                          ? osz::spec::build_specn(taps.data(), hdr[0], sos.data(), hdr[1], hdr[2] != 0)
                                      : !strcmp(argv[3], "zpn")
                                          ? osz::spec::build_zpn(taps.data(), hdr[0], sos.data(), hdr[1], hdr[2] != 0, 15360 - 1024)
                                          : osz::spec::build_zp(taps.data(), hdr[0], sos.data(), hdr[1], hdr[2] != 0);
        f = fopen(argv[2], "wb");
        if (!f) return 2;
        const int32_t out[8] = {T.eligible, T.NR, T.NM, T.nm, T.R, T.nh, T.Rf, T.NS};
        fwrite(out, sizeof(int32_t), 8, f);
        fwrite(&T.fit_ratio, sizeof(double), 1, f);
        for (const std::vector<double> *v : {&T.H, &T.M, &T.P, &T.L}) {
            const int64_t n = (int64_t)v->size();
            fwrite(&n, sizeof(int64_t), 1, f);
            fwrite(v->data(), sizeof(double), v->size(), f);
        }
        fclose(f);
        return 0;
    }
    const osz::spec::Tables T = osz::spec::build(taps.data(), hdr[0], sos.data(), hdr[1], hdr[2] != 0);
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    const int32_t out[5] = {T.eligible, T.NR, T.NM, T.nm, T.R};
    fwrite(out, sizeof(int32_t), 5, f);
    fwrite(&T.fit_ratio, sizeof(double), 1, f);
    for (const std::vector<double> *v : {&T.H, &T.M, &T.P, &T.L}) {
        const int64_t n = (int64_t)v->size();
        fwrite(&n, sizeof(int64_t), 1, f);
        fwrite(v->data(), sizeof(double), v->size(), f);
    }
    fclose(f);
    return 0;
}
