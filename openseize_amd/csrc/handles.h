// handles.h -- the opaque handles of the C ABI as the library's translation
// units see them (fir.hip, sos.hip own them; chain.hip drives both).
#pragma once

#include <cstdint>
#include <functional>
#include <map>
#include <vector>

#include "fft4096.h"
#include "sos_tile.h"

namespace osz {
struct ChainSpec;   // chain_spec.hip: a FIR and a SOS handle that run the spectral kernel together
struct ChainZp;     // chain_zp.hip: the same pair on the zero-phase kernel
}

struct FirPart {
    int ntaps, step;
    double *dH;         // [4096][2]
    double *dHn;        // fir_nega_kernel: the response at the quarter-shifted bins, [r][t] (fft::cube2::bin), or null
    int stepn;          // its block length: 256 rows-per-block
    double *dstate[2];  // ping-pong carried tails [nch][ntaps-1]
    int cur;
};

struct osz_fir_s {
    int device;         // HIP device the handle's buffers live on
    int ntaps, nch;
    std::vector<FirPart> parts;
    double *dtails;     // run-tail workspace [nch][nruns_cap][<= 2048]
    int64_t tails_cap;  // doubles
    double *dD;         // partitioned: deferred sums [nch][(P-1)*kFirPart]
    int64_t dlen;
    double *dW;         // partitioned: work rows
    int64_t w_cap;      // doubles
    osz::fft::Tables tb;
    std::vector<double> htaps;   // host copy of the taps
    osz::ChainSpec *spec;        // set while the handle's stream runs in the spectral chain kernel
    osz::ChainZp *zp;            // tables and carried sequences of the zero-phase kernel
};


struct osz_sos_s {
    int device;         // HIP device the handle's buffers live on
    int T, NW;          // kernel geometry: samples per lane, waves per workgroup
    int64_t warm_len;   // samples of the sosfiltfilt warm-up that matter (see sos_warmup_len)
    int nsec, nch;
    osz::SosSection *dsec;   // device, built for T samples per lane
    osz::SosSection *dsec_t[33];   // tables for other tile geometries (index = T), lazily built
    double *dtab2_t[33];           // lane tables A^(T k) for other T ([nsec][4][66]), lazily built
    double coef[osz::kSosMaxSec * 6];   // host copy of the sections (b0 b1 b2 1 a1 a2)
    double *dstate;     // device (nsec, nch, 2): carried forward state (current)
    double *dstate_alt; // the other half of the ping-pong: a forward pass reads dstate and
                        // writes dstate_alt (its time segments are separate workgroups with
                        // no ordering between them), then the two swap
    double *dtmp;       // device (nsec, nch, 2): warm-up state of sosfiltfilt
    double *dcarry;     // device (nsec, nch, 2): state between the main and remainder launches
    double *dzi;        // device (nsec, 2): sosfilt_zi of this cascade
    int touch;          // tuning knob OSZ_SOS_TOUCH: touch-prefetch of the next tile
    double *dtab2;      // device [nsec][4][66]: A^(T k) per section for sos_body2, or null
    // osz_chain_step: the backward pass runs beside the fused forward kernel on a
    // stream of the handle's own, with scratch of its own (created on first use)
    hipStream_t side;
    hipEvent_t side_go, side_done[2];     // done: ping-pong, side_cur names the last recorded
    int side_cur;
    double *dtmp_side, *dcarry_side;
    bool side_busy;                       // a deferred backward pass may still be running
    const double *side_in[2];             // fa, fb of that pass: (nch, n) views with row pitch ld
    int64_t side_ld[2], side_n[2];
    osz::ChainSpec *spec;                 // see osz_fir_s
    osz::ChainZp *zp;
    double zp_tol;                        // osz_chain_zp_tolerance: where the bursts are cut (0: the default)
};

namespace osz {
// the NaN reach of the reference is reproduced (always; the A/B knob of round 2 is gone)
bool sos_nanfix();
// NaN reach of a forward pass cut into runs (sos.hip): a small launch behind the pass
int sos_seal_launch(double *y, int64_t ldy, int64_t n, int nseg, int64_t A, int64_t B, int64_t C, double *state,
                    int nsec, int nch, double *carry, int64_t ldcarry, int64_t ncarry, hipStream_t st);
// process-wide twiddle tables on the device (fir.hip)
int get_fft_tables(fft::Tables &out);
// per-section tables of a cascade for a tile of T samples per lane, built on
// first use and owned by the handle (sos.hip)
int sos_tables_for(osz_sos_s *h, int T, const SosSection **dsec);
// per-lane scan matrices A^(T k) ([nsec][4][kSos2Tab]) for a tile of T samples per
// lane (sos_tile_full2), built on first use and owned by the handle; *dtab = null
// when the cascade has more sections than the LDS table holds
int sos_lane_table_for(osz_sos_s *h, int T, const double **dtab);
// osz_sosfiltfilt_chunk with the warm-up state and the main/remainder carry in
// caller-named scratch ((nsec, nch, 2) doubles each): two of them may be in flight
// on two streams (sos.hip)
int sosfiltfilt_chunk_on(osz_sos_s *h, const double *fa, int64_t ldfa, int64_t na, const double *fb,
                         int64_t ldfb, int64_t nb, double *y, int64_t ldy, double *tmp, double *carry,
                         hipStream_t st);
// ---- chain_spec.hip: FIR -> forward cascade in the spectrum.  While a pair of handles
// runs it, their own states (overlap tail, section states) are not kept up to date:
// every entry point that reads them calls spec_settle first, every one that changes
// them spec_touch.
int spec_route(osz_fir_s *fir, osz_sos_s *sos, hipStream_t st, int *route);
int spec_try_forward(osz_fir_s *fir, osz_sos_s *sos, const double *x, int64_t ldx, int64_t n, double *f,
                     int64_t ldf, hipStream_t st, const std::function<int()> &between, bool *taken);
int spec_settle(ChainSpec *s, hipStream_t st);
int spec_touch(ChainSpec *s, hipStream_t st);
void spec_unlink(ChainSpec *s);
// the entry points osz_fir_push / osz_sos_forward without those calls (fir.hip, sos.hip)
int fir_push_raw(osz_fir_s *h, const double *x, int64_t ldx, int64_t n, double *y, int64_t ldy,
                 int64_t skip, hipStream_t st);
int sos_forward_raw(osz_sos_s *h, const double *x, int64_t ldx, double *y, int64_t ldy, int64_t n,
                    hipStream_t st);
// a backward pass over (nch, n) from the section states in `state` ((nsec, nch, 2), device);
// the handle's own state is not touched (sos.hip)
int sos_backward_raw(osz_sos_s *h, const double *x, int64_t ldx, double *y, int64_t ldy, int64_t n,
                     const double *state, hipStream_t st);
void zp_unlink(ChainZp *s);
}  // namespace osz
