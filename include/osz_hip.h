/*
 * osz_hip.h -- C ABI of libosz_hip.so: the MI355X (gfx950) implementation of
 * Openseize's chunked DSP hot path.
 *
 * The reference (mscaudill/openseize, 100 % Python) has no FFI of its own: its
 * hot loops are calls into SciPy/NumPy compiled code from
 * src/openseize/core/numerical.py.  Each entry point below replaces one of
 * those call sites (cited per function) and is what a ctypes binding inside
 * the reference would bind -- see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - plain C, no exceptions, no ownership crossing the ABI;
 *   - every function returns 0 on success and a negative osz_status otherwise;
 *     osz_last_error() gives a thread-local message;
 *   - all signal data is float64, device resident, laid out (channels, samples)
 *     row-major with a row pitch `ld` given in ELEMENTS.  Complex outputs are
 *     interleaved (re, im) float64 pairs; their pitch is in complex elements;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls
 *     only enqueue work unless documented otherwise;
 *   - one opaque handle per ITERATOR (the reference's generators each own
 *     their carried state); handles are not thread-safe.
 */
#ifndef OSZ_HIP_H
#define OSZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    OSZ_OK = 0,
    OSZ_ERR_INVALID = -1,   /* bad argument            -> ValueError   */
    OSZ_ERR_HIP = -2,       /* HIP / rocFFT failure    -> RuntimeError */
    OSZ_ERR_NOMEM = -3,     /* allocation failure      -> MemoryError  */
    OSZ_ERR_STATE = -4,     /* call order violated     -> RuntimeError */
    OSZ_ERR_UNSUPPORTED = -5
} osz_status;

typedef struct osz_sos_s *osz_sos_t;
typedef struct osz_fir_s *osz_fir_t;
typedef struct osz_poly_s *osz_poly_t;
typedef struct osz_spec_s *osz_spec_t;

/* ---- library / device ------------------------------------------------- */
int osz_version(void);
const char *osz_last_error(void);
/* name may be NULL; reports the current HIP device. */
int osz_device_info(int *cu_count, size_t *hbm_bytes, char *name, int name_len);

/* Memory and stream helpers so that a host without PyTorch can drive the
 * library (the Python host uses torch tensors' data_ptr() instead). */
int osz_malloc(void **dptr, size_t bytes);
int osz_free(void *dptr);
int osz_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int osz_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int osz_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
int osz_memset(void *dst, int value, size_t bytes, void *stream);
int osz_stream_sync(void *stream);
/* HIP-event timing on `stream` (torch.cuda.Event only sees torch's stream). */
int osz_event_create(void **ev);
int osz_event_destroy(void *ev);
int osz_event_record(void *ev, void *stream);
int osz_event_elapsed_ms(void *start, void *stop, float *ms); /* syncs stop */

/* Per-kernel timing with HIP events on the launch stream.  While enabled every
 * kernel launch of the library is bracketed by two events; query syncs and
 * returns the launches and summed milliseconds recorded under `name`
 * ("fir_oa", "fir_seam", "sos_fwd", "sos_warmup", "sos_bwd", "poly",
 * "spec_prep", "spec_rocfft", "spec_post"). */
int osz_profile_enable(int on);
int osz_profile_reset(void);
int osz_profile_query(const char *name, int64_t *launches, double *total_ms);

/* ---- K2/K3: cascaded second-order sections ---------------------------- */
/*
 * Replaces scipy.signal.sosfilt as called by the reference at
 *   core/numerical.py:334        (sosfilt: forward, zi carried per chunk)
 *   core/numerical.py:399,402,410 (sosfiltfilt: backward passes on flips)
 * sos: nsec x 6 host doubles (b0 b1 b2 a0 a1 a2; a0 is divided out).
 * The handle owns the carried state z (nsec, nch, 2) on the device.
 */
int osz_sos_create(osz_sos_t *h, const double *sos, int nsec, int nch);
int osz_sos_destroy(osz_sos_t h);
/* zi/zf host arrays laid out (nsec, nch, 2) like the reference's zi argument
 * (core/numerical.py:313-329).  zi == NULL zeroes the state.  Synchronous -- or, handed a
 * DEVICE array of the same layout, a copy ordered on `stream` that nothing waits for (a
 * snapshot the caller may never need: core/numerical.py's zero-phase stream, before its end). */
int osz_sos_set_state(osz_sos_t h, const double *zi, void *stream);
int osz_sos_get_state(osz_sos_t h, double *zf, void *stream);
/* zi_unit (nsec, 2) is the steady-state unit-step state the reference gets from
 * scipy.signal.sosfilt_zi at core/numerical.py:378.  The library derives it
 * from the coefficients at create time; this call overrides it (host doubles).
 * Synchronous. */
int osz_sos_set_zi_unit(osz_sos_t h, const double *zi_unit);
/* state[s, c, :] = zi_unit[s, :] * x[c, col]  -- the steady-state start the
 * reference builds at core/numerical.py:374-386 (zi * x0). x: device. */
int osz_sos_set_state_scaled(osz_sos_t h, const double *x, int64_t ldx,
                             int64_t col, void *stream);
/* y[c, 0:n] = cascade(x[c, 0:n]) continuing from the carried state, which is
 * advanced (core/numerical.py:332-335). x == y (in place) is allowed. */
int osz_sos_forward(osz_sos_t h, const double *x, int64_t ldx, double *y,
                    int64_t ldy, int64_t n, void *stream);
/*
 * Backward sweep of sosfiltfilt for one chunk (core/numerical.py:390-411).
 *   fa (nch, na): forward-filtered chunk i;  fb (nch, nb): forward-filtered
 *   chunk i+1 or NULL for the last chunk.
 * fb != NULL: warm-up = back-filter fb from zi_unit * fb[:, nb-1], keep only
 * its final state (:397-399), then back-filter fa from that state (:401-403).
 * fb == NULL: back-filter fa from zi_unit * fa[:, na-1] (:408-411).
 * y (nch, na) receives the result in natural sample order. The handle's
 * carried forward state is not touched.
 */
/* Samples of forward chunk i+1 (counted from its start) that the chunk-local
 * backward warm-up of chunk i reads: beyond them the state has forgotten its start
 * to below 1e-16 (1 << 62: the whole chunk).  fb / nb of the calls below may be cut
 * to this length. */
int64_t osz_sos_warm_len(osz_sos_t h);

/* The warm-up over fb stops once further samples cannot change a float64
 * state: after warmup_len samples, the smallest multiple of the kernel tile
 * with ||M^len||_inf < 1e-18 (M: state-transition matrix of the cascade).
 * For slowly decaying cascades warmup_len exceeds the chunk and the whole of
 * fb is used, exactly as the reference does.  set(0) forces the full chunk. */
int64_t osz_sos_warmup_len(osz_sos_t h);
int osz_sos_set_warmup_len(osz_sos_t h, int64_t len);
int osz_sosfiltfilt_chunk(osz_sos_t h, const double *fa, int64_t ldfa,
                          int64_t na, const double *fb, int64_t ldfb,
                          int64_t nb, double *y, int64_t ldy, void *stream);

/*
 * One steady-state step of sosfiltfilt in a single launch: forward-filter the
 * incoming chunk x (nch, nx) -> f (advancing the carried state) AND back-filter
 * an earlier forward chunk fa with the warm-up over fb (as
 * osz_sosfiltfilt_chunk) -> y.  The two passes are independent and share the
 * GPU (two workgroups per CU).  Equivalent to osz_sos_forward followed by
 * osz_sosfiltfilt_chunk, which it falls back to for ragged chunk lengths.
 */
int osz_sosfiltfilt_step(osz_sos_t h, const double *x, int64_t ldx, int64_t nx,
                         double *f, int64_t ldf, const double *fa, int64_t ldfa,
                         int64_t na, const double *fb, int64_t ldfb, int64_t nb,
                         double *y, int64_t ldy, void *stream);

/* ---- K1: streaming FFT overlap-add FIR -------------------------------- */
/*
 * Replaces the np.fft.rfft / irfft circular convolution and overlap add of
 * oaconvolve (core/numerical.py:229-251, hot loop :254-298).  The handle
 * carries the (nch, ntaps-1) overlap tail.  The stream of FULL linear
 * convolution samples is produced; boundary modes (:143-150) are applied by
 * the caller through `skip` (left cut) and by trimming the flush.
 */
int osz_fir_create(osz_fir_t *h, const double *taps, int ntaps, int nch);
int osz_fir_destroy(osz_fir_t h);
int osz_fir_reset(osz_fir_t h, void *stream);
/* Checkpoint / resume (the de-facto state of the reference's
 * generator is the overlap tail, core/numerical.py:223, :269): the carried
 * tail(s) as osz_fir_state_size() host doubles.  Synchronous; with a device array instead,
 * ordered on `stream` and not waited for (as osz_sos_get_state). */
int64_t osz_fir_state_size(osz_fir_t h);
int osz_fir_get_state(osz_fir_t h, double *state, void *stream);
int osz_fir_set_state(osz_fir_t h, const double *state, void *stream);
/* Consumes x (nch, n) and writes the n next samples of the full convolution,
 * minus the first `skip` of them, to y (nch, n - skip); 0 <= skip <= n. */
int osz_fir_push(osz_fir_t h, const double *x, int64_t ldx, int64_t n,
                 double *y, int64_t ldy, int64_t skip, void *stream);
/* Writes tail[skip : ntaps-1-drop] (the samples after the last input) to y.
 * Does not modify the carried tail. */
int osz_fir_flush(osz_fir_t h, double *y, int64_t ldy, int64_t skip,
                  int64_t drop, void *stream);

/* ---- K1 + K2 fused: FIR feeding the forward pass of a cascade --------- */
/*
 * The forward half of the chain the reference builds from two generators --
 * oaconvolve (core/numerical.py:158-298, 'full' stream, no left cut) into
 * sosfilt / the forward pass of sosfiltfilt (:301-335, :374-386) -- in one
 * kernel: f = sosfilt(fir(x)) for the next n samples of every channel, both
 * handles' carried states advanced exactly as osz_fir_push(skip = 0) followed
 * by osz_sos_forward would.  The FIR output never reaches HBM (16 instead of
 * 32 B per channel-sample).  Which kernel takes the chunk: osz_chain_forward_route
 * below.  What no fused kernel takes (partitioned FIRs, chunks of fewer than four
 * block pairs on route 0, the ragged head of a chunk there) runs through the
 * separate kernels inside the call: same results, same carried states.
 */
int osz_chain_forward(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx,
                      int64_t n, double *f, int64_t ldf, void *stream);
/* Which kernel osz_chain_forward runs the whole blocks of this pair of handles on:
 * 2 = FIR and cascade as one multiplication per bin, one 8192-sample block per 4096-point
 * transform (chain_zpn_kernel<.., ZP = false>); 1 = the same with a pair of blocks per
 * transform (chain_spec_kernel: cascades the first does not take); 0 = the FIR in the
 * spectrum and the cascade as a scan in time (chain_kernel); -1 on error.  Builds the
 * pair's tables on first use (cached on the handles); a handle paired otherwise before
 * gets its own carried state back on `stream`.  Diagnostic: tests and coverage tables. */
int osz_chain_forward_route(osz_fir_t fir, osz_sos_t sos, void *stream);

/*
 * One steady-state step of the FIR -> sosfiltfilt chain: osz_chain_forward of the
 * incoming chunk x -> f on `stream` and, BESIDE it on a stream of the SOS handle's
 * own, osz_sosfiltfilt_chunk of an earlier forward chunk fa (warmed up over fb, or
 * NULL) -> y.  The backward pass starts behind everything queued on `stream` before
 * the call.  f must not alias fa or fb.
 *   flags = 0: `stream` is ordered behind the backward pass when the call returns --
 *     one stream-ordered operation for the caller.
 *   flags = OSZ_CHAIN_DEFER: the backward pass is left running: y, and the right to
 *     overwrite fa / fb, are the caller's only after the NEXT osz_chain_step or an
 *     osz_chain_wait on this handle has been queued on `stream`.  The next step
 *     waits for the pass only if its f overlaps fa or fb (a ring of four forward
 *     buffers never does).
 * Replaces, for a FIR producer feeding sosfiltfilt, core/numerical.py:158-298 +
 * :374-411 of the reference.
 */
#define OSZ_CHAIN_DEFER 1
int osz_chain_step(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx,
                   int64_t n, double *f, int64_t ldf, const double *fa,
                   int64_t ldfa, int64_t na, const double *fb, int64_t ldfb,
                   int64_t nb, double *y, int64_t ldy, int flags, void *stream);

/* `stream` is ordered behind the deferred backward pass of the last osz_chain_step. */
int osz_chain_wait(osz_sos_t sos, void *stream);

/* ---- K1 + K2 + K3 as one multiplication per bin: FIR -> sosfiltfilt ----- */
/*
 * The whole chain the reference builds from oaconvolve (core/numerical.py:158-298) and
 * sosfiltfilt (:338-411), for streams whose chunks are much longer than the cascade's
 * memory (osz_sos_warm_len): there the chunk-local backward pass of the reference equals,
 * to 1e-18, the backward pass over the whole rest of the stream, and FIR, forward and
 * backward cascade are the zero-phase filter H_fir |H_iir|^2 -- one spectrum multiply in
 * the FIR's 4096-point transform plus the cascade's ringing on both sides of each block,
 * put back by mode bursts (csrc/chain_zp.hip).  The forward stream never exists: 8 B read
 * and 8 B written per channel-sample for what SURVEY 8d books at 48.
 *
 * The output stream runs osz_chain_zp_lag() samples behind the input: a block's backward
 * ringing reaches that far into what the previous block has produced.  The two ends of a
 * stream (where the reference's start state sosfilt_zi * x[0] and its last chunk's
 * backward start matter) are the caller's, with the separate kernels: open after the
 * start state has been set on the SOS handle, finish before the last chunks.
 *
 *   osz_chain_zp_lag        samples of delay (a multiple of 256), or -1 when the pair
 *                           of filters does not take this kernel (poles that repeat,
 *                           ringing longer than the transform's guard rows -- twelve rows
 *                           of 256 samples at most --, a partitioned FIR, i.e. one of more
 *                           than 2049 taps)
 *   osz_chain_zp_tolerance  where the bursts are cut, relative to the norm of the composite
 *                           impulse response (0: the default, 1e-15; before osz_chain_zp_lag /
 *                           _open; also the cut of osz_chain_forward's spectral kernels for
 *                           this cascade).  What is cut off scales with the INPUT's magnitude,
 *                           about 0.3 tol max|x|: the default keeps it at float64's own
 *                           rounding on any input; a caller whose streams are zero-mean may
 *                           relax it (1e-12: one burst row less each way, 2 % less time, 6e-13
 *                           of the output scale on such data).
 *   osz_chain_zp_reach      how far a non-finite INPUT sample reaches in the reference's FIR: it works
 *                           in segments of `step` input samples, one FFT each (core/numerical.py:
 *                           202-217, 258-283), so the forward stream is bad from the start of the
 *                           segment that holds the sample -- osz_chain_zp_seal then counts chunks
 *                           from there (0, the default: from the sample itself; the kernels record
 *                           the exact sample)
 *   osz_chain_zp_min_chunk  shortest chunk osz_chain_zp_step takes (two blocks)
 *   osz_chain_zp_open       starts a stream at sample 0: the FIR's overlap tail must be
 *                           zero; the forward cascade starts from the state on the SOS
 *                           handle (osz_sos_set_state*) at stream sample `skip` -- the
 *                           FIR's left cut, which the cascade never sees -- and that state
 *                           is consumed
 *   osz_chain_zp_step       the next n input samples; the n output samples
 *                           [pos - lag, pos + n - lag) of the stream (pos: samples stepped
 *                           before; the first lag outputs of a stream mean nothing) go to
 *                           y0[0, n0) and y[0, n - n0): a caller that cuts the output into
 *                           chunks of its own passes the tail of the previous chunk and
 *                           the head of the current one (n0 = 0: everything to y).
 *                           Non-finite input: the step records where the forward stream
 *                           went bad, every later step writes NaN only, and
 *                           osz_chain_zp_seal settles the chunk it happened in and the one
 *                           before -- outputs are final once sealed
 *   osz_chain_zp_seal       NaN reach of sosfiltfilt: y holds output samples [s0, s0 + n)
 *                           of a stream cut into chunks of cs samples from `origin` on; a
 *                           chunk is NaN as a whole when the forward stream went bad in it
 *                           or in the chunk after it (positions the steps so far have seen)
 *   osz_chain_zp_finish     ends the zero-phase part: the handles' own states (FIR overlap
 *                           tail, forward section states) become those at the end of the
 *                           samples stepped -- osz_fir_push / osz_sos_forward /
 *                           osz_chain_forward continue the forward stream from there --
 *                           and y[0, ny) receives the next ny output samples, for which
 *                           the first m samples of what follows are read (m >= min_chunk,
 *                           ny <= m - min_chunk / 2: the last pair of blocks of those m
 *                           samples is not what the continued stream would have there)
 */
int64_t osz_chain_zp_lag(osz_fir_t fir, osz_sos_t sos);
int osz_chain_zp_tolerance(osz_fir_t fir, osz_sos_t sos, double tol);
int osz_chain_zp_reach(osz_fir_t fir, osz_sos_t sos, int64_t step);
int64_t osz_chain_zp_min_chunk(osz_fir_t fir, osz_sos_t sos);
int osz_chain_zp_open(osz_fir_t fir, osz_sos_t sos, int64_t skip, void *stream);
int osz_chain_zp_step(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx,
                      int64_t n, double *y0, int64_t ldy0, int64_t n0, double *y,
                      int64_t ldy, void *stream);
int osz_chain_zp_seal(osz_fir_t fir, osz_sos_t sos, double *y, int64_t ldy, int64_t n,
                      int64_t s0, int64_t origin, int64_t cs, void *stream);
int osz_chain_zp_finish(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx,
                        int64_t m, double *y, int64_t ldy, int64_t ny, void *stream);

/* ---- K4: polyphase rational resampler --------------------------------- */
/*
 * Replaces scipy.signal.resample_poly(padded, L, M, window=h) as called at
 * core/numerical.py:610,631 together with the overhang bookkeeping :590-632.
 * Output sample j of the whole stream is
 *    sum_k L*h[k] * xup[j*M + half - k],  half = (ntaps-1)/2
 * The handle carries the input history it still needs.
 */
int osz_poly_create(osz_poly_t *h, const double *taps, int ntaps, int L, int M,
                    int nch);
/* The same with the tap that lines up with an output named (`half` above = centre): what
 * resample_poly does to the window before it filters -- zeros in front (n_pre_pad) and behind
 * (n_post_pad, every phase to one count of taps; scipy/signal/_signaltools.py, resample_poly) --
 * handed in as taps.  The kernels multiply every tap they are given, zero-valued ones included,
 * and no other: a non-finite sample is lost to exactly the outputs SciPy (and so the reference,
 * core/numerical.py:610,631) loses it to.  osz_poly_create(taps, n) = ..._centred(taps, n, (n-1)/2). */
int osz_poly_create_centred(osz_poly_t *h, const double *taps, int ntaps, int centre, int L, int M,
                            int nch);
int osz_poly_destroy(osz_poly_t h);
int osz_poly_reset(osz_poly_t h, void *stream);
/* Checkpoint / resume: [samples consumed, samples produced, input history]
 * as osz_poly_state_size() host doubles.  Synchronous. */
int64_t osz_poly_state_size(osz_poly_t h);
int osz_poly_get_state(osz_poly_t h, double *state, void *stream);
int osz_poly_set_state(osz_poly_t h, const double *state, void *stream);
/* Number of output samples a push of n more input samples will produce
 * (final != 0: the stream ends with this push, total = ceil(N*L/M)). */
int64_t osz_poly_out_count(osz_poly_t h, int64_t n, int final);
int osz_poly_push(osz_poly_t h, const double *x, int64_t ldx, int64_t n,
                  int final, double *y, int64_t ldy, int64_t *n_out,
                  void *stream);

/* ---- K5/K6: segmenter + detrend + window + rFFT (+ power, + average) -- */
/*
 * Replaces _spectra_estimatives' FIFO segmenter (core/numerical.py:799-849),
 * modified_dft (:635-718: scipy.signal.detrend, get_window, np.fft.rfft,
 * scaling) and periodogram (:721-796), and the running segment average of
 * spectra/estimators.py:149-152.
 */
typedef enum {
    OSZ_SPEC_PSD_MEAN = 0,     /* accumulate sum of periodograms (psd)        */
    OSZ_SPEC_PSD_SEGMENTS = 1, /* one (nch, nfreq) f64 periodogram / segment  */
    OSZ_SPEC_DFT_SEGMENTS = 2  /* one (nch, nfreq) c128 modified DFT / segment */
} osz_spec_mode;
enum { OSZ_DETREND_CONSTANT = 0, OSZ_DETREND_LINEAR = 1 };

/* Segments are nwin samples long, `stride` apart, zero-padded to nfft >= nwin
 * (welch/stft: nwin == nfft; periodogram with nfft > samples: :697-699).
 * window: nwin host doubles (scipy.signal.get_window, periodic);
 * scale: sqrt(norm) of core/numerical.py:703-716, computed by the caller. */
int osz_spec_create(osz_spec_t *h, int nwin, int nfft, int stride,
                    const double *window, double scale, int detrend, int mode,
                    int nch);
int osz_spec_destroy(osz_spec_t h);
int osz_spec_reset(osz_spec_t h, void *stream);
/* Number of complete segments a push of n more samples will emit. */
int64_t osz_spec_seg_count(osz_spec_t h, int64_t n);
/* Consumes x (nch, n).  SEGMENTS modes: writes nseg estimates to out laid out
 * (nseg, nch, nfreq) contiguous (f64 or interleaved c128).  PSD_MEAN: out is
 * ignored and the handle's accumulator advances. */
int osz_spec_push(osz_spec_t h, const double *x, int64_t ldx, int64_t n,
                  void *out, int64_t *nseg, void *stream);
/* PSD_MEAN: raw sum of periodograms (nch, nfreq) f64 on the device and the
 * segment count so far -- a multi-GPU time split reduces these (RCCL) before
 * dividing.  The pointer stays owned by the handle. */
int osz_spec_sum(osz_spec_t h, double **dsum, int64_t *count);
/* PSD_MEAN: copies the raw sum into a caller-owned device buffer (nch, nfreq),
 * e.g. a tensor handed to an RCCL all-reduce. Asynchronous on `stream`. */
int osz_spec_export_sum(osz_spec_t h, double *dst, int64_t *count, void *stream);
/* PSD_MEAN: mean = sum / count into host array (nch, nfreq). Synchronous. */
int osz_spec_mean(osz_spec_t h, double *mean, int64_t *count, void *stream);

/* PSD_MEAN: mean = sum / count into a caller-owned DEVICE buffer (nch, nfreq):
 * the estimate never leaves HBM.  Asynchronous on `stream`. */
int osz_spec_mean_device(osz_spec_t h, double *dmean, int64_t *count, void *stream);
/* Checkpoint / resume: [carried samples, segment count, FIFO remainder
 * (core/numerical.py:821), periodogram sum] as osz_spec_state_size() host
 * doubles.  Synchronous. */
int64_t osz_spec_state_size(osz_spec_t h);
int osz_spec_get_state(osz_spec_t h, double *state, void *stream);
int osz_spec_set_state(osz_spec_t h, const double *state, void *stream);

/* ---- the one collective: Welch segment average over a time split ------ */
/*
 * The reference averages the periodograms of the whole stream with a running
 * mean (spectra/estimators.py:149-152).  When the stream of a channel block is
 * split in time across GPUs (one process per GPU), each rank's PSD_MEAN handle
 * holds the sum over its own segments; osz_welch_reduce all-reduces (sum) the
 * (nch, nfreq) accumulator and the segment count over the ranks of an RCCL
 * communicator, in place: afterwards osz_spec_mean / _mean_device /
 * _export_sum on ANY rank give the average over the whole stream.
 * comm is an ncclComm_t (void*) of the RCCL instance this library binds at run
 * time: the one already mapped in the process (a PyTorch host), else
 * /opt/rocm/lib/librccl.so.1.  A host without PyTorch creates the
 * communicator through the three helpers below; the 128-byte unique id
 * travels from rank 0 to the others by the host's own means (file, socket).
 * osz_welch_reduce synchronises `stream` (the count comes back to the host).
 */
#define OSZ_RCCL_ID_BYTES 128
int osz_rccl_bind(const char *librccl_path /* NULL: default search */);
int osz_rccl_unique_id(char *id128);
int osz_rccl_comm_create(void **comm, int nranks, int rank, const char *id128);
int osz_rccl_comm_destroy(void *comm);
int osz_rccl_comm_size(void *comm, int *nranks);
int osz_welch_reduce(osz_spec_t h, void *comm, void *stream);

/* ---- host side of host-fed streams -------------------------------------- */
/* dst[r][0, row_bytes) = src[r][0, row_bytes), r < rows; rows dst_pitch / src_pitch bytes
 * apart, host memory both: packs a chunk of a host ndarray -- a column range of a C-ordered
 * array, what the reference's ArrayProducer slices (core/producer.py:289-295) -- into a
 * pinned staging buffer over a few persistent threads; returns when the rows are in place. */
int osz_host_copy2d(void *dst, int64_t dst_pitch, const void *src, int64_t src_pitch,
                    int64_t rows, int64_t row_bytes);

/* ---- K7: mask compaction ---------------------------------------------- */
/* y[c, j] = x[c, idx[j]], j < nidx: np.take(arr, np.flatnonzero(mask), axis)
 * of MaskedProducer.__iter__ (core/producer.py:432). idx: device int64. */
int osz_take(const double *x, int64_t ldx, int nch, const int64_t *idx,
             int64_t nidx, double *y, int64_t ldy, void *stream);

/* ---- producer-level arithmetic on device chunks -- */
/*
 * Streaming per-channel moments: replaces the chunk loops of protools.mean /
 * protools.std (core/protools.py:500-545, :547-592).  push folds one chunk
 * x (nch, n): per channel the sum, the sum of squares and the count of its
 * samples (NaNs skipped when ignore_nan, numpy.nanmean; deterministic order),
 * then A += n * sum/count, B += n * sumsq/count, L += n -- the reference
 * weights every chunk's (nan)mean by the chunk length.  finish writes
 * mean = A/L and std = sqrt(B/L - mean^2) (:592) to device arrays (nch);
 * either may be NULL.
 */
typedef struct osz_moments_s *osz_moments_t;
int osz_moments_create(osz_moments_t *h, int nch);
int osz_moments_destroy(osz_moments_t h);
int osz_moments_reset(osz_moments_t h, void *stream);
int osz_moments_push(osz_moments_t h, const double *x, int64_t ldx, int64_t n,
                     int ignore_nan, void *stream);
int osz_moments_finish(osz_moments_t h, double *dmean, double *dstd, void *stream);

/* Mean / standard deviation along the FIRST axis of a (nred, ncols) matrix
 * (numpy.(nan)mean, numpy.(nan)std per column): what protools.mean / std do
 * chunk by chunk when the reduced axis is not the production axis
 * (core/protools.py:538-545, :586-592).  Device outputs (ncols); either NULL. */
int osz_col_moments(const double *x, int64_t ldx, int nred, int64_t ncols, int ignore_nan,
                    double *dmean, double *dstd, void *stream);

/* Elementwise y = x (op) operand with NumPy-style broadcasting reduced to the
 * (channels, samples) layout: protools.add / multiply / multiply_along_axis
 * (core/protools.py:72-180, :334-384), standardize's (x - mean) / std
 * (:660-671) and power_norm's division (spectra/metrics.py:141).  a (and b
 * for STANDARDIZE) are DEVICE arrays: one value, one per channel (row), one
 * per sample (column), or a full (nch, n) matrix with pitch ldab. */
enum { OSZ_EW_ADD = 0, OSZ_EW_MUL = 1, OSZ_EW_DIV = 2, OSZ_EW_STANDARDIZE = 3 };
enum { OSZ_BCAST_SCALAR = 0, OSZ_BCAST_ROW = 1, OSZ_BCAST_COL = 2, OSZ_BCAST_FULL = 3 };
int osz_ew(int op, const double *x, int64_t ldx, int nch, int64_t n, const double *a,
           const double *b, int kind, int64_t ldab, double *y, int64_t ldy, void *stream);

/* z = re + i im (interleaved c128, pitch in complex elements): the analytic
 * signal x + i H(x) of experimental/coupling/transforms.py:186-192. */
int osz_complex_join(const double *re, int64_t ldre, const double *im, int64_t ldim,
                     int nch, int64_t n, double *z, int64_t ldz, void *stream);
/* mag = |z| (numpy.abs), phase = numpy.angle(z) mapped to [0, 2 pi)
 * (transforms.py:75-107); either output may be NULL. */
int osz_magphase(const double *z, int64_t ldz, int nch, int64_t n, double *mag,
                 double *phase, int64_t ldo, void *stream);
/* out[c] = scipy.integrate.simpson(p[c, a : a + m], dx = dx): the band power
 * of spectra/metrics.py:80-87 on the (nch, nfreq) device estimate. */
int osz_simpson(const double *p, int64_t ldp, int nch, int64_t a, int64_t m, double dx,
                double *out, void *stream);

/* ---- EDF record decode ----------------------------- */
/*
 * Replaces the host-side unpacking of edf.Reader (reference
 * file_io/edf.py:452-483 _records, :382-419 _decipher, :506-556 _read_array):
 * the file's little-endian int16 records go to the device as they are (2 B per
 * sample over PCIe instead of 8) and are de-interleaved and scaled there.
 * raw: device int16, records [rec0, ...) of `reclen` samples each; per-channel
 * device arrays choff/spr (int32), slope/offset (f64), len (int64: samples the
 * channel can fill; the rest of the `width` columns get padvalue*slope+offset,
 * as the reference pads before deciphering).  out: (nch, width) f64.
 */
int osz_edf_decode(const int16_t *raw, int reclen, int nch, const int32_t *choff,
                   const int32_t *spr, const double *slope, const double *offset,
                   const int64_t *len, int64_t rec0, int64_t start, int64_t width,
                   double padvalue, double *out, int64_t ldo, void *stream);

/* ---- synthetic device-resident source (benchmarks, tests) ------------- */
/* x[c, j] = N(0,1) keyed by (seed, ch0 + c, n0 + j): counter-based, so any
 * shard or chunk is reproducible on CPU and GPU alike. */
int osz_synth_normal(double *x, int64_t ldx, int nch, int64_t n, uint64_t seed,
                     int64_t ch0, int64_t n0, void *stream);
/* 64-bit order-independent checksum of a (nch, n) block: sum of the bit
 * patterns, plus the float sum; both written to host. Synchronous. */
int osz_checksum(const double *x, int64_t ldx, int nch, int64_t n,
                 uint64_t *bits, double *fsum, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OSZ_HIP_H */
