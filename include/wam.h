/*
 * wam.h - C ABI of libwam_hip.so, the MI355X (gfx950) backend that replaces the
 * export-render and analysis hot path of entrepeneur4lyf/Web-Audio-Mastering.
 *
 * Every entry point cites the reference interface it replaces (paths under the
 * reference checkout, file:line).  Conventions, all entry points:
 *   - plain C, no exceptions; return 0 (WAM_OK) or a negative wam_status, and
 *     wam_last_error() then holds a message for the calling thread;
 *   - audio is planar float32: `ch[c]` points at `n` samples of channel c;
 *   - `*_dev` entry points take DEVICE pointers (hipMalloc memory) and enqueue on
 *     the HIP stream given (NULL = default stream); the others take HOST pointers,
 *     copy in, run the same kernels, copy out and synchronise;
 *   - the caller owns every buffer; inputs are never written (the reference's
 *     interface transfers copies, web/workers/worker-interface.js:116-132);
 *   - there is no CPU fallback: without a usable HIP device every compute entry
 *     point returns WAM_ERR_NO_DEVICE.
 */
#ifndef WAM_H
#define WAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  WAM_OK = 0,
  WAM_ERR_INVALID = -1,    /* bad argument (NULL, n <= 0, nch not 1..2, ...) */
  WAM_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime failure at init */
  WAM_ERR_HIP = -3,        /* a HIP call failed; message in wam_last_error() */
  WAM_ERR_NOMEM = -4,
  WAM_ERR_CANCELLED = -5,  /* progress callback returned non-zero ('Cancelled', web/ui/encoder.js:195-197) */
  WAM_ERR_TOO_LARGE = -6
} wam_status;

/* Flat settings object of the reference UI, web/ui/controls.js:66-89 (same
 * meaning per key; booleans are 0/1).  auto_level is accepted and ignored, as
 * in the reference (read at controls.js:80, never consumed). */
typedef struct wam_settings {
  int32_t normalize_loudness;   /* normalizeLoudness */
  double  target_lufs;          /* targetLufs; 0 disables, dsp-worker.js:1602 */
  int32_t true_peak_limit;      /* truePeakLimit */
  double  true_peak_ceiling;    /* truePeakCeiling dB; 0 or NaN means -1, dsp-worker.js:1612 */
  int32_t clean_low_end;        /* cleanLowEnd: 30 Hz HPF */
  int32_t glue_compression;     /* glueCompression */
  int32_t deharsh;              /* deharsh */
  double  stereo_width;         /* stereoWidth in percent (100 = unchanged); NaN means 100 */
  int32_t center_bass;          /* centerBass */
  int32_t cut_mud;              /* cutMud */
  int32_t add_air;              /* addAir (exciter) */
  int32_t tape_warmth;          /* tapeWarmth */
  int32_t auto_level;           /* autoLevel (ignored) */
  int32_t add_punch;            /* addPunch */
  double  input_gain;           /* inputGain dB */
  double  eq_low, eq_low_mid, eq_mid, eq_high_mid, eq_high; /* dB */
} wam_settings;

enum { WAM_MODE_PREVIEW = 0, WAM_MODE_EXPORT = 1 }; /* dsp-worker.js:1473,1530 */

typedef struct wam_result {
  double lufs;            /* result.lufs          (dsp-worker.js:1649,1660) */
  double measured_lufs;   /* result.measuredLufs  (dsp-worker.js:1661)      */
  double true_peak_db;    /* findTruePeak of the output (extra; true-peak.js:46) */
  double lufs_in;         /* loudness measured by stage 8 before the gain (normalizer.js:39); NaN if stage 8 did not run */
  double gain_db;         /* stage-8 gain applied, 0 if none (normalizer.js:50) */
} wam_result;

/* Progress callback, same fractions/status strings as sendProgress(),
 * dsp-worker.js:69-71,1475-1664, in the same order - the look-ahead limiter's own messages (:302,364,427,448)
 * included, wherever the worker runs that limiter (RENDER_FULL_CHAIN, NORMALIZE, APPLY_LIMITER).  Return non-zero to cancel. */
typedef int (*wam_progress_cb)(double progress, const char* status, void* user);

/* ---- library ------------------------------------------------------------ */
const char* wam_last_error(void);
/* Initialise device `device` (hipSetDevice) for the calling thread; idempotent.
 * Replaces DSPWorkerInterface.init(), worker-interface.js:23. */
int wam_init(int device);
int wam_device_count(int* count);
const char* wam_version(void);
/* Version of THIS header's signatures.  Bumped whenever an existing entry point changes its parameter list (2: wam_normalize
 * and wam_limiter gained (wam_progress_cb, void*); 3: round 3, no signature changed, entry points added): a binding compares
 * wam_abi_version() with the WAM_ABI_VERSION it was written against before its first call and refuses to run on a mismatch,
 * instead of passing a stale argument list.  The in-tree bindings (napi/wam_napi.cc, python/wam.py) do. */
#define WAM_ABI_VERSION 3
int wam_abi_version(void);

/* ---- metering ----------------------------------------------------------- */
/* measureLUFS(audioBuffer, fallbackLufs), web/lib/dsp/lufs.js:17-82 (worker twin
 * dsp-worker.js:77-137, message MEASURE_LUFS :1355).  `lufs` gets -INFINITY for
 * silence and `fallback` for audio shorter than 0.4 s.  Optional outputs (may be
 * NULL): block_ms[nblocks] mean-square per 400 ms block; gate_mask[nblocks]
 * 0 = dropped by absolute gate, 1 = dropped by relative gate, 2 = kept;
 * nblocks_out = number of blocks.  Capacity of both arrays: wam_lufs_num_blocks(). */
int wam_measure_lufs(const float* const* ch, int nch, int64_t n, double sample_rate,
                     double fallback, double* lufs,
                     double* block_ms, uint8_t* gate_mask, int64_t* nblocks_out);
int64_t wam_lufs_num_blocks(int64_t n, double sample_rate);

/* findTruePeak(audioBuffer), web/lib/dsp/true-peak.js:46-71 (worker twin
 * dsp-worker.js:169-192, message FIND_TRUE_PEAK :1361).  dBTP, -INFINITY for silence. */
int wam_true_peak_db(const float* const* ch, int nch, int64_t n, double* dbtp);

/* measureShortTermLUFS(audioBuffer, position) / measureMomentaryLUFS(audioBuffer, position), web/lib/dsp/lufs.js:91-119
 * and :128-171: the 3 s window at `position` seconds measured like a buffer of its own (filters from zero state, both
 * gates), and the ungated K-weighted level of the 400 ms window.  -INFINITY when less than half a window is left. */
int wam_short_term_lufs(const float* const* ch, int nch, int64_t n, double sample_rate, double position, double* lufs);
int wam_momentary_lufs(const float* const* ch, int nch, int64_t n, double sample_rate, double position, double* lufs);

/* The same two measures at npos positions of one buffer in one call (one upload): lufs[i] = measureShortTermLUFS(ab,
 * positions[i]) when `momentary` is 0, measureMomentaryLUFS(ab, positions[i]) otherwise (lufs.js:91-171).  This is the series
 * a meter display polls; every window is still measured like a buffer of its own, exactly as the reference does per call. */
int wam_lufs_series(const float* const* ch, int nch, int64_t n, double sample_rate, const double* positions, int npos,
                    int momentary, double* lufs);

/* resampleAudioBuffer(sourceBuffer, targetSampleRate), web/ui/renderer.js:172-187.  The reference renders the buffer through
 * Chromium's OfflineAudioContext at the target rate; that engine is not in the image and the reference holds no fixtures of
 * its output, so this entry is PARITY UNPINNED and self-defined: out[i] is the signal at source position i * src / dst
 * (zero outside the buffer), quality 0 = linear interpolation, 1 = Kaiser-windowed sinc (cutoff 0.98 x the lower Nyquist
 * rate, 64 zero crossings per side, beta 8.6).  n_out must equal wam_resample_length() = ceil(n / src * dst) (:178). */
int64_t wam_resample_length(int64_t n, double src_rate, double dst_rate);
int wam_resample(const float* const* ch, int nch, int64_t n, double src_rate, double dst_rate, int quality,
                 float* const* out, int64_t n_out);

/* extractPeaks(audioBuffer, numPeaks), web/ui/waveform.js:164-181: peaks[i] = max |x| over window i of floor(n / num_peaks)
 * samples of channel 0 (the values are f32 magnitudes, exact); all zero when n < num_peaks. */
int wam_extract_peaks(const float* ch0, int64_t n, int num_peaks, float* peaks);

/* One pass over the input for both numbers (what app.js:865-866 asks for on load). */
int wam_meter(const float* const* ch, int nch, int64_t n, double sample_rate,
              double fallback, double* lufs, double* dbtp);

/* detectDCOffsetBuffer / removeDCOffset, web/lib/dsp/dc-offset.js:38-77: mean per
 * channel into dc[nch]; when `remove` is set, out[c] = ch[c] - dc[c] where
 * |dc[c]| > 1e-10 (copy otherwise).  out may be NULL when remove == 0. */
int wam_dc_offset(const float* const* ch, int nch, int64_t n, double* dc, int remove, float* const* out);

/* ---- rendering ---------------------------------------------------------- */
/* RENDER_FULL_CHAIN, web/workers/dsp-worker.js:1468-1667, reached through
 * DSPWorkerInterface.renderFullChain(), worker-interface.js:259-277.
 * out[c] receives n samples.  Differences from the reference, both documented in
 * the 5-band EQ / Cut Mud stage works (the reference's is broken,
 * SURVEY.md 0.1) and follows the Web Audio biquad formulas with the parameters of
 * dsp-worker.js:1236-1264. */
int wam_render_chain(const float* const* ch, int nch, int64_t n, double sample_rate,
                     const wam_settings* settings, int mode, float* const* out,
                     wam_result* result, wam_progress_cb progress, void* user);

/* Batch form: ntracks tracks of identical shape, track t channel c at
 * in[t * nch + c] / out[t * nch + c]; results[t].  Tracks are independent
 * (SURVEY.md 8e); the caller shards track ranges over devices, one process or
 * thread per device, and calls this once per shard. */
int wam_batch_render(const float* const* in, int ntracks, int nch, int64_t n, double sample_rate,
                     const wam_settings* settings, int mode, float* const* out, wam_result* results);

/* The same batch over several devices of one node (the north-star's "multi-track batches shard track-wise across the GPUs";
 * it batches what DSPWorkerInterface.renderFullChain(), worker-interface.js:259-277, does one track at a time): entry i of
 * `devices` renders the contiguous track range [ceil(i * ntracks / ndev), ceil((i + 1) * ntracks / ndev)) on HIP device
 * devices[i], from its own host thread with its own streams; per-track results are written at their track index.  No
 * collective, no peer traffic.  A device may be listed more than once.  The calling thread keeps its device. */
int wam_batch_render_multi(const float* const* in, int ntracks, int nch, int64_t n, double sample_rate,
                           const wam_settings* settings, int mode, float* const* out, wam_result* results,
                           const int* devices, int ndev);

/* Device-resident batch: `in_dev` / `out_dev` are device pointers to
 * [ntracks][nch][n] float32, contiguous.  Enqueues on `stream` and returns after
 * the results have been copied back (it synchronises the stream once). */
int wam_batch_render_dev(const float* in_dev, int ntracks, int nch, int64_t n, double sample_rate,
                         const wam_settings* settings, int mode, float* out_dev,
                         wam_result* results, void* stream);
int wam_batch_meter_dev(const float* in_dev, int ntracks, int nch, int64_t n, double sample_rate,
                        double fallback, double* lufs, double* dbtp, void* stream);

/* wam_batch_meter_dev for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code that runs under
 * HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same kernels as wam_batch_meter_dev; LUFS / dBTP agree with
 * it to ~1e-12 (the closing log10 runs on the device here, on the host there).  Differences: (1) what that entry point takes from the library's workspace pool (plan, per-hop sums, peaks, block energies, gate
 * results) lives in a fixed 64 MiB device array that every GPU holds from the moment it loads the library until the process
 * ends; (2) the results go to the CALLER's device arrays lufs_dev / dbtp_dev [ntracks] (either may be NULL), so there is
 * no device-to-host copy and no synchronisation: the call only enqueues on `stream`; (3) it calls no hipMalloc / hipFree or
 * other allocating routine - call wam_init(device) once beforehand, which looks the array's address up on that device.
 * A batch whose scratch does not fit (1,024 three-minute stereo tracks need 44.5 MB) is refused with WAM_ERR_TOO_LARGE.
 * All calls on one GPU share that GPU's array, so they are ordered on the device: each call's stream first waits (no host
 * synchronisation) for an event recorded behind the previous call's last kernel on that GPU - calls from two streams or two
 * host threads run one after the other.  Under stream capture the event is not used: two graphs that contain this call must
 * not run at the same time on one GPU. */
int wam_batch_meter_dev_noalloc(const float* in_dev, int ntracks, int nch, int64_t n, double sample_rate,
                                double fallback, double* lufs_dev, double* dbtp_dev, void* stream);

/* What web/app.js:823-872 does on every file load, as one call with one upload: detectDCOffsetBuffer
 * (dc-offset.js:38-60); removeDCOffset on every channel when any channel's |mean| > 0.001 (app.js:828-834); measureLUFS and
 * findTruePeak of the result (:865-866); normalizeToLUFS(buffer, target_lufs, ceiling_db) with the library limiter when the
 * projected peak exceeds the ceiling (normalizer.js:38-94).  cleaned / normalized (either may be NULL) receive the buffer
 * after DC removal (a copy of the input when none was needed) and the normalised buffer (the cleaned buffer when loudness
 * is not finite, as the reference returns its input). */
typedef struct wam_load_analysis {
  double dc[2];                 /* mean per channel */
  double dc_percent;            /* average |mean| * 100 (dcInfo.average.percent) */
  double original_lufs, original_true_peak_db;
  double gain_db;               /* applied by the normalisation (0 when not measured) */
  int32_t dc_significant, dc_removed, measured, limiter_applied;
} wam_load_analysis;
int wam_analyze_on_load(const float* const* ch, int nch, int64_t n, double sample_rate, double target_lufs, double ceiling_db,
                        float* const* cleaned, float* const* normalized, wam_load_analysis* res);

/* applyGain(audioBuffer, gainDB), web/lib/dsp/normalizer.js:103-121: out = f32(in * 10^(gainDB/20)). */
int wam_apply_gain(const float* const* ch, int nch, int64_t n, double gain_db, float* const* out);

/* normalizeChannelsToLUFS, dsp-worker.js:457-527 (message NORMALIZE :1367;
 * DSPWorkerInterface.normalize(), worker-interface.js:197).  Applies the limiter
 * when the projected peak exceeds ceiling_db. */
typedef struct wam_normalize_result {
  double current_lufs, final_lufs, peak_db, gain_applied_db;
  int32_t limiter_applied;
  int32_t measured;   /* 0 when loudness was not finite and the input was returned unchanged */
} wam_normalize_result;
int wam_normalize(const float* const* ch, int nch, int64_t n, double sample_rate,
                  double target_lufs, double ceiling_db, float* const* out, wam_normalize_result* res,
                  wam_progress_cb progress, void* user);   /* the message's own PROGRESS stream (:458,471,491 + the limiter's), may be NULL */

/* applyLookaheadLimiterToChannels, dsp-worker.js:286-452 (message APPLY_LIMITER
 * :1380; DSPWorkerInterface.applyLimiter(), worker-interface.js:227). */
int wam_limiter(const float* const* ch, int nch, int64_t n, double sample_rate,
                double ceiling_linear, double lookahead_ms, double release_ms, double knee_db,
                int preserve_transients, float* const* out,
                wam_progress_cb progress, void* user);     /* dsp-worker.js:302,364,427,448, then 1.0 'Complete' (:1673); may be NULL */

/* applyLookaheadLimiter(audioBuffer, ...), web/lib/dsp/limiter.js:155-362: the library twin, which the sync API and
 * normalizeToLUFS (normalizer.js:83) call.  Same stages; its oversampled soft-knee pass starts the running maximum at 0
 * instead of |y2| (limiter.js:107 vs dsp-worker.js:260). */
int wam_limiter_lib(const float* const* ch, int nch, int64_t n, double sample_rate,
                    double ceiling_linear, double lookahead_ms, double release_ms, double knee_db,
                    int preserve_transients, float* const* out);

/* Single stages, same arithmetic as inside wam_render_chain (lib/dsp/index.js names). */
int wam_final_filters(const float* const* ch, int nch, int64_t n, double sample_rate,
                      int highpass, int lowpass, float* const* out);          /* final-filters.js:96 */
int wam_soft_clip(const float* const* ch, int nch, int64_t n, double sample_rate,
                  double ceiling_db, double lookahead_ms, double release_ms, double drive,
                  float* const* out);                                         /* soft-clipper.js:165 */
int wam_glue_compressor(const float* const* ch, int nch, int64_t n, double sample_rate,
                        double threshold_db, double ratio, double attack_s, double release_s,
                        float* const* out);                                   /* dsp-worker.js:1272 */
int wam_stereo(const float* const* ch, int64_t n, double sample_rate, double width, int bass_mono,
               double bass_freq, float* const* out);                          /* dsp-worker.js:1122 */
int wam_parametric_eq(const float* const* ch, int nch, int64_t n, double sample_rate,
                      const double gains_db[5], int cut_mud, float* const* out); /* dsp-worker.js:1222 */
int wam_exciter(const float* const* ch, int nch, int64_t n, double sample_rate, float* const* out);   /* exciter.js:223 */
int wam_deharsh(const float* const* ch, int nch, int64_t n, double sample_rate, float* const* out);   /* dynamic-processor.js:595 ('mastering') */
/* processHybridDynamic(buffer, preset), dynamic-processor.js:595-609: 'gentle' (:538-549; also what any unknown name and no
 * name at all select, :604-606), 'aggressive' (:555-566), 'mastering' (:572-585, = wam_deharsh).  The first two are fully wet
 * (dryWetMix 1.0, :268): samples after the last whole frame fade out with the synthesis window, as the reference's do. */
enum { WAM_DEHARSH_GENTLE = 0, WAM_DEHARSH_AGGRESSIVE = 1, WAM_DEHARSH_MASTERING = 2 };
int wam_deharsh_preset(const float* const* ch, int nch, int64_t n, double sample_rate, int preset, float* const* out);
int wam_tape_warmth(const float* const* ch, int nch, int64_t n, double sample_rate, float* const* out); /* multiband-saturation.js:365 */
int wam_multiband_transient(const float* const* ch, int nch, int64_t n, double sample_rate, float* const* out); /* multiband-transient.js:15 */

/* encodeWAV 24-bit branch, web/ui/encoder.js:116-123: interleaved little-endian
 * 24-bit PCM, 3 * nch * n bytes, no header (wam_wav_header writes the 44 bytes of
 * encoder.js:74-86). */
int wam_encode_pcm24(const float* const* ch, int nch, int64_t n, uint8_t* out);
int wam_wav_header(int nch, int64_t n, int sample_rate, int bits, uint8_t out44[44]);

/* encodeWAV 16-bit branch, web/ui/encoder.js:97-115: interleaved little-endian int16, 2 * nch * n bytes, no header.
 * dither_mode 0 = 'none', 1 = 'tpdf', 2 = 'noise-shaped' (error feedback 0.85, encoder.js:16,110-113).
 * dither = nch * n values in LSB units, the values triangularDither() would return, one per sample-channel in
 * interleaved order (the reference draws them from Math.random, encoder.js:10-30: pass that stream for identical
 * bytes); NULL = an internal TPDF generator of the same distribution, keyed by `seed`. */
enum { WAM_DITHER_NONE = 0, WAM_DITHER_TPDF = 1, WAM_DITHER_NOISE_SHAPED = 2 };
int wam_encode_pcm16(const float* const* ch, int nch, int64_t n, int dither_mode, const float* dither, uint32_t seed, uint8_t* out);

/* Spectrogram.process() per frame, web/ui/spectrogram.js:113-143,217-257, batched:
 * frames start at f * hop of the mono mix 0.5 (L + R) (the analyser's down-mix),
 * periodic Hann, dB per bin as spectrogram.js:135-142.  out[nframes][nfft / 2];
 * nframes = (n - nfft) / hop + 1.  columns (optional) [nframes][256] = the
 * log-frequency column of updateLatestColumn(), spectrogram.js:177-215. */
int wam_stft_db(const float* const* ch, int nch, int64_t n, double sample_rate, int nfft, int hop,
                float* out_db, float* columns);
int64_t wam_stft_num_frames(int64_t n, int nfft, int hop);
/* Device-resident batch (BASELINE.json configs[4]: the corpus through the same per-frame transform):
 * in_dev [ntracks][nch][n], out_db_dev [ntracks][nframes][nfft/2], columns_dev [ntracks][nframes][256] or NULL. */
int wam_batch_stft_db_dev(const float* in_dev, int ntracks, int nch, int64_t n, double sample_rate, int nfft, int hop,
                          float* out_db_dev, float* columns_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WAM_H */
