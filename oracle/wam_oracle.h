/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT.
 *
 * CPU restatement (plain C, IEEE double, f32 rounding on every store the
 * reference makes into a Float32Array) of the reference's export chain and
 * meters.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker.  Pinned by
 * tests/golden/data/golden.json (outputs of the reference's own JavaScript run
 * under Node by tests/golden/gen_golden.mjs) in tests/test_oracle_golden.py.
 * Unpinned parts, stated where they occur: 5-band EQ / Cut Mud (the reference's
 * stage is non-functional; Web Audio spec formulas) and the DFT inside de-harsh
 * (npm fft.js@4.0.4 is not in the reference checkout; a radix-2 double DFT).
 */
#ifndef WAM_ORACLE_H
#define WAM_ORACLE_H
#include <stdint.h>
#include "../include/wam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_synth_opts {
  double amp, dc, click_amp;
  int64_t quiet_from, quiet_to; double quiet_gain;
  int64_t quiet2_from, quiet2_to; double quiet2_gain;
} ora_synth_opts;
void ora_synth(int idx, int ch, int64_t n0, int64_t count, double sr, const ora_synth_opts* o, float* out);

typedef struct ora_biquad { double b0, b1, b2, a1, a2; } ora_biquad;
void ora_calc_high_shelf(double sr, double f, double gain_db, double q, ora_biquad* c);
void ora_calc_high_pass(double sr, double f, double q, ora_biquad* c);
void ora_biquad_filter(const float* in, int64_t n, const ora_biquad* c, float* out);

double ora_measure_lufs(const float* const* ch, int nch, int64_t n, double sr, double fallback,
                        double* block_ms, uint8_t* gate_mask, int64_t* nblocks);
double ora_true_peak_db(const float* const* ch, int nch, int64_t n);
double ora_true_peak_sample(const double y[4]);
double ora_short_term_lufs(const float* const* ch, int nch, int64_t n, double sr, double position);
double ora_momentary_lufs(const float* const* ch, int nch, int64_t n, double sr, double position);
void ora_extract_peaks(const float* x, int64_t n, int num_peaks, float* peaks);
double ora_dc_detect(const float* x, int64_t n);
double ora_dc_remove(float* x, int64_t n);

void ora_final_filters(const float* const* in, int nch, int64_t n, double sr, int highpass, int lowpass, float* const* out);
void ora_parametric_eq(const float* const* in, int nch, int64_t n, double sr, const double gains[5], int cut_mud, float* const* out);
void ora_glue(const float* const* in, int nch, int64_t n, double sr, double thr_db, double ratio, double attack, double release, float* const* out);
void ora_stereo(const float* const* in, int64_t n, double sr, double width, int bass_mono, double bass_freq, float* const* out);
/* returns 1 if gain applied, 0 if loudness not finite (out = copy of in) */
int ora_normalize_skip_limiter(const float* const* in, int nch, int64_t n, double sr, double target, float* const* out,
                               double* lufs_in, double* tp_in);
void ora_soft_clip(const float* const* in, int nch, int64_t n, double sr, double ceiling_db, double lookahead_ms,
                   double release_ms, double drive, float* const* out);
void ora_limiter(const float* const* in, int nch, int64_t n, double sr, double ceiling_lin, double lookahead_ms,
                 double release_ms, double knee_db, int preserve_transients, float* const* out);
void ora_limiter_lib(const float* const* in, int nch, int64_t n, double sr, double ceiling_lin, double lookahead_ms,
                     double release_ms, double knee_db, int preserve, float* const* out);
void ora_soft_knee_oversampled(const float* in, int64_t n, double ceiling, double knee_db, float* out);
void ora_normalize(const float* const* in, int nch, int64_t n, double sr, double target, double ceiling_db,
                   float* const* out, wam_normalize_result* res);
void ora_exciter(const float* const* in, int nch, int64_t n, double sr, float* const* out);
void ora_deharsh(const float* const* in, int nch, int64_t n, double sr, float* const* out);
void ora_deharsh_preset(const float* const* in, int nch, int64_t n, double sr, int preset, float* const* out);   /* 0 gentle, 1 aggressive, 2 mastering */
void ora_warmth(const float* const* in, int nch, int64_t n, double sr, float* const* out);
void ora_punch(const float* const* in, int nch, int64_t n, double sr, float* const* out);
void ora_pcm24(const float* const* in, int nch, int64_t n, uint8_t* out);
void ora_pcm16(const float* const* in, int nch, int64_t n, int mode, const float* dither, uint8_t* out);
void ora_wav_header(int nch, int64_t n, int sr, int bits, uint8_t out[44]);
void ora_stft_db(const float* const* in, int nch, int64_t n, double sr, int nfft, int hop, float* out_db, float* columns);
int ora_render_chain(const float* const* in, int nch, int64_t n, double sr, const wam_settings* s, int mode,
                     float* const* out, wam_result* res);

#ifdef __cplusplus
}
#endif
#endif
