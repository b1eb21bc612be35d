/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT (see wam_oracle.h).
 *
 * Each function restates one reference function; "ref:" gives the file:line under
 * the reference checkout.  Arithmetic convention of the reference: values read
 * from a Float32Array widen to double, scalar math in IEEE double in source
 * order, one rounding to f32 at each store into a Float32Array.  Build with
 * -ffp-contract=off so the compiler never fuses a*b+c.
 */
#include "wam_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define F32(x) ((float)(x))
static float* falloc(int64_t n) { return (float*)calloc((size_t)(n > 0 ? n : 1), sizeof(float)); }
static double js_sign(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : v); }
static double db_to_linear(double db) { return pow(10, db / 20); }                 /* ref: utils.js:80-82 */
static double linear_to_db(double l) { return l > 0 ? 20 * log10(l) : -INFINITY; } /* ref: utils.js:89-91 */

/* ref: web/lib/dsp/utils.js:143-153 interpolateCatmullRom */
static double catmull(double y0, double y1, double y2, double y3, double t) {
  const double a0 = -0.5 * y0 + 1.5 * y1 - 1.5 * y2 + 0.5 * y3;
  const double a1 = y0 - 2.5 * y1 + 2 * y2 - 0.5 * y3;
  const double a2 = -0.5 * y0 + 0.5 * y2;
  const double a3 = y1;
  const double t2 = t * t;
  const double t3 = t2 * t;
  return a0 * t3 + a1 * t2 + a2 * t + a3;
}

/* ref: web/lib/dsp/utils.js:35-50 calcHighShelfCoeffs */
void ora_calc_high_shelf(double sr, double f, double gain_db, double q, ora_biquad* c) {
  const double A = pow(10, gain_db / 40);
  const double w0 = 2 * M_PI * f / sr;
  const double cosw = cos(w0), sinw = sin(w0);
  const double alpha = sinw / (2 * q);
  const double b0 = A * ((A + 1) + (A - 1) * cosw + 2 * sqrt(A) * alpha);
  const double b1 = -2 * A * ((A - 1) + (A + 1) * cosw);
  const double b2 = A * ((A + 1) + (A - 1) * cosw - 2 * sqrt(A) * alpha);
  const double a0 = (A + 1) - (A - 1) * cosw + 2 * sqrt(A) * alpha;
  const double a1 = 2 * ((A - 1) - (A + 1) * cosw);
  const double a2 = (A + 1) - (A - 1) * cosw - 2 * sqrt(A) * alpha;
  c->b0 = b0 / a0; c->b1 = b1 / a0; c->b2 = b2 / a0; c->a1 = a1 / a0; c->a2 = a2 / a0;
}

/* ref: web/lib/dsp/utils.js:59-73 calcHighPassCoeffs; same formulas at
 * final-filters.js:22-43, multiband-transient.js:137-169, multiband-saturation.js:63-96 */
static void rbj(double sr, double f, double q, int highpass, ora_biquad* c) {
  const double w0 = 2 * M_PI * f / sr;
  const double cosw = cos(w0), sinw = sin(w0);
  const double alpha = sinw / (2 * q);
  double b0, b1, b2;
  if (highpass) { b0 = (1 + cosw) / 2; b1 = -(1 + cosw); b2 = (1 + cosw) / 2; }
  else          { b0 = (1 - cosw) / 2; b1 = 1 - cosw;    b2 = (1 - cosw) / 2; }
  const double a0 = 1 + alpha, a1 = -2 * cosw, a2 = 1 - alpha;
  c->b0 = b0 / a0; c->b1 = b1 / a0; c->b2 = b2 / a0; c->a1 = a1 / a0; c->a2 = a2 / a0;
}
void ora_calc_high_pass(double sr, double f, double q, ora_biquad* c) { rbj(sr, f, q, 1, c); }

/* ref: web/lib/dsp/utils.js:12-25 applyBiquadFilter (DF-I, double state, f32 store).
 * in == out is allowed (final-filters.js:51-59 filters in place). */
void ora_biquad_filter(const float* in, int64_t n, const ora_biquad* c, float* out) {
  double x1 = 0, x2 = 0, y1 = 0, y2 = 0;
  const double b0 = c->b0, b1 = c->b1, b2 = c->b2, a1 = c->a1, a2 = c->a2;
  for (int64_t i = 0; i < n; i++) {
    const double x0 = in[i];
    const double y0 = b0 * x0 + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
    out[i] = F32(y0);
    x2 = x1; x1 = x0; y2 = y1; y1 = y0;
  }
}

/* ref: web/lib/dsp/lufs.js:17-82 measureLUFS (worker twin dsp-worker.js:77-137) */
double ora_measure_lufs(const float* const* ch, int nch, int64_t n, double sr, double fallback,
                        double* block_ms, uint8_t* gate_mask, int64_t* nblocks_out) {
  if (nblocks_out) *nblocks_out = 0;
  const double duration = (double)n / sr;
  if (duration < 0.4) return fallback;
  ora_biquad hs, hp;
  ora_calc_high_shelf(sr, 1681.97, 4.0, 0.71, &hs);
  ora_calc_high_pass(sr, 38.14, 0.5, &hp);
  float** f = (float**)malloc(sizeof(float*) * nch);
  for (int c = 0; c < nch; c++) {
    float* t = falloc(n);
    f[c] = falloc(n);
    ora_biquad_filter(ch[c], n, &hs, t);
    ora_biquad_filter(t, n, &hp, f[c]);
    free(t);
  }
  const int64_t bs = (int64_t)floor(sr * 0.4);
  const int64_t hop = (int64_t)floor(sr * 0.4 * (1 - 0.75));
  int64_t nb = 0;
  for (int64_t st = 0; st + bs <= n; st += hop) nb++;
  double* blocks = (double*)malloc(sizeof(double) * (nb > 0 ? nb : 1));
  int64_t k = 0;
  for (int64_t st = 0; st + bs <= n; st += hop) {
    double ss = 0;
    for (int c = 0; c < nch; c++) {
      const float* d = f[c];
      for (int64_t i = st; i < st + bs; i++) ss += (double)d[i] * (double)d[i];
    }
    blocks[k++] = ss / (double)(bs * nch);
  }
  for (int c = 0; c < nch; c++) free(f[c]);
  free(f);
  if (nblocks_out) *nblocks_out = nb;
  if (block_ms) memcpy(block_ms, blocks, sizeof(double) * nb);
  if (gate_mask) memset(gate_mask, 0, (size_t)nb);
  double result = -INFINITY;
  if (nb > 0) {
    double s1 = 0; int64_t c1 = 0;
    for (int64_t i = 0; i < nb; i++) if (blocks[i] > 1e-7) { s1 += blocks[i]; c1++; if (gate_mask) gate_mask[i] = 1; }
    if (c1 > 0) {
      const double mean1 = s1 / (double)c1;
      double s2 = 0; int64_t c2 = 0;
      for (int64_t i = 0; i < nb; i++)
        if (blocks[i] > 1e-7 && blocks[i] > mean1 * 0.1) { s2 += blocks[i]; c2++; if (gate_mask) gate_mask[i] = 2; }
      if (c2 > 0) result = -0.691 + 10 * log10(s2 / (double)c2);
    }
  }
  free(blocks);
  return result;
}

/* ref: web/lib/dsp/true-peak.js:13-38 calculateTruePeakSample */
double ora_true_peak_sample(const double y[4]) {
  const double y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3];
  double peak = fabs(y2);
  const double a0 = -0.5 * y0 + 1.5 * y1 - 1.5 * y2 + 0.5 * y3;
  const double a1 = y0 - 2.5 * y1 + 2 * y2 - 0.5 * y3;
  const double a2 = -0.5 * y0 + 0.5 * y2;
  const double a3 = y1;
  for (int i = 1; i <= 3; i++) {
    const double t = i * 0.25, t2 = t * t, t3 = t2 * t;
    const double v = fabs(a0 * t3 + a1 * t2 + a2 * t + a3);
    if (v > peak) peak = v;
  }
  return peak;
}

/* ref: web/lib/dsp/true-peak.js:46-71 findTruePeak */
static double true_peak_lin(const float* const* ch, int nch, int64_t n) {
  double mx = 0;
  for (int c = 0; c < nch; c++) {
    double p[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; i++) {
      p[0] = p[1]; p[1] = p[2]; p[2] = p[3]; p[3] = ch[c][i];
      if (i >= 3) { const double t = ora_true_peak_sample(p); if (t > mx) mx = t; }
    }
  }
  return mx;
}
double ora_true_peak_db(const float* const* ch, int nch, int64_t n) {
  const double mx = true_peak_lin(ch, nch, n);
  return mx > 0 ? 20 * log10(mx) : -INFINITY;
}

/* ref: web/lib/dsp/lufs.js:91-119 measureShortTermLUFS */
double ora_short_term_lufs(const float* const* ch, int nch, int64_t n, double sr, double position) {
  const int64_t win = (int64_t)floor(sr * 3000 / 1000);
  const int64_t st = (int64_t)floor(position * sr);
  int64_t en = st + win; if (en > n) en = n;
  if ((double)(en - st) < (double)win * 0.5) return -INFINITY;
  const float* sub[64];                          /* any channel count the reference's loop takes (lufs.js:102-116); 64 here */
  if (nch > 64) return NAN;
  for (int c = 0; c < nch; c++) sub[c] = ch[c] + st;
  return ora_measure_lufs(sub, nch, en - st, sr, -14, NULL, NULL, NULL);
}

/* ref: web/lib/dsp/lufs.js:128-171 measureMomentaryLUFS */
double ora_momentary_lufs(const float* const* ch, int nch, int64_t n, double sr, double position) {
  const int64_t win = (int64_t)floor(sr * 400 / 1000);
  const int64_t st = (int64_t)floor(position * sr);
  int64_t en = st + win; if (en > n) en = n;
  if ((double)(en - st) < (double)win * 0.5) return -INFINITY;
  const int64_t m = en - st;
  ora_biquad hs, hp;
  ora_calc_high_shelf(sr, 1681.97, 4.0, 0.71, &hs);
  ora_calc_high_pass(sr, 38.14, 0.5, &hp);
  double ss = 0;
  float* a = falloc(m); float* b = falloc(m);
  for (int c = 0; c < nch; c++) {
    ora_biquad_filter(ch[c] + st, m, &hs, a);
    ora_biquad_filter(a, m, &hp, b);
    for (int64_t i = 0; i < m; i++) ss += (double)b[i] * (double)b[i];
  }
  free(a); free(b);
  const double ms = ss / (double)(m * nch);
  if (ms <= 0) return -INFINITY;
  return -0.691 + 10 * log10(ms);
}

/* ref: web/ui/waveform.js:164-181 extractPeaks (channel 0 only) */
void ora_extract_peaks(const float* x, int64_t n, int num_peaks, float* peaks) {
  const int64_t per = n / num_peaks;                       /* Math.floor(length / numPeaks) */
  for (int i = 0; i < num_peaks; i++) {
    const int64_t st = (int64_t)i * per;
    int64_t en = st + per; if (en > n) en = n;
    float mx = 0;
    for (int64_t j = st; j < en; j++) { const float a = fabsf(x[j]); if (a > mx) mx = a; }
    peaks[i] = mx;
  }
}

/* ref: web/lib/dsp/dc-offset.js:13-31 detectDCOffset, :67-77 removeDCOffset */
double ora_dc_detect(const float* x, int64_t n) {
  double s = 0;
  for (int64_t i = 0; i < n; i++) s += x[i];
  return s / (double)n;
}
double ora_dc_remove(float* x, int64_t n) {
  const double dc = ora_dc_detect(x, n);
  if (fabs(dc) > 1e-10) for (int64_t i = 0; i < n; i++) x[i] = F32((double)x[i] - dc);
  return dc;
}

/* ref: web/lib/dsp/final-filters.js:96-131 applyFinalFilters, :14-60, :68-78 */
void ora_final_filters(const float* const* in, int nch, int64_t n, double sr, int highpass, int lowpass, float* const* out) {
  for (int c = 0; c < nch; c++) {
    float* o = out[c];
    memcpy(o, in[c], sizeof(float) * (size_t)n);
    if (highpass) { ora_biquad b; rbj(sr, 30, 0.707, 1, &b); ora_biquad_filter(o, n, &b, o); }
    if (lowpass) {
      const double rc = 1 / (2 * M_PI * 18000);
      const double dt = 1 / sr;
      const double alpha = dt / (rc + dt);
      double y = (n > 0 && o[0] == o[0] && o[0] != 0) ? (double)o[0] : 0;   /* data[0] || 0 */
      for (int64_t i = 0; i < n; i++) { y = y + alpha * ((double)o[i] - y); o[i] = F32(y); }
    }
  }
}

/* a10.  The reference stage (ref: web/workers/dsp-worker.js:1222-1267) cannot run
 * (SURVEY.md 0.1): PARITY UNPINNED.  Stated from the Web Audio API BiquadFilterNode
 * formulas with the reference's parameters (dsp-worker.js:1236-1264, encoder.js:342-372):
 * lowshelf 80 Hz, peaking 250/1k/4k Hz Q 1, highshelf 12 kHz, Cut Mud peaking 250 Hz -3 dB Q 1.5;
 * each band a DF-I biquad per channel (utils.js:12-25), f32 between bands, skipped at 0 dB. */
static void webaudio_coeffs(int type, double sr, double f, double gain_db, double q, ora_biquad* c) {
  const double A = pow(10, gain_db / 40);
  const double w0 = 2 * M_PI * f / sr;
  const double cw = cos(w0), sw = sin(w0);
  double b0, b1, b2, a0, a1, a2;
  if (type == 0) {          /* lowshelf, S = 1 */
    const double aS = sw / 2 * sqrt((A + 1 / A) * (1 / 1.0 - 1) + 2);
    const double k = 2 * aS * sqrt(A);
    b0 = A * ((A + 1) - (A - 1) * cw + k); b1 = 2 * A * ((A - 1) - (A + 1) * cw); b2 = A * ((A + 1) - (A - 1) * cw - k);
    a0 = (A + 1) + (A - 1) * cw + k; a1 = -2 * ((A - 1) + (A + 1) * cw); a2 = (A + 1) + (A - 1) * cw - k;
  } else if (type == 2) {   /* highshelf, S = 1 */
    const double aS = sw / 2 * sqrt((A + 1 / A) * (1 / 1.0 - 1) + 2);
    const double k = 2 * aS * sqrt(A);
    b0 = A * ((A + 1) + (A - 1) * cw + k); b1 = -2 * A * ((A - 1) + (A + 1) * cw); b2 = A * ((A + 1) + (A - 1) * cw - k);
    a0 = (A + 1) - (A - 1) * cw + k; a1 = 2 * ((A - 1) - (A + 1) * cw); a2 = (A + 1) - (A - 1) * cw - k;
  } else {                  /* peaking */
    const double aQ = sw / (2 * q);
    b0 = 1 + aQ * A; b1 = -2 * cw; b2 = 1 - aQ * A; a0 = 1 + aQ / A; a1 = -2 * cw; a2 = 1 - aQ / A;
  }
  c->b0 = b0 / a0; c->b1 = b1 / a0; c->b2 = b2 / a0; c->a1 = a1 / a0; c->a2 = a2 / a0;
}
void ora_parametric_eq(const float* const* in, int nch, int64_t n, double sr, const double g[5], int cut_mud, float* const* out) {
  static const int types[5] = {0, 1, 1, 1, 2};
  static const double freqs[5] = {80, 250, 1000, 4000, 12000};
  for (int c = 0; c < nch; c++) {
    memcpy(out[c], in[c], sizeof(float) * (size_t)n);
    for (int b = 0; b < 5; b++) {
      if (g[b] != 0) { ora_biquad q; webaudio_coeffs(types[b], sr, freqs[b], g[b], 1.0, &q); ora_biquad_filter(out[c], n, &q, out[c]); }
    }
    if (cut_mud) { ora_biquad q; webaudio_coeffs(1, sr, 250, -3.0, 1.5, &q); ora_biquad_filter(out[c], n, &q, out[c]); }
  }
}

/* ref: web/workers/dsp-worker.js:1272-1343 applyGlueCompressor */
void ora_glue(const float* const* in, int nch, int64_t n, double sr, double thr_db, double ratio, double attack, double release, float* const* out) {
  float* sc = falloc(n);
  if (nch >= 2) for (int64_t i = 0; i < n; i++) sc[i] = F32(((double)in[0][i] + (double)in[1][i]) * 0.5);
  else memcpy(sc, in[0], sizeof(float) * (size_t)n);
  const double thr = pow(10, thr_db / 20);
  const double ac = exp(-1 / (attack * sr));
  const double rc = exp(-1 / (release * sr));
  double env = 0;
  float* gc = falloc(n);
  for (int64_t i = 0; i < n; i++) {
    const double a = fabs((double)sc[i]);
    if (a > env) env = ac * env + (1 - ac) * a;
    else env = rc * env + (1 - rc) * a;
    double gain = 1.0;
    if (env > thr) {
      const double over = 20 * log10(env / thr);
      const double red = over * (1 - 1 / ratio);
      gain = pow(10, -red / 20);
    }
    gc[i] = F32(gain);
  }
  for (int c = 0; c < nch; c++)
    for (int64_t i = 0; i < n; i++) out[c][i] = F32((double)in[c][i] * (double)gc[i]);
  free(sc); free(gc);
}

/* ref: web/workers/dsp-worker.js:1122-1217 applyStereoProcessingToChannels
 * (balance 0, midGain 0, sideGain 0 as called at :1590-1595) */
void ora_stereo(const float* const* in, int64_t n, double sr, double width, int bass_mono, double bass_freq, float* const* out) {
  float* mid = falloc(n); float* side = falloc(n); float* ps = side;
  for (int64_t i = 0; i < n; i++) {
    mid[i] = F32(((double)in[0][i] + (double)in[1][i]) * 0.5);
    side[i] = F32(((double)in[0][i] - (double)in[1][i]) * 0.5);
  }
  if (bass_mono && bass_freq > 0) {
    const double om = 2 * M_PI * bass_freq / sr;
    const double cs = cos(om), sn = sin(om);
    const double alpha = sn / (2 * 0.7071);
    ora_biquad b;
    b.b0 = ((1 + cs) / 2) / (1 + alpha); b.b1 = (-(1 + cs)) / (1 + alpha); b.b2 = ((1 + cs) / 2) / (1 + alpha);
    b.a1 = (-2 * cs) / (1 + alpha); b.a2 = (1 - alpha) / (1 + alpha);
    ps = falloc(n);
    ora_biquad_filter(side, n, &b, ps);
  }
  const double mg = pow(10, 0 / 20.0);
  const double sg = pow(10, 0 / 20.0) * width;
  for (int64_t i = 0; i < n; i++) {
    const float pm = F32((double)mid[i] * mg);
    const float pf = F32((double)ps[i] * sg);
    float l = F32((double)pm + (double)pf);
    float r = F32((double)pm - (double)pf);
    l = F32((double)l * 1.0); r = F32((double)r * 1.0);   /* balance 0 */
    out[0][i] = l; out[1][i] = r;
  }
  if (ps != side) free(ps);
  free(mid); free(side);
}

/* ref: web/lib/dsp/normalizer.js:38-94 normalizeToLUFS with {skipLimiter:true} */
int ora_normalize_skip_limiter(const float* const* in, int nch, int64_t n, double sr, double target, float* const* out,
                               double* lufs_in, double* tp_in) {
  const double cur = ora_measure_lufs(in, nch, n, sr, target, NULL, NULL, NULL);
  const double tp = ora_true_peak_db(in, nch, n);
  if (lufs_in) *lufs_in = cur;
  if (tp_in) *tp_in = tp;
  if (!isfinite(cur)) {
    for (int c = 0; c < nch; c++) if (out[c] != in[c]) memcpy(out[c], in[c], sizeof(float) * (size_t)n);
    return 0;
  }
  const double gdb = target - cur;
  const double g = pow(10, gdb / 20);
  for (int c = 0; c < nch; c++) for (int64_t i = 0; i < n; i++) out[c][i] = F32((double)in[c][i] * g);
  return 1;
}

/* ref: web/lib/dsp/soft-clipper.js:165-294 applyMasteringSoftClip */
void ora_soft_clip(const float* const* in, int nch, int64_t n, double sr, double ceiling_db, double lookahead_ms,
                   double release_ms, double drive, float* const* out) {
  const double ceil_lin = db_to_linear(ceiling_db);
  const double thr = db_to_linear(ceiling_db + 3);
  const int64_t L = (int64_t)floor(sr * lookahead_ms / 1000);
  const double rc = exp(-1 / (release_ms * sr / 1000));
  float* g = falloc(n);
  for (int64_t i = 0; i < n; i++) g[i] = 1.0f;
  for (int c = 0; c < nch; c++) {
    const float* x = in[c];
    for (int64_t i = 0; i < n; i++) {
      const double a = fabs((double)x[i]);
      if (a > thr) {
        const double excess = a - thr;
        const double range = ceil_lin - thr;
        const double normalized = excess / fmax(range, 0.001);
        const double sat = tanh(normalized * drive);
        const double target = thr + sat * range;
        const double req = target / a;
        const int64_t st = i - L > 0 ? i - L : 0;
        for (int64_t j = st; j <= i; j++) {
          const double t = (double)(j - st) / (double)(i - st > 1 ? i - st : 1);
          const double ig = 1.0 + (req - 1.0) * t;
          g[j] = F32(fmin((double)g[j], ig));
        }
      }
    }
  }
  double cur = 1.0;
  for (int64_t i = 0; i < n; i++) {
    if ((double)g[i] < cur) cur = g[i];
    else { cur = rc * cur + (1 - rc) * 1.0; cur = fmin(cur, 1.0); }
    g[i] = F32(cur);
  }
#define XIN(k) (((k) >= 0 && (k) < n && x[k] == x[k]) ? (double)x[k] : 0.0)               /* input[k] || 0 */
#define GIN(k) (((k) >= 0 && (k) < n && g[k] == g[k] && g[k] != 0) ? (double)g[k] : 1.0)  /* gainEnvelope[k] || 1 */
  for (int c = 0; c < nch; c++) {
    const float* x = in[c];
    for (int64_t i = 0; i < n; i++) {
      const double p1 = XIN(i - 1), p2 = x[i], p3 = XIN(i + 1), p4 = XIN(i + 2);
      const double g1 = GIN(i - 1), g2 = g[i], g3 = GIN(i + 1), g4 = GIN(i + 2);
      const double s1 = p2 * g2;
      const double sB = catmull(p1 * g1, p2 * g2, p3 * g3, p4 * g4, 0.5);
      double v[2] = {s1, sB};
      for (int k = 0; k < 2; k++) {
        const double a = fabs(v[k]);
        if (a > ceil_lin) {
          const double ex = (a - ceil_lin) / ceil_lin;
          const double red = 1 - tanh(ex * 2) * 0.1;
          v[k] = js_sign(v[k]) * fmin(a * red, ceil_lin);
        }
      }
      out[c][i] = F32((v[0] + v[1]) * 0.5);
    }
  }
#undef XIN
#undef GIN
  free(g);
}

/* ref: web/workers/dsp-worker.js:197-229 applySoftKneeCurve */
static double soft_knee_curve(double s, double ceiling, double knee_db) {
  const double a = fabs(s);
  if (a <= ceiling * 0.9) return s;
  const double kr = pow(10, knee_db / 20);
  const double ks = ceiling / kr;
  if (a <= ks) return s;
  if (a <= ceiling) {
    const double t = (a - ks) / (ceiling - ks);
    const double blend = t * t * (3 - 2 * t);
    const double o = a + (ceiling - a) * blend * 0.5;
    return js_sign(s) * o;
  }
  const double excess = a - ceiling;
  const double nrm = excess / ceiling;
  const double comp = 1 - tanh(nrm * 2) * 0.1;
  const double o = fmin(ceiling, a * comp);
  return js_sign(s) * o;
}

/* ref: web/workers/dsp-worker.js:234-281 applySoftKneeOversampled (worker twin: seeds |y2|) */
void ora_soft_knee_oversampled(const float* in, int64_t n, double ceiling, double knee_db, float* out) {
  double p[4] = {0, 0, 0, 0};
  for (int64_t i = 0; i < n; i++) {
    p[0] = p[1]; p[1] = p[2]; p[2] = p[3]; p[3] = in[i];
    if (i < 3) { out[i] = F32(soft_knee_curve(in[i], ceiling, knee_db)); continue; }
    const double m = ora_true_peak_sample(p);
    if (m > ceiling) { const double gr = ceiling / m; out[i] = F32(soft_knee_curve((double)in[i] * gr, ceiling, knee_db)); }
    else out[i] = F32(soft_knee_curve(in[i], ceiling, knee_db));
  }
}

/* ref: web/lib/dsp/limiter.js:81-128 applySoftKneeOversampled (lib twin: the running maximum starts at 0, :107, and the
 * interpolation is interpolateCatmullRom, utils.js:143-153) */
void ora_soft_knee_oversampled_lib(const float* in, int64_t n, double ceiling, double knee_db, float* out) {
  double p[4] = {0, 0, 0, 0};
  for (int64_t i = 0; i < n; i++) {
    p[0] = p[1]; p[1] = p[2]; p[2] = p[3]; p[3] = in[i];
    if (i < 3) { out[i] = F32(soft_knee_curve(in[i], ceiling, knee_db)); continue; }
    const double a0 = -0.5 * p[0] + 1.5 * p[1] - 1.5 * p[2] + 0.5 * p[3];
    const double a1 = p[0] - 2.5 * p[1] + 2 * p[2] - 0.5 * p[3];
    const double a2 = -0.5 * p[0] + 0.5 * p[2];
    double m = 0;
    for (int j = 1; j <= 3; j++) {
      const double t = j * 0.25, t2 = t * t, t3 = t2 * t;
      const double v = fabs(a0 * t3 + a1 * t2 + a2 * t + p[1]);
      if (v > m) m = v;
    }
    if (m > ceiling) { const double gr = ceiling / m; out[i] = F32(soft_knee_curve((double)in[i] * gr, ceiling, knee_db)); }
    else out[i] = F32(soft_knee_curve(in[i], ceiling, knee_db));
  }
}

static void limiter_impl(const float* const* in, int nch, int64_t n, double sr, double ceil_lin, double lookahead_ms,
                         double release_ms, double knee_db, int preserve, float* const* out, int lib_twin);
/* ref: web/workers/dsp-worker.js:286-452 applyLookaheadLimiterToChannels */
void ora_limiter(const float* const* in, int nch, int64_t n, double sr, double ceil_lin, double lookahead_ms,
                 double release_ms, double knee_db, int preserve, float* const* out) {
  limiter_impl(in, nch, n, sr, ceil_lin, lookahead_ms, release_ms, knee_db, preserve, out, 0);
}
/* ref: web/lib/dsp/limiter.js:155-362 applyLookaheadLimiter - the same stages; only its soft-knee pass differs (above) */
void ora_limiter_lib(const float* const* in, int nch, int64_t n, double sr, double ceil_lin, double lookahead_ms,
                     double release_ms, double knee_db, int preserve, float* const* out) {
  limiter_impl(in, nch, n, sr, ceil_lin, lookahead_ms, release_ms, knee_db, preserve, out, 1);
}
static void limiter_impl(const float* const* in, int nch, int64_t n, double sr, double ceil_lin, double lookahead_ms,
                         double release_ms, double knee_db, int preserve, float* const* out, int lib_twin) {
  const int64_t L = (int64_t)floor(sr * lookahead_ms / 1000);
  const double rc = exp(-1 / (release_ms * sr / 1000));
  float* tm = NULL;
  if (preserve) {
    tm = falloc(n);
    const int64_t W = (int64_t)floor(sr * 10 / 1000);
    float* rms = falloc(n);
    double rsum = 0;
    for (int64_t i = 0; i < n; i++) {
      double ss = 0;
      for (int c = 0; c < nch; c++) ss += (double)in[c][i] * (double)in[c][i];
      ss /= nch;
      rsum += ss;
      if (i >= W) {
        double os = 0;
        for (int c = 0; c < nch; c++) os += (double)in[c][i - W] * (double)in[c][i - W];
        rsum -= os / nch;
      }
      const int64_t ws = i + 1 < W ? i + 1 : W;
      rms[i] = F32(sqrt(rsum / (double)ws));
    }
    float* pk = falloc(n);
    const double pa = exp(-1 / (0.001 * sr)), pr = exp(-1 / (0.050 * sr));
    double lvl = 0;
    for (int64_t i = 0; i < n; i++) {
      double p = 0;
      for (int c = 0; c < nch; c++) p = fmax(p, fabs((double)in[c][i]));
      if (p > lvl) lvl = pa * lvl + (1 - pa) * p;
      else lvl = pr * lvl + (1 - pr) * p;
      pk[i] = F32(lvl);
    }
    for (int64_t i = 0; i < n; i++) {
      if (rms[i] > 0.0001) {
        const double crest = 20 * log10((double)pk[i] / (double)rms[i]);
        tm[i] = F32(fmin(1, fmax(0, (crest - 6) / (10 - 6))));
      } else tm[i] = 0;
    }
    free(rms); free(pk);
  }
  float* g = falloc(n);
  for (int64_t i = 0; i < n; i++) g[i] = 1.0f;
  double pl[4] = {0, 0, 0, 0}, prr[4] = {0, 0, 0, 0};
  for (int64_t i = 0; i < n; i++) {
    pl[0] = pl[1]; pl[1] = pl[2]; pl[2] = pl[3]; pl[3] = in[0][i];
    double tp = 0;
    if (i >= 3) tp = ora_true_peak_sample(pl);
    if (nch > 1) {
      prr[0] = prr[1]; prr[1] = prr[2]; prr[2] = prr[3]; prr[3] = in[1][i];
      if (i >= 3) tp = fmax(tp, ora_true_peak_sample(prr));
    }
    double ec = ceil_lin;
    if (preserve && tm && tm[i] > 0.5) ec = ceil_lin * pow(10, (double)tm[i] * 0.5 / 20);
    double req = 1.0;
    if (tp > ec) req = ec / tp;
    const int64_t t = i - L > 0 ? i - L : 0;
    if (req < (double)g[t]) {
      for (int64_t j = t; j <= i; j++) {
        const double prog = (double)(j - t) / (double)L;
        const double sm = (double)g[t] + (req - (double)g[t]) * prog;
        g[j] = F32(fmin((double)g[j], sm));
      }
    }
  }
  double cur = 1.0;
  for (int64_t i = 0; i < n; i++) {
    if ((double)g[i] < cur) cur = g[i];
    else { cur = rc * cur + (1 - rc) * 1.0; cur = fmin(cur, 1.0); }
    g[i] = F32(cur);
  }
  float* tmp = falloc(n);
  for (int c = 0; c < nch; c++) {
    for (int64_t i = 0; i < n; i++) tmp[i] = F32((double)in[c][i] * (double)g[i]);
    if (lib_twin) ora_soft_knee_oversampled_lib(tmp, n, ceil_lin, knee_db, out[c]);
    else ora_soft_knee_oversampled(tmp, n, ceil_lin, knee_db, out[c]);
  }
  free(tmp); free(g); free(tm);
}

/* ref: web/workers/dsp-worker.js:457-527 normalizeChannelsToLUFS (message NORMALIZE) */
void ora_normalize(const float* const* in, int nch, int64_t n, double sr, double target, double ceiling_db,
                   float* const* out, wam_normalize_result* res) {
  const double cur = ora_measure_lufs(in, nch, n, sr, target, NULL, NULL, NULL);
  const double pk = ora_true_peak_db(in, nch, n);
  memset(res, 0, sizeof(*res));
  res->current_lufs = cur;
  if (!isfinite(cur)) {
    for (int c = 0; c < nch; c++) memcpy(out[c], in[c], sizeof(float) * (size_t)n);
    res->final_lufs = cur; res->peak_db = pk; res->measured = 0;
    return;
  }
  res->measured = 1;
  const double gdb = target - cur;
  const double g = pow(10, gdb / 20);
  const double proj = pk + gdb;
  const double ceil_lin = pow(10, ceiling_db / 20);
  res->gain_applied_db = gdb;
  float* tmp[2];
  for (int c = 0; c < nch; c++) { tmp[c] = falloc(n); for (int64_t i = 0; i < n; i++) tmp[c][i] = F32((double)in[c][i] * g); }
  if (proj > ceiling_db) {
    ora_limiter((const float* const*)tmp, nch, n, sr, ceil_lin, 3, 100, 3, 1, out);
    res->peak_db = ora_true_peak_db((const float* const*)out, nch, n);
    res->final_lufs = ora_measure_lufs((const float* const*)out, nch, n, sr, -14, NULL, NULL, NULL);
    res->limiter_applied = 1;
  } else {
    for (int c = 0; c < nch; c++) memcpy(out[c], tmp[c], sizeof(float) * (size_t)n);
    res->final_lufs = ora_measure_lufs((const float* const*)out, nch, n, sr, -14, NULL, NULL, NULL);
    res->peak_db = proj;
  }
  for (int c = 0; c < nch; c++) free(tmp[c]);
}

/* ref: web/lib/dsp/exciter.js:80-214 Exciter.process (defaults constants.js:36-42), HPF :15-55 */
static void exciter_hpf(const float* in, int64_t n, const ora_biquad* c, float* out) {
  double x1 = 0, x2 = 0, y1 = 0, y2 = 0;
  for (int64_t i = 0; i < n; i++) {
    const double x0 = in[i];
    const double y0 = c->b0 * x0 + c->b1 * x1 + c->b2 * x2 - c->a1 * y1 - c->a2 * y2;
    x2 = x1; x1 = x0; y2 = y1;
    y1 = fabs(y0) < 1e-25 ? 0 : y0;
    out[i] = F32(y1);
  }
}
void ora_exciter(const float* const* in, int nch, int64_t n, double sr, float* const* out) {
  const double omega = tan(M_PI * 6000 / sr);
  const double om2 = omega * omega;
  const double nn = 1 / (1 + M_SQRT2 * omega + om2);
  ora_biquad c = { nn, -2 * nn, nn, 2 * (om2 - 1) * nn, (1 - M_SQRT2 * omega + om2) * nn };
  const double mix = 0.15 / 100, drive = 0.5, bias = 0.2;
  float* f1 = falloc(n); float* f = falloc(n);
  for (int ch = 0; ch < nch; ch++) {
    exciter_hpf(in[ch], n, &c, f1);
    exciter_hpf(f1, n, &c, f);
#define FIN(k) (((k) >= 0 && (k) < n && f[k] == f[k]) ? (double)f[k] : 0.0)
    for (int64_t i = 0; i < n; i++) {
      const double dry = in[ch][i];
      const double p1 = FIN(i - 1), p2 = FIN(i), p3 = FIN(i + 1), p4 = FIN(i + 2);
      const double wetA = tanh(p2 * drive + bias);
      const double sB = catmull(p1, p2, p3, p4, 0.5);
      const double wetB = tanh(sB * drive + bias);
      const double wet = (wetA + wetB) * 0.5;
      out[ch][i] = F32(dry + wet * mix);
    }
#undef FIN
  }
  free(f1); free(f);
}

/* Radix-2 double DFT standing in for npm fft.js@4.0.4 (absent from the reference
 * checkout, package-lock.json:3466-3470): unnormalised forward, 1/N inverse.
 * PARITY UNPINNED at this boundary; any exact DFT agrees to ~1e-13 relative. */
typedef struct { int n, bits; uint32_t* rev; double* cs; double* sn; } dft_plan;
static dft_plan* dft_new(int n) {
  dft_plan* p = (dft_plan*)malloc(sizeof(dft_plan));
  p->n = n; p->bits = 0; while ((1 << p->bits) < n) p->bits++;
  p->rev = (uint32_t*)malloc(sizeof(uint32_t) * n);
  for (int i = 0; i < n; i++) { uint32_t r = 0, x = (uint32_t)i; for (int b = 0; b < p->bits; b++) { r = (r << 1) | (x & 1); x >>= 1; } p->rev[i] = r; }
  p->cs = (double*)malloc(sizeof(double) * n / 2); p->sn = (double*)malloc(sizeof(double) * n / 2);
  for (int i = 0; i < n / 2; i++) { const double a = -2 * M_PI * i / n; p->cs[i] = cos(a); p->sn[i] = sin(a); }
  return p;
}
static void dft_free(dft_plan* p) { free(p->rev); free(p->cs); free(p->sn); free(p); }
static void dft_run(const dft_plan* p, double* out, const double* data, int inv) {
  const int n = p->n;
  for (int i = 0; i < n; i++) { const uint32_t j = p->rev[i]; out[2 * j] = data[2 * i]; out[2 * j + 1] = data[2 * i + 1]; }
  for (int size = 2; size <= n; size <<= 1) {
    const int half = size >> 1, step = n / size;
    for (int i = 0; i < n; i += size) {
      for (int j = i, k = 0; j < i + half; j++, k += step) {
        const double c = p->cs[k], s = inv ? -p->sn[k] : p->sn[k];
        const double xr = out[2 * (j + half)], xi = out[2 * (j + half) + 1];
        const double tr = xr * c - xi * s, ti = xr * s + xi * c;
        out[2 * (j + half)] = out[2 * j] - tr; out[2 * (j + half) + 1] = out[2 * j + 1] - ti;
        out[2 * j] += tr; out[2 * j + 1] += ti;
      }
    }
  }
  if (inv) for (int i = 0; i < 2 * n; i++) out[i] /= n;
}

/* ref: web/lib/dsp/dynamic-processor.js:595-609 processHybridDynamic('mastering'):
 * preset :572-585, process :393-427, _processChannel :444-500, _computeBinGains :319-385,
 * ResonanceDetector.analyze :194-233, GainComputer :75-90, EnvelopeFollower :33-43;
 * FFT wrapper web/lib/dsp/fft.js:41-68; Hann utils.js:98-104 */
/* preset: 0 'gentle' (:538-549, the default of processHybridDynamic :595,:604-606), 1 'aggressive' (:555-566), 2 'mastering' */
void ora_deharsh_preset(const float* const* in, int nch, int64_t n, double sr, int preset, float* const* out) {
  enum { N = 2048, HOP = 512, NB = 1024, NBANDS = 7 };
  static const double lo[NBANDS] = {0, 80, 250, 1000, 3000, 6000, 12000};
  static const double hi[NBANDS] = {80, 250, 1000, 3000, 6000, 12000, 20000};
  static const double att[NBANDS] = {30, 20, 10, 8, 2, 2, 5};
  static const double rel[NBANDS] = {200, 150, 100, 80, 40, 30, 50};
  static const double thr0[NBANDS] = {-12, -15, -18, -20, -26, -28, -22};
  static const double rat0[NBANDS] = {2, 2.5, 3, 3.5, 5, 6, 3};
  static const double knee0[NBANDS] = {10, 8, 6, 6, 4, 3, 6};
  double thr[NBANDS], rat[NBANDS], knee[NBANDS], ks[NBANDS], ke[NBANDS], ac[NBANDS], rcf[NBANDS];
  for (int b = 0; b < NBANDS; b++) {
    if (preset == 2) { thr[b] = thr0[b] + 12; knee[b] = knee0[b] + 8; rat[b] = fmax(1.2, rat0[b] - 1.5); }
    else if (preset == 1) { thr[b] = thr0[b] - 6; knee[b] = knee0[b]; rat[b] = rat0[b] + 2; }
    else { thr[b] = thr0[b] + 6; knee[b] = knee0[b]; rat[b] = fmax(2, rat0[b] - 1); }
    ks[b] = thr[b] - knee[b] / 2; ke[b] = thr[b] + knee[b] / 2;
    ac[b] = exp(-1 / (sr * att[b] / 1000)); rcf[b] = exp(-1 / (sr * rel[b] / 1000));
  }
  /* constructor defaults :268-276 (dryWetMix 1.0, aiArtifactMode true) under what each preset sets */
  const double mix = preset == 2 ? 0.15 : 1.0, sens = preset == 2 ? 0.2 : preset == 1 ? 0.8 : 0.3, maxcut = preset == 2 ? -3 : preset == 1 ? -18 : -6;
  const int ai = preset != 2;
  float window[N];
  for (int i = 0; i < N; i++) window[i] = F32(0.5 * (1 - cos(2 * M_PI * i / (N - 1))));
  int band_of[NB];
  for (int bin = 0; bin < NB; bin++) {
    const double fr = bin * sr / N;
    int bi = NBANDS - 1;
    for (int b = 0; b < NBANDS; b++) if (fr >= lo[b] && fr < hi[b]) { bi = b; break; }
    band_of[bin] = bi;
  }
  dft_plan* plan = dft_new(N);
  double* tin = (double*)malloc(sizeof(double) * 2 * N);
  double* spec = (double*)malloc(sizeof(double) * 2 * N);
  double* inv = (double*)malloc(sizeof(double) * 2 * N);
  float mags[NB], avg[NB], gains[NB], res[NB];
  const int64_t nframes = n >= N ? (int64_t)floor((double)(n - N) / HOP) : 0;
  for (int c = 0; c < nch; c++) {
    float* o = out[c];
    memset(o, 0, sizeof(float) * (size_t)n);
    double env[NBANDS]; for (int b = 0; b < NBANDS; b++) env[b] = 0;
    int initialized = 0;
    memset(avg, 0, sizeof(avg));
    for (int64_t fr = 0; fr < nframes; fr++) {
      const int64_t pos = fr * HOP;
      for (int i = 0; i < N; i++) {
        const float v = in[c][pos + i];
        const float w = F32((double)((v == v) ? v : 0.0f) * (double)window[i]);
        tin[2 * i] = w; tin[2 * i + 1] = 0;
      }
      dft_run(plan, spec, tin, 0);
      for (int i = 1; i < N / 2; i++) { spec[2 * (N - i)] = spec[2 * i]; spec[2 * (N - i) + 1] = -spec[2 * i + 1]; }
      for (int b = 0; b < NB; b++) mags[b] = F32(sqrt(spec[2 * b] * spec[2 * b] + spec[2 * b + 1] * spec[2 * b + 1]));
      float be[NBANDS], bc[NBANDS], bg[NBANDS];
      for (int b = 0; b < NBANDS; b++) { be[b] = 0; bc[b] = 0; bg[b] = 0; }
      for (int b = 0; b < NB; b++) { be[band_of[b]] = F32((double)be[band_of[b]] + (double)mags[b] * (double)mags[b]); bc[band_of[b]] += 1; }
      for (int b = 0; b < NBANDS; b++) if (bc[b] > 0) be[b] = F32(sqrt((double)be[b] / (double)bc[b]));
      for (int b = 0; b < NBANDS; b++) {
        const double a = fabs((double)be[b]);
        if (a > env[b]) env[b] = ac[b] * env[b] + (1 - ac[b]) * a;
        else env[b] = rcf[b] * env[b] + (1 - rcf[b]) * a;
        const double edb = linear_to_db(env[b]);
        double gdb;
        if (edb <= ks[b]) gdb = 0;
        else if (edb >= ke[b]) { const double ex = edb - thr[b]; const double cp = ex / rat[b]; gdb = cp - ex; }
        else { const double kp = (edb - ks[b]) / knee[b]; const double full = (edb - thr[b]) * (1 - 1 / rat[b]); gdb = -full * kp * kp; }
        bg[b] = F32(gdb);
      }
      memset(res, 0, sizeof(res));
      if (!initialized) { for (int b = 0; b < NB; b++) avg[b] = mags[b]; initialized = 1; }
      else {
        for (int b = 0; b < NB; b++) {
          const double mag = mags[b];
          avg[b] = F32(0.3 * mag + (1 - 0.3) * (double)avg[b]);
          const double av = avg[b];
          if (av > 1e-10 && mag > av * 1.5) { const double ex = mag / av; res[b] = F32(fmin(1, (ex - 1.5) / 2)); }
        }
      }
      for (int b = 0; b < NB; b++) {
        double gdb = bg[band_of[b]];
        if (res[b] > 0) { const double cut = maxcut * (double)res[b] * sens; gdb += cut; }
        if (ai) { const double fq = b * sr / N; if (fq >= 5000 && fq <= 12000) gdb += gdb * 0.3; }           /* :375-378 */
        gains[b] = F32(db_to_linear(gdb));
      }
      for (int b = 0; b < NB; b++) {
        spec[2 * b] *= (double)gains[b]; spec[2 * b + 1] *= (double)gains[b];
        if (b > 0) { spec[2 * (N - b)] = spec[2 * b]; spec[2 * (N - b) + 1] = -spec[2 * b + 1]; }
      }
      dft_run(plan, inv, spec, 1);
      for (int i = 0; i < N; i++) {
        const float pr = F32(inv[2 * i] * (double)window[i]);
        if (pos + i < n) o[pos + i] = F32((double)o[pos + i] + (double)pr);
      }
    }
    for (int64_t i = 0; i < n; i++) o[i] = F32((double)o[i] / 4.0);
    if (mix < 1) for (int64_t i = 0; i < n; i++) o[i] = F32((double)in[c][i] * (1 - mix) + (double)o[i] * mix);   /* :419-423 */
  }
  free(tin); free(spec); free(inv); dft_free(plan);
}
void ora_deharsh(const float* const* in, int nch, int64_t n, double sr, float* const* out) { ora_deharsh_preset(in, nch, n, sr, 2, out); }

/* ref: web/lib/dsp/multiband-saturation.js:365-368 applyTapeWarmth: process :320-355,
 * createBypassEnvelope :204-269, saturateBand :280-311, splitBands :125-148, preset :27-36 */
void ora_warmth(const float* const* in, int nch, int64_t n, double sr, float* const* out) {
  const int64_t ws = (int64_t)floor(sr * 100 / 1000);
  const int64_t nw = (n + ws - 1) / ws;
  float* envl = falloc(nw); float* sm = falloc(nw); float* la = falloc(nw);
  const double thr = db_to_linear(-24), kst = db_to_linear(-24 - 6);
  for (int64_t w = 0; w < nw; w++) {
    const int64_t st = w * ws; int64_t en = st + ws; if (en > n) en = n;
    double peak = 0;
    for (int64_t i = st; i < en; i += 4) {
      double s = fabs((double)in[0][i]);
      if (nch > 1) s = fmax(s, fabs((double)in[1][i]));
      if (s > peak) peak = s;
    }
    if (peak >= thr) envl[w] = 1.0f;
    else if (peak >= kst) { const double r = (peak - kst) / (thr - kst); envl[w] = F32(r * r); }
    else envl[w] = 0.0f;
  }
  if (nw > 0) sm[0] = envl[0];
  for (int64_t i = 1; i < nw; i++) {
    const double coef = envl[i] > sm[i - 1] ? 0.5 : 0.03;
    sm[i] = F32((double)sm[i - 1] + ((double)envl[i] - (double)sm[i - 1]) * coef);
  }
  const int64_t lw = (int64_t)ceil(sr * 5 / 1000 / (double)ws);
  const float* env = sm;
  if (lw > 0) { for (int64_t i = 0; i < nw; i++) { int64_t f = i + lw; if (f > nw - 1) f = nw - 1; la[i] = sm[f]; } env = la; }
  static const double drive[3] = {0.2, 0.4, 0.3}, bias[3] = {0.0, 0.1, 0.05}, mixb[3] = {0.3, 0.5, 0.4};
  ora_biquad lp1, hp1, lp2, hp2;
  rbj(sr, 200, 0.7071067811865476, 0, &lp1); rbj(sr, 200, 0.7071067811865476, 1, &hp1);
  rbj(sr, 4000, 0.7071067811865476, 0, &lp2); rbj(sr, 4000, 0.7071067811865476, 1, &hp2);
  float* t = falloc(n); float* band[3]; float* mh = falloc(n);
  for (int b = 0; b < 3; b++) band[b] = falloc(n);
  for (int c = 0; c < nch; c++) {
    ora_biquad_filter(in[c], n, &lp1, t); ora_biquad_filter(t, n, &lp1, band[0]);
    ora_biquad_filter(in[c], n, &hp1, t); ora_biquad_filter(t, n, &hp1, mh);
    ora_biquad_filter(mh, n, &lp2, t); ora_biquad_filter(t, n, &lp2, band[1]);
    ora_biquad_filter(mh, n, &hp2, t); ora_biquad_filter(t, n, &hp2, band[2]);
    for (int b = 0; b < 3; b++) {
      const double bo = tanh(drive[b] * bias[b]);
      const double gl = pow(10, 0 / 20.0);
      const double mk = 1 / tanh(fmax(drive[b], 0.1));
      for (int64_t i = 0; i < n; i++) {
        const double dry = band[b][i];
        const double wet = (tanh(drive[b] * (dry + bias[b])) - bo) * mk;
        int64_t wi = i / ws; if (wi > nw - 1) wi = nw - 1;
        const double sa = (double)env[wi] * mixb[b];
        band[b][i] = F32((dry * (1 - sa) + wet * sa) * gl);
      }
    }
    for (int64_t i = 0; i < n; i++) out[c][i] = F32((double)band[0][i] + (double)band[1][i] + (double)band[2][i]);
  }
  free(t); free(mh); for (int b = 0; b < 3; b++) free(band[b]);
  free(envl); free(sm); free(la);
}

/* ref: web/lib/dsp/multiband-transient.js:15-89 applyMultibandTransient, splitBands :94-132,
 * processTransients :194-263 */
static void punch_band(const float* band, int64_t n, double sr, double fa, double fr, double sa, double srl,
                       double tg_db, double sg_db, double smooth, float* out) {
  const double fac = exp(-1 / (fa * sr)), frc = exp(-1 / (fr * sr));
  const double sac = exp(-1 / (sa * sr)), src = exp(-1 / (srl * sr));
  const double smc = exp(-1 / (smooth * sr));
  const double tg = pow(10, tg_db / 20), sg = pow(10, sg_db / 20);
  double fe = 0, se = 0, g = 1;
  for (int64_t i = 0; i < n; i++) {
    const double s = band[i];
    const double a = fabs(s);
    if (a > fe) fe = fac * fe + (1 - fac) * a; else fe = frc * fe + (1 - frc) * a;
    if (a > se) se = sac * se + (1 - sac) * a; else se = src * se + (1 - src) * a;
    const double diff = fe - se;
    double target;
    if (diff > 0) { const double st = se > 0.0001 ? fmin(diff / se, 1) : fmin(diff / 0.0001, 1); target = 1 + (tg - 1) * st; }
    else { const double st = se > 0.0001 ? fmin(-diff / se, 1) : fmin(-diff / 0.0001, 1); target = 1 + (sg - 1) * st; }
    g = smc * g + (1 - smc) * target;
    out[i] = F32(s * g);
  }
}
void ora_punch(const float* const* in, int nch, int64_t n, double sr, float* const* out) {
  ora_biquad lp1, hp1, lp2, hp2;
  rbj(sr, 200, 0.707, 0, &lp1); rbj(sr, 200, 0.707, 1, &hp1); rbj(sr, 4000, 0.707, 0, &lp2); rbj(sr, 4000, 0.707, 1, &hp2);
  float* t1 = falloc(n); float* t2 = falloc(n); float* lo = falloc(n); float* mid = falloc(n); float* hi = falloc(n);
  float* pl = falloc(n); float* pm = falloc(n); float* ph = falloc(n);
  for (int c = 0; c < nch; c++) {
    ora_biquad_filter(in[c], n, &lp1, t1); ora_biquad_filter(t1, n, &lp1, lo);
    ora_biquad_filter(in[c], n, &hp1, t1); ora_biquad_filter(t1, n, &hp1, t2);
    ora_biquad_filter(t2, n, &lp2, t1); ora_biquad_filter(t1, n, &lp2, mid);
    ora_biquad_filter(t2, n, &hp2, t1); ora_biquad_filter(t1, n, &hp2, hi);
    punch_band(lo, n, sr, 0.005, 0.050, 0.025, 0.250, 2, -1, 0.020, pl);
    punch_band(mid, n, sr, 0.001, 0.040, 0.020, 0.200, 6, 0, 0.020, pm);
    punch_band(hi, n, sr, 0.003, 0.030, 0.015, 0.150, 2, 0, 0.020, ph);
    for (int64_t i = 0; i < n; i++) out[c][i] = F32((double)pl[i] + (double)pm[i] + (double)ph[i]);
  }
  free(t1); free(t2); free(lo); free(mid); free(hi); free(pl); free(pm); free(ph);
}

/* ref: web/ui/encoder.js:92-124 encodeWAV, 24-bit branch */
void ora_pcm24(const float* const* in, int nch, int64_t n, uint8_t* out) {
  int64_t off = 0;
  for (int64_t i = 0; i < n; i++) {
    for (int c = 0; c < nch; c++) {
      double s = in[c][i];
      s = fmax(-1, fmin(1, s));
      const double scaled = s * 8388607;
      double r = floor(scaled);
      if (scaled - r >= 0.5) r += 1;            /* Math.round: half toward +inf */
      if (r < -8388607) r = -8388607;
      if (r > 8388607) r = 8388607;
      const int32_t v = (int32_t)r;
      out[off] = (uint8_t)(v & 0xFF); out[off + 1] = (uint8_t)((v >> 8) & 0xFF); out[off + 2] = (uint8_t)((v >> 16) & 0xFF);
      off += 3;
    }
  }
}

/* ref: web/ui/encoder.js:92-115 encodeWAV, 16-bit branch.  mode 0 = 'none', 1 = 'tpdf', 2 = 'noise-shaped'.
 * dither[j] = the j-th value triangularDither() returns (encoder.js:18-30), one per sample-channel in interleaved
 * order; ignored for mode 0.  Math.max / Math.min propagate NaN, C's fmax / fmin do not: js_max / js_min below. */
static double js_max(double a, double b) { if (a != a || b != b) return NAN; return a > b ? a : b; }
static double js_min(double a, double b) { if (a != a || b != b) return NAN; return a < b ? a : b; }
void ora_pcm16(const float* const* in, int nch, int64_t n, int mode, const float* dither, uint8_t* out) {
  float err[2] = {0, 0};                         /* noiseShapeError: Float32Array(numChannels) */
  int64_t off = 0, di = 0;
  for (int64_t i = 0; i < n; i++) {
    for (int c = 0; c < nch; c++) {
      const double sample = js_max(-1, js_min(1, (double)in[c][i]));
      const double scaled = sample * 32767;
      double q = scaled;
      if (mode == 2) q += (double)err[c];
      if (mode != 0) q += (double)dither[di++];
      double r = floor(q);
      if (q - r >= 0.5) r += 1;                  /* Math.round */
      double cl = js_max(-32768, js_min(32767, r));
      int16_t v = (cl != cl) ? 0 : (int16_t)cl;  /* DataView.setInt16(NaN) stores 0 */
      out[off] = (uint8_t)(v & 0xFF); out[off + 1] = (uint8_t)((v >> 8) & 0xFF);
      if (mode == 2) err[c] = (float)js_max(-2, js_min(2, (q - cl) * 0.85));
      off += 2;
    }
  }
}

/* ref: web/ui/encoder.js:74-86 WAV header */
void ora_wav_header(int nch, int64_t n, int sr, int bits, uint8_t h[44]) {
  const uint32_t bps = (uint32_t)bits / 8, data = (uint32_t)(n * nch * bps);
  memcpy(h, "RIFF", 4); uint32_t v = 36 + data; memcpy(h + 4, &v, 4);
  memcpy(h + 8, "WAVEfmt ", 8); v = 16; memcpy(h + 16, &v, 4);
  uint16_t s = 1; memcpy(h + 20, &s, 2); s = (uint16_t)nch; memcpy(h + 22, &s, 2);
  v = (uint32_t)sr; memcpy(h + 24, &v, 4); v = (uint32_t)sr * nch * bps; memcpy(h + 28, &v, 4);
  s = (uint16_t)(nch * bps); memcpy(h + 32, &s, 2); s = (uint16_t)bits; memcpy(h + 34, &s, 2);
  memcpy(h + 36, "data", 4); memcpy(h + 40, &data, 4);
}

/* ref: web/ui/spectrogram.js:113-143 process, :217-257 fft, :40-48 tables, :177-215 column */
void ora_stft_db(const float* const* in, int nch, int64_t n, double sr, int nfft, int hop, float* out_db, float* columns) {
  int bits = 0; while ((1 << bits) < nfft) bits++;
  float* st = falloc(nfft); float* ct = falloc(nfft); float* win = falloc(nfft);
  float* re = falloc(nfft); float* im = falloc(nfft);
  for (int i = 0; i < nfft; i++) {
    const double ang = -2 * M_PI * i / nfft;
    st[i] = F32(sin(ang)); ct[i] = F32(cos(ang));
    win[i] = F32(0.5 * (1 - cos(2 * M_PI * i / nfft)));
  }
  const int half = nfft >> 1;
  const double corr = 10 * log10(16), norm = -20 * log10(nfft);
  const int64_t nframes = n >= nfft ? (n - nfft) / hop + 1 : 0;
  for (int64_t f = 0; f < nframes; f++) {
    const int64_t pos = f * hop;
    for (int i = 0; i < nfft; i++) {
      float m;
      if (nch >= 2) m = F32(0.5 * ((double)in[0][pos + i] + (double)in[1][pos + i]));   /* analyser mono down-mix */
      else m = in[0][pos + i];
      re[i] = F32((double)m * (double)win[i]); im[i] = 0;
    }
    for (int i = 0; i < nfft; i++) {
      int r = 0, x = i;
      for (int b = 0; b < bits; b++) { r = (r << 1) | (x & 1); x >>= 1; }
      if (r > i) { float t = re[i]; re[i] = re[r]; re[r] = t; t = im[i]; im[i] = im[r]; im[r] = t; }
    }
    for (int size = 2; size <= nfft; size <<= 1) {
      const int hs = size >> 1, step = nfft / size;
      for (int i = 0; i < nfft; i += size) {
        for (int j = i, k = 0; j < i + hs; j++, k += step) {
          const double cw = ct[k], sw = st[k];
          const double tr = (double)re[j + hs] * cw - (double)im[j + hs] * sw;
          const double ti = (double)re[j + hs] * sw + (double)im[j + hs] * cw;
          re[j + hs] = F32((double)re[j] - tr); im[j + hs] = F32((double)im[j] - ti);
          re[j] = F32((double)re[j] + tr); im[j] = F32((double)im[j] + ti);
        }
      }
    }
    float* sp = out_db + f * half;
    for (int i = 0; i < half; i++) {
      const double pw = (double)re[i] * (double)re[i] + (double)im[i] * (double)im[i];
      const double db = 10 * log10(pw + 1e-24) + corr + norm;
      sp[i] = F32(db < -144 ? -144 : db);
    }
    if (columns) {
      const double lmin = log10(20), lmax = log10(22000), lr = lmax - lmin, nyq = sr / 2;
      for (int y = 0; y < 256; y++) {
        const double fr = pow(10, lmax - ((double)y / 255) * lr);
        double v = -144;
        if (fr >= 20 && fr <= nyq) {
          const double bf = (fr * nfft) / sr;
          const int b1 = (int)floor(bf);
          if (b1 < half) { const int b2 = b1 + 1 < half ? b1 + 1 : b1; const double frac = bf - b1; v = (double)sp[b1] + ((double)sp[b2] - (double)sp[b1]) * frac; }
        }
        columns[f * 256 + y] = F32(v);
      }
    }
  }
  free(st); free(ct); free(win); free(re); free(im);
}

/* ref: web/workers/dsp-worker.js:1468-1667 RENDER_FULL_CHAIN */
int ora_render_chain(const float* const* in, int nch, int64_t n, double sr, const wam_settings* s, int mode,
                     float* const* out, wam_result* res) {
  float* a[2]; float* b[2];
  for (int c = 0; c < nch; c++) { a[c] = falloc(n); b[c] = falloc(n); memcpy(a[c], in[c], sizeof(float) * (size_t)n); }
#define SWAP() do { for (int c_ = 0; c_ < nch; c_++) { float* t_ = a[c_]; a[c_] = b[c_]; b[c_] = t_; } } while (0)
#define CA ((const float* const*)a)
  if (res) { res->lufs_in = NAN; res->gain_db = 0; }
  const double ig = (s->input_gain == s->input_gain) ? s->input_gain : 0;
  if (ig != 0) {
    const double gl = pow(10, ig / 20);
    for (int c = 0; c < nch; c++) for (int64_t i = 0; i < n; i++) a[c][i] = F32((double)a[c][i] * gl);
  }
  if (s->deharsh) { ora_deharsh(CA, nch, n, sr, b); SWAP(); }
  if (s->add_air) { ora_exciter(CA, nch, n, sr, b); SWAP(); }
  if (s->tape_warmth) { ora_warmth(CA, nch, n, sr, b); SWAP(); }
  if (s->add_punch) { ora_punch(CA, nch, n, sr, b); SWAP(); }
  if (mode == WAM_MODE_EXPORT) {
    ora_final_filters(CA, nch, n, sr, !!s->clean_low_end, 1, b); SWAP();
    { const double g[5] = {s->eq_low, s->eq_low_mid, s->eq_mid, s->eq_high_mid, s->eq_high};
      int any = s->cut_mud != 0; for (int k = 0; k < 5; k++) if (g[k] != 0 && g[k] == g[k]) any = 1;
      if (any) { double gg[5]; for (int k = 0; k < 5; k++) gg[k] = (g[k] == g[k]) ? g[k] : 0; ora_parametric_eq(CA, nch, n, sr, gg, s->cut_mud, b); SWAP(); } }
    if (s->glue_compression) { ora_glue(CA, nch, n, sr, -18, 3, 0.02, 0.25, b); SWAP(); }
    if (nch == 2) {
      const double w = isfinite(s->stereo_width) ? s->stereo_width / 100 : 1.0;
      const double cw = fmax(0, fmin(2, w));
      if (s->center_bass || fabs(cw - 1.0) > 1e-6) { ora_stereo(CA, n, sr, cw, !!s->center_bass, 200, b); SWAP(); }
    }
    if (s->normalize_loudness && s->target_lufs != 0 && s->target_lufs == s->target_lufs) {
      double li = 0, ti = 0;
      const int applied = ora_normalize_skip_limiter(CA, nch, n, sr, s->target_lufs, b, &li, &ti);
      if (res) { res->lufs_in = li; res->gain_db = applied ? s->target_lufs - li : 0; }
      SWAP();
    }
    if (s->true_peak_limit) {
      const double ceiling = (s->true_peak_ceiling != 0 && s->true_peak_ceiling == s->true_peak_ceiling) ? s->true_peak_ceiling : -1;
      ora_soft_clip(CA, nch, n, sr, ceiling, 0.5, 10, 1.5, b); SWAP();
      ora_limiter(CA, nch, n, sr, pow(10, ceiling / 20), 3, 100, 3, 1, b); SWAP();
    }
  }
  const double fl = ora_measure_lufs(CA, nch, n, sr, -14, NULL, NULL, NULL);
  if (res) { res->lufs = fl; res->measured_lufs = fl; res->true_peak_db = ora_true_peak_db(CA, nch, n); }
  for (int c = 0; c < nch; c++) { memcpy(out[c], a[c], sizeof(float) * (size_t)n); free(a[c]); free(b[c]); }
#undef SWAP
#undef CA
  return 0;
}
