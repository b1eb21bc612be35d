// extras.hip - the callers and data formats either side of the chain:
//   wam_dc_offset     detectDCOffsetBuffer / removeDCOffset          web/lib/dsp/dc-offset.js:13-77
//   wam_normalize     normalizeChannelsToLUFS (NORMALIZE message)    web/workers/dsp-worker.js:457-527
//   wam_encode_pcm24  encodeWAV 24-bit sample loop                   web/ui/encoder.js:92-124
//   wam_wav_header    encodeWAV header                               web/ui/encoder.js:74-86
//   wam_analyze_on_load  the load-time bundle: DC + LUFS + TP + normalizeToLUFS   web/app.js:823-872
//   wam_resample      resampleAudioBuffer (unpinned: Chromium's engine)  web/ui/renderer.js:172-187
//   wam_extract_peaks extractPeaks                                   web/ui/waveform.js:164-181
//   wam_stft_db       Spectrogram.process / fft / updateLatestColumn web/ui/spectrogram.js:113-257, batched over frames
#include "chain.hpp"
#include <cmath>
#include <cstring>
#include <vector>

namespace wam {

// ------------------------------------------------------------------ DC offset
// partial sums: 256 blocks per channel, fixed order -> deterministic
__global__ __launch_bounds__(256) void k_dc_partial(const float* __restrict__ in, int64_t n, double* __restrict__ part) {
  __shared__ double red[4];
  const int64_t chan = blockIdx.y;
  const float* x = in + chan * n;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t st = (int64_t)blockIdx.x * per;
  int64_t en = st + per; if (en > n) en = n;
  double s = 0;
  for (int64_t i = st + threadIdx.x; i < en; i += 256) s += (double)x[i];
  s = wave_sum_lane63(s);
  if (lane_id() == 63) red[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[chan * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void k_dc_final(const double* __restrict__ part, int nparts, int64_t n, double* __restrict__ dc) {
  const int chan = blockIdx.x;
  if (threadIdx.x != 0) return;
  double s = 0;
  for (int i = 0; i < nparts; i++) s += part[chan * nparts + i];
  dc[chan] = s / (double)n;
}
__global__ __launch_bounds__(256) void k_dc_sub(const float* __restrict__ in, float* __restrict__ out, int64_t n, const double* __restrict__ dc) {
  const int64_t chan = blockIdx.y;
  const double d = dc[chan];
  const bool sub = fabs(d) > 1e-10;                                  // dc-offset.js:70
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = in[chan * n + i];
    out[chan * n + i] = sub ? f32r((double)v - d) : v;
  }
}

// ------------------------------------------------------------------ waveform peaks
// extractPeaks (web/ui/waveform.js:164-181): max |x| over num_peaks windows of floor(n / num_peaks) samples of channel 0;
// NaN never wins the `abs > max` comparison, the tail beyond num_peaks windows is not looked at.  One workgroup per window.
__global__ __launch_bounds__(256) void k_peaks(const float* __restrict__ x, int64_t per, float* __restrict__ peaks) {
  __shared__ float red[4];
  const int64_t st = (int64_t)blockIdx.x * per;
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < per; i += 256) { const float a = fabsf(x[st + i]); if (a > m) m = a; }
  for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_down(m, off); if (o > m) m = o; }
  if (lane_id() == 0) red[wave_id()] = m;
  __syncthreads();
  if (threadIdx.x == 0) peaks[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------ sample-rate conversion
// resampleAudioBuffer (web/ui/renderer.js:172-187) hands the work to Chromium's OfflineAudioContext: an AudioBufferSourceNode
// whose buffer rate differs from the context rate.  Neither the engine nor fixtures of its output exist here: PARITY
// UNPINNED.  Two self-defined kernels, both evaluated at source position p = i * src / dst for output sample i, samples
// outside [0, n) being zero:
//   quality 0  linear interpolation between the two neighbouring source samples (what a buffer source node does per sample)
//   quality 1  Kaiser-windowed sinc, cutoff 0.98 x the lower of the two Nyquist rates, 64 zero crossings per side, beta = 8.6
//              (flat to 20 kHz, -0.2 dB at 21 kHz, -88 dB from 23 kHz for 48 -> 44.1 kHz)
__global__ __launch_bounds__(256) void k_resample_linear(const float* __restrict__ in, int64_t n, double ratio, float* __restrict__ out, int64_t n_out) {
  const float* x = in + (int64_t)blockIdx.y * n;
  float* y = out + (int64_t)blockIdx.y * n_out;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
    const double p = (double)i * ratio;
    const int64_t k = (int64_t)floor(p);
    const double f = p - (double)k;
    const double s1 = k < n ? (double)x[k] : 0.0, s2 = k + 1 < n ? (double)x[k + 1] : 0.0;
    y[i] = f32r((1.0 - f) * s1 + f * s2);
  }
}
__device__ __forceinline__ double bessel_i0(double v) {               // power series, |v| <= 9: 30 terms reach 1e-17 relative
  const double q = 0.25 * v * v;
  double term = 1.0, sum = 1.0;
#pragma unroll 1
  for (int k = 1; k <= 30; k++) { term *= q / ((double)k * (double)k); sum += term; }
  return sum;
}
constexpr double kSincBeta = 8.6;
constexpr int kSincZeros = 64;
constexpr double kSincBandwidth = 0.98;
__global__ __launch_bounds__(256) void k_resample_sinc(const float* __restrict__ in, int64_t n, double ratio, double fc, double inv_i0b,
                                                       float* __restrict__ out, int64_t n_out) {
  const float* x = in + (int64_t)blockIdx.y * n;
  float* y = out + (int64_t)blockIdx.y * n_out;
  const double H = (double)kSincZeros / fc;                            // half width in source samples
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
    const double p = (double)i * ratio;
    int64_t k0 = (int64_t)ceil(p - H), k1 = (int64_t)floor(p + H);
    if (k0 < 0) k0 = 0;
    if (k1 > n - 1) k1 = n - 1;
    double acc = 0;
    for (int64_t k = k0; k <= k1; k++) {
      const double t = p - (double)k, u = t / H;
      const double w = u * u < 1.0 ? bessel_i0(kSincBeta * sqrt(1.0 - u * u)) * inv_i0b : 0.0;
      const double a = fc * t;
      const double sc = a == 0.0 ? 1.0 : sinpi(a) / (M_PI * a);
      acc += (double)x[k] * (fc * sc * w);
    }
    y[i] = f32r(acc);
  }
}

// ------------------------------------------------------------------ 24-bit PCM
__device__ __forceinline__ int32_t quant24(float s) {
  double v = s;
  if (!(v == v)) return 0;                                            // NaN ends up as 0 through `& 0xFF`
  v = fmax(-1.0, fmin(1.0, v));
  const double r = floor(v * 8388607.0 + 0.5);                        // Math.round: half toward +inf
  return (int32_t)fmax(-8388607.0, fmin(8388607.0, r));
}
// 12 output bytes (3 dwords) per thread: 2 stereo frames or 4 mono frames
__global__ __launch_bounds__(256) void k_pcm24(const float* __restrict__ in, int nch, int64_t n, uint8_t* __restrict__ out) {
  const int fpt = 4 / nch;                                            // frames per thread
  const int64_t groups = (n + fpt - 1) / fpt;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
    uint32_t w[4];
    bool full = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t f = g * fpt + k / nch;
      const int c = k % nch;
      if (f < n) w[k] = (uint32_t)quant24(in[(int64_t)c * n + f]) & 0xFFFFFFu; else { w[k] = 0; full = false; }
    }
    uint8_t* o = out + g * 12;
    if (full) {
      uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
      o32[0] = w[0] | (w[1] << 24);
      o32[1] = (w[1] >> 8) | (w[2] << 16);
      o32[2] = (w[2] >> 16) | (w[3] << 8);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int64_t f = g * fpt + k / nch;
        if (f < n) { o[3 * k] = w[k] & 0xFF; o[3 * k + 1] = (w[k] >> 8) & 0xFF; o[3 * k + 2] = (w[k] >> 16) & 0xFF; }
      }
    }
  }
}

// ------------------------------------------------------------------ 16-bit quantiser (encoder.js:92-115)
// One thread per run of kPcm16Run frames, both channels, in the reference's loop order.  'none' and 'tpdf' have no
// state.  'noise-shaped' feeds 0.85 x the quantisation error of a channel into its next sample (f32 state).  A run
// rebuilds that state by starting kPcm16Warm frames early from zero error.  Two walks over the same input and dither
// that start from different errors differ by exactly 0.85 x the previous difference for as long as both round to the
// same integer, and meet for good once the difference drops under the f32 grain of the state (~110 such steps); a
// step on which they round differently re-expands the difference to at most 0.85, and the chance of that is about the
// size of the difference.  So this is NOT a strict contraction: convergence within the warm-up is probabilistic (the
// difference has to survive several hundred frames at a per-frame escape rate that shrinks geometrically), and it is
// validated rather than proven: identical bytes to the sequential walk on inputs that span many runs, in every mode
// (tests/test_gpu_extras.py::test_pcm16_modes_vs_reference_and_oracle).
constexpr int kPcm16Run = 2048, kPcm16Warm = 512;
__device__ __forceinline__ double js_max(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
__device__ __forceinline__ double js_min(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
__device__ __forceinline__ float tpdf_seeded(uint32_t seed, uint64_t j) {      // (u1 + u2) - 1, u in [0, 1): the shape of encoder.js:13
  auto h = [&](uint32_t a) { uint32_t x = (a + 1u) * 0x9E3779B1u ^ seed * 0x85EBCA77u ^ (uint32_t)(j >> 31) * 0xC2B2AE3Du;
                             x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; };
  const double u1 = (double)h((uint32_t)(2 * j)) * (1.0 / 4294967296.0), u2 = (double)h((uint32_t)(2 * j + 1)) * (1.0 / 4294967296.0);
  return f32r((u1 + u2) - 1);
}
// first NaN sample per channel (n if none): in 'noise-shaped' mode a NaN turns the channel's error state into NaN for
// good (encoder.js:110-113), so everything after it encodes as 0 - a dependence no warm-up can rebuild
__global__ __launch_bounds__(256) void k_first_nan(const float* __restrict__ in, int64_t n, unsigned long long* __restrict__ first) {
  const int64_t chan = blockIdx.y;
  unsigned long long best = (unsigned long long)n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = in[chan * n + i];
    if (v != v) { best = (unsigned long long)i; break; }
  }
  if (best < (unsigned long long)n) atomicMin(&first[chan], best);
}
__global__ __launch_bounds__(256) void k_pcm16(const float* __restrict__ in, int nch, int64_t n, int mode, const float* __restrict__ dither,
                                                uint32_t seed, const unsigned long long* __restrict__ first_nan, uint8_t* __restrict__ out) {
  const int64_t run = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i0 = run * kPcm16Run;
  if (i0 >= n) return;
  int64_t i1 = i0 + kPcm16Run; if (i1 > n) i1 = n;
  int64_t iw = i0; if (mode == 2) { iw = i0 - kPcm16Warm; if (iw < 0) iw = 0; }
  float err[2] = {0.0f, 0.0f};
  if (mode == 2) for (int c = 0; c < nch; c++) if ((unsigned long long)iw > first_nan[c]) err[c] = NAN;
  for (int64_t i = iw; i < i1; i++) {
#pragma unroll 2
    for (int c = 0; c < nch; c++) {
      const double sample = js_max(-1.0, js_min(1.0, (double)in[(int64_t)c * n + i]));
      double q = sample * 32767;
      if (mode == 2) q += (double)err[c];
      if (mode != 0) { const uint64_t j = (uint64_t)i * nch + c; q += (double)(dither ? dither[j] : tpdf_seeded(seed, j)); }
      double r = floor(q);
      if (q - r >= 0.5) r += 1;                                       // Math.round
      const double cl = js_max(-32768.0, js_min(32767.0, r));
      if (mode == 2) err[c] = f32r(js_max(-2.0, js_min(2.0, (q - cl) * 0.85)));
      if (i >= i0) {
        const int v = (cl != cl) ? 0 : (int)cl;
        uint8_t* o = out + 2 * ((int64_t)i * nch + c);
        o[0] = (uint8_t)(v & 0xFF); o[1] = (uint8_t)((v >> 8) & 0xFF);
      }
    }
  }
}

// ------------------------------------------------------------------ spectrogram STFT
struct StftRow { int bin1, bin2; double frac; int valid; };

// 10 log10(pw + 1e-24) + add_db, floored at -144 (spectrogram.js:135-141), for pw >= 0 finite.  x = 2^E m, m in [0.5, 1); the top
// five mantissa bits pick one of 32 intervals with centre c, the table holds (1 / c, 10 log10 c) - the second for the double that
// 1 / c actually is, from the host's long double log10 - and r = m (1 / c) - 1 (one FMA, |r| <= 1 / 64) goes through the series of
// 10 log10(1 + r) to r^8 (truncation 6e-18 in the natural log): ~22 instructions against ~33 for a division and an atanh series
// and ~90 for ocml's log10.  The result is an f32 store, and all three STFT kernels share this function (they must agree to the bit).
constexpr int kLogTab = 32;
#ifndef WAM_STFT_DB_GROUP
#define WAM_STFT_DB_GROUP 4
#endif
// (in two halves, so that a kernel with nothing else to hide the table read's latency can start the next value's read first)
struct StftDbHead { double m; int E; double2 t; };
__device__ __forceinline__ StftDbHead stft_db_head(double pw, const double2* __restrict__ lt) {
  const double x = pw + 1e-24;
  StftDbHead h;
  h.m = __builtin_amdgcn_frexp_mant(x);                   // [0.5, 1)
  h.E = __builtin_amdgcn_frexp_exp(x);
  h.t = lt[(__double2hiint(h.m) >> 15) & (kLogTab - 1)];
  return h;
}
__device__ __forceinline__ float stft_db_tail(const StftDbHead& h, double add_db) {
  const double r = fma(h.m, h.t.x, -1.0);
  constexpr double K = 4.34294481903251827651;            // 10 / ln 10
  double q = -K / 8;
  q = fma(q, r, K / 7);
  q = fma(q, r, -K / 6);
  q = fma(q, r, K / 5);
  q = fma(q, r, -K / 4);
  q = fma(q, r, K / 3);
  q = fma(q, r, -K / 2);
  q = fma(q, r, K);
  const double db = fma((double)h.E, 3.01029995663981195214, fma(r, q, h.t.y)) + add_db;
  return f32r(db < -144 ? -144.0 : db);
}
__device__ __forceinline__ float stft_db(double pw, double add_db, const double2* __restrict__ lt) { return stft_db_tail(stft_db_head(pw, lt), add_db); }
__global__ __launch_bounds__(256) void k_stft(const float* __restrict__ in, int nch, int64_t n, int nfft, int bits, int hop, int64_t nframes,
                                              const float* __restrict__ cosT, const float* __restrict__ sinT, const float* __restrict__ win,
                                              double add_db, float* __restrict__ out_db, const StftRow* __restrict__ rows, float* __restrict__ cols,
                                              const double2* __restrict__ logt) {
  extern __shared__ float sm[];
  float* re = sm; float* im = sm + nfft; float* sp = sm + 2 * nfft;
  const int half = nfft >> 1;
  in += (int64_t)blockIdx.y * nch * n;                                 // track
  out_db += (int64_t)blockIdx.y * nframes * half;
  if (cols) cols += (int64_t)blockIdx.y * nframes * 256;
  for (int64_t f = blockIdx.x; f < nframes; f += gridDim.x) {
    const int64_t pos = f * hop;
    __syncthreads();
    for (int i = threadIdx.x; i < nfft; i += 256) {
      float m;
      if (nch >= 2) m = f32r(0.5 * ((double)in[pos + i] + (double)in[n + pos + i])); else m = in[pos + i];
      const unsigned r = __brev((unsigned)i) >> (32 - bits);
      re[r] = f32r((double)m * (double)win[i]);                       // spectrogram.js:121-123, bit reversal :222-235
      im[r] = 0.0f;
    }
    __syncthreads();
    for (int size = 2; size <= nfft; size <<= 1) {                    // spectrogram.js:238-256, f32 store after every butterfly
      const int hs = size >> 1, step = nfft / size;
      for (int b = threadIdx.x; b < half; b += 256) {
        const int k = b % hs, j = (b / hs) * size + k;
        const double cw = cosT[k * step], sw = sinT[k * step];
        const double xr = re[j + hs], xi = im[j + hs];
        const double tr = xr * cw - xi * sw;
        const double ti = xr * sw + xi * cw;
        const double ar = re[j], ai = im[j];
        re[j + hs] = f32r(ar - tr); im[j + hs] = f32r(ai - ti);
        re[j] = f32r(ar + tr); im[j] = f32r(ai + ti);
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < half; i += 256) {
      const double pw = (double)re[i] * (double)re[i] + (double)im[i] * (double)im[i];
      const float v = stft_db(pw, add_db, logt);                      // spectrogram.js:135-141
      sp[i] = v;
      out_db[f * half + i] = v;
    }
    if (cols) {
      __syncthreads();
      for (int yy = threadIdx.x; yy < 256; yy += 256) {
        const StftRow r = rows[yy];
        double v = -144;
        if (r.valid) v = (double)sp[r.bin1] + ((double)sp[r.bin2] - (double)sp[r.bin1]) * r.frac;   // spectrogram.js:196-200
        cols[f * 256 + yy] = f32r(v);
      }
    }
  }
}

// ---- the same transform, register-blocked -----------------------------------------------------------------
// k_stft above walks the reference's loop nest literally: one LDS round trip and one barrier per radix-2 stage.
// For the two sizes that matter (2048: the reference's analyser, 4096: BASELINE configs[4]) the stages are grouped:
// a thread holds R = 2^NS points that are closed under NS consecutive stages (stride D between them), runs those
// stages in registers, and the workgroup meets in LDS only between groups (4096: 4+4+4 stages, 2048: 3+3+3+2).
// Every butterfly is the same expression on the same operands with the same f32 store as spectrogram.js:238-256,
// so the results are bit-identical to k_stft; only the order in which independent butterflies run differs.
__device__ __forceinline__ int padq(int i) { return i + (i >> 4); }

// NS stages on v[0..R): element m sits at position base + D m of the (bit-reversed-input) array; low = base mod D.
template <int R, int NS, int LOGN>
__device__ __forceinline__ void stft_stages(float2 v[R], int logD, int low, const float2* __restrict__ tw) {
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const int h = 1 << k;
#pragma unroll
    for (int m = 0; m < R; m++) {
      if (m & h) continue;
      const int kk = low + ((m & (h - 1)) << logD);                   // k of spectrogram.js:243 for this butterfly
      const float2 w = tw[kk << (LOGN - 1 - logD - k)];               // (cos, sin)[k * step]
      const double cw = w.x, sw = w.y;
      const double xr = v[m + h].x, xi = v[m + h].y;
      const double tr = xr * cw - xi * sw;
      const double ti = xr * sw + xi * cw;
      const double ar = v[m].x, ai = v[m].y;
      v[m + h] = make_float2(f32r(ar - tr), f32r(ai - ti));
      v[m] = make_float2(f32r(ar + tr), f32r(ai + ti));
    }
  }
}

template <int LOGN>
__global__ __launch_bounds__(256) void k_stft_fast(const float* __restrict__ in, int nch, int64_t n, int hop, int64_t nframes,
                                                   const float* __restrict__ cosT, const float* __restrict__ sinT, const float* __restrict__ win,
                                                   double add_db, float* __restrict__ out_db, const StftRow* __restrict__ rows, float* __restrict__ cols,
                                                   const double2* __restrict__ logt) {
  constexpr int N = 1 << LOGN, PT = N / 256, LP = LOGN - 8, half = N / 2;
  __shared__ float2 z[N + N / 16];
  __shared__ float2 tw[N / 2];
  __shared__ float sp[N / 2];
  const int t = threadIdx.x;
  in += (int64_t)blockIdx.y * nch * n;
  out_db += (int64_t)blockIdx.y * nframes * half;
  if (cols) cols += (int64_t)blockIdx.y * nframes * 256;
  for (int i = t; i < half; i += 256) tw[i] = make_float2(cosT[i], sinT[i]);
  __syncthreads();
  for (int64_t f = blockIdx.x; f < nframes; f += gridDim.x) {
    const int64_t pos = f * hop;
    float2 v[PT];
    // group 1: positions t PT .. t PT + PT - 1 of the bit-reversed array (spectrogram.js:121-123, :222-235)
#pragma unroll
    for (int m = 0; m < PT; m++) {
      const int i = (int)(__brev((unsigned)(t * PT + m)) >> (32 - LOGN));
      float mm;
      if (nch >= 2) mm = f32r(0.5 * ((double)in[pos + i] + (double)in[n + pos + i])); else mm = in[pos + i];
      v[m] = make_float2(f32r((double)mm * (double)win[i]), 0.0f);
    }
    stft_stages<PT, LP, LOGN>(v, 0, 0, tw);
#pragma unroll
    for (int m = 0; m < PT; m++) z[padq(t * PT + m)] = v[m];
    __syncthreads();
    // middle groups (4096: one, 2048: two): stride D = PT^g, R = PT points each; a task rewrites exactly the positions
    // it read, so the workgroup only has to meet between groups
    constexpr int LASTS = (LOGN % LP) ? (LOGN % LP) : LP, MID = (LOGN - LP - LASTS) / LP;
    int logD = LP;
#pragma unroll
    for (int g = 0; g < MID; g++) {
      const int D = 1 << logD;
      const int low = t & (D - 1), base = ((t >> logD) << (logD + LP)) + low;
#pragma unroll
      for (int m = 0; m < PT; m++) v[m] = z[padq(base + (m << logD))];
      stft_stages<PT, LP, LOGN>(v, logD, low, tw);
#pragma unroll
      for (int m = 0; m < PT; m++) z[padq(base + (m << logD))] = v[m];
      __syncthreads();
      logD += LP;
    }
    // last group: the remaining LOGN - logD stages, stride D = 2^logD, natural-order output
    constexpr int RL = 1 << LASTS, TASKS = N / RL / 256;
#pragma unroll
    for (int q = 0; q < TASKS; q++) {
      const int u = t + 256 * q;                       // = low: D = N / RL > u
      float2 w4[RL];
#pragma unroll
      for (int m = 0; m < RL; m++) w4[m] = z[padq(u + m * (N / RL))];
      stft_stages<RL, LASTS, LOGN>(w4, LOGN - LASTS, u, tw);
#pragma unroll
      for (int m = 0; m < RL / 2; m++) {               // bins below N / 2
        const int i = u + m * (N / RL);
        const double pw = (double)w4[m].x * (double)w4[m].x + (double)w4[m].y * (double)w4[m].y;
        const float o = stft_db(pw, add_db, logt);                      // spectrogram.js:135-141
        sp[i] = o;
        out_db[f * half + i] = o;
      }
    }
    __syncthreads();                                   // sp complete; z free for the next frame
    if (cols) {
      const StftRow r = rows[t];
      double o = -144;
      if (r.valid) o = (double)sp[r.bin1] + ((double)sp[r.bin2] - (double)sp[r.bin1]) * r.frac;   // spectrogram.js:196-200
      cols[f * 256 + t] = f32r(o);
      __syncthreads();
    }
  }
}


// ---- the 4096-point transform, one WAVEFRONT per frame ------------------------------------------------------------------
// BASELINE configs[4] (4096 / hop 1024 over the corpus).  A lane holds 64 points: positions 64 lane .. 64 lane + 63 of the
// bit-reversed array for stages 1-6 (all in registers), then - after ONE transposition through a wavefront-private LDS buffer -
// positions lane + 64 m for stages 7-12.  No workgroup barrier inside the frame loop (the eight wavefronts of a workgroup only
// share the read-only twiddle table); a wavefront that waits for its loads leaves its SIMD to the other resident one.
// Every butterfly is the expression of spectrogram.js:238-256 on the same operands with the same f32 stores, so the result
// equals k_stft's to the bit; what is NOT computed is what cannot change a bit:
//   * twiddle (1, -0) (k = 0 of every stage): tr = xr, ti = xi exactly, and the sum of two f32 values rounded to double and then
//     to f32 is the f32 sum (double rounding is innocuous for +, -, x when the wide format has >= 2 p + 2 bits: 53 >= 50);
//   * the imaginary parts are exactly zero until stage 2's k = 1 butterflies write the first non-zero ones;
//   * mono mix and window, f32(0.5 (a + b)) and f32(m w): the same theorem (0.5 x is exact);
//   * the twiddles of stages 1-6 are the same 32 numbers for every lane: scalar loads of a pre-widened table.
constexpr int kSwN = 4096, kSwPad = kSwN + 64, kSwWaves = 4;
__host__ __device__ constexpr int brev6c(int m) { return ((m & 1) << 5) | ((m & 2) << 3) | ((m & 4) << 1) | ((m & 8) >> 1) | ((m & 16) >> 3) | ((m & 32) >> 5); }
__device__ __forceinline__ void stft_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// stage S (1..6) on the lane's 64 consecutive positions; IMZ: every imaginary input of the stage is zero
template <int S, bool IMZ>
__device__ __forceinline__ void stft_local_stage(float re[64], float im[64], const double2* __restrict__ tw6) {
#pragma clang fp contract(off)       // an f32 sum must not be fused with the f32 product that made its operand (that product IS rounded)
  constexpr int h = 1 << (S - 1);
  int done = 0;
#pragma unroll
  for (int m = 0; m < 64; m++) {
    if (m & h) continue;
    // eight butterflies in flight, not all thirty-two: their widened operands would not fit beside the 128 registers of data (measured per
    // 1,024 tracks: 4 at a time 157 ms, 8: 152.5, 16: 157)
    if ((done++ & 7) == 0) __builtin_amdgcn_sched_barrier(0);
    const int kk = m & (h - 1);
    if (kk == 0) {
      const float ar = re[m], xr = re[m + h];
      re[m] = ar + xr; re[m + h] = ar - xr;
      if (!IMZ) { const float ai = im[m], xi = im[m + h]; im[m] = ai + xi; im[m + h] = ai - xi; }
    } else {
      const double2 w = tw6[kk << (6 - S)];              // (cos, sin)[kk 4096 / 2^S], widened on the host: a scalar load
      const double xr = re[m + h], ar = re[m];
      double tr, ti, ai;
      if (IMZ) { tr = xr * w.x; ti = xr * w.y; ai = 0.0; }             // xi = 0: its two products are zeros
      else { const double xi = im[m + h]; tr = xr * w.x - xi * w.y; ti = xr * w.y + xi * w.x; ai = im[m]; }
      re[m + h] = f32r(ar - tr); im[m + h] = f32r(ai - ti);
      re[m] = f32r(ar + tr); im[m] = f32r(ai + ti);
    }
  }
}
// stage 7 + J on positions lane + 64 m: butterfly (m, m + 2^J) has k = lane + 64 (m mod 2^J), table stride 2^(5 - J).
// Thirty-two butterflies in twiddle-major order (t = m mod h outer), eight at a time between scheduling fences; the twiddles of
// the NEXT eight are read from LDS before these eight are computed.  With one wavefront per SIMD nothing else covers an LDS
// round trip: read where they are used, the stage-12 twiddles alone (one per butterfly) exposed thirty-two of them per frame.
template <int J>
__device__ __forceinline__ void stft_cross_stage(float re[64], float im[64], const float2* twl, int lane) {
  constexpr int h = 1 << J, per_t = 32 / h;              // butterflies per twiddle
  constexpr int GB = 8, NG = 32 / GB;                    // butterflies per group, groups
  constexpr int TW = per_t >= GB ? 1 : GB / per_t;       // twiddles per group: 1, 1, 1, 2, 4, 8
  float2 cur[TW], nxt[TW];
  auto fetch = [&](int g, float2 w[TW]) {
#pragma unroll
    for (int i = 0; i < TW; i++) w[i] = twl[(lane + 64 * ((g * GB) / per_t + i)) << (5 - J)];
  };
  fetch(0, cur);
#pragma unroll
  for (int g = 0; g < NG; g++) {
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < NG) fetch(g + 1, nxt);
#pragma unroll
    for (int q = 0; q < GB; q++) {
      const int b = g * GB + q, t = b / per_t, m = t + 2 * h * (b % per_t);
      const float2 wf = cur[t - (g * GB) / per_t];
      const double cw = wf.x, sw = wf.y;
      const double xr = re[m + h], xi = im[m + h];
      const double tr = xr * cw - xi * sw;
      const double ti = xr * sw + xi * cw;
      const double ar = re[m], ai = im[m];
      re[m + h] = f32r(ar - tr); im[m + h] = f32r(ai - ti);
      re[m] = f32r(ar + tr); im[m] = f32r(ai + ti);
    }
    if (g + 1 < NG) {
#pragma unroll
      for (int i = 0; i < TW; i++) cur[i] = nxt[i];
    }
  }
}
template <int NCHT, bool COLS>
__global__ __launch_bounds__(64 * kSwWaves) void k_stft_wave(const float* __restrict__ in, int64_t n, int hop, int64_t nframes,
                                                             const float* __restrict__ cosT, const float* __restrict__ sinT, const float* __restrict__ win,
                                                             const double2* __restrict__ tw6, double add_db, float* __restrict__ out_db,
                                                             const StftRow* __restrict__ rows, float* __restrict__ cols, const double2* __restrict__ logt) {
#pragma clang fp contract(off)       // (see stft_local_stage; the double butterflies' products are exact, fused or not)
  __shared__ float2 twl[kSwN / 2];
  __shared__ float winl[kSwN];
  __shared__ float exch[kSwWaves][kSwPad];
  __shared__ double2 logl[kLogTab];
  if (threadIdx.x < kLogTab) logl[threadIdx.x] = logt[threadIdx.x];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int i = threadIdx.x; i < kSwN / 2; i += 64 * kSwWaves) twl[i] = make_float2(cosT[i], sinT[i]);
  for (int i = threadIdx.x; i < kSwN; i += 64 * kSwWaves) winl[i] = win[i];
  __syncthreads();
  float* buf = exch[wv];
  in += (int64_t)blockIdx.y * NCHT * n;
  out_db += (int64_t)blockIdx.y * nframes * (kSwN / 2);
  if (COLS) cols += (int64_t)blockIdx.y * nframes * 256;
  const unsigned rl = __brev((unsigned)lane) >> 26;
  // One wavefront per SIMD (frame, widened operands and the allocator's slack take ~430 of the 512 registers), so nothing else hides
  // a load: the NEXT frame's samples are requested when this frame's last stage is done - its 128 data registers are about to die -
  // and arrive while the dB values are computed; the stores that follow are younger than those loads, so the wait at the top of
  // the loop does not include them.
  const int64_t fstep = (int64_t)gridDim.x * kSwWaves;
  int64_t f = (int64_t)blockIdx.x * kSwWaves + wv;
  // Channel 0 lands in registers; channel 1 would be 64 more, which the allocator does not fit, so it goes straight from memory
  // into this wavefront's exchange buffer (global_load_lds: no register touched), which is idle from the transposition to the
  // next frame's: sample (lane, m) at buf[64 m + lane].  (Issued after the frame's last LDS read: the compiler orders every LDS
  // read behind outstanding loads of this kind.)
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  float pa[64];
  if (f < nframes) {
    const float* xa = in + f * hop;                    // uniform bases (scalar registers) + 32-bit lane offsets: four address registers, not 128
    const float* xb = xa + n;
#pragma unroll
    for (int m = 0; m < 64; m++) {
      const unsigned i = rl + (unsigned)(brev6c(m) << 6);
      pa[m] = xa[i];
      if (NCHT >= 2) __builtin_amdgcn_global_load_lds((gptr_t)(xb + i), (lptr_t)(buf + 64 * m), 4, 0, 0);
    }
  }
  for (; f < nframes; f += fstep) {
    // The window, the twiddles of both halves and the row table do not depend on the frame: left alone, the compiler hoists their
    // ~250 registers of loads out of this loop and spills.  An opaque zero added to the indices keeps each load where it is used.
    // (Likewise every index derived from the lane number: hoisted, the 190 of them are spilled; recomputed, each costs one add.)
    int zero = 0;
    asm volatile("" : "+s"(zero));
    unsigned rlo = rl;
    int lo = lane;
    asm volatile("" : "+v"(rlo), "+v"(lo));
    const float* winf = winl + zero;
    const double2* tw6f = tw6 + zero;
    const float2* twlf = twl + zero;
    const double2* logf = logl + zero;
    float re[64], im[64];
    // position 64 lane + m of the bit-reversed array holds sample brev12(64 lane + m) = 64 brev6(m) + brev6(lane)  (spectrogram.js:121-123, :222-235)
#pragma unroll
    for (int m = 0; m < 64; m++) {
      const unsigned i = rlo + (unsigned)(brev6c(m) << 6);
      float mm;
      if (NCHT >= 2) mm = 0.5f * (pa[m] + buf[64 * m + lo]); else mm = pa[m];
      re[m] = mm * winf[i];
      im[m] = 0.0f;
    }
    stft_local_stage<1, true>(re, im, tw6f);
    stft_local_stage<2, true>(re, im, tw6f);
    stft_local_stage<3, false>(re, im, tw6f);
    stft_local_stage<4, false>(re, im, tw6f);
    stft_local_stage<5, false>(re, im, tw6f);
    stft_local_stage<6, false>(re, im, tw6f);
    // transposition: position p sits at buf[p + (p >> 6)] (one pad per 64: both directions touch 64 different banks)
    stft_wave_sync();                                    // the previous frame's reads of buf precede these writes
#pragma unroll
    for (int m = 0; m < 64; m++) buf[65 * lo + m] = re[m];
    stft_wave_sync();
#pragma unroll
    for (int m = 0; m < 64; m++) re[m] = buf[lo + 65 * m];
    stft_wave_sync();
#pragma unroll
    for (int m = 0; m < 64; m++) buf[65 * lo + m] = im[m];
    stft_wave_sync();
#pragma unroll
    for (int m = 0; m < 64; m++) im[m] = buf[lo + 65 * m];
    stft_cross_stage<0>(re, im, twlf, lo);
    stft_cross_stage<1>(re, im, twlf, lo);
    stft_cross_stage<2>(re, im, twlf, lo);
    stft_cross_stage<3>(re, im, twlf, lo);
    stft_cross_stage<4>(re, im, twlf, lo);
    stft_cross_stage<5>(re, im, twlf, lo);
    float* o = out_db + f * (kSwN / 2) + lo;
    if (f + fstep < nframes) {                           // the next frame's samples: in flight during the dB stage
      const float* xa = in + (f + fstep) * hop;
      const float* xb = xa + n;
      if (NCHT >= 2 && !COLS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the transposition's reads of buf have returned
#pragma unroll
      for (int m = 0; m < 64; m++) {
        const unsigned i = rlo + (unsigned)(brev6c(m) << 6);
        pa[m] = xa[i];
        if (NCHT >= 2 && !COLS) __builtin_amdgcn_global_load_lds((gptr_t)(xb + i), (lptr_t)(buf + 64 * m), 4, 0, 0);
      }
    }
    if (COLS) stft_wave_sync();                          // the reads of buf above precede the writes below
    // bins lane + 64 m < 2048, in groups of four: the table reads of the next group are issued before this group's polynomials
    // (one wavefront per SIMD: nothing else would cover their latency)
    constexpr int G = WAM_STFT_DB_GROUP;
    StftDbHead hd[G];
#pragma unroll
    for (int i = 0; i < G; i++) hd[i] = stft_db_head((double)re[i] * (double)re[i] + (double)im[i] * (double)im[i], logf);
#pragma unroll
    for (int g = 0; g < 32 / G; g++) {
      __builtin_amdgcn_sched_barrier(0);
      StftDbHead nx[G];
      if (g + 1 < 32 / G) {
#pragma unroll
        for (int i = 0; i < G; i++) { const int m = G * (g + 1) + i; nx[i] = stft_db_head((double)re[m] * (double)re[m] + (double)im[m] * (double)im[m], logf); }
      }
#pragma unroll
      for (int i = 0; i < G; i++) {
        const int m = G * g + i;
        const float v = stft_db_tail(hd[i], add_db);
        o[64 * m] = v;
        if (COLS) buf[lo + 64 * m] = v;
      }
      if (g + 1 < 32 / G) {
#pragma unroll
        for (int i = 0; i < G; i++) hd[i] = nx[i];
      }
    }
    if (COLS) {
      stft_wave_sync();
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const StftRow r = rows[lo + 64 * q];
        double v = -144;
        if (r.valid) v = (double)buf[r.bin1] + ((double)buf[r.bin2] - (double)buf[r.bin1]) * r.frac;   // spectrogram.js:196-200
        cols[f * 256 + lo + 64 * q] = f32r(v);
      }
      if (NCHT >= 2 && f + fstep < nframes) {            // buf held the spectrum until here: channel 1 of the next frame only now
        const float* xb = in + (f + fstep) * hop + n;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < 64; m++) __builtin_amdgcn_global_load_lds((gptr_t)(xb + rlo + (unsigned)(brev6c(m) << 6)), (lptr_t)(buf + 64 * m), 4, 0, 0);
      }
    }
  }
}

static int check_in(const float* const* ch, int nch, int64_t n, int max_nch = 2) {
  if (!ch || nch < 1 || nch > max_nch || n < 0) { set_error("invalid audio arguments (nch=%d, n=%lld)", nch, (long long)n); return WAM_ERR_INVALID; }
  for (int c = 0; c < nch; c++) if (!ch[c] && n > 0) { set_error("channel %d is NULL", c); return WAM_ERR_INVALID; }
  return WAM_OK;
}
static int upload_chans(const float* const* ch, int nch, int64_t n, DevBuf& b) {
  int rc = b.alloc(sizeof(float) * (size_t)nch * (size_t)(n > 0 ? n : 1)); if (rc) return rc;
  std::vector<HostSeg> segs;
  for (int c = 0; c < nch && n > 0; c++) segs.push_back(HostSeg{ (void*)ch[c], b.as<float>() + (size_t)c * n, sizeof(float) * (size_t)n });
  return staged_copy(segs.data(), (int)segs.size(), true, nullptr);
}

}  // namespace wam

using namespace wam;

extern "C" {

int wam_dc_offset(const float* const* ch, int nch, int64_t n, double* dc, int remove, float* const* out) {
  int rc = check_in(ch, nch, n, 64); if (rc) return rc;               // detectDCOffsetBuffer loops over numberOfChannels (dc-offset.js:38-60)
  if (!dc || (remove && !out)) { set_error("dc / out is NULL"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  if (n == 0) { for (int c = 0; c < nch; c++) dc[c] = NAN; return WAM_OK; }     // 0 / 0 in the reference
  DevBuf in, o, part, dcd;
  rc = upload_chans(ch, nch, n, in); if (rc) return rc;
  const int nparts = 256;
  rc = part.alloc(sizeof(double) * nparts * nch); if (rc) return rc;
  rc = dcd.alloc(sizeof(double) * nch); if (rc) return rc;
  WAM_LAUNCH("k_dc_partial", k_dc_partial, dim3(nparts, nch), dim3(256), nullptr, (const float*)in.as<float>(), n, part.as<double>());
  WAM_LAUNCH("k_dc_final", k_dc_final, dim3(nch), dim3(64), nullptr, (const double*)part.as<double>(), nparts, n, dcd.as<double>());
  if (remove) {
    rc = o.alloc(sizeof(float) * (size_t)nch * (size_t)n); if (rc) return rc;
    int64_t bx = (n + 255) / 256; if (bx > 2048) bx = 2048;
    WAM_LAUNCH("k_dc_sub", k_dc_sub, dim3((unsigned)bx, nch), dim3(256), nullptr, (const float*)in.as<float>(), o.as<float>(), n, (const double*)dcd.as<double>());
    for (int c = 0; c < nch; c++) WAM_HIP(hipMemcpyAsync(out[c], o.as<float>() + (size_t)c * n, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, nullptr));
  }
  WAM_HIP(hipMemcpyAsync(dc, dcd.p, sizeof(double) * nch, hipMemcpyDeviceToHost, nullptr));
  WAM_HIP(hipStreamSynchronize(nullptr));
  return WAM_OK;
}

int64_t wam_resample_length(int64_t n, double src_rate, double dst_rate) {
  if (n <= 0 || !(src_rate > 0) || !(dst_rate > 0)) return 0;
  return (int64_t)std::ceil(((double)n / src_rate) * dst_rate);       // Math.ceil(sourceBuffer.duration * targetSampleRate), renderer.js:178
}

int wam_resample(const float* const* ch, int nch, int64_t n, double src_rate, double dst_rate, int quality, float* const* out, int64_t n_out) {
  int rc = check_in(ch, nch, n); if (rc) return rc;
  if (!out || !(src_rate > 0) || !(dst_rate > 0) || (quality != 0 && quality != 1)) { set_error("invalid resample arguments"); return WAM_ERR_INVALID; }
  if (n_out != wam_resample_length(n, src_rate, dst_rate)) {
    set_error("n_out must be wam_resample_length() = %lld, got %lld", (long long)wam_resample_length(n, src_rate, dst_rate), (long long)n_out);
    return WAM_ERR_INVALID;
  }
  for (int c = 0; c < nch; c++) if (!out[c] && n_out > 0) { set_error("out[%d] is NULL", c); return WAM_ERR_INVALID; }
  if (n_out == 0) return WAM_OK;
  rc = ensure_device(); if (rc) return rc;
  DevBuf in, o;
  rc = upload_chans(ch, nch, n, in); if (rc) return rc;
  rc = o.alloc(sizeof(float) * (size_t)nch * (size_t)n_out); if (rc) return rc;
  const double ratio = src_rate / dst_rate;
  int64_t bx = (n_out + 255) / 256; if (bx > 4096) bx = 4096;
  if (quality == 0) {
    WAM_LAUNCH("k_resample_linear", k_resample_linear, dim3((unsigned)bx, nch), dim3(256), nullptr, (const float*)in.as<float>(), n, ratio, o.as<float>(), n_out);
  } else {
    const double fc = (dst_rate < src_rate ? dst_rate / src_rate : 1.0) * kSincBandwidth;
    double i0b = 1, term = 1; const double q = 0.25 * kSincBeta * kSincBeta;
    for (int k = 1; k <= 30; k++) { term *= q / ((double)k * (double)k); i0b += term; }
    WAM_LAUNCH("k_resample_sinc", k_resample_sinc, dim3((unsigned)bx, nch), dim3(256), nullptr, (const float*)in.as<float>(), n, ratio, fc, 1.0 / i0b,
               o.as<float>(), n_out);
  }
  for (int c = 0; c < nch; c++) WAM_HIP(hipMemcpyAsync(out[c], o.as<float>() + (size_t)c * n_out, sizeof(float) * (size_t)n_out, hipMemcpyDeviceToHost, nullptr));
  WAM_HIP(hipStreamSynchronize(nullptr));
  return WAM_OK;
}

int wam_extract_peaks(const float* ch0, int64_t n, int num_peaks, float* peaks) {
  if (n < 0 || num_peaks < 0 || (n > 0 && !ch0) || (num_peaks > 0 && !peaks)) { set_error("invalid peak arguments"); return WAM_ERR_INVALID; }
  if (num_peaks == 0) return WAM_OK;
  const int64_t per = n / num_peaks;                                   // waveform.js:166
  if (per == 0) { std::memset(peaks, 0, sizeof(float) * (size_t)num_peaks); return WAM_OK; }   // empty windows: max stays 0
  int rc = ensure_device(); if (rc) return rc;
  DevBuf in, pk;
  const int64_t used = per * num_peaks;
  rc = in.alloc(sizeof(float) * (size_t)used); if (rc) return rc;
  rc = pk.alloc(sizeof(float) * (size_t)num_peaks); if (rc) return rc;
  WAM_HIP(hipMemcpyAsync(in.p, ch0, sizeof(float) * (size_t)used, hipMemcpyHostToDevice, nullptr));
  WAM_LAUNCH("k_peaks", k_peaks, dim3((unsigned)num_peaks), dim3(256), nullptr, (const float*)in.as<float>(), per, pk.as<float>());
  WAM_HIP(hipMemcpyAsync(peaks, pk.p, sizeof(float) * (size_t)num_peaks, hipMemcpyDeviceToHost, nullptr));
  WAM_HIP(hipStreamSynchronize(nullptr));
  return WAM_OK;
}

int wam_normalize(const float* const* ch, int nch, int64_t n, double sr, double target_lufs, double ceiling_db, float* const* out,
                  wam_normalize_result* res, wam_progress_cb progress, void* user) {
  int rc = check_in(ch, nch, n); if (rc) return rc;
  if (!out || !res || !(sr > 0)) { set_error("invalid normalize arguments"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  std::memset(res, 0, sizeof(*res));
  const Progress prog{progress, user, nullptr};
  rc = prog(0.05, "Measuring loudness..."); if (rc) return rc;                                 // dsp-worker.js:458
  MeterPlan mp; rc = make_meter_plan(nch, n, sr, target_lufs, &mp); if (rc) return rc;      // measure with fallback = target (:460)
  MeterPlan mp14 = mp; mp14.fallback = -14.0;                                                 // final measurements use the default (:504,:517)
  DevBuf in, tmp, o, ms, ms2, gd;
  rc = upload_chans(ch, nch, n, in); if (rc) return rc;
  rc = ms.alloc(meter_scratch_bytes(mp, 1, false)); if (rc) return rc;
  MeterScratch M; carve_meter_scratch(mp, 1, false, ms.p, &M);
  rc = enqueue_meter(in.as<float>(), 1, mp, n, M, nullptr); if (rc) return rc;
  GateOut g;
  WAM_HIP(hipMemcpy(&g, M.gate, sizeof(g), hipMemcpyDeviceToHost));
  const double cur = finish_lufs(mp, g), pk = finish_peak_db(g);
  res->current_lufs = cur;
  const size_t bytes = sizeof(float) * (size_t)(n > 0 ? n : 1);
  if (!std::isfinite(cur)) {                                                                  // :466-469
    for (int c = 0; c < nch && n > 0; c++) std::memcpy(out[c], ch[c], sizeof(float) * (size_t)n);
    res->final_lufs = cur; res->peak_db = pk; res->measured = 0;
    return prog(1.0, "Complete");                                                             // the handler's closing message, :1673
  }
  res->measured = 1;
  rc = prog(0.15, "Applying gain..."); if (rc) return rc;                                     // :471
  const double gdb = target_lufs - cur;                                                       // :473-476
  const double gl = std::pow(10, gdb / 20);
  const double proj = pk + gdb;
  res->gain_applied_db = gdb;
  rc = tmp.alloc(bytes * nch); if (rc) return rc;
  rc = gd.alloc(sizeof(double)); if (rc) return rc;
  WAM_HIP(hipMemcpy(gd.p, &gl, sizeof(double), hipMemcpyHostToDevice));
  rc = ms2.alloc(meter_scratch_bytes(mp14, 1, false)); if (rc) return rc;
  MeterScratch M2; carve_meter_scratch(mp14, 1, false, ms2.p, &M2);
  WAM_HIP(hipMemcpyAsync(M2.plan, &mp14, sizeof(mp14), hipMemcpyHostToDevice, nullptr));
  const float* result = tmp.as<float>();
  if (proj > ceiling_db) {                                                                    // :490-515
    rc = prog(0.25, "Applying limiter..."); if (rc) return rc;                                // :491
    LimiterPlan lp; rc = make_limiter_plan(sr, std::pow(10, ceiling_db / 20), 3, 100, 3, 1, &lp); if (rc) return rc;
    DevBuf plan, tmb, gv;
    rc = plan.alloc(sizeof(lp)); if (rc) return rc;
    WAM_HIP(hipMemcpyAsync(plan.p, &lp, sizeof(lp), hipMemcpyHostToDevice, nullptr));
    rc = tmb.alloc(bytes); if (rc) return rc;
    rc = gv.alloc(bytes); if (rc) return rc;
    rc = o.alloc(bytes * nch); if (rc) return rc;
    rc = enqueue_gain_meter(in.as<float>(), tmp.as<float>(), 1, nch, n, gd.as<double>(), mp14, M2, nullptr); if (rc) return rc;
    WAM_HIP(hipMemsetAsync(M2.hopsums, 0, M2.hopsums_bytes, nullptr));
    rc = enqueue_limiter(tmp.as<float>(), o.as<float>(), tmb.as<float>(), gv.as<float>(), 1, nch, n, plan.as<LimiterPlan>(), lp.L, 1,
                         M2.plan, M2.hopsums, M2.hop_stride, M2.peaks, sr, nullptr, &prog); if (rc) return rc;
    rc = enqueue_gate(mp14, 1, M2, nullptr); if (rc) return rc;
    WAM_HIP(hipMemcpy(&g, M2.gate, sizeof(g), hipMemcpyDeviceToHost));
    res->peak_db = finish_peak_db(g);
    res->final_lufs = finish_lufs(mp14, g);
    res->limiter_applied = 1;
    result = o.as<float>();
    for (int c = 0; c < nch && n > 0; c++) WAM_HIP(hipMemcpy(out[c], result + (size_t)c * n, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    return prog(1.0, "Complete");
  }
  rc = enqueue_gain_meter(in.as<float>(), tmp.as<float>(), 1, nch, n, gd.as<double>(), mp14, M2, nullptr); if (rc) return rc;
  WAM_HIP(hipMemcpy(&g, M2.gate, sizeof(g), hipMemcpyDeviceToHost));
  res->final_lufs = finish_lufs(mp14, g);
  res->peak_db = proj;
  for (int c = 0; c < nch && n > 0; c++) WAM_HIP(hipMemcpy(out[c], result + (size_t)c * n, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  return prog(1.0, "Complete");
}

// The analysis the app runs on every file load (web/app.js:823-872), as one call with one upload: DC detection
// (detectDCOffsetBuffer, dc-offset.js:38-60), removal from every channel when any channel is "significant" (:828-834), loudness
// and true peak of the result (:865-866), and normalizeToLUFS(buffer, target) with the LIBRARY limiter when the projected peak
// exceeds the ceiling (normalizer.js:38-94).
int wam_analyze_on_load(const float* const* ch, int nch, int64_t n, double sr, double target_lufs, double ceiling_db,
                        float* const* cleaned, float* const* normalized, wam_load_analysis* res) {
  int rc = check_in(ch, nch, n); if (rc) return rc;
  if (!res || !(sr > 0)) { set_error("invalid load-analysis arguments"); return WAM_ERR_INVALID; }
  for (int c = 0; c < nch; c++) if (((cleaned && !cleaned[c]) || (normalized && !normalized[c])) && n > 0) { set_error("output channel %d is NULL", c); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  std::memset(res, 0, sizeof(*res));
  if (n == 0) { res->dc[0] = res->dc[1] = NAN; res->original_lufs = -14.0; res->original_true_peak_db = -INFINITY; return WAM_OK; }
  const size_t bytes = sizeof(float) * (size_t)n;
  DevBuf in, cl, part, dcd, ms, nb, gd;
  rc = upload_chans(ch, nch, n, in); if (rc) return rc;
  // 1. DC offset
  const int nparts = 256;
  rc = part.alloc(sizeof(double) * nparts * nch); if (rc) return rc;
  rc = dcd.alloc(sizeof(double) * nch); if (rc) return rc;
  WAM_LAUNCH("k_dc_partial", k_dc_partial, dim3(nparts, nch), dim3(256), nullptr, (const float*)in.as<float>(), n, part.as<double>());
  WAM_LAUNCH("k_dc_final", k_dc_final, dim3(nch), dim3(64), nullptr, (const double*)part.as<double>(), nparts, n, dcd.as<double>());
  WAM_HIP(hipMemcpy(res->dc, dcd.p, sizeof(double) * nch, hipMemcpyDeviceToHost));
  double tot = 0;
  for (int c = 0; c < nch; c++) { tot += std::fabs(res->dc[c]); if (std::fabs(res->dc[c]) > 0.001) res->dc_significant = 1; }   // dc-offset.js:29,58
  res->dc_percent = tot / nch * 100;                                                                                            // :50-56
  const float* cur = in.as<float>();
  if (res->dc_significant) {                                                                                                   // app.js:828-834
    rc = cl.alloc(bytes * nch); if (rc) return rc;
    int64_t bx = (n + 255) / 256; if (bx > 2048) bx = 2048;
    WAM_LAUNCH("k_dc_sub", k_dc_sub, dim3((unsigned)bx, nch), dim3(256), nullptr, (const float*)in.as<float>(), cl.as<float>(), n, (const double*)dcd.as<double>());
    cur = cl.as<float>();
    res->dc_removed = 1;
  }
  if (cleaned) for (int c = 0; c < nch; c++) WAM_HIP(hipMemcpyAsync(cleaned[c], cur + (size_t)c * n, bytes, hipMemcpyDeviceToHost, nullptr));
  // 2. loudness and peak of the (cleaned) buffer; normalizeToLUFS measures again with fallback = target, which differs from
  //    measureLUFS(buffer) only for audio shorter than 0.4 s
  MeterPlan mp; rc = make_meter_plan(nch, n, sr, -14.0, &mp); if (rc) return rc;
  rc = ms.alloc(meter_scratch_bytes(mp, 1, false)); if (rc) return rc;
  MeterScratch M; carve_meter_scratch(mp, 1, false, ms.p, &M);
  rc = enqueue_meter(cur, 1, mp, n, M, nullptr); if (rc) return rc;
  GateOut g;
  WAM_HIP(hipMemcpy(&g, M.gate, sizeof(g), hipMemcpyDeviceToHost));
  res->original_lufs = finish_lufs(mp, g);
  res->original_true_peak_db = finish_peak_db(g);
  // 3. normalizeToLUFS(buffer, target, ceiling)
  const double lufs_n = mp.too_short ? target_lufs : res->original_lufs;
  if (!std::isfinite(lufs_n)) {                                                                                                // normalizer.js:44-47: the same buffer
    if (normalized) for (int c = 0; c < nch; c++) WAM_HIP(hipMemcpyAsync(normalized[c], cur + (size_t)c * n, bytes, hipMemcpyDeviceToHost, nullptr));
    WAM_HIP(hipStreamSynchronize(nullptr));
    return WAM_OK;
  }
  res->measured = 1;
  res->gain_db = target_lufs - lufs_n;
  const double gl = std::pow(10, res->gain_db / 20);
  rc = nb.alloc(bytes * nch); if (rc) return rc;
  rc = gd.alloc(sizeof(double)); if (rc) return rc;
  WAM_HIP(hipMemcpy(gd.p, &gl, sizeof(double), hipMemcpyHostToDevice));
  rc = enqueue_gain_meter(cur, nb.as<float>(), 1, nch, n, gd.as<double>(), mp, M, nullptr); if (rc) return rc;                  // (the meter by-product is not used)
  const float* result = nb.as<float>();
  DevBuf o, plan, tmb, gv;
  if (res->original_true_peak_db + res->gain_db > ceiling_db) {                                                                 // :81-91, library limiter
    LimiterPlan lp; rc = make_limiter_plan(sr, std::pow(10, ceiling_db / 20), 3, 100, 3, 1, &lp); if (rc) return rc;
    lp.lib_twin = 1;
    rc = plan.alloc(sizeof(lp)); if (rc) return rc;
    WAM_HIP(hipMemcpyAsync(plan.p, &lp, sizeof(lp), hipMemcpyHostToDevice, nullptr));
    rc = tmb.alloc(bytes); if (rc) return rc;
    rc = gv.alloc(bytes); if (rc) return rc;
    rc = o.alloc(bytes * nch); if (rc) return rc;
    rc = enqueue_limiter(nb.as<float>(), o.as<float>(), tmb.as<float>(), gv.as<float>(), 1, nch, n, plan.as<LimiterPlan>(), lp.L, 1,
                         nullptr, nullptr, 0, nullptr, sr, nullptr); if (rc) return rc;
    result = o.as<float>();
    res->limiter_applied = 1;
  }
  if (normalized) for (int c = 0; c < nch; c++) WAM_HIP(hipMemcpyAsync(normalized[c], result + (size_t)c * n, bytes, hipMemcpyDeviceToHost, nullptr));
  WAM_HIP(hipStreamSynchronize(nullptr));
  return WAM_OK;
}

int wam_apply_gain(const float* const* ch, int nch, int64_t n, double gain_db, float* const* out) {
  int rc = check_in(ch, nch, n); if (rc) return rc;
  if (!out) { set_error("out is NULL"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  if (n == 0) return WAM_OK;
  MeterPlan mp; rc = make_meter_plan(nch, n, 48000.0, -14.0, &mp); if (rc) return rc;     // the metering by-product is discarded
  DevBuf in, o, ms, gd;
  rc = upload_chans(ch, nch, n, in); if (rc) return rc;
  rc = o.alloc(sizeof(float) * (size_t)nch * (size_t)n); if (rc) return rc;
  rc = ms.alloc(meter_scratch_bytes(mp, 1, false)); if (rc) return rc;
  MeterScratch M; carve_meter_scratch(mp, 1, false, ms.p, &M);
  WAM_HIP(hipMemcpyAsync(M.plan, &mp, sizeof(mp), hipMemcpyHostToDevice, nullptr));
  const double gl = std::pow(10, gain_db / 20);                                            // normalizer.js:104
  rc = gd.alloc(sizeof(double)); if (rc) return rc;
  WAM_HIP(hipMemcpy(gd.p, &gl, sizeof(double), hipMemcpyHostToDevice));
  rc = enqueue_gain_meter(in.as<float>(), o.as<float>(), 1, nch, n, gd.as<double>(), mp, M, nullptr); if (rc) return rc;
  for (int c = 0; c < nch; c++) WAM_HIP(hipMemcpy(out[c], o.as<float>() + (size_t)c * n, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  return WAM_OK;
}

int wam_encode_pcm24(const float* const* ch, int nch, int64_t n, uint8_t* out) {
  int rc = check_in(ch, nch, n); if (rc) return rc;
  if (!out && n > 0) { set_error("out is NULL"); return WAM_ERR_INVALID; }       // (an empty buffer's output has no address: a header-only file)
  rc = ensure_device(); if (rc) return rc;
  if (n == 0) return WAM_OK;
  DevBuf in, o;
  rc = upload_chans(ch, nch, n, in); if (rc) return rc;
  const size_t obytes = (size_t)3 * nch * (size_t)n;
  rc = o.alloc(obytes + 16); if (rc) return rc;
  const int fpt = 4 / nch;
  const int64_t groups = (n + fpt - 1) / fpt;
  int64_t bx = (groups + 255) / 256; if (bx > 4096) bx = 4096;
  WAM_LAUNCH("k_pcm24", k_pcm24, dim3((unsigned)bx), dim3(256), nullptr, (const float*)in.as<float>(), nch, n, o.as<uint8_t>());
  WAM_HIP(hipMemcpy(out, o.p, obytes, hipMemcpyDeviceToHost));
  return WAM_OK;
}

int wam_encode_pcm16(const float* const* ch, int nch, int64_t n, int dither_mode, const float* dither, uint32_t seed, uint8_t* out) {
  int rc = check_in(ch, nch, n); if (rc) return rc;
  if ((!out && n > 0) || dither_mode < 0 || dither_mode > 2) { set_error("invalid 16-bit encoder arguments (out NULL, or dither_mode outside 0..2)"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  if (n == 0) return WAM_OK;
  DevBuf in, o, dd;
  rc = upload_chans(ch, nch, n, in); if (rc) return rc;
  const size_t obytes = (size_t)2 * nch * (size_t)n;
  rc = o.alloc(obytes); if (rc) return rc;
  if (dither && dither_mode != 0) {
    rc = dd.alloc(sizeof(float) * (size_t)nch * (size_t)n); if (rc) return rc;
    WAM_HIP(hipMemcpyAsync(dd.p, dither, sizeof(float) * (size_t)nch * (size_t)n, hipMemcpyHostToDevice, nullptr));
  }
  DevBuf fn;
  rc = fn.alloc(sizeof(unsigned long long) * 2); if (rc) return rc;
  if (dither_mode == 2) {
    const unsigned long long init[2] = { (unsigned long long)n, (unsigned long long)n };
    WAM_HIP(hipMemcpyAsync(fn.p, init, sizeof(init), hipMemcpyHostToDevice, nullptr));
    int64_t bx = (n + 255) / 256; if (bx > 1024) bx = 1024;
    WAM_LAUNCH("k_first_nan", k_first_nan, dim3((unsigned)bx, (unsigned)nch), dim3(256), nullptr, (const float*)in.as<float>(), n, fn.as<unsigned long long>());
  }
  const int64_t runs = (n + kPcm16Run - 1) / kPcm16Run;
  WAM_LAUNCH("k_pcm16", k_pcm16, dim3((unsigned)((runs + 63) / 64)), dim3(64), nullptr, (const float*)in.as<float>(), nch, n, dither_mode,
             (const float*)(dither && dither_mode != 0 ? dd.as<float>() : nullptr), seed, (const unsigned long long*)fn.as<unsigned long long>(), o.as<uint8_t>());
  WAM_HIP(hipMemcpy(out, o.p, obytes, hipMemcpyDeviceToHost));
  return WAM_OK;
}

int wam_wav_header(int nch, int64_t n, int sample_rate, int bits, uint8_t h[44]) {
  if (!h || nch < 1 || n < 0 || (bits != 16 && bits != 24)) { set_error("invalid header arguments"); return WAM_ERR_INVALID; }
  const uint32_t bps = (uint32_t)bits / 8, data = (uint32_t)((uint64_t)n * nch * bps);
  std::memcpy(h, "RIFF", 4); uint32_t v = 36 + data; std::memcpy(h + 4, &v, 4);
  std::memcpy(h + 8, "WAVEfmt ", 8); v = 16; std::memcpy(h + 16, &v, 4);
  uint16_t s = 1; std::memcpy(h + 20, &s, 2); s = (uint16_t)nch; std::memcpy(h + 22, &s, 2);
  v = (uint32_t)sample_rate; std::memcpy(h + 24, &v, 4); v = (uint32_t)sample_rate * nch * bps; std::memcpy(h + 28, &v, 4);
  s = (uint16_t)(nch * bps); std::memcpy(h + 32, &s, 2); s = (uint16_t)bits; std::memcpy(h + 34, &s, 2);
  std::memcpy(h + 36, "data", 4); std::memcpy(h + 40, &data, 4);
  return WAM_OK;
}

int64_t wam_stft_num_frames(int64_t n, int nfft, int hop) {
  if (nfft <= 0 || hop <= 0 || n < nfft) return 0;
  return (n - nfft) / hop + 1;
}

// tables + launch for [ntracks][nch][n] on the device; out_db_dev [ntracks][nframes][nfft/2], cols_dev [ntracks][nframes][256] or NULL
static int stft_enqueue(const float* in_dev, int ntracks, int nch, int64_t n, double sr, int nfft, int hop, float* out_db_dev, float* cols_dev,
                        hipStream_t stream) {
  using namespace wam;
  const StreamScope scope(stream);
  const int64_t nframes = wam_stft_num_frames(n, nfft, hop);
  if (nframes == 0 || ntracks == 0) return WAM_OK;
  int bits = 0; while ((1 << bits) < nfft) bits++;
  const int half = nfft >> 1;
  std::vector<float> tab(3 * (size_t)nfft);
  for (int i = 0; i < nfft; i++) {                                       // spectrogram.js:40-48
    const double ang = -2 * M_PI * i / nfft;
    tab[i] = (float)std::cos(ang); tab[nfft + i] = (float)std::sin(ang);
    tab[2 * nfft + i] = (float)(0.5 * (1 - std::cos(2 * M_PI * i / nfft)));
  }
  std::vector<StftRow> rows(256);
  {
    const double lmin = std::log10(20.0), lmax = std::log10(22000.0), lr = lmax - lmin, nyq = sr / 2;   // spectrogram.js:177-203
    for (int y = 0; y < 256; y++) {
      const double fr = std::pow(10, lmax - ((double)y / 255) * lr);
      StftRow r{0, 0, 0.0, 0};
      if (fr >= 20 && fr <= nyq) {
        const double bf = (fr * nfft) / sr;
        const int b1 = (int)std::floor(bf);
        if (b1 < half) { r.bin1 = b1; r.bin2 = b1 + 1 < half ? b1 + 1 : b1; r.frac = bf - b1; r.valid = 1; }
      }
      rows[y] = r;
    }
  }
  const double add_db = 10 * std::log10(16.0) + (-20 * std::log10((double)nfft));   // correctionAC + fftNorm, :135-136
  DevBuf tabs, rowsd;
  int rc = tabs.alloc(sizeof(float) * 3 * nfft); if (rc) return rc;
  WAM_HIP(hipMemcpyAsync(tabs.p, tab.data(), sizeof(float) * 3 * nfft, hipMemcpyHostToDevice, stream));
  rc = rowsd.alloc(sizeof(StftRow) * 256); if (rc) return rc;
  WAM_HIP(hipMemcpyAsync(rowsd.p, rows.data(), sizeof(StftRow) * 256, hipMemcpyHostToDevice, stream));
  int64_t bx = nframes < 4096 ? nframes : 4096;
  if (ntracks >= 64 && bx > 128) bx = 128;                               // enough workgroups from the track dimension
  const size_t shmem = sizeof(float) * (size_t)(2 * nfft + half);
  const float* ct = tabs.as<float>(); const float* st = ct + nfft; const float* wn = ct + 2 * nfft;
  // stft_db's table: 32 intervals of the mantissa, (1 / c, 10 log10 c) with the logarithm taken of the double 1 / c itself
  DevBuf logtb;
  {
    double lt[2 * kLogTab];
    for (int i = 0; i < kLogTab; i++) {
      const double c = 0.5 + (i + 0.5) / (2.0 * kLogTab), rcp = 1.0 / c;
      lt[2 * i] = rcp; lt[2 * i + 1] = (double)(-10.0L * log10l((long double)rcp));
    }
    rc = logtb.alloc(sizeof(lt)); if (rc) return rc;
    WAM_HIP(hipMemcpyAsync(logtb.p, lt, sizeof(lt), hipMemcpyHostToDevice, stream));
    WAM_HIP(hipStreamSynchronize(stream));                               // lt is a local
  }
  const double2* logt = (const double2*)logtb.as<double2>();
  DevBuf tw6b;
  if (nfft == 4096) {                                                    // stages 1-6 of the wavefront kernel: (cos, sin)[64 k], k < 32, widened
    std::vector<double> t6(64);
    for (int k = 0; k < 32; k++) { t6[2 * k] = (double)tab[64 * k]; t6[2 * k + 1] = (double)tab[nfft + 64 * k]; }
    rc = tw6b.alloc(sizeof(double) * 64); if (rc) return rc;
    WAM_HIP(hipMemcpyAsync(tw6b.p, t6.data(), sizeof(double) * 64, hipMemcpyHostToDevice, stream));
    WAM_HIP(hipStreamSynchronize(stream));                               // t6 is a local
  }
  const int variant = (int)tuning().stft_literal;                        // 0: fastest kernel for the size; 1: the literal loop nest; 2: the register-blocked workgroup kernel
  const int prof_ticket = prof_begin("k_stft", stream);
  if (nfft == 4096 && variant == 0) {
    int64_t gx = (nframes + kSwWaves - 1) / kSwWaves;
    if (gx > 64) gx = 64;
    if (ntracks < 64) { const int64_t want = (2048 + ntracks - 1) / ntracks; gx = (nframes + kSwWaves - 1) / kSwWaves; if (gx > want) gx = want; }
#define WAM_STFT_WAVE(NC, CL) hipLaunchKernelGGL((k_stft_wave<NC, CL>), dim3((unsigned)gx, (unsigned)ntracks), dim3(64 * kSwWaves), 0, stream, in_dev, n, hop, nframes, \
                                                 ct, st, wn, (const double2*)tw6b.as<double2>(), add_db, out_db_dev, (const StftRow*)rowsd.as<StftRow>(), cols_dev, logt)
    if (nch == 2) { if (cols_dev) WAM_STFT_WAVE(2, true); else WAM_STFT_WAVE(2, false); }
    else { if (cols_dev) WAM_STFT_WAVE(1, true); else WAM_STFT_WAVE(1, false); }
#undef WAM_STFT_WAVE
  } else if (nfft == 4096 && variant != 1)
    hipLaunchKernelGGL(k_stft_fast<12>, dim3((unsigned)bx, (unsigned)ntracks), dim3(256), 0, stream, in_dev, nch, n, hop, nframes, ct, st, wn, add_db,
                       out_db_dev, (const StftRow*)rowsd.as<StftRow>(), cols_dev, logt);
  else if (nfft == 2048 && variant != 1)
    hipLaunchKernelGGL(k_stft_fast<11>, dim3((unsigned)bx, (unsigned)ntracks), dim3(256), 0, stream, in_dev, nch, n, hop, nframes, ct, st, wn, add_db,
                       out_db_dev, (const StftRow*)rowsd.as<StftRow>(), cols_dev, logt);
  else
    hipLaunchKernelGGL(k_stft, dim3((unsigned)bx, (unsigned)ntracks), dim3(256), shmem, stream, in_dev, nch, n, nfft, bits, hop, nframes, ct, st, wn,
                       add_db, out_db_dev, (const StftRow*)rowsd.as<StftRow>(), cols_dev, logt);
  prof_end(prof_ticket, stream);
  WAM_HIP(hipGetLastError());
  WAM_HIP(hipStreamSynchronize(stream));                                 // tables (host vectors, pool blocks) are released on return
  return WAM_OK;
}

int wam_batch_stft_db_dev(const float* in_dev, int ntracks, int nch, int64_t n, double sr, int nfft, int hop, float* out_db_dev,
                          float* columns_dev, void* stream) {
  if (!in_dev || !out_db_dev || ntracks < 1 || nch < 1 || nch > 2 || n < 0 || nfft < 16 || nfft > 8192 || (nfft & (nfft - 1)) || hop < 1 || !(sr > 0)) {
    wam::set_error("invalid STFT arguments (nfft must be a power of two in 16..8192)"); return WAM_ERR_INVALID;
  }
  int rc = wam::ensure_device(); if (rc) return rc;
  return stft_enqueue(in_dev, ntracks, nch, n, sr, nfft, hop, out_db_dev, columns_dev, (hipStream_t)stream);
}

int wam_stft_db(const float* const* ch, int nch, int64_t n, double sr, int nfft, int hop, float* out_db, float* columns) {
  int rc = check_in(ch, nch, n); if (rc) return rc;
  if (!out_db || nfft < 16 || nfft > 8192 || (nfft & (nfft - 1)) || hop < 1 || !(sr > 0)) { set_error("invalid STFT arguments (nfft must be a power of two in 16..8192)"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  const int64_t nframes = wam_stft_num_frames(n, nfft, hop);
  if (nframes == 0) return WAM_OK;
  const int half = nfft >> 1;
  DevBuf in, od, cd;
  rc = upload_chans(ch, nch, n, in); if (rc) return rc;
  rc = od.alloc(sizeof(float) * (size_t)nframes * half); if (rc) return rc;
  if (columns) { rc = cd.alloc(sizeof(float) * (size_t)nframes * 256); if (rc) return rc; }
  rc = stft_enqueue(in.as<float>(), 1, nch, n, sr, nfft, hop, od.as<float>(), columns ? cd.as<float>() : nullptr, nullptr); if (rc) return rc;
  WAM_HIP(hipMemcpy(out_db, od.p, sizeof(float) * (size_t)nframes * half, hipMemcpyDeviceToHost));
  if (columns) WAM_HIP(hipMemcpy(columns, cd.p, sizeof(float) * (size_t)nframes * 256, hipMemcpyDeviceToHost));
  return WAM_OK;
}

}  // extern "C"
