// chain_deharsh.hip - stage 1 of the export chain: processHybridDynamic(buffer, 'mastering'),
// web/lib/dsp/dynamic-processor.js:595-609 (preset :572-585, _processChannel :444-500,
// _computeBinGains :319-385, ResonanceDetector :194-233, GainComputer :75-90,
// EnvelopeFollower :33-43; STFT wrapper web/lib/dsp/fft.js:41-68; Hann utils.js:98-104).
//
// 2048-point STFT, hop 512, per channel:
//   k_dh_bands : every frame in parallel -> windowed FFT -> 7 band RMS values          (parallel over frames)
//   k_dh_follow: per (channel, band) the per-frame envelope follower + soft-knee gain  (serial over frames, tiny)
//   k_dh_apply : one workgroup per channel walks the frames in order: FFT, per-bin resonance EMA (state per
//                bin in registers), gains, inverse FFT, synthesis window, overlap-add in LDS, 85/15 dry/wet mix.
// The FFT runs in float32 in LDS (radix-2 stages fused in pairs, DIF forward / DIT inverse, no bit-reversal
// pass); the reference's FFT is double (npm fft.js, absent: parity unpinned at that boundary).  The wet path
// is scaled by 0.15 * 0.375, so float32 FFT rounding (~1e-6 relative) is ~1e-7 of full scale in the output.
#include "chain.hpp"
#include <cmath>
#include <cstring>
#include <vector>

namespace wam {

constexpr int kN = 2048, kHop = 512, kBins = 1024, kBands = 7;

struct DeharshPlan {
  double thr[kBands], ks[kBands], ke[kBands], knee[kBands], ratio[kBands], A[kBands], R[kBands];
  float cnt[kBands];
  int64_t nframes;
  uint8_t band_of[kBins];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)

// ---- the transform ------------------------------------------------------------------------------------
// A 2048-point real frame is transformed as a 1024-point complex FFT of z[m] = x[2m] + i x[2m+1] plus an O(N)
// split step (and the mirror image of that for the inverse), which halves the butterfly work and LDS traffic.
// The 1024-point FFT is a radix-4 Stockham autosort (natural order in and out, no bit reversal): 256 threads, one
// butterfly per thread and pass, five passes.  Every pass reads z[g + 256 r] (unit stride across lanes) and
// writes 4 (g - j) + j + r Ls; the array is padded by 4 entries every 16 so that those strided writes spread over
// the LDS banks.  Tables: twA[k] = exp(-2 pi i k / 1024), k < 768; twB[k] = exp(-2 pi i k / 2048), k <= 512.
constexpr int kM = 1024, kZ = kM + kM / 4, kTwA = 768, kTwB = 513;
__device__ __forceinline__ int padz(int i) { return i + ((i >> 4) << 2); }

template <bool INV>
__device__ __forceinline__ void r4_butterfly(float2 a[4]) {
  const float2 s02 = make_float2(a[0].x + a[2].x, a[0].y + a[2].y), d02 = make_float2(a[0].x - a[2].x, a[0].y - a[2].y);
  const float2 s13 = make_float2(a[1].x + a[3].x, a[1].y + a[3].y), d13 = make_float2(a[1].x - a[3].x, a[1].y - a[3].y);
  a[0] = make_float2(s02.x + s13.x, s02.y + s13.y);
  a[2] = make_float2(s02.x - s13.x, s02.y - s13.y);
  if (!INV) {   // W4 = -i
    a[1] = make_float2(d02.x + d13.y, d02.y - d13.x);
    a[3] = make_float2(d02.x - d13.y, d02.y + d13.x);
  } else {
    a[1] = make_float2(d02.x - d13.y, d02.y + d13.x);
    a[3] = make_float2(d02.x + d13.y, d02.y - d13.x);
  }
}

// a[r] = input element g + 256 r (in registers).  Passes alternate between two LDS arrays (one barrier per pass):
// pass 0 writes z0, then z0 -> z1 -> z0 -> z1 -> z0; the result (padded, natural order) is in z0 and visible after the
// final barrier.  Unnormalised in both directions.  The caller guarantees that nobody still reads z0 or z1 on entry.
template <bool INV>
__device__ __forceinline__ void fft1024(float2 a[4], float2* z0, float2* z1, const float2* twA) {
  const int g = threadIdx.x;
  r4_butterfly<INV>(a);
#pragma unroll
  for (int r = 0; r < 4; r++) z0[padz(4 * g + r)] = a[r];
  __syncthreads();
#pragma unroll
  for (int s = 1; s < 5; s++) {
    const int Ls = 1 << (2 * s), shift = 8 - 2 * s;
    const float2* src = (s & 1) ? z0 : z1;
    float2* dst = (s & 1) ? z1 : z0;
#pragma unroll
    for (int r = 0; r < 4; r++) a[r] = src[padz(g + 256 * r)];
    const int j = g & (Ls - 1);
#pragma unroll
    for (int r = 1; r < 4; r++) {
      const float2 w = twA[(r * j) << shift];
      a[r] = INV ? cmulc(a[r], w) : cmul(a[r], w);
    }
    r4_butterfly<INV>(a);
    const int base = ((g - j) << 2) + j;
#pragma unroll
    for (int r = 0; r < 4; r++) dst[padz(base + r * Ls)] = a[r];
    __syncthreads();
  }
}

// Split step: from Z = FFT1024(z) to the spectrum X of the real frame for the pair of bins (k, 1024 - k), 1 <= k <= 511.
//   Xe = (Z[k] + conj Z[M-k]) / 2,  Xo = -i (Z[k] - conj Z[M-k]) / 2,  X[k] = Xe + W^k Xo,  X[M-k] = conj(Xe - W^k Xo)
struct BinPair { float2 xe, t; };      // X[k] = xe + t, conj(X[M-k]) = xe - t
__device__ __forceinline__ BinPair split_pair(float2 zk, float2 zm, float2 w) {
  const float2 bq = make_float2(zm.x, -zm.y);
  BinPair p;
  p.xe = make_float2(0.5f * (zk.x + bq.x), 0.5f * (zk.y + bq.y));
  const float2 d = make_float2(0.5f * (zk.x - bq.x), 0.5f * (zk.y - bq.y));
  p.t = cmul(make_float2(d.y, -d.x), w);
  return p;
}

// Band RMS of every frame.  grid = (frame groups, channels); each workgroup handles frames f = blockIdx.x, += gridDim.x.
__global__ __launch_bounds__(256) void k_dh_bands(const float* __restrict__ in, int64_t n, const DeharshPlan* __restrict__ dp,
                                                  const float2* __restrict__ twg, const float* __restrict__ wing,
                                                  float* __restrict__ band_rms, double pre_gain) {
  __shared__ float2 z[kZ], zb[kZ];
  __shared__ float2 tw[kTwA + kTwB];
  __shared__ float part[4][kBands];
  const DeharshPlan& P = *dp;
  const float2* twA = tw; const float2* twB = tw + kTwA;
  const float* win = wing;                       // read through L1: coalesced, the same 8 KB for every frame
  const int64_t chan = blockIdx.y;
  const float* x = in + chan * n;
  const int t = threadIdx.x;
  for (int i = t; i < kTwA + kTwB; i += 256) tw[i] = twg[i];
  // bins of this thread: pairs k1 = t (t = 0: DC, and bin 512 in the second slot) and k2 = t + 256
  const int k1 = t, k2 = t + 256;
  const int bnd[4] = { P.band_of[k1], P.band_of[t == 0 ? 512 : kM - k1], P.band_of[k2], P.band_of[kM - k2] };
  __syncthreads();
  for (int64_t f = blockIdx.x; f < P.nframes; f += gridDim.x) {
    const int64_t pos = f * kHop;
    float2 a[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = 2 * (t + 256 * r);
      float v0 = x[pos + i], v1 = x[pos + i + 1];
      if (pre_gain != 1.0) { v0 = f32r((double)v0 * pre_gain); v1 = f32r((double)v1 * pre_gain); }
      if (!(v0 == v0)) v0 = 0.0f;                                  // (input[i] || 0)
      if (!(v1 == v1)) v1 = 0.0f;
      a[r] = make_float2(f32r((double)v0 * (double)win[i]), f32r((double)v1 * (double)win[i + 1]));   // fft.js:43-45
    }
    fft1024<false>(a, z, zb, twA);
    float m2[4];                                                   // |X|^2 of the four bins (dynamic-processor.js:456-460)
    {
      const float2 zk = z[padz(k2)], zm = z[padz(kM - k2)];
      const BinPair p = split_pair(zk, zm, twB[k2]);
      const float2 xk = make_float2(p.xe.x + p.t.x, p.xe.y + p.t.y), xm = make_float2(p.xe.x - p.t.x, p.xe.y - p.t.y);
      m2[2] = xk.x * xk.x + xk.y * xk.y; m2[3] = xm.x * xm.x + xm.y * xm.y;
    }
    if (t == 0) {
      const float2 z0 = z[padz(0)], zh = z[padz(512)];
      const float x0 = z0.x + z0.y;
      m2[0] = x0 * x0; m2[1] = zh.x * zh.x + zh.y * zh.y;
    } else {
      const float2 zk = z[padz(k1)], zm = z[padz(kM - k1)];
      const BinPair p = split_pair(zk, zm, twB[k1]);
      const float2 xk = make_float2(p.xe.x + p.t.x, p.xe.y + p.t.y), xm = make_float2(p.xe.x - p.t.x, p.xe.y - p.t.y);
      m2[0] = xk.x * xk.x + xk.y * xk.y; m2[1] = xm.x * xm.x + xm.y * xm.y;
    }
    float be[kBands];
#pragma unroll
    for (int b = 0; b < kBands; b++) be[b] = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float mag = sqrtf(m2[q]);                              // the reference squares the f32 magnitude (:335-337)
#pragma unroll
      for (int b = 0; b < kBands; b++) if (b == bnd[q]) be[b] += mag * mag;
    }
#pragma unroll
    for (int b = 0; b < kBands; b++) {
      float v = be[b];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
      if (lane_id() == 0) part[wave_id()][b] = v;
    }
    __syncthreads();
    if (t < kBands) {
      const float s = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
      const float rms = P.cnt[t] > 0 ? f32r(sqrt((double)s / (double)P.cnt[t])) : s;     // :335-339
      band_rms[(chan * P.nframes + f) * kBands + t] = rms;
    }
    // the next frame's first write to z / part comes after this frame's reads: fft1024 writes z only after all threads
    // passed the barrier above, and part is rewritten after the barriers inside fft1024
  }
}

// GainComputer.computeGainDb, dynamic-processor.js:75-90, on the follower's envelope; the reference keeps the result in a Float32Array
__device__ __forceinline__ float band_gain_db_of(double env, const DeharshPlan& P, int b) {
  const double edb = env > 0 ? 20 * log10(env) : -INFINITY;
  double gdb;
  if (edb <= P.ks[b]) gdb = 0;
  else if (edb >= P.ke[b]) { const double ex = edb - P.thr[b]; gdb = ex / P.ratio[b] - ex; }
  else { const double kp = (edb - P.ks[b]) / P.knee[b]; const double full = (edb - P.thr[b]) * (1 - 1 / P.ratio[b]); gdb = -full * kp * kp; }
  return f32r(gdb);
}

// Per (channel, band): the envelope follower over frames (EnvelopeFollower.process, :33-43).  band_env: [chan][frame][7]
// double.  The release coefficient is per SAMPLE but the follower is stepped per frame, so its memory spans the whole
// track (no warm-up segmentation); instead one workgroup per (channel, band) walks the frames in tiles of 4,096 with
// the same policy-iteration scan as the per-sample followers (an hour at 96 kHz is 337 k frames = 83 tiles).  Only
// the recurrence runs here - the logarithm and the knee are evaluated by the consumer.
__global__ __launch_bounds__(kThreads) void k_dh_follow(const float* __restrict__ band_rms, const DeharshPlan* __restrict__ dp, int nchan,
                                                        double* __restrict__ band_env) {
  __shared__ double lds_scan[kSlotDoubles];
  const DeharshPlan& P = *dp;
  const int chan = blockIdx.x / kBands, b = blockIdx.x % kBands;
  const double A = P.A[b], R = P.R[b];
  const int64_t F = P.nframes;
  const float* src = band_rms + (int64_t)chan * F * kBands + b;
  double* dst = band_env + (int64_t)chan * F * kBands + b;
  LdsSlots slots(lds_scan);
  double Senv = 0;
  const int64_t ntiles = (F + kTile - 1) / kTile;
  for (int64_t t = 0; t < ntiles; t++) {
    const int64_t base = t * kTile + (int64_t)threadIdx.x * C;
    float a[C];
#pragma unroll
    for (int k = 0; k < C; k++) a[k] = (base + k < F) ? fabsf(src[(base + k) * kBands]) : 0.0f;
    // frames past the end: a = 0 keeps the recurrence defined; nothing is stored for them
    double e = follower_solve(a, A, R, Senv, slots.next());
#pragma unroll
    for (int k = 0; k < C; k++) {
      e = follower_step(e, (double)a[k], A, R);
      if (base + k < F) dst[(base + k) * kBands] = e;
    }
  }
}

// Synthesis.  grid = (segments, channels): a workgroup owns the frames [f0, f1) of its segment and emits the samples
// [f0 * 512, f1 * 512) (the last segment also the tail).  Frame-to-frame state is the per-bin resonance EMA
// (avg' = 0.3 mag + 0.7 avg) and the overlap-add ring; a segment rebuilds both by re-running `warm` frames before
// f0: the EMA forgets its start value as 0.7^warm, and the ring needs the three frames before f0.  Warm-up frames
// more than three before f0 skip the inverse transform.
__global__ __launch_bounds__(256) void k_dh_apply(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                  const DeharshPlan* __restrict__ dp, const float2* __restrict__ twg,
                                                  const float* __restrict__ wing, const double* __restrict__ band_env, double pre_gain,
                                                  int64_t seg_frames, int64_t warm) {
  __shared__ float2 z[kZ], zb[kZ];
  __shared__ float2 tw[kTwA + kTwB];
  __shared__ float xin[kN];      // input ring: sample s at xin[s & 2047]
  __shared__ float ola[kN];      // overlap-add ring
  __shared__ float bg[kBands];
  const DeharshPlan& P = *dp;
  const float2* twA = tw; const float2* twB = tw + kTwA;
  const int64_t chan = blockIdx.y;
  const float* x = in + chan * n;
  float* y = out + chan * n;
  const int t = threadIdx.x;
  const double dry = 1 - 0.15, wetm = 0.15;                         // preset dryWetMix 0.15, dynamic-processor.js:419-423
  const float* win = wing;                       // read through L1: coalesced, the same 8 KB for every frame
  for (int i = t; i < kTwA + kTwB; i += 256) tw[i] = twg[i];
  for (int i = t; i < kN; i += 256) ola[i] = 0.0f;
  auto load_in = [&](int64_t s) -> float {
    float v = (s < n) ? x[s] : 0.0f;
    if (pre_gain != 1.0) v = f32r((double)v * pre_gain);
    return v;
  };
  const int64_t F = P.nframes;
  const int64_t f0 = (int64_t)blockIdx.x * seg_frames;
  int64_t f1 = f0 + seg_frames; if (f1 > F) f1 = F;
  const int64_t fs = f0 - warm > 0 ? f0 - warm : 0;
  const bool last_seg = f1 >= F;
  if (F > 0) for (int i = t; i < kN - kHop; i += 256) { const int64_t s = fs * kHop + i; xin[s & (kN - 1)] = load_in(s); }   // each frame adds 512
  // bins of this thread: the pairs (k1, 1024 - k1) and (k2, 1024 - k2), k1 = t, k2 = t + 256; thread 0 holds DC and, in
  // the slot of "1024 - 0", bin 512.  avg[] = the resonance detector's running mean of those four bins.
  const int k1 = t, k2 = t + 256;
  const int bnd[4] = { P.band_of[k1], P.band_of[t == 0 ? 512 : kM - k1], P.band_of[k2], P.band_of[kM - k2] };
  float avg[4] = {0, 0, 0, 0};
  for (int64_t f = fs; f < f1; f++) {
    const int64_t pos = f * kHop;
    const bool synth = f + 3 >= f0;        // this frame's output reaches samples the segment emits
    __syncthreads();
    // newest 512 samples of this frame: [pos + 1536, pos + 2048)
    for (int i = t; i < kHop; i += 256) { const int64_t s = pos + (kN - kHop) + i; xin[s & (kN - 1)] = load_in(s); }
    if (t < kBands) bg[t] = band_gain_db_of(band_env[(chan * F + f) * kBands + t], P, t);
    __syncthreads();
    float2 a[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = 2 * (t + 256 * r);
      const int o = (int)((pos + i) & (kN - 1));               // pos is even: o + 1 does not wrap
      float v0 = xin[o], v1 = xin[o + 1];
      if (!(v0 == v0)) v0 = 0.0f;
      if (!(v1 == v1)) v1 = 0.0f;
      a[r] = make_float2(f32r((double)v0 * (double)win[i]), f32r((double)v1 * (double)win[i + 1]));
    }
    fft1024<false>(a, z, zb, twA);
    // resonance detector (:194-233) and bin gain (:354-382) of one bin; returns the linear gain
    const bool first = f == fs;            // the detector's first frame (:203-207); a warm start for later segments
    auto bin_gain = [&](float mag, float& av_state, int band) -> float {
      float res = 0.0f;
      if (first) av_state = mag;
      else {
        av_state = f32r(0.3 * (double)mag + (1 - 0.3) * (double)av_state);
        const double av = av_state;
        if (av > 1e-10 && (double)mag > av * 1.5) res = f32r(fmin(1.0, ((double)mag / av - 1.5) / 2));
      }
      double gdb = bg[band];
      if (res > 0) gdb += -3.0 * (double)res * 0.2;
      return exp2f((float)(gdb * 0.16609640474436813));        // 10^(gdb/20), stored as f32 by the reference
    };
    // scale the bins (mirror implied, :471-476) and fold the spectrum back into the 1024-point input of the inverse:
    //   Ye = (Y[k] + conj Y[M-k]) / 2, Yo = (Y[k] - conj Y[M-k]) / 2 conj(W^k), Z'[k] = Ye + i Yo, Z'[M-k] = conj Ye + i conj Yo
    auto do_pair = [&](int k, float& av_k, float& av_m, int band_k, int band_m) {
      const float2 zk = z[padz(k)], zm = z[padz(kM - k)];
      const float2 w = twB[k];
      const BinPair p = split_pair(zk, zm, w);
      const float2 xk = make_float2(p.xe.x + p.t.x, p.xe.y + p.t.y), xmc = make_float2(p.xe.x - p.t.x, p.xe.y - p.t.y);
      const float gk = bin_gain(sqrtf(xk.x * xk.x + xk.y * xk.y), av_k, band_k);
      const float gm = bin_gain(sqrtf(xmc.x * xmc.x + xmc.y * xmc.y), av_m, band_m);
      if (!synth) return;
      const float2 yk = make_float2(xk.x * gk, xk.y * gk), ymc = make_float2(xmc.x * gm, xmc.y * gm);
      const float2 ye = make_float2(0.5f * (yk.x + ymc.x), 0.5f * (yk.y + ymc.y));
      const float2 yo = cmulc(make_float2(0.5f * (yk.x - ymc.x), 0.5f * (yk.y - ymc.y)), w);
      z[padz(k)] = make_float2(ye.x - yo.y, ye.y + yo.x);
      z[padz(kM - k)] = make_float2(ye.x + yo.y, yo.x - ye.y);
    };
    do_pair(k2, avg[2], avg[3], bnd[2], bnd[3]);
    if (t == 0) {
      const float2 z0 = z[padz(0)], zh = z[padz(512)];
      const float x0 = z0.x + z0.y, xny = z0.x - z0.y;          // X[0] and X[1024] are real; the Nyquist bin is not scaled
      const float g0 = bin_gain(fabsf(x0), avg[0], bnd[0]);
      const float gh = bin_gain(sqrtf(zh.x * zh.x + zh.y * zh.y), avg[1], bnd[1]);
      if (synth) {
        const float y0 = x0 * g0;
        z[padz(0)] = make_float2(0.5f * (y0 + xny), 0.5f * (y0 - xny));
        z[padz(512)] = make_float2(zh.x * gh, zh.y * gh);
      }
    } else {
      do_pair(k1, avg[0], avg[1], bnd[0], bnd[1]);
    }
    if (!synth) continue;                   // uniform across the workgroup
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) a[r] = z[padz(t + 256 * r)];
    fft1024<true>(a, zb, z, twA);             // pass 0 writes zb while others may still read z: no barrier in between
    // synthesis window + overlap-add (fft.js:63-65, dynamic-processor.js:483-487): z[m] = 1024 (y[2m], y[2m+1])
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int m = t + 256 * r, i = 2 * m;
      const float2 zz = zb[padz(m)];
      const float pr0 = f32r((double)(zz.x * (1.0f / kM)) * (double)win[i]);
      const float pr1 = f32r((double)(zz.y * (1.0f / kM)) * (double)win[i + 1]);
      const int64_t s = pos + i;
      const int o = (int)(s & (kN - 1));
      if (s < n) ola[o] = f32r((double)ola[o] + (double)pr0);
      if (s + 1 < n) ola[o + 1] = f32r((double)ola[o + 1] + (double)pr1);
    }
    __syncthreads();
    // samples [pos, pos + 512) are complete: normalise by 4, mix, emit, and free the ring slots
    for (int i = t; i < kHop; i += 256) {
      const int64_t s = pos + i;
      const int o = (int)(s & (kN - 1));
      if (f >= f0) {
        const float w = f32r((double)ola[o] / 4.0);
        y[s] = f32r((double)xin[o] * dry + (double)w * wetm);
      }
      ola[o] = 0.0f;
    }
  }
  __syncthreads();
  if (!last_seg) return;
  // tail: samples after the last frame's first hop keep their partial overlap-add sums (or none)
  const int64_t done = F > 0 ? F * kHop : 0;
  const int64_t ring_end = F > 0 ? (F - 1) * kHop + kN : 0;           // samples below this are in the ring
  for (int64_t s = done + t; s < n; s += 256) {
    float xv, w = 0.0f;
    if (s < ring_end) { const int o = (int)(s & (kN - 1)); xv = xin[o]; w = f32r((double)ola[o] / 4.0); }
    else xv = load_in(s);
    y[s] = f32r((double)xv * dry + (double)w * wetm);
  }
}

// ---------------------------------------------------------------------------- host
static void make_deharsh_plan(double sr, int64_t n, DeharshPlan* p) {
  std::memset(p, 0, sizeof(*p));
  static const double lo[kBands] = {0, 80, 250, 1000, 3000, 6000, 12000}, hi[kBands] = {80, 250, 1000, 3000, 6000, 12000, 20000};
  static const double att[kBands] = {30, 20, 10, 8, 2, 2, 5}, rel[kBands] = {200, 150, 100, 80, 40, 30, 50};
  static const double thr0[kBands] = {-12, -15, -18, -20, -26, -28, -22}, rat0[kBands] = {2, 2.5, 3, 3.5, 5, 6, 3}, knee0[kBands] = {10, 8, 6, 6, 4, 3, 6};
  for (int b = 0; b < kBands; b++) {            // createMasteringPreset, dynamic-processor.js:572-585
    p->thr[b] = thr0[b] + 12; p->knee[b] = knee0[b] + 8; p->ratio[b] = std::fmax(1.2, rat0[b] - 1.5);
    p->ks[b] = p->thr[b] - p->knee[b] / 2; p->ke[b] = p->thr[b] + p->knee[b] / 2;
    p->A[b] = std::exp(-1 / (sr * att[b] / 1000)); p->R[b] = std::exp(-1 / (sr * rel[b] / 1000));
  }
  for (int k = 0; k < kBins; k++) {             // _getBandForFreq :306-313
    const double fr = k * sr / kN;
    int bi = kBands - 1;
    for (int b = 0; b < kBands; b++) if (fr >= lo[b] && fr < hi[b]) { bi = b; break; }
    p->band_of[k] = (uint8_t)bi; p->cnt[bi] += 1;
  }
  p->nframes = n >= kN ? (int64_t)std::floor((double)(n - kN) / kHop) : 0;    // :445
}

int enqueue_deharsh(const float* in, float* out, int ntracks, int nch, int64_t n, double sr, double pre_gain, hipStream_t s) {
  if (n <= 0) return WAM_OK;
  DeharshPlan dp; make_deharsh_plan(sr, n, &dp);
  const int nchan = ntracks * nch;
  const size_t nf = (size_t)(dp.nframes > 0 ? dp.nframes : 1);
  // per-call device scratch (freed after the stream has consumed it)
  DevBuf plan, twb, winb, rms, gdb;
  int rc;
  rc = plan.alloc(sizeof(dp)); if (rc) return rc;
  rc = twb.alloc(sizeof(float2) * (kTwA + kTwB)); if (rc) return rc;
  rc = winb.alloc(sizeof(float) * kN); if (rc) return rc;
  rc = rms.alloc(sizeof(float) * nf * kBands * nchan); if (rc) return rc;
  rc = gdb.alloc(sizeof(double) * nf * kBands * nchan); if (rc) return rc;
  std::vector<float2> tw(kTwA + kTwB);
  std::vector<float> win(kN);
  for (int k = 0; k < kTwA; k++) { const double a = -2 * M_PI * k / kM; tw[k] = make_float2((float)std::cos(a), (float)std::sin(a)); }
  for (int k = 0; k < kTwB; k++) { const double a = -2 * M_PI * k / kN; tw[kTwA + k] = make_float2((float)std::cos(a), (float)std::sin(a)); }
  for (int i = 0; i < kN; i++) win[i] = (float)(0.5 * (1 - std::cos(2 * M_PI * i / (kN - 1))));     // utils.js:98-104
  WAM_HIP(hipMemcpyAsync(plan.p, &dp, sizeof(dp), hipMemcpyHostToDevice, s));
  WAM_HIP(hipMemcpyAsync(twb.p, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice, s));
  WAM_HIP(hipMemcpyAsync(winb.p, win.data(), sizeof(float) * win.size(), hipMemcpyHostToDevice, s));
  if (dp.nframes > 0) {
    int64_t gx = dp.nframes < 2048 ? dp.nframes : 2048;
    WAM_LAUNCH("k_dh_bands", k_dh_bands, dim3((unsigned)gx, (unsigned)nchan), dim3(256), s, in, n, plan.as<DeharshPlan>(), twb.as<float2>(),
               winb.as<float>(), rms.as<float>(), pre_gain);
    WAM_LAUNCH("k_dh_follow", k_dh_follow, dim3((unsigned)(nchan * kBands)), dim3(kThreads), s, (const float*)rms.as<float>(),
               plan.as<DeharshPlan>(), nchan, gdb.as<double>());
  }
  const int64_t warm = tuning().dh_warm_frames;
  const int64_t seg = pick_segment_tiles(nchan, dp.nframes, warm, tuning().dh_seg_frames, 8192);
  const int64_t nseg = dp.nframes > 0 ? (dp.nframes + seg - 1) / seg : 1;
  WAM_LAUNCH("k_dh_apply", k_dh_apply, dim3((unsigned)nseg, (unsigned)nchan), dim3(256), s, in, out, n, plan.as<DeharshPlan>(), twb.as<float2>(),
             winb.as<float>(), (const double*)gdb.as<double>(), pre_gain, seg, warm);
  WAM_HIP(hipStreamSynchronize(s));           // the scratch above is released on return
  return WAM_OK;
}

}  // namespace wam
