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

__device__ __forceinline__ unsigned brev11(unsigned v) { return __brev(v) >> 21; }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)

// forward DIF, natural order in -> bit-reversed order out.  tw[k] = exp(-2 pi i k / N), k < N/2.  256 threads.
__device__ __forceinline__ void fft_fwd(float2* z, const float2* tw) {
  const int t = threadIdx.x;
  for (int m = kN; m >= 8; m >>= 2) {          // fused stages of block size m and m/2
    const int q = m >> 2, step = kN / m;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int g = t + r * 256;               // 512 groups of 4 points
      const int j = g % q, b = (g / q) * m;
      const int p0 = b + j, p1 = p0 + q, p2 = p0 + 2 * q, p3 = p0 + 3 * q;
      const float2 a = z[p0], bb = z[p1], c = z[p2], d = z[p3];
      // stage m: (a,c) twiddle W_m^j, (b,d) twiddle W_m^(j+q)
      float2 s0 = make_float2(a.x + c.x, a.y + c.y), d0 = make_float2(a.x - c.x, a.y - c.y);
      float2 s1 = make_float2(bb.x + d.x, bb.y + d.y), d1 = make_float2(bb.x - d.x, bb.y - d.y);
      d0 = cmul(d0, tw[j * step]);
      d1 = cmul(d1, tw[(j + q) * step]);
      // stage m/2: (s0,s1) and (d0,d1) twiddle W_(m/2)^j
      const float2 w2 = tw[j * step * 2];
      z[p0] = make_float2(s0.x + s1.x, s0.y + s1.y);
      z[p1] = cmul(make_float2(s0.x - s1.x, s0.y - s1.y), w2);
      z[p2] = make_float2(d0.x + d1.x, d0.y + d1.y);
      z[p3] = cmul(make_float2(d0.x - d1.x, d0.y - d1.y), w2);
    }
    __syncthreads();
  }
  // last stage, block size 2 (2048 = 4^5 * 2)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int p = 2 * (t + r * 256);
    const float2 a = z[p], b = z[p + 1];
    z[p] = make_float2(a.x + b.x, a.y + b.y);
    z[p + 1] = make_float2(a.x - b.x, a.y - b.y);
  }
  __syncthreads();
}

// inverse DIT, bit-reversed order in -> natural order out, unscaled (caller multiplies by 1/N)
__device__ __forceinline__ void fft_inv(float2* z, const float2* tw) {
  const int t = threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int p = 2 * (t + r * 256);
    const float2 a = z[p], b = z[p + 1];
    z[p] = make_float2(a.x + b.x, a.y + b.y);
    z[p + 1] = make_float2(a.x - b.x, a.y - b.y);
  }
  __syncthreads();
  for (int m = 8; m <= kN; m <<= 2) {          // fused stages of block size m/2 and m
    const int q = m >> 2, step = kN / m;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int g = t + r * 256;
      const int j = g % q, b = (g / q) * m;
      const int p0 = b + j, p1 = p0 + q, p2 = p0 + 2 * q, p3 = p0 + 3 * q;
      const float2 a = z[p0], bb = z[p1], c = z[p2], d = z[p3];
      // stage m/2: (a,b) and (c,d) with conj W_(m/2)^j
      const float2 w2 = tw[j * step * 2];
      const float2 tb = cmulc(bb, w2), td = cmulc(d, w2);
      const float2 s0 = make_float2(a.x + tb.x, a.y + tb.y), d0 = make_float2(a.x - tb.x, a.y - tb.y);
      const float2 s1 = make_float2(c.x + td.x, c.y + td.y), d1 = make_float2(c.x - td.x, c.y - td.y);
      // stage m: (s0,s1) with conj W_m^j, (d0,d1) with conj W_m^(j+q)
      const float2 t1 = cmulc(s1, tw[j * step]);
      const float2 t3 = cmulc(d1, tw[(j + q) * step]);
      z[p0] = make_float2(s0.x + t1.x, s0.y + t1.y);
      z[p2] = make_float2(s0.x - t1.x, s0.y - t1.y);
      z[p1] = make_float2(d0.x + t3.x, d0.y + t3.y);
      z[p3] = make_float2(d0.x - t3.x, d0.y - t3.y);
    }
    __syncthreads();
  }
}

// Band RMS of every frame.  grid = (frame groups, channels); each workgroup handles frames f = blockIdx.x, += gridDim.x.
__global__ __launch_bounds__(256) void k_dh_bands(const float* __restrict__ in, int64_t n, const DeharshPlan* __restrict__ dp,
                                                  const float2* __restrict__ twg, const float* __restrict__ wing,
                                                  float* __restrict__ band_rms, double pre_gain) {
  __shared__ float2 z[kN];
  __shared__ float2 tw[kN / 2];
  __shared__ float part[4][kBands];
  const DeharshPlan& P = *dp;
  const int64_t chan = blockIdx.y;
  const float* x = in + chan * n;
  const int t = threadIdx.x;
  for (int i = t; i < kN / 2; i += 256) tw[i] = twg[i];
  for (int64_t f = blockIdx.x; f < P.nframes; f += gridDim.x) {
    const int64_t pos = f * kHop;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int i = t + r * 256;
      float v = x[pos + i];
      if (pre_gain != 1.0) v = f32r((double)v * pre_gain);
      if (!(v == v)) v = 0.0f;                                     // (input[i] || 0)
      z[i] = make_float2(f32r((double)v * (double)wing[i]), 0.0f);   // fft.js:43-45
    }
    __syncthreads();
    fft_fwd(z, tw);
    float be[kBands];
#pragma unroll
    for (int b = 0; b < kBands; b++) be[b] = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int k = t + r * 256;
      const float2 c = z[brev11(k)];
      const float mag = sqrtf(c.x * c.x + c.y * c.y);            // dynamic-processor.js:456-460
      const int bnd = P.band_of[k];
#pragma unroll
      for (int b = 0; b < kBands; b++) if (b == bnd) be[b] += mag * mag;
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
  }
}

// Per (channel, band): follower over frames, gain computer.  band_gain_db: [chan][frame][7] float.
__global__ void k_dh_follow(const float* __restrict__ band_rms, const DeharshPlan* __restrict__ dp, int nchan,
                            float* __restrict__ band_gain_db) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= nchan * kBands) return;
  const DeharshPlan& P = *dp;
  const int chan = id / kBands, b = id % kBands;
  double env = 0;
  const double A = P.A[b], R = P.R[b];
  for (int64_t f = 0; f < P.nframes; f++) {
    const double a = fabs((double)band_rms[((int64_t)chan * P.nframes + f) * kBands + b]);
    env = (a > env) ? A * env + (1 - A) * a : R * env + (1 - R) * a;                    // :33-43
    const double edb = env > 0 ? 20 * log10(env) : -INFINITY;
    double gdb;
    if (edb <= P.ks[b]) gdb = 0;                                                         // :75-90
    else if (edb >= P.ke[b]) { const double ex = edb - P.thr[b]; gdb = ex / P.ratio[b] - ex; }
    else { const double kp = (edb - P.ks[b]) / P.knee[b]; const double full = (edb - P.thr[b]) * (1 - 1 / P.ratio[b]); gdb = -full * kp * kp; }
    band_gain_db[((int64_t)chan * P.nframes + f) * kBands + b] = f32r(gdb);
  }
}

// Synthesis.  grid = (segments, channels): a workgroup owns the frames [f0, f1) of its segment and emits the samples
// [f0 * 512, f1 * 512) (the last segment also the tail).  Frame-to-frame state is the per-bin resonance EMA
// (avg' = 0.3 mag + 0.7 avg) and the overlap-add ring; a segment rebuilds both by re-running `warm` frames before
// f0: the EMA forgets its start value as 0.7^warm, and the ring needs the three frames before f0.  Warm-up frames
// more than three before f0 skip the inverse transform.
__global__ __launch_bounds__(256) void k_dh_apply(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                  const DeharshPlan* __restrict__ dp, const float2* __restrict__ twg,
                                                  const float* __restrict__ wing, const float* __restrict__ band_gain_db, double pre_gain,
                                                  int64_t seg_frames, int64_t warm) {
  __shared__ float2 z[kN];
  __shared__ float2 tw[kN / 2];
  __shared__ float win[kN];
  __shared__ float xin[kN];      // input ring: sample s at xin[s & 2047]
  __shared__ float ola[kN];      // overlap-add ring
  __shared__ float bg[kBands];
  const DeharshPlan& P = *dp;
  const int64_t chan = blockIdx.y;
  const float* x = in + chan * n;
  float* y = out + chan * n;
  const int t = threadIdx.x;
  const double dry = 1 - 0.15, wetm = 0.15;                         // preset dryWetMix 0.15, dynamic-processor.js:419-423
  for (int i = t; i < kN / 2; i += 256) tw[i] = twg[i];
  for (int i = t; i < kN; i += 256) { win[i] = wing[i]; ola[i] = 0.0f; }
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
  float avg[4] = {0, 0, 0, 0};
  for (int64_t f = fs; f < f1; f++) {
    const int64_t pos = f * kHop;
    const bool synth = f + 3 >= f0;        // this frame's output reaches samples the segment emits
    __syncthreads();
    // newest 512 samples of this frame: [pos + 1536, pos + 2048)
    for (int i = t; i < kHop; i += 256) { const int64_t s = pos + (kN - kHop) + i; xin[s & (kN - 1)] = load_in(s); }
    if (t < kBands) bg[t] = band_gain_db[(chan * F + f) * kBands + t];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int i = t + r * 256;
      float v = xin[(pos + i) & (kN - 1)];
      if (!(v == v)) v = 0.0f;
      z[i] = make_float2(f32r((double)v * (double)win[i]), 0.0f);
    }
    __syncthreads();
    fft_fwd(z, tw);
    // gains per bin (dynamic-processor.js:354-382), resonance detector (:194-233)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int k = t + r * 256;
      const unsigned p = brev11(k);
      const float2 c = z[p];
      const float mag = sqrtf(c.x * c.x + c.y * c.y);
      float res = 0.0f;
      if (f == fs) avg[r] = mag;            // the detector's first frame (:203-207); a warm start for later segments
      else {
        avg[r] = f32r(0.3 * (double)mag + (1 - 0.3) * (double)avg[r]);
        const double av = avg[r];
        if (av > 1e-10 && (double)mag > av * 1.5) res = f32r(fmin(1.0, ((double)mag / av - 1.5) / 2));
      }
      if (synth) {
        double gdb = bg[P.band_of[k]];
        if (res > 0) gdb += -3.0 * (double)res * 0.2;
        const float g = exp2f((float)(gdb * 0.16609640474436813));      // 10^(gdb/20), stored as f32 by the reference
        const float2 s = make_float2(c.x * g, c.y * g);
        z[p] = s;
        if (k > 0) z[brev11(kN - k)] = make_float2(s.x, -s.y);          // mirror, :471-476
      }
    }
    if (!synth) continue;                   // uniform across the workgroup
    __syncthreads();
    fft_inv(z, tw);
    // synthesis window + overlap-add (fft.js:63-65, dynamic-processor.js:483-487)
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int i = t + r * 256;
      const float pr = f32r((double)(z[i].x * (1.0f / kN)) * (double)win[i]);
      const int64_t s = pos + i;
      if (s < n) { const int o = (int)(s & (kN - 1)); ola[o] = f32r((double)ola[o] + (double)pr); }
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
  rc = twb.alloc(sizeof(float2) * (kN / 2)); if (rc) return rc;
  rc = winb.alloc(sizeof(float) * kN); if (rc) return rc;
  rc = rms.alloc(sizeof(float) * nf * kBands * nchan); if (rc) return rc;
  rc = gdb.alloc(sizeof(float) * nf * kBands * nchan); if (rc) return rc;
  std::vector<float2> tw(kN / 2);
  std::vector<float> win(kN);
  for (int k = 0; k < kN / 2; k++) { const double a = -2 * M_PI * k / kN; tw[k] = make_float2((float)std::cos(a), (float)std::sin(a)); }
  for (int i = 0; i < kN; i++) win[i] = (float)(0.5 * (1 - std::cos(2 * M_PI * i / (kN - 1))));     // utils.js:98-104
  WAM_HIP(hipMemcpyAsync(plan.p, &dp, sizeof(dp), hipMemcpyHostToDevice, s));
  WAM_HIP(hipMemcpyAsync(twb.p, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice, s));
  WAM_HIP(hipMemcpyAsync(winb.p, win.data(), sizeof(float) * win.size(), hipMemcpyHostToDevice, s));
  if (dp.nframes > 0) {
    int64_t gx = dp.nframes < 2048 ? dp.nframes : 2048;
    WAM_LAUNCH("k_dh_bands", k_dh_bands, dim3((unsigned)gx, (unsigned)nchan), dim3(256), s, in, n, plan.as<DeharshPlan>(), twb.as<float2>(),
               winb.as<float>(), rms.as<float>(), pre_gain);
    WAM_LAUNCH("k_dh_follow", k_dh_follow, dim3((unsigned)((nchan * kBands + 63) / 64)), dim3(64), s, (const float*)rms.as<float>(),
               plan.as<DeharshPlan>(), nchan, gdb.as<float>());
  }
  const int64_t seg = tuning().dh_seg_frames, warm = tuning().dh_warm_frames;
  const int64_t nseg = dp.nframes > 0 ? (dp.nframes + seg - 1) / seg : 1;
  WAM_LAUNCH("k_dh_apply", k_dh_apply, dim3((unsigned)nseg, (unsigned)nchan), dim3(256), s, in, out, n, plan.as<DeharshPlan>(), twb.as<float2>(),
             winb.as<float>(), (const float*)gdb.as<float>(), pre_gain, seg, warm);
  WAM_HIP(hipStreamSynchronize(s));           // the scratch above is released on return
  return WAM_OK;
}

}  // namespace wam
