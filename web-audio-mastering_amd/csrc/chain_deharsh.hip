// chain_deharsh.hip - stage 1 of the export chain: processHybridDynamic(buffer, 'mastering'),
// web/lib/dsp/dynamic-processor.js:595-609 (presets :538-585, _processChannel :444-500,
// _computeBinGains :319-385, ResonanceDetector :194-233, GainComputer :75-90,
// EnvelopeFollower :33-43; STFT wrapper web/lib/dsp/fft.js:41-68; Hann utils.js:98-104).
//
// 2048-point STFT, hop 512, per channel.  The default kernels ("exact mode", second half of this file) compute in double and
// round to f32 wherever the reference stores into a Float32Array:
//   k_dh_bands_x: every frame in parallel -> windowed FFT -> the frame's 1,024 f32 magnitudes into a scratch array
//   k_dh_bandsum: one LANE per frame walks its magnitudes bin by bin: the reference's sequential f32 band sums -> 7 band RMS values
//   k_dh_follow : per (channel, band) the per-frame envelope follower (serial over frames, tiny)
//   k_dh_gain   : the soft-knee gain of every envelope value
//   k_dh_apply_x: one wavefront per (channel, segment of frames) walks its frames in order: FFT, per-bin resonance EMA and
//                 overlap-add ring in registers, gains, inverse FFT, synthesis window, dry/wet mix (<true>: presets other than
//                 'mastering', their parameters from the plan).
// The float32 pair k_dh_bands / k_dh_apply (first half; wam_tune("dh_exact", 0)) has the same frame structure with a float32
// transform, one wavefront per frame, sixteen points per lane (see "the transform" below); the reference's FFT is double (npm
// fft.js, absent: parity unpinned at that boundary).  The mastering preset's wet path is scaled by 0.15 * 0.375, so float32 FFT
// rounding (~1e-6 relative) is ~1e-7 of full scale in the stage's output (DESIGN.md 2 for what the soft clipper makes of it
// on hard-driven masters).
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
  int band_start[kBands + 1];    // band b = bins [band_start[b], band_start[b + 1]) (the bands are ascending frequency ranges)
  uint8_t band_of[kBins];
  // what the presets other than 'mastering' vary (the mastering kernels carry its values as constants):
  double max_cut, sens, mix;     // dynamicEqMaxCut, dynamicEqSensitivity, dryWetMix (:268-273)
  int ai_lo, ai_hi;              // aiArtifactMode: bins with 5000 <= freq <= 12000 (:375), empty range when off
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)

// ---- the transform ------------------------------------------------------------------------------------
// A 2048-point real frame is transformed as a 1024-point complex FFT of z[m] = x[2m] + i x[2m+1] plus an O(N)
// split step (and the mirror image of that for the inverse), which halves the butterfly work and LDS traffic.
//
// One WAVEFRONT transforms one frame: 1024 = 16 x 16 x 4, sixteen points per lane.
//   1. lane l holds z[64 r + l], r = 0..15: a 16-point DFT over r in registers            -> A[k1], k1 = 0..15
//   2. twiddle A[k1] *= W1024^(l k1)   (per-lane constants, read from an LDS table with unit stride across lanes)
//   3. one exchange through LDS (rows of k1, 64 + 4 columns): lane (k1' = lane >> 2, q = lane & 3) takes
//      A'[k1'][q + 4 m], m = 0..15
//   4. a second 16-point DFT over m in registers                                            -> B[j1]
//   5. twiddle B[j1] *= W64^(q j1)
//   6. a radix-4 butterfly across the four lanes of a quad (DPP quad_perm moves, no LDS): lane (k1', q) ends up with
//      Z[k1' + 16 j1 + 256 rev2(q)], j1 = 0..15.
// That is 2 LDS round trips per transform instead of the 5 of a radix-4 pass structure over 256 threads, and the only
// synchronisation inside a transform is the wavefront-local one between the exchange's writes and reads.
constexpr int kM = 1024, kTw1 = 1024, kTwB = 513;    // tables: W1024^k, k < 1024; W2048^k, k <= 512
constexpr int kExRow = 68;                           // exchange row stride (float2): 16 rows land on distinct banks
constexpr int kWaveBuf = 16 * kExRow;                // float2 per wavefront buffer (>= 1024)

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

// a * W16^P (forward) or a * conj(W16^P) (inverse), P a compile-time constant
template <bool INV, int P>
__device__ __forceinline__ float2 mul_w16(float2 a) {
  constexpr float h = 0.70710678118654752f, c = 0.92387953251128674f, s = 0.38268343236508977f;
  if (P == 0) return a;
  if (P == 4) return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
  if (P == 2) return INV ? make_float2(h * (a.x - a.y), h * (a.x + a.y)) : make_float2(h * (a.x + a.y), h * (a.y - a.x));
  if (P == 6) return INV ? make_float2(-h * (a.x + a.y), h * (a.x - a.y)) : make_float2(h * (a.y - a.x), -h * (a.x + a.y));
  // W16^1 = (c, -s), W16^3 = (s, -c), W16^9 = (-c, s)
  const float wr = P == 1 ? c : (P == 3 ? s : -c), wi = P == 1 ? -s : (P == 3 ? -c : s);
  return INV ? cmulc(a, make_float2(wr, wi)) : cmul(a, make_float2(wr, wi));
}

// 16-point DFT in registers, natural order in and out: index r = r0 + 4 r1 in, k = ka + 4 kb out;
// W16^(r k) = W16^(r0 ka) W4^(r0 kb) W4^(r1 ka).
template <bool INV>
__device__ __forceinline__ void dft16(float2 v[16]) {
  float2 T[4][4];
#pragma unroll
  for (int r0 = 0; r0 < 4; r0++) {
    float2 a[4] = { v[r0], v[r0 + 4], v[r0 + 8], v[r0 + 12] };
    r4_butterfly<INV>(a);
#pragma unroll
    for (int ka = 0; ka < 4; ka++) T[r0][ka] = a[ka];
  }
  T[1][1] = mul_w16<INV, 1>(T[1][1]); T[1][2] = mul_w16<INV, 2>(T[1][2]); T[1][3] = mul_w16<INV, 3>(T[1][3]);
  T[2][1] = mul_w16<INV, 2>(T[2][1]); T[2][2] = mul_w16<INV, 4>(T[2][2]); T[2][3] = mul_w16<INV, 6>(T[2][3]);
  T[3][1] = mul_w16<INV, 3>(T[3][1]); T[3][2] = mul_w16<INV, 6>(T[3][2]); T[3][3] = mul_w16<INV, 9>(T[3][3]);
#pragma unroll
  for (int ka = 0; ka < 4; ka++) {
    float2 b[4] = { T[0][ka], T[1][ka], T[2][ka], T[3][ka] };
    r4_butterfly<INV>(b);
#pragma unroll
    for (int kb = 0; kb < 4; kb++) v[ka + 4 * kb] = b[kb];
  }
}

// value of lane ^ 2 / lane ^ 1 within each quad (DPP quad_perm [2,3,0,1] = 0x4E, [1,0,3,2] = 0xB1)
__device__ __forceinline__ float quad_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float quad_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }

// Twiddles live in LDS: twl[64 (k - 1) + lane] = W1024^(lane k), k = 1..15 (unit stride across lanes), and
// twq[16 q + k] = W64^(q k) (lanes of equal q read the same word: a broadcast).  Keeping the 15 per-lane values in
// registers instead costs 30 VGPRs and with them a wavefront of occupancy.
constexpr int kTwl = 15 * 64, kTwq = 64;
struct WaveTw { const float2* l; const float2* q; };
__device__ __forceinline__ void load_wave_tw(const float2* __restrict__ tw1, WaveTw& T, float2* tws) {
  for (int i = threadIdx.x; i < kTwl; i += 256) tws[i] = tw1[(i & 63) * ((i >> 6) + 1)];
  if (threadIdx.x < 64) tws[kTwl + threadIdx.x] = tw1[16 * (threadIdx.x >> 4) * (threadIdx.x & 15)];
  T.l = tws; T.q = tws + kTwl;
}
// output slot of this lane's j1-th result: k1' + 16 j1 + 256 rev2(q)
__device__ __forceinline__ int wave_out_base() { const int lane = lane_id(), q = lane & 3; return (lane >> 2) + 256 * (((q & 1) << 1) | (q >> 1)); }

// v[r] = z[64 r + lane] on entry; on return v[j1] = Z[wave_out_base() + 16 j1].  `ex` = this wavefront's LDS buffer
// (kWaveBuf float2), touched by no other wavefront.  Unnormalised both ways.
// Orders this wavefront's LDS writes before its later LDS reads (other lanes' data) without a workgroup barrier: the
// LDS unit executes one wavefront's instructions in order, so only the compiler has to be kept from reordering.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool INV>
__device__ __forceinline__ void fft1024_wave(float2 v[16], float2* ex, const WaveTw& T) {
  const int lane = lane_id(), q = lane & 3;
  dft16<INV>(v);
#pragma unroll
  for (int k = 1; k < 16; k++) { const float2 w = T.l[64 * (k - 1) + lane]; v[k] = INV ? cmulc(v[k], w) : cmul(v[k], w); }
#pragma unroll
  for (int k = 0; k < 16; k++) ex[k * kExRow + lane] = v[k];
  wave_sync();
  const float2* row = ex + (lane >> 2) * kExRow + q;
#pragma unroll
  for (int m = 0; m < 16; m++) v[m] = row[4 * m];
  dft16<INV>(v);
#pragma unroll
  for (int k = 1; k < 16; k++) { const float2 w = T.q[16 * q + k]; v[k] = INV ? cmulc(v[k], w) : cmul(v[k], w); }
  // radix-4 across the quad: lanes 0,1 get the sums (v0 + v2, v1 + v3), lanes 2,3 the differences; lane 3 turns its
  // difference by -i (forward) / +i (inverse); then even lanes add, odd lanes subtract: lane q holds output rev2(q)
  const float sg1 = (q & 2) ? -1.0f : 1.0f, sg2 = (q & 1) ? -1.0f : 1.0f;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    float2 a = make_float2(fmaf(sg1, v[k].x, quad_xor2(v[k].x)), fmaf(sg1, v[k].y, quad_xor2(v[k].y)));
    if (q == 3) a = INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
    v[k] = make_float2(fmaf(sg2, a.x, quad_xor1(a.x)), fmaf(sg2, a.y, quad_xor1(a.y)));
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

// analysis frame of this lane: raw[r] = (x[pos + 2 m], x[pos + 2 m + 1]), m = 64 r + lane - loads only, so that a caller can issue
// them a frame ahead
__device__ __forceinline__ void load_frame_raw(const float* __restrict__ x, int64_t pos, float2 v[16]) {
  const float* xp = x + pos + 2 * lane_id();
  if (((uintptr_t)xp & 7) == 0) {                                  // uniform: pos is even, so this is the alignment of x
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = *(const float2*)(xp + 128 * r);
  } else {
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = make_float2(xp[128 * r], xp[128 * r + 1]);
  }
}
// input gain, (input[i] || 0), analysis window (fft.js:43-45)
__device__ __forceinline__ void window_frame(float2 v[16], const float* __restrict__ win, double pre_gain) {
  const float2* wp = (const float2*)win + lane_id();               // the window table is 8-byte aligned, 2 m is even
#pragma unroll
  for (int r = 0; r < 16; r++) {
    float v0 = v[r].x, v1 = v[r].y;
    if (pre_gain != 1.0) { v0 = f32r((double)v0 * pre_gain); v1 = f32r((double)v1 * pre_gain); }
    if (!(v0 == v0)) v0 = 0.0f;
    if (!(v1 == v1)) v1 = 0.0f;
    const float2 w = wp[64 * r];
    v[r] = make_float2(v0 * w.x, v1 * w.y);                        // one f32 rounding of the exact product, as f32(double product)
  }
}
__device__ __forceinline__ void load_frame(const float* __restrict__ x, int64_t pos, const float* __restrict__ win, double pre_gain, float2 v[16]) {
  load_frame_raw(x, pos, v);
  window_frame(v, win, pre_gain);
}

// Band RMS of every frame.  grid = (frame groups, channels); wavefront w of a workgroup takes frame 4 g + w of its groups
// g = blockIdx.x, += gridDim.x; lane l then holds the bin pairs (k, 1024 - k), k = l + 64 j, j = 0..7.
__global__ __launch_bounds__(256) void k_dh_bands(const float* __restrict__ in, int64_t n, const DeharshPlan* __restrict__ dp,
                                                  const float2* __restrict__ twg, const float* __restrict__ wing,
                                                  float* __restrict__ band_rms, double pre_gain) {
  __shared__ float2 buf[4][kWaveBuf];
  __shared__ float2 tws[kTwl + kTwq];
  __shared__ float bacc[4][8 * 64];
  const DeharshPlan& P = *dp;
  const float2* __restrict__ twB = twg + kTw1;
  const int64_t chan = blockIdx.y;
  const float* x = in + chan * n;
  const int lane = lane_id(), wave = wave_uniform();
  float2* z = buf[wave];
  float* ba = bacc[wave];
#pragma unroll
  for (int b = 0; b < 8; b++) ba[b * 64 + lane] = 0.0f;
  WaveTw T; load_wave_tw(twg, T, tws);
  lds_barrier();
  const int ob = wave_out_base();
  uint64_t bands = 0;                           // 3 bits per bin: bins k = lane + 64 j at 3 j, bins 1024 - k at 24 + 3 j
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = lane + 64 * j;
    bands |= (uint64_t)P.band_of[k] << (3 * j) | (uint64_t)P.band_of[k == 0 ? 512 : kM - k] << (24 + 3 * j);
  }
  const int64_t ngroups = (P.nframes + 3) / 4;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t f = 4 * g + wave;
    if (f >= P.nframes) break;                                     // wavefronts are independent here: no workgroup barriers below
    float2 v[16];
    load_frame(x, f * kHop, wing, pre_gain, v);
    wave_sync();                                                   // the previous frame's reads of z precede the writes below
    fft1024_wave<false>(v, z, T);
#pragma unroll
    for (int j = 0; j < 16; j++) z[ob + 16 * j] = v[j];
    wave_sync();
    // band sums: every lane adds its 16 squared magnitudes into its own column of a [band][lane] LDS array (no
    // contention, no selects), then the 8 rows are reduced across the wavefront by a halving butterfly (10 exchanges
    // instead of 7 x 6): lane L ends up with the total of band 4 (L & 1) + 2 (L >> 1 & 1) + (L >> 2 & 1).
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = lane + 64 * j;
      float m2k, m2m;                                              // |X|^2 of bins k and 1024 - k (k = 0: DC and bin 512)
      if (k == 0) {
        const float2 z0 = z[0], zh = z[512];
        const float x0 = z0.x + z0.y;
        m2k = x0 * x0; m2m = zh.x * zh.x + zh.y * zh.y;
      } else {
        const BinPair p = split_pair(z[k], z[kM - k], twB[k]);
        const float2 xk = make_float2(p.xe.x + p.t.x, p.xe.y + p.t.y), xm = make_float2(p.xe.x - p.t.x, p.xe.y - p.t.y);
        m2k = xk.x * xk.x + xk.y * xk.y; m2m = xm.x * xm.x + xm.y * xm.y;
      }
      const float mk = sqrtf(m2k), mm = sqrtf(m2m);                // the reference squares the f32 magnitude (:335-337, :456-460)
      const int bk = (int)(bands >> (3 * j)) & 7, bm = (int)(bands >> (24 + 3 * j)) & 7;
      ba[bk * 64 + lane] += mk * mk;
      ba[bm * 64 + lane] += mm * mm;
    }
    float r8[8];
#pragma unroll
    for (int b = 0; b < 8; b++) { r8[b] = ba[b * 64 + lane]; ba[b * 64 + lane] = 0.0f; }
    float r4[4], r2[2], r1;
    {
      const bool hi = lane & 1;
#pragma unroll
      for (int i = 0; i < 4; i++) { const float send = hi ? r8[i] : r8[i + 4], keep = hi ? r8[i + 4] : r8[i]; r4[i] = keep + __shfl_xor(send, 1); }
    }
    {
      const bool hi = lane & 2;
#pragma unroll
      for (int i = 0; i < 2; i++) { const float send = hi ? r4[i] : r4[i + 2], keep = hi ? r4[i + 2] : r4[i]; r2[i] = keep + __shfl_xor(send, 2); }
    }
    {
      const bool hi = lane & 4;
      const float send = hi ? r2[0] : r2[1], keep = hi ? r2[1] : r2[0];
      r1 = keep + __shfl_xor(send, 4);
    }
    r1 += __shfl_xor(r1, 8); r1 += __shfl_xor(r1, 16); r1 += __shfl_xor(r1, 32);
    const int band = ((lane & 1) << 2) | (lane & 2) | ((lane >> 2) & 1);
    if (lane < 8 && band < kBands) {
      const float rms = P.cnt[band] > 0 ? f32r(sqrt((double)r1 / (double)P.cnt[band])) : r1;   // :335-339
      band_rms[(chan * P.nframes + f) * kBands + band] = rms;
    }
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

// Per (channel, band): the envelope follower over frames (EnvelopeFollower.process, :33-43).  The release coefficient is per
// SAMPLE but the follower is stepped per frame, so its memory spans the whole track (no warm-up segmentation); instead one
// workgroup per (channel, band) walks the frames in tiles of 4,096 with the same policy-iteration scan as the per-sample
// followers (an hour at 96 kHz is 337 k frames = 83 tiles).  The walk is the only serial part, so nothing else is on it: the
// next tile's levels are loaded before this tile is solved, and the envelope leaves as doubles - its soft-knee gain (a log10
// per frame and band) is k_dh_gain's, over the whole grid.  env: [chan][frame][7] double.
__global__ __launch_bounds__(kThreads) void k_dh_follow(const float* __restrict__ band_rms, const DeharshPlan* __restrict__ dp, int nchan,
                                                        double* __restrict__ env) {
  __shared__ double lds_scan[kSlotDoubles];
  const DeharshPlan& P = *dp;
  const int chan = blockIdx.x / kBands, b = blockIdx.x % kBands;
  const double A = P.A[b], R = P.R[b];
  const int64_t F = P.nframes;
  const float* src = band_rms + (int64_t)chan * F * kBands + b;
  double* dst = env + (int64_t)chan * F * kBands + b;
  LdsSlots slots(lds_scan);
  __shared__ double flT[C + 2];                                   // follower_al_table of this band's follower
  follower_al_table(A, R, flT);
  lds_barrier();
  double Senv = 0;
  const int64_t ntiles = (F + kTile - 1) / kTile;
  float nx[C];
  // frames past the end: a = 0 keeps the recurrence defined; nothing is stored for them
  auto fetch = [&](int64_t t) {
    const int64_t base = t * kTile + (int64_t)threadIdx.x * C;
#pragma unroll
    for (int k = 0; k < C; k++) nx[k] = (base + k < F) ? fabsf(src[(base + k) * kBands]) : 0.0f;
  };
  fetch(0);
  for (int64_t t = 0; t < ntiles; t++) {
    const int64_t base = t * kTile + (int64_t)threadIdx.x * C;
    float a[C];
#pragma unroll
    for (int k = 0; k < C; k++) a[k] = nx[k];
    if (t + 1 < ntiles) fetch(t + 1);
    double e = follower_solve(a, A, R, Senv, flT, slots.next());
#pragma unroll
    for (int k = 0; k < C; k++) {
      e = follower_step(e, (double)a[k], A, R);
      if (base + k < F) dst[(base + k) * kBands] = e;
    }
  }
}
// the soft-knee gain of every envelope value.  band_gdb: [chan][frame][7] float (the reference keeps the gains in a Float32Array)
__global__ __launch_bounds__(256) void k_dh_gain(const double* __restrict__ env, const DeharshPlan* __restrict__ dp, int64_t total,
                                                 float* __restrict__ band_gdb) {
  const DeharshPlan& P = *dp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    band_gdb[i] = band_gain_db_of(env[i], P, (int)(i % kBands));
}

// Synthesis.  Every WAVEFRONT owns one segment [f0, f1) of one channel's frames and emits the samples [f0 * 512, f1 * 512)
// (the last segment also the tail); the four wavefronts of a workgroup take four neighbouring segments and share nothing
// but the twiddle tables, so after set-up there is no workgroup barrier at all: a wavefront that waits (global load, LDS
// round trip) leaves its SIMD to the wavefronts of other workgroups instead of stalling three siblings at a barrier.
// Frame-to-frame state, all in registers: the per-bin resonance EMA (avg' = 0.3 mag + 0.7 avg) of the lane's bin pairs
// (k, 1024 - k), k = lane + 64 j, and the overlap-add ring slots lane + 64 j, j = 0..31.  A segment rebuilds both by
// re-running `warm` frames before f0: the EMA forgets its start value as 0.7^warm, and the ring needs the three frames
// before f0.  Warm-up frames more than three before f0 skip the inverse transform.  Per frame:
//   windowed frame -> FFT -> Z in LDS -> split, resonance, gain, fold back (own slots) -> inverse FFT, 1/1024,
//   synthesis window -> the frame's 2048 samples in LDS -> overlap-add into the ring registers, emit the 512 samples the
//   frame completes (normalise, 85/15 mix), clear their slots.
__global__ __launch_bounds__(256, 2) void k_dh_apply(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                     const DeharshPlan* __restrict__ dp, const float2* __restrict__ twg,
                                                     const float* __restrict__ wing, const float* __restrict__ band_gdb, double pre_gain,
                                                     int64_t seg_frames, int64_t nseg, int64_t warm) {
  __shared__ float2 buf[4][kWaveBuf];
  __shared__ float2 tws[kTwl + kTwq];
  __shared__ float2 twBs[kM / 2];              // W2048^k, k < 512
  __shared__ float bgs[4][8];
  const DeharshPlan& P = *dp;
  const int64_t chan = blockIdx.y;
  const float* x = in + chan * n;
  float* y = out + chan * n;
  const int lane = lane_id(), wave = wave_uniform();
  const double dry = 1 - 0.15, wetm = 0.15;                         // preset dryWetMix 0.15, dynamic-processor.js:419-423 (doubles: 0.85f is not 0.85)
  const float* win = wing;                       // read through L1: coalesced, the same 8 KB for every frame
  WaveTw T; load_wave_tw(twg, T, tws);
  for (int i = threadIdx.x; i < kM / 2; i += 256) twBs[i] = twg[kTw1 + i];
  lds_barrier();
  const int64_t seg = (int64_t)blockIdx.x * 4 + wave;
  if (seg >= nseg) return;                       // no workgroup barrier below
  auto load_in = [&](int64_t s) -> float {
    float v = (s < n) ? x[s] : 0.0f;
    if (pre_gain != 1.0) v = f32r((double)v * pre_gain);
    return v;
  };
  const int64_t F = P.nframes;
  const int64_t f0 = seg * seg_frames;
  int64_t f1 = f0 + seg_frames; if (f1 > F) f1 = F;
  const int64_t fs = f0 - warm > 0 ? f0 - warm : 0;
  const bool last_seg = f1 >= F;
  const int ob = wave_out_base();
  float2* zw = buf[wave];
  float* bg = bgs[wave];
  uint64_t bands = 0;                           // 3 bits per bin: bins k = lane + 64 j at 3 j, bins 1024 - k at 24 + 3 j
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = lane + 64 * j;
    bands |= (uint64_t)P.band_of[k] << (3 * j) | (uint64_t)P.band_of[k == 0 ? 512 : kM - k] << (24 + 3 * j);
  }
  float avg[16], acc[32];
#pragma unroll
  for (int j = 0; j < 16; j++) avg[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < 32; j++) acc[j] = 0.0f;
  for (int64_t f = fs; f < f1; f++) {
    const bool synth = f + 3 >= f0;            // this frame's output reaches samples the segment emits
    const bool first = f == fs;                // the detector's first frame (:203-207); a warm start for later segments
    float2 v[16];
    load_frame(x, f * kHop, win, pre_gain, v);
    if (lane < kBands) bg[lane] = band_gdb[(chan * F + f) * kBands + lane];
    wave_sync();                               // the previous frame's reads of zw precede the exchange's writes
    fft1024_wave<false>(v, zw, T);
#pragma unroll
    for (int j = 0; j < 16; j++) zw[ob + 16 * j] = v[j];
    wave_sync();
    // resonance detector (:194-233) and bin gain (:354-382) of one bin; returns the linear gain
    // (single precision throughout: the spectrum itself is a float32 FFT, the detector's thresholds are continuous)
    auto bin_gain = [&](float mag, float& av_state, int band) -> float {
      float gdb = bg[band];
      if (first) av_state = mag;
      else {
        av_state = fmaf(0.3f, mag, 0.7f * av_state);
        const float av = av_state;
        if (av > 1e-10f && mag > av * 1.5f) {
          const float res = fminf(1.0f, (mag * __builtin_amdgcn_rcpf(av) - 1.5f) * 0.5f);
          gdb = fmaf(-0.6f, res, gdb);                             // -3 dB * res * 0.2 (:372-376)
        }
      }
      return exp2f(gdb * 0.16609640474436813f);                    // 10^(gdb/20)
    };
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if ((j & 1) == 0) __builtin_amdgcn_sched_barrier(0);      // two pairs in flight, not eight: register pressure
      const int k = lane + 64 * j;
      const int bk = (int)(bands >> (3 * j)) & 7, bm = (int)(bands >> (24 + 3 * j)) & 7;
      if (j == 0 && k == 0) {
        const float2 z0 = zw[0], zh = zw[512];
        const float x0 = z0.x + z0.y, xny = z0.x - z0.y;          // X[0] and X[1024] are real; the Nyquist bin is not scaled
        const float g0 = bin_gain(fabsf(x0), avg[0], bk);
        const float gh = bin_gain(sqrtf(zh.x * zh.x + zh.y * zh.y), avg[1], bm);
        if (synth) {
          const float y0 = x0 * g0;
          zw[0] = make_float2(0.5f * (y0 + xny), 0.5f * (y0 - xny));
          zw[512] = make_float2(zh.x * gh, zh.y * gh);
        }
      } else {
        // scale the bins (mirror implied, :471-476) and fold the spectrum back into the 1024-point input of the inverse:
        //   Ye = (Y[k] + conj Y[M-k]) / 2, Yo = (Y[k] - conj Y[M-k]) / 2 conj(W^k), Z'[k] = Ye + i Yo, Z'[M-k] = conj Ye + i conj Yo
        const float2 zk = zw[k], zm = zw[kM - k];
        const float2 wtw = twBs[k];
        const BinPair p = split_pair(zk, zm, wtw);
        const float2 xk = make_float2(p.xe.x + p.t.x, p.xe.y + p.t.y), xmc = make_float2(p.xe.x - p.t.x, p.xe.y - p.t.y);
        const float gk = bin_gain(sqrtf(xk.x * xk.x + xk.y * xk.y), avg[2 * j], bk);
        const float gm = bin_gain(sqrtf(xmc.x * xmc.x + xmc.y * xmc.y), avg[2 * j + 1], bm);
        if (synth) {
          const float2 yk = make_float2(xk.x * gk, xk.y * gk), ymc = make_float2(xmc.x * gm, xmc.y * gm);
          const float2 ye = make_float2(0.5f * (yk.x + ymc.x), 0.5f * (yk.y + ymc.y));
          const float2 yo = cmulc(make_float2(0.5f * (yk.x - ymc.x), 0.5f * (yk.y - ymc.y)), wtw);
          zw[k] = make_float2(ye.x - yo.y, ye.y + yo.x);
          zw[kM - k] = make_float2(ye.x + yo.y, yo.x - ye.y);
        }
      }
    }
    if (!synth) continue;                      // uniform across the wavefront
    wave_sync();
    // inverse: z[m] = 1024 (y[2m], y[2m+1]); synthesis window (fft.js:63-65)
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = zw[64 * r + lane];
    fft1024_wave<true>(v, zw, T);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int m = ob + 16 * j;
      const float2 wv = ((const float2*)win)[m];
      zw[m] = make_float2((v[j].x * (1.0f / kM)) * wv.x, (v[j].y * (1.0f / kM)) * wv.y);
    }
    wave_sync();
    // overlap-add (dynamic-processor.js:483-487): ring slot o = lane + 64 j holds sample s with s & 2047 = o; this frame
    // covers s = pos + i, i = (o - pos) & 2047, and completes the 512 samples whose slots are j >> 3 = f & 3
    const float* yw = (const float*)zw;
    const int64_t pos = f * kHop;
    const int rot = (int)(pos & (kN - 1)), jdone = (int)(f & 3);
#pragma unroll
    for (int j = 0; j < 32; j++) {
      if ((j & 7) == 0) __builtin_amdgcn_sched_barrier(0);
      const int i = (lane + 64 * j - rot) & (kN - 1);
      acc[j] += yw[i];
      if ((j >> 3) == jdone) {                 // uniform
        if (f >= f0) {
          const int64_t s = pos + i;
          y[s] = f32r((double)load_in(s) * (double)dry + (double)(acc[j] * 0.25f) * (double)wetm);
        }
        acc[j] = 0.0f;
      }
    }
  }
  if (!last_seg) return;
  // tail: samples after the last frame's first hop keep their partial overlap-add sums (or none)
  wave_sync();
  float* ring = (float*)zw;
#pragma unroll
  for (int j = 0; j < 32; j++) ring[lane + 64 * j] = acc[j];
  wave_sync();
  const int64_t done = F > 0 ? F * kHop : 0;
  const int64_t ring_end = F > 0 ? (F - 1) * kHop + kN : 0;           // samples below this are in the ring
  for (int64_t s = done + lane; s < n; s += 64) {
    float w = 0.0f;
    if (s < ring_end) w = ring[(int)(s & (kN - 1))] * 0.25f;
    y[s] = f32r((double)load_in(s) * (double)dry + (double)w * (double)wetm);
  }
}


// =====================================================================================================================
// Exact mode (wam_tune("dh_exact", 1), the default).  The kernels above keep the stage within ~1e-7 of the reference, which is
// one f32 ulp on ~9 % of its output samples; the mastering soft clipper downstream multiplies a one-ulp difference by up to ~550
// on hard-driven material.  The kernels below reproduce every Float32Array store of the reference instead: the
// transform in double (same wavefront-per-frame structure), magnitudes rounded to f32 from the double spectrum, the band
// energies as the reference's SEQUENTIAL f32 sums in bin order (dynamic-processor.js:327-332: k_dh_bandsum, one lane per
// frame), resonance / gain arithmetic in double with f32 stores where the reference has Float32Arrays.
struct dcx { double x, y; };
__device__ __forceinline__ dcx dmul(dcx a, dcx b) { return dcx{ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x }; }
__device__ __forceinline__ dcx dmulc(dcx a, dcx b) { return dcx{ a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y }; }

template <bool INV>
__device__ __forceinline__ void r4_butterfly_d(dcx a[4]) {
  const dcx s02{ a[0].x + a[2].x, a[0].y + a[2].y }, d02{ a[0].x - a[2].x, a[0].y - a[2].y };
  const dcx s13{ a[1].x + a[3].x, a[1].y + a[3].y }, d13{ a[1].x - a[3].x, a[1].y - a[3].y };
  a[0] = dcx{ s02.x + s13.x, s02.y + s13.y };
  a[2] = dcx{ s02.x - s13.x, s02.y - s13.y };
  if (!INV) { a[1] = dcx{ d02.x + d13.y, d02.y - d13.x }; a[3] = dcx{ d02.x - d13.y, d02.y + d13.x }; }
  else { a[1] = dcx{ d02.x - d13.y, d02.y + d13.x }; a[3] = dcx{ d02.x + d13.y, d02.y - d13.x }; }
}
template <bool INV, int P>
__device__ __forceinline__ dcx mul_w16_d(dcx a) {
  constexpr double h = 0.70710678118654752440, c = 0.92387953251128675613, s = 0.38268343236508977173;
  if (P == 0) return a;
  if (P == 4) return INV ? dcx{ -a.y, a.x } : dcx{ a.y, -a.x };
  if (P == 2) return INV ? dcx{ h * (a.x - a.y), h * (a.x + a.y) } : dcx{ h * (a.x + a.y), h * (a.y - a.x) };
  if (P == 6) return INV ? dcx{ -h * (a.x + a.y), h * (a.x - a.y) } : dcx{ h * (a.y - a.x), -h * (a.x + a.y) };
  const double wr = P == 1 ? c : (P == 3 ? s : -c), wi = P == 1 ? -s : (P == 3 ? -c : s);
  return INV ? dmulc(a, dcx{ wr, wi }) : dmul(a, dcx{ wr, wi });
}
template <bool INV>
__device__ __forceinline__ void dft16_d(dcx v[16]) {
  dcx T[4][4];
#pragma unroll
  for (int r0 = 0; r0 < 4; r0++) {
    dcx a[4] = { v[r0], v[r0 + 4], v[r0 + 8], v[r0 + 12] };
    r4_butterfly_d<INV>(a);
#pragma unroll
    for (int ka = 0; ka < 4; ka++) T[r0][ka] = a[ka];
  }
  T[1][1] = mul_w16_d<INV, 1>(T[1][1]); T[1][2] = mul_w16_d<INV, 2>(T[1][2]); T[1][3] = mul_w16_d<INV, 3>(T[1][3]);
  T[2][1] = mul_w16_d<INV, 2>(T[2][1]); T[2][2] = mul_w16_d<INV, 4>(T[2][2]); T[2][3] = mul_w16_d<INV, 6>(T[2][3]);
  T[3][1] = mul_w16_d<INV, 3>(T[3][1]); T[3][2] = mul_w16_d<INV, 6>(T[3][2]); T[3][3] = mul_w16_d<INV, 9>(T[3][3]);
#pragma unroll
  for (int ka = 0; ka < 4; ka++) {
    dcx b[4] = { T[0][ka], T[1][ka], T[2][ka], T[3][ka] };
    r4_butterfly_d<INV>(b);
#pragma unroll
    for (int kb = 0; kb < 4; kb++) v[ka + 4 * kb] = b[kb];
  }
}
// tables (double): twl[64 (k - 1) + lane] = W1024^(lane k), twq[16 q + k] = W64^(q k), twB[k] = W2048^k.  The kernels keep twq and
// twl's k = 1..8 in LDS (what it has room for beside the exchange buffers) and read twB from global memory.
struct WaveTwD { const dcx* l; const dcx* q; };
constexpr int kTwlLds = 6;                                         // W1024^(lane k) kept in LDS for k = 1..6
constexpr int kTwqRow = 17;                                        // W64^(q k), k < 16, in rows of 17: the four q of a quad read four different 16-byte
constexpr int kTwqLds = 4 * kTwqRow;                               // slots of the 256-byte bank row (in rows of 16 every read was a 4-way conflict)
constexpr int kTwsLds = kTwqLds + kTwlLds * 64;                    // dcx entries: W64^(qk), then those
// Bank conflicts of the 16-byte accesses (MI355X: ds_read_b128 is served in four groups of 16 lanes over a 256-byte bank row,
// ds_write_b128 in eight groups of 8 contiguous lanes over 128 bytes; SQ_LDS_BANK_CONFLICT was 44 % of the LDS cycles of the two
// default kernels before).  The spectrum Z[p], p < 1024, lives at slot p + 2 (p >> 6): a lane's results have p = a + 64 q + ...
// (a = lane >> 2, q = lane & 3), so eight contiguous lanes store to slots a + 2 q (mod 8): all different, where the linear layout
// put the four q of a quad on ONE slot (4-way on every store); rows of 64 stay contiguous for the bin-pair reads.
__device__ __forceinline__ constexpr int zslot(int p) { return p + 2 * (p >> 6); }
// The analysis / synthesis window.  The f32 Hann table is symmetric (win[i] == win[2047 - i], checked on the host when the
// table is built), so its first half fits the LDS beside the twiddles: 512 float2; otherwise the kernels read the full
// table from global memory.
// SYM (the table mirrors in f32: its first half is in LDS) is a template parameter, not a flag: with a run-time flag the compiler
// turns `sym ? LDS : global` into a select of generic pointers and every window read into a FLAT load, which counts on vmcnt and
// lgkmcnt alike and so makes the frame wait for the next frame's samples that are meant to stay in flight.
template <bool SYM> struct WinSrc { const float* g; const float2* h; };
template <bool LOW, bool SYM>                                      // LOW: m < 512 known at compile time
__device__ __forceinline__ float2 window_pair(const WinSrc<SYM>& W, int m) {   // (win[2 m], win[2 m + 1])
  if constexpr (!SYM) return ((const float2*)W.g)[m];
  if (LOW) return W.h[m];
  const float2 t = W.h[kM - 1 - m];
  return make_float2(t.y, t.x);
}
__device__ __forceinline__ void load_tables_lds(dcx* tws, float2* winh, const dcx* __restrict__ twd, const float* __restrict__ wing) {
  for (int i = threadIdx.x; i < kTwsLds; i += 256) {
    if (i < kTwqLds) { const int q = i / kTwqRow, k = i % kTwqRow; tws[i] = twd[kTwl + 16 * q + (k < 16 ? k : 0)]; }
    else tws[i] = twd[i - kTwqLds];
  }
  for (int i = threadIdx.x; i < kM / 2; i += 256) winh[i] = ((const float2*)wing)[i];
  lds_barrier();
}
template <bool INV>
__device__ __forceinline__ void fft1024_wave_d(dcx v[16], dcx* ex, const WaveTwD& T) {
  const int lane = lane_id(), q = lane & 3;
  dft16_d<INV>(v);
  dcx w[9];                                                       // W1024^(lane k): k = 1..6 from the table (LDS), then W^7 = W^4 W^3, W^8 = W^4 W^4,
#pragma unroll                                                    // W^(8 + j) = W^8 W^j: at most three products deep
  for (int k = 1; k <= kTwlLds; k++) w[k] = T.l[64 * (k - 1) + lane];
  w[7] = dmul(w[4], w[3]); w[8] = dmul(w[4], w[4]);
#pragma unroll
  for (int k = 1; k <= 8; k++) v[k] = INV ? dmulc(v[k], w[k]) : dmul(v[k], w[k]);
#pragma unroll
  for (int k = 9; k < 16; k++) { const dcx wk = dmul(w[8], w[k - 8]); v[k] = INV ? dmulc(v[k], wk) : dmul(v[k], wk); }
#pragma unroll
  for (int k = 0; k < 16; k++) ex[k * kExRow + lane] = v[k];
  wave_sync();
  const dcx* row = ex + (lane >> 2) * kExRow + q;
#pragma unroll
  for (int m = 0; m < 16; m++) v[m] = row[4 * m];
  dft16_d<INV>(v);
#pragma unroll
  for (int k = 1; k < 16; k++) { const dcx w = T.q[kTwqRow * q + k]; v[k] = INV ? dmulc(v[k], w) : dmul(v[k], w); }
  // Last stage: the radix-4 over the four lanes of a quad.  In double a DPP exchange costs two moves per 32 bits moved (8 per
  // complex value and stage: 256 moves + selects per transform), so the quad is transposed through LDS instead: lane
  // (a, q) = (lane >> 2, lane & 3) stores its sixteen k2-values into row a, slot 4 k2 + ((q + (k2 >> 2)) & 3) - the skew
  // makes both directions conflict-free - and reads back, for its four k2 = 4 q + i, all four q-values; the butterflies
  // then run in registers.  On return v[i + 4 k3] = Z[a + 64 q + 16 i + 256 k3] = Z[wave_out_index_d(i + 4 k3)].
  dcx* tr = ex + (lane >> 2) * kExRow;
#pragma unroll
  for (int k = 0; k < 16; k++) tr[4 * k + ((q + (k >> 2)) & 3)] = v[k];
  wave_sync();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    dcx x[4];
#pragma unroll
    for (int qq = 0; qq < 4; qq++) x[qq] = tr[4 * (4 * q + i) + ((qq + q) & 3)];
    r4_butterfly_d<INV>(x);
#pragma unroll
    for (int k3 = 0; k3 < 4; k3++) v[i + 4 * k3] = x[k3];
  }
}
// W2048^(lane + 64 j) = W2048^lane x W32^j: the lane's own table entry (one load per frame, issued early) times one of eight
// literals, instead of a table load in front of every bin pair's arithmetic (their latency sat on the critical path of the
// pair loop).  One rounding more than the table value (~1e-16), in the noise of the transform.
__device__ __forceinline__ dcx tw2048_lane_j(dcx wl, int j) {
  constexpr dcx W32[8] = { { 1.00000000000000000000, -0.00000000000000000000 }, { 0.98078528040323043058, -0.19509032201612824808 }, { 0.92387953251128673848, -0.38268343236508978178 }, { 0.83146961230254523567, -0.55557023301960217765 }, { 0.70710678118654757274, -0.70710678118654746172 }, { 0.55557023301960228867, -0.83146961230254523567 }, { 0.38268343236508983729, -0.92387953251128673848 }, { 0.19509032201612833135, -0.98078528040323043058 } };
  return j == 0 ? wl : dmul(wl, W32[j]);
}
// output[i] = input[i] * (1 - mix) + output[i] * mix on a Float32Array (dynamic-processor.js:419-423).  Where the stage changes a
// sample by exactly ten f32 ulps the sum sits on an f32 rounding tie (0.15 x 10 = 1.5 ulps) and the last bit of each PRODUCT
// decides: three separately rounded operations as JavaScript does them, not a fused multiply-add (one sample in a thousand otherwise).
__device__ __forceinline__ float mix_dry_wet(float dry_in, double dry, float wet_in, double wet) {
#pragma clang fp contract(off)
  const double a = (double)dry_in * dry, b = (double)wet_in * wet;
  return f32r(a + b);
}
// index p of this lane's j-th result of fft1024_wave_d, and its slot zslot(p) (p >> 6 = q + 4 (j >> 2))
__device__ __forceinline__ int wave_out_base_d() { const int lane = lane_id(); return (lane >> 2) + 64 * (lane & 3); }
__device__ __forceinline__ int wave_out_index_d(int ob, int j) { return ob + 16 * (j & 3) + 256 * (j >> 2); }
__device__ __forceinline__ int wave_out_slot_base_d() { const int lane = lane_id(); return (lane >> 2) + 66 * (lane & 3); }
__device__ __forceinline__ int wave_out_slot_d(int os, int j) { return os + 16 * (j & 3) + 264 * (j >> 2); }
struct BinPairD { dcx xe, t; };
__device__ __forceinline__ BinPairD split_pair_d(dcx zk, dcx zm, dcx w) {
  const dcx bq{ zm.x, -zm.y };
  BinPairD p;
  p.xe = dcx{ 0.5 * (zk.x + bq.x), 0.5 * (zk.y + bq.y) };
  const dcx d{ 0.5 * (zk.x - bq.x), 0.5 * (zk.y - bq.y) };
  p.t = dmul(dcx{ d.y, -d.x }, w);
  return p;
}
// the windowed frame is a Float32Array in the reference (fft.js:30,44): the f32 product IS that store (exact product, one rounding)
// (split from the loads: the kernels issue a frame's loads while the previous frame is still being finished - load_frame_raw -
// and window it here when its turn comes)
template <bool SYM>
__device__ __forceinline__ void window_frame_d(const float2 vf[16], const WinSrc<SYM>& W, double pre_gain, dcx v[16]) {
  const int lane = lane_id();
#pragma unroll
  for (int r = 0; r < 16; r++) {
    float v0 = vf[r].x, v1 = vf[r].y;
    if (pre_gain != 1.0) { v0 = f32r((double)v0 * pre_gain); v1 = f32r((double)v1 * pre_gain); }
    if (!(v0 == v0)) v0 = 0.0f;                                    // (input[i] || 0)
    if (!(v1 == v1)) v1 = 0.0f;
    const float2 w = r < 8 ? window_pair<true, SYM>(W, 64 * r + lane) : window_pair<false, SYM>(W, 64 * r + lane);
    v[r] = dcx{ (double)(v0 * w.x), (double)(v1 * w.y) };
  }
}
// magnitude of a bin as the reference stores it: Float32Array of sqrt(re^2 + im^2) of the double spectrum (:456-460)
// One Goldschmidt step on v_rsq_f64's 2^-24 seed leaves 1.5 seed^2 ~ 4e-15 relative: what the rounding to f32 needs (a value
// within that of a rounding boundary, odds 6e-8 per magnitude, may land on the other side - far rarer than the transform's own
// rounding noise does that on the weak bins); sqrt_fast's two correction steps are for the last bit of the double.
__device__ __forceinline__ float mag_f32(dcx X) {
  const double x = fmax(X.x * X.x + X.y * X.y, 1e-300);          // sqrt(1e-300) rounds to 0.0f: no select for silence
  const double y = __builtin_amdgcn_rsq(x);
  const double g = x * y, h = 0.5 * y;
  return f32r(fma(g, fma(-h, g, 0.5), g));
}

// ---- the reference's sequential f32 band sums ------------------------------------------------------------------------------
// bandEnergies[b] += mag * mag on a Float32Array (dynamic-processor.js:327-332) is s <- RN24(RN53(s + a)) bin after bin: a 512-step
// dependent chain in the widest band that no reordering reproduces.  What is parallel is the FRAMES: k_dh_bands_x leaves the f32
// magnitudes of every frame in a scratch array (4 KB per frame), and k_dh_bandsum walks them with one LANE per frame - sixty-four
// independent chains per wavefront, the band boundaries uniform across the lanes - after transposing [64 frames][32 bins] blocks
// through LDS so that both the loads and the walk are contiguous.  The walk is three dependent instructions per bin and is bound
// by reading the scratch array once (8 B per sample-channel written and read on top of the stage's own traffic).
constexpr int kSumCols = 32, kSumStride = kSumCols + 4;          // floats per LDS row: 16-byte aligned rows

__global__ __launch_bounds__(256) void k_dh_bandsum(const float* __restrict__ mags, const DeharshPlan* __restrict__ dp, float* __restrict__ band_rms) {
  __shared__ float tile[4][64 * kSumStride];
  const DeharshPlan& P = *dp;
  const int lane = lane_id(), wave = wave_uniform();
  const int64_t F = P.nframes, chan = blockIdx.y;
  const int64_t fb = ((int64_t)blockIdx.x * 4 + wave) * 64;      // this wavefront's 64 frames
  if (fb >= F) return;                                           // (no workgroup barrier in this kernel)
  const int64_t f = fb + lane;
  const float* src = mags + chan * F * kBins;
  float* rms_out = band_rms + (chan * F + (f < F ? f : F - 1)) * kBands;
  float* t = tile[wave];
  const int lr = lane >> 3, lc = lane & 7;                       // loader role: rows lr + 8 r, 16-byte column lc
  const float* lsrc[8];
#pragma unroll
  for (int r = 0; r < 8; r++) { int64_t fr = fb + lr + 8 * r; if (fr >= F) fr = F - 1; lsrc[r] = src + fr * kBins + 4 * lc; }
  // (eight named registers, not an array: carried around the block loop an array lands in scratch memory)
#define WAM_SUM_ROWS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define WAM_SUM_LOAD(r) float4 nx##r = *(const float4*)lsrc[r];
  WAM_SUM_ROWS(WAM_SUM_LOAD)
  auto emit = [&](int b, double s) {                             // RMS per band, a Float32Array store (:335-339)
    const float cnt = P.cnt[b];
    const float rms = cnt > 0 ? f32r(sqrt(s / (double)cnt)) : (float)s;
    if (f < F) rms_out[b] = rms;
  };
  int b = 0;
  while (b < kBands && P.band_start[b + 1] == 0) { emit(b, 0.0); b++; }
  int kend = b < kBands ? P.band_start[b + 1] : kBins + 1;       // bands are ascending bin ranges (make_deharsh_plan)
  double s = 0.0;
  auto step = [&](float mf, int knext) {                         // knext = this bin's index + 1
    const double md = mf;
    s = (double)f32r(fma(md, md, s));                            // m * m is exact in double: fma = RN53(s + m * m)
    if (knext == kend) {                                         // uniform: the band ends here
      emit(b, s);
      s = 0.0; b++;
      while (b < kBands && P.band_start[b + 1] == kend) { emit(b, 0.0); b++; }         // empty bands
      kend = b < kBands ? P.band_start[b + 1] : kBins + 1;
    }
  };
  for (int kb = 0; kb < kBins; kb += kSumCols) {
    wave_sync();                                                 // the previous block's rows have been read
#define WAM_SUM_PUT(r) *(float4*)(t + (lr + 8 * r) * kSumStride + 4 * lc) = nx##r;
    WAM_SUM_ROWS(WAM_SUM_PUT)
    wave_sync();
    const int kn = kb + kSumCols < kBins ? kb + kSumCols : kb;    // the next block in flight during the walk (the last one re-reads itself)
#define WAM_SUM_NEXT(r) nx##r = *(const float4*)(lsrc[r] + kn);
    WAM_SUM_ROWS(WAM_SUM_NEXT)
#pragma unroll
    for (int q = 0; q < kSumCols / 4; q++) {
      const float4 v = *(const float4*)(t + lane * kSumStride + 4 * q);
      step(v.x, kb + 4 * q + 1); step(v.y, kb + 4 * q + 2); step(v.z, kb + 4 * q + 3); step(v.w, kb + 4 * q + 4);
    }
  }
#undef WAM_SUM_ROWS
#undef WAM_SUM_LOAD
#undef WAM_SUM_PUT
#undef WAM_SUM_NEXT
}

template <bool SYM>
__global__ __launch_bounds__(256, 2) void k_dh_bands_x(const float* __restrict__ in, int64_t n, const DeharshPlan* __restrict__ dp,
                                                       const dcx* __restrict__ twd, const float* __restrict__ wing,
                                                       float* __restrict__ mags_out, double pre_gain) {
  __shared__ dcx buf[4][kWaveBuf];
  __shared__ dcx tws[kTwsLds];
  __shared__ float2 winh[kM / 2];
  load_tables_lds(tws, winh, twd, wing);
  const DeharshPlan& P = *dp;
  const int64_t chan = blockIdx.y;
  const float* x = in + chan * n;
  const int lane = lane_id(), wave = wave_uniform();
  dcx* z = buf[wave];
  const int os = wave_out_slot_base_d();
  const int c0 = lane == 0 ? 2 : 0;                               // zslot(1024 - k), k = lane + 64 j: 1024 - k + 2 (15 - j) + c0
  const int64_t ngroups = (P.nframes + 3) / 4;
  float2 vf[16];                                                   // the next frame's samples, in flight while this one is transformed
  if (4 * (int64_t)blockIdx.x + wave < P.nframes) load_frame_raw(x, (4 * (int64_t)blockIdx.x + wave) * kHop, vf);
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t f = 4 * g + wave;
    if (f >= P.nframes) break;
    dcx v[16];
    const dcx* twf = reload_each_tile(twd);                       // tables re-read per frame, not hoisted into ~200 registers
    int toff = 0;
    asm volatile("" : "+s"(toff));                                 // same for the LDS tables (an opaque offset keeps them ds_reads)
    const WaveTwD T{ tws + toff + kTwqLds, tws + toff };
    const dcx wl = twf[kTwl + kTwq + lane];                        // W2048^lane
    const WinSrc<SYM> Wn{ reload_each_tile(wing), winh + toff };
    window_frame_d(vf, Wn, pre_gain, v);
    {
      const int64_t fn = f + 4 * (int64_t)gridDim.x;                // this wavefront's next frame
      if (fn < P.nframes) load_frame_raw(x, fn * kHop, vf);
    }
    wave_sync();
    fft1024_wave_d<false>(v, z, T);
#pragma unroll
    for (int j = 0; j < 16; j++) z[wave_out_slot_d(os, j)] = v[j];
    wave_sync();
    float* mg = mags_out + (chan * P.nframes + f) * kBins;         // magnitudes[bin], a Float32Array (:456-460): 256-byte runs per store
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = lane + 64 * j;
      if (k == 0) {
        const dcx z0 = z[0], zh = z[zslot(512)];
        mg[0] = f32r(fabs(z0.x + z0.y)); mg[512] = mag_f32(zh);        // X[0] = Re + Im of Z[0] (real); bin 512 = conj Z[512]
      } else {
        const BinPairD p = split_pair_d(z[k + 2 * j], z[kM - k + 2 * (15 - j) + c0], tw2048_lane_j(wl, j));
        mg[k] = mag_f32(dcx{ p.xe.x + p.t.x, p.xe.y + p.t.y });
        mg[kM - k] = mag_f32(dcx{ p.xe.x - p.t.x, p.xe.y - p.t.y });
      }
    }
    wave_sync();                                                   // everybody has read Z before the next frame's exchange
  }
}

// GEN = false: the 'mastering' preset of the export chain, its constants folded.  GEN = true: max cut, sensitivity, the AI-artifact
// range and the dry/wet mix come from the plan ('gentle', 'aggressive': fully wet, so there is no mix step at all, :419).
template <bool GEN, bool SYM>
__global__ __launch_bounds__(256, 2) void k_dh_apply_x(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                       const DeharshPlan* __restrict__ dp, const dcx* __restrict__ twd,
                                                       const float* __restrict__ wing, const float* __restrict__ band_gdb, double pre_gain,
                                                       int64_t seg_frames, int64_t nseg, int64_t warm) {
  __shared__ dcx buf[4][kWaveBuf];
  __shared__ dcx tws[kTwsLds];
  __shared__ float2 winh[kM / 2];
  __shared__ float bgs[4][8];
  __shared__ double bls[4][8];                                     // mastering preset: 10^(band gain / 20) of the frame, in double
  load_tables_lds(tws, winh, twd, wing);
  const DeharshPlan& P = *dp;
  const int64_t chan = blockIdx.y;
  const float* x = in + chan * n;
  float* y = out + chan * n;
  const int lane = lane_id(), wave = wave_uniform();
  const double wetm = GEN ? P.mix : 0.15, dry = 1 - wetm;
  const bool mixed = !GEN || wetm < 1;
  const int64_t seg = (int64_t)blockIdx.x * 4 + wave;
  if (seg >= nseg) return;
  auto load_in = [&](int64_t s) -> float {
    float v = (s < n) ? x[s] : 0.0f;
    if (pre_gain != 1.0) v = f32r((double)v * pre_gain);
    return v;
  };
  auto mix_out = [&](int64_t s, float wet) -> float {              // :419-423
    return mixed ? mix_dry_wet(load_in(s), dry, wet, wetm) : wet;
  };
  const int64_t F = P.nframes;
  const int64_t f0 = seg * seg_frames;
  int64_t f1 = f0 + seg_frames; if (f1 > F) f1 = F;
  const int64_t fs = f0 - warm > 0 ? f0 - warm : 0;
  const bool last_seg = f1 >= F;
  const int ob = wave_out_base_d(), os = wave_out_slot_base_d();
  const int c0 = lane == 0 ? 2 : 0;                               // zslot(1024 - k), k = lane + 64 j: 1024 - k + 2 (15 - j) + c0
  dcx* zw = buf[wave];
  float* bg = bgs[wave];
  double* bl = bls[wave];
  uint64_t bands = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = lane + 64 * j;
    bands |= (uint64_t)P.band_of[k] << (3 * j) | (uint64_t)P.band_of[k == 0 ? 512 : kM - k] << (24 + 3 * j);
  }
  float avg[16], acc[32];
#pragma unroll
  for (int j = 0; j < 16; j++) avg[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < 32; j++) acc[j] = 0.0f;
  for (int64_t f = fs; f < f1; f++) {
    const bool synth = f + 3 >= f0;
    const bool first = f == fs;
    dcx v[16];
    // the tables are re-read every frame through opaque pointers: hoisted out of the frame loop (they are loop-invariant
    // global loads) they would occupy ~200 registers and force one wavefront per SIMD
    const dcx* twf = reload_each_tile(twd);
    int toff = 0;
    asm volatile("" : "+s"(toff));                                 // same for the LDS tables (an opaque offset keeps them ds_reads)
    WaveTwD T{ tws + toff + kTwqLds, tws + toff };

    WinSrc<SYM> Wn{ reload_each_tile(wing), winh + toff };
    {                                                              // (loads a frame ahead cost this kernel 32 registers it does not have)
      float2 vf[16];
      load_frame_raw(x, f * kHop, vf);
      window_frame_d(vf, Wn, pre_gain, v);
    }
    if (lane < kBands) {
      const float gb = band_gdb[(chan * F + f) * kBands + lane];
      bg[lane] = gb;
      if (!GEN) bl[lane] = exp_fast((double)gb * (2.30258509299404568402 / 20));
    }
    wave_sync();
    fft1024_wave_d<false>(v, zw, T);
#pragma unroll
    for (int j = 0; j < 16; j++) zw[wave_out_slot_d(os, j)] = v[j];
    wave_sync();
    // ResonanceDetector.analyze (:194-233) and the gain of one bin (:361-382), with the reference's Float32Array stores
    auto bin_gain = [&](float mag, float& av_state, int band, int bin) -> double {
      float res = 0.0f;
      if (first) av_state = mag;
      else {
        av_state = f32r(0.3 * (double)mag + (1 - 0.3) * (double)av_state);
        const double av = av_state;
        if (av > 1e-10 && (double)mag > av * 1.5) res = f32r(fmin(1.0, (div_fast((double)mag, av) - 1.5) / 2));
      }
      if (!GEN) {
        // 10^((g + cut) / 20) = 10^(g / 20) e^z with z = ln10 / 20 x cut in [-0.07, 0] (cut = -3 dB x res x 0.2, :369-372): the band's
        // factor once per frame (seven lanes), the cut's by a degree-8 Taylor sum (z^9 / 9! < 1e-16); z = 0 gives the band's factor
        // itself.  Differs from exp((g + cut) ln10 / 20) by ~1e-16 relative, far below the f32 rounding of the store (:320,:381).
        const double z = res > 0 ? (-3.0 * 0.2 * (2.30258509299404568402 / 20)) * (double)res : 0.0;
        double e = 1.0 / 40320;
        e = fma(e, z, 1.0 / 5040); e = fma(e, z, 1.0 / 720); e = fma(e, z, 1.0 / 120); e = fma(e, z, 1.0 / 24);
        e = fma(e, z, 1.0 / 6); e = fma(e, z, 0.5); e = fma(e, z, 1.0); e = fma(e, z, 1.0);
        return (double)f32r(bl[band] * e);
      }
      double gdb = bg[band];
      if (res > 0) gdb += P.max_cut * (double)res * P.sens;                                          // :369-372
      if (bin >= P.ai_lo && bin <= P.ai_hi) gdb += gdb * 0.3;                                        // :375-378
      return (double)f32r(exp_fast(gdb * (2.30258509299404568402 / 20)));   // 10^(gdb / 20) (a division by 20 would be a full IEEE
                                                                            // sequence per bin); gains is a Float32Array (:320,:381)
    };
    const dcx wl = twf[kTwl + kTwq + lane];                        // W2048^lane (the table itself stays in global memory: no LDS left)
#pragma unroll 2                                                   // two bodies in flight (all eight: their loads are hoisted and spill)
    for (int j = 0; j < 8; j++) {
      const int k = lane + 64 * j;
      const int bk = (int)(bands >> (3 * j)) & 7, bm = (int)(bands >> (24 + 3 * j)) & 7;
      if (j == 0 && k == 0) {
        const dcx z0 = zw[0], zh = zw[zslot(512)];
        const double x0 = z0.x + z0.y, xny = z0.x - z0.y;         // X[0] and X[1024] are real; the Nyquist bin is not scaled
        const double g0 = bin_gain(f32r(fabs(x0)), avg[0], bk, 0);
        const double gh = bin_gain(mag_f32(zh), avg[1], bm, 512);
        if (synth) {
          const double y0 = x0 * g0;
          zw[0] = dcx{ 0.5 * (y0 + xny), 0.5 * (y0 - xny) };
          zw[zslot(512)] = dcx{ zh.x * gh, zh.y * gh };
        }
      } else {
        const int sk = k + 2 * j, sm = kM - k + 2 * (15 - j) + c0;       // zslot(k), zslot(1024 - k)
        const dcx zk = zw[sk], zm = zw[sm];
        const dcx wtw = tw2048_lane_j(wl, j);
        const BinPairD p = split_pair_d(zk, zm, wtw);
        const dcx xk{ p.xe.x + p.t.x, p.xe.y + p.t.y }, xmc{ p.xe.x - p.t.x, p.xe.y - p.t.y };
        const double gk = bin_gain(mag_f32(xk), avg[2 * j], bk, k);
        const double gm = bin_gain(mag_f32(xmc), avg[2 * j + 1], bm, kM - k);
        if (synth) {
          const dcx yk{ xk.x * gk, xk.y * gk }, ymc{ xmc.x * gm, xmc.y * gm };
          const dcx ye{ 0.5 * (yk.x + ymc.x), 0.5 * (yk.y + ymc.y) };
          const dcx yo = dmulc(dcx{ 0.5 * (yk.x - ymc.x), 0.5 * (yk.y - ymc.y) }, wtw);
          zw[sk] = dcx{ ye.x - yo.y, ye.y + yo.x };
          zw[sm] = dcx{ ye.x + yo.y, yo.x - ye.y };
        }
      }
    }
    if (!synth) continue;                                          // a warm-up frame: analysis only
    wave_sync();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = zw[66 * r + lane];      // zslot(64 r + lane)
    twf = reload_each_tile(twf);
    asm volatile("" : "+s"(toff));
    T = WaveTwD{ tws + toff + kTwqLds, tws + toff };
    fft1024_wave_d<true>(v, zw, T);
    Wn = WinSrc<SYM>{ reload_each_tile(wing), winh + toff };
    wave_sync();                                                   // the exchange's reads precede the float writes below (other layout)
    float2* yf = (float2*)zw;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int m = wave_out_index_d(ob, j);
      // outputBuffer[i] = inverseBuffer[2 i] * window[i], a Float32Array (fft.js:63-65); inverse includes 1 / N
      // m = ob + 16 (j & 3) + 256 (j >> 2) with ob + 16 (j & 3) < 256: the half of the table is known from j alone
      const float2 ws = j < 8 ? window_pair<true, SYM>(Wn, m) : window_pair<false, SYM>(Wn, m);   // the inverse's 1 / N is an exact scaling: applied to the window
      // pair m at float2 slot m + 4 (m >> 6) (m >> 6 = q + 4 (j >> 2)): sixteen contiguous lanes - four a, four q - then store to
      // sixteen different slots of the 128-byte row ds_write_b64 banks over, not to four
      yf[m + 4 * (lane & 3) + 16 * (j >> 2)] = make_float2(f32r(v[j].x * (double)(ws.x * (1.0f / kM))), f32r(v[j].y * (double)(ws.y * (1.0f / kM))));
    }
    wave_sync();
    const float* yw = (const float*)zw;
    const int64_t pos = f * kHop;
    const int rot = (int)(pos & (kN - 1)), jdone = (int)(f & 3);
    // This frame completes its first hop: ring slots 8 jdone .. 8 jdone + 7 of every lane = samples pos + lane + 64 jj.  Their dry
    // inputs are loaded together and ahead of the ring's LDS reads (inside the per-slot branch each load sat in front of its use).
    const bool emit = f >= f0;
    float dryv[8];
    if (emit && mixed) {
#pragma unroll
      for (int jj = 0; jj < 8; jj++) dryv[jj] = load_in(pos + lane + 64 * jj);
    }
#pragma unroll
    for (int j = 0; j < 32; j++) { const int i = (lane + 64 * j - rot) & (kN - 1); acc[j] += yw[i + 8 * (i >> 7)]; }   // (float i of the frame at i + 8 (i >> 7))
#define WAM_DH_FINISH(G)                                                                              \
    _Pragma("unroll") for (int jj = 0; jj < 8; jj++) {                                                \
      if (emit) {                                                                                     \
        const float wet = f32r((double)acc[8 * G + jj] / 4.0);                                        \
        y[pos + lane + 64 * jj] = mixed ? mix_dry_wet(dryv[jj], dry, wet, wetm) : wet;               \
      }                                                                                               \
      acc[8 * G + jj] = 0.0f;                                                                         \
    }
    if (jdone == 0) { WAM_DH_FINISH(0) } else if (jdone == 1) { WAM_DH_FINISH(1) } else if (jdone == 2) { WAM_DH_FINISH(2) } else { WAM_DH_FINISH(3) }
#undef WAM_DH_FINISH
  }
  if (!last_seg) return;
  wave_sync();
  float* ring = (float*)zw;
#pragma unroll
  for (int j = 0; j < 32; j++) ring[lane + 64 * j] = acc[j];
  wave_sync();
  const int64_t done = F > 0 ? F * kHop : 0;
  const int64_t ring_end = F > 0 ? (F - 1) * kHop + kN : 0;
  for (int64_t s = done + lane; s < n; s += 64) {
    float w = 0.0f;
    if (s < ring_end) w = f32r((double)ring[(int)(s & (kN - 1))] / 4.0);
    y[s] = mix_out(s, w);
  }
}

// ---------------------------------------------------------------------------- host
static void make_deharsh_plan(double sr, int64_t n, int preset, DeharshPlan* p) {
  std::memset(p, 0, sizeof(*p));
  static const double lo[kBands] = {0, 80, 250, 1000, 3000, 6000, 12000}, hi[kBands] = {80, 250, 1000, 3000, 6000, 12000, 20000};
  static const double att[kBands] = {30, 20, 10, 8, 2, 2, 5}, rel[kBands] = {200, 150, 100, 80, 40, 30, 50};
  static const double thr0[kBands] = {-12, -15, -18, -20, -26, -28, -22}, rat0[kBands] = {2, 2.5, 3, 3.5, 5, 6, 3}, knee0[kBands] = {10, 8, 6, 6, 4, 3, 6};
  const bool mastering = preset == WAM_DEHARSH_MASTERING, aggressive = preset == WAM_DEHARSH_AGGRESSIVE;
  for (int b = 0; b < kBands; b++) {            // createMasteringPreset :572-585, createAggressivePreset :555-566, createGentlePreset :538-549
    if (mastering) { p->thr[b] = thr0[b] + 12; p->knee[b] = knee0[b] + 8; p->ratio[b] = std::fmax(1.2, rat0[b] - 1.5); }
    else if (aggressive) { p->thr[b] = thr0[b] - 6; p->knee[b] = knee0[b]; p->ratio[b] = rat0[b] + 2; }
    else { p->thr[b] = thr0[b] + 6; p->knee[b] = knee0[b]; p->ratio[b] = std::fmax(2.0, rat0[b] - 1); }
    p->ks[b] = p->thr[b] - p->knee[b] / 2; p->ke[b] = p->thr[b] + p->knee[b] / 2;
    p->A[b] = std::exp(-1 / (sr * att[b] / 1000)); p->R[b] = std::exp(-1 / (sr * rel[b] / 1000));
  }
  for (int k = 0; k < kBins; k++) {             // _getBandForFreq :306-313
    const double fr = k * sr / kN;
    int bi = kBands - 1;
    for (int b = 0; b < kBands; b++) if (fr >= lo[b] && fr < hi[b]) { bi = b; break; }
    p->band_of[k] = (uint8_t)bi; p->cnt[bi] += 1;
  }
  for (int b = 0; b <= kBands; b++) p->band_start[b] = kBins;
  for (int k = kBins - 1; k >= 0; k--) p->band_start[p->band_of[k]] = k;        // first bin of every band that has bins
  for (int b = kBands - 1; b >= 0; b--) if (p->band_start[b] > p->band_start[b + 1]) p->band_start[b] = p->band_start[b + 1];   // empty bands
  p->nframes = n >= kN ? (int64_t)std::floor((double)(n - kN) / kHop) : 0;    // :445
  p->max_cut = mastering ? -3 : aggressive ? -18 : -6;
  p->sens = mastering ? 0.2 : aggressive ? 0.8 : 0.3;
  p->mix = mastering ? 0.15 : 1.0;                                            // constructor default :268
  p->ai_lo = 1; p->ai_hi = 0;
  if (!mastering) {                                                           // aiArtifactMode (default true :276): an interval of bins
    p->ai_lo = kBins; p->ai_hi = -1;
    for (int k = 0; k < kBins; k++) { const double fr = k * sr / kN; if (fr >= 5000 && fr <= 12000) { if (k < p->ai_lo) p->ai_lo = k; p->ai_hi = k; } }
  }
}

int enqueue_deharsh(const float* in, float* out, int ntracks, int nch, int64_t n, double sr, double pre_gain, hipStream_t s, int preset) {
  if (n <= 0) return WAM_OK;
  const bool mastering = preset == WAM_DEHARSH_MASTERING;
  DeharshPlan dp; make_deharsh_plan(sr, n, preset, &dp);
  const int nchan = ntracks * nch;
  const size_t nf = (size_t)(dp.nframes > 0 ? dp.nframes : 1);
  // per-call device scratch (freed after the stream has consumed it)
  DevBuf plan, twb, winb, rms, gdb;
  int rc;
  rc = plan.alloc(sizeof(dp)); if (rc) return rc;
  rc = twb.alloc(sizeof(float2) * (kTw1 + kTwB)); if (rc) return rc;
  rc = winb.alloc(sizeof(float) * kN); if (rc) return rc;
  rc = rms.alloc(sizeof(float) * nf * kBands * nchan); if (rc) return rc;
  rc = gdb.alloc(sizeof(float) * nf * kBands * nchan); if (rc) return rc;
  DevBuf envd;
  rc = envd.alloc(sizeof(double) * nf * kBands * nchan); if (rc) return rc;
  std::vector<float2> tw(kTw1 + kTwB);
  std::vector<float> win(kN);
  for (int k = 0; k < kTw1; k++) { const double a = -2 * M_PI * k / kM; tw[k] = make_float2((float)std::cos(a), (float)std::sin(a)); }
  for (int k = 0; k < kTwB; k++) { const double a = -2 * M_PI * k / kN; tw[kTw1 + k] = make_float2((float)std::cos(a), (float)std::sin(a)); }
  for (int i = 0; i < kN; i++) win[i] = (float)(0.5 * (1 - std::cos(2 * M_PI * i / (kN - 1))));     // utils.js:98-104
  int win_sym = 1;                                                // the exact kernels keep half the table in LDS if it mirrors (it does with glibc's cos)
  for (int i = 0; i < kN / 2; i++) if (win[i] != win[kN - 1 - i]) win_sym = 0;
  if (!tuning().dh_win_lds) win_sym = 0;
  WAM_HIP(hipMemcpyAsync(plan.p, &dp, sizeof(dp), hipMemcpyHostToDevice, s));
  WAM_HIP(hipMemcpyAsync(twb.p, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice, s));
  WAM_HIP(hipMemcpyAsync(winb.p, win.data(), sizeof(float) * win.size(), hipMemcpyHostToDevice, s));
  const bool exact = tuning().dh_exact != 0 || !mastering;     // fully wet presets: nothing scales the float32 transform's rounding down
  DevBuf twd;
  if (exact) {                                     // double tables: twl[15][64] lane-major, twq[4][16], twB[513]
    std::vector<dcx> td(kTwl + kTwq + kTwB);
    auto W = [](double num, double den) { const double a = -2 * M_PI * num / den; return dcx{ std::cos(a), std::sin(a) }; };
    for (int i = 0; i < kTwl; i++) td[i] = W((double)((i & 63) * ((i >> 6) + 1)), kM);
    for (int i = 0; i < kTwq; i++) td[kTwl + i] = W((double)(16 * (i >> 4) * (i & 15)), kM);
    for (int k = 0; k < kTwB; k++) td[kTwl + kTwq + k] = W((double)k, kN);
    rc = twd.alloc(sizeof(dcx) * td.size()); if (rc) return rc;
    WAM_HIP(hipMemcpyAsync(twd.p, td.data(), sizeof(dcx) * td.size(), hipMemcpyHostToDevice, s));
    WAM_HIP(hipStreamSynchronize(s));              // td is a host temporary
  }
  if (dp.nframes > 0) {
    const int64_t ngroups = (dp.nframes + 3) / 4;
    int64_t gx = ngroups < 2048 ? ngroups : 2048;
    if (exact) {
      // the magnitudes wait in scratch between the two kernels: 4 KB per frame, so a large batch goes through in groups of channels
      const int64_t cap = tuning().dh_mag_bytes > 0 ? tuning().dh_mag_bytes : (int64_t)4 << 30;
      int64_t sub = cap / ((int64_t)sizeof(float) * kBins * dp.nframes); if (sub < 1) sub = 1; if (sub > nchan) sub = nchan;
      DevBuf mags;
      rc = mags.alloc(sizeof(float) * kBins * (size_t)dp.nframes * (size_t)sub); if (rc) return rc;
      for (int64_t c0 = 0; c0 < nchan; c0 += sub) {
        const int64_t nc = nchan - c0 < sub ? nchan - c0 : sub;
        int64_t gxx = 8192 / nc; if (gxx < 1) gxx = 1; if (gxx > ngroups) gxx = ngroups;
        if (win_sym) WAM_LAUNCH("k_dh_bands_x", k_dh_bands_x<true>, dim3((unsigned)gxx, (unsigned)nc), dim3(256), s, in + c0 * n, n, plan.as<DeharshPlan>(), (const dcx*)twd.as<dcx>(),
                   winb.as<float>(), mags.as<float>(), pre_gain);
        else WAM_LAUNCH("k_dh_bands_x", k_dh_bands_x<false>, dim3((unsigned)gxx, (unsigned)nc), dim3(256), s, in + c0 * n, n, plan.as<DeharshPlan>(), (const dcx*)twd.as<dcx>(),
                   winb.as<float>(), mags.as<float>(), pre_gain);
        WAM_LAUNCH("k_dh_bandsum", k_dh_bandsum, dim3((unsigned)((dp.nframes + 255) / 256), (unsigned)nc), dim3(256), s, (const float*)mags.as<float>(),
                   plan.as<DeharshPlan>(), rms.as<float>() + c0 * dp.nframes * kBands);
      }
      WAM_HIP(hipStreamSynchronize(s));            // mags is released here
    }
    else WAM_LAUNCH("k_dh_bands", k_dh_bands, dim3((unsigned)gx, (unsigned)nchan), dim3(256), s, in, n, plan.as<DeharshPlan>(), twb.as<float2>(),
               winb.as<float>(), rms.as<float>(), pre_gain);
    WAM_LAUNCH("k_dh_follow", k_dh_follow, dim3((unsigned)(nchan * kBands)), dim3(kThreads), s, (const float*)rms.as<float>(),
               plan.as<DeharshPlan>(), nchan, envd.as<double>());
    const int64_t total = (int64_t)dp.nframes * kBands * nchan;
    int64_t gg = (total + 255) / 256; if (gg > 16384) gg = 16384;
    WAM_LAUNCH("k_dh_gain", k_dh_gain, dim3((unsigned)gg), dim3(256), s, (const double*)envd.as<double>(), plan.as<DeharshPlan>(), total, gdb.as<float>());
  }
  // segments are walked by wavefronts, four to a workgroup: aim at ~16 k wavefronts (12 are resident per CU)
  // exact mode: the resonance EMA (an f32 state) must forget its start far below one f32 rounding, not to ~1e-7:
  // 0.7^96 = 1.4e-15 is 2e-8 of an ulp, the odds per bin and segment that the warm-started state rounds differently
  const int64_t warm = exact && tuning().dh_warm_frames < 96 ? 96 : tuning().dh_warm_frames;
  const int64_t seg = pick_segment_tiles(nchan, dp.nframes, warm, tuning().dh_seg_frames, exact ? 8192 : 16384, 2048);
  const int64_t nseg = dp.nframes > 0 ? (dp.nframes + seg - 1) / seg : 1;
#define WAM_DH_APPLY_X(name, GEN, SYM) WAM_LAUNCH(name, (k_dh_apply_x<GEN, SYM>), dim3((unsigned)((nseg + 3) / 4), (unsigned)nchan), dim3(256), s, in, out, n, \
    plan.as<DeharshPlan>(), (const dcx*)twd.as<dcx>(), winb.as<float>(), (const float*)gdb.as<float>(), pre_gain, seg, nseg, warm)
  if (exact && mastering) { if (win_sym) WAM_DH_APPLY_X("k_dh_apply_x", false, true); else WAM_DH_APPLY_X("k_dh_apply_x", false, false); }
  else if (exact) { if (win_sym) WAM_DH_APPLY_X("k_dh_apply_xg", true, true); else WAM_DH_APPLY_X("k_dh_apply_xg", true, false); }
#undef WAM_DH_APPLY_X
  else WAM_LAUNCH("k_dh_apply", k_dh_apply, dim3((unsigned)((nseg + 3) / 4), (unsigned)nchan), dim3(256), s, in, out, n, plan.as<DeharshPlan>(), twb.as<float2>(),
                  winb.as<float>(), (const float*)gdb.as<float>(), pre_gain, seg, nseg, warm);
  WAM_HIP(hipStreamSynchronize(s));           // the scratch above is released on return
  return WAM_OK;
}

}  // namespace wam
