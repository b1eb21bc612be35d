// chain_heavy.hip - the "heavy FX" stages 2..4 of the export chain (dsp-worker.js:1511-1527):
//   k_exciter  : Add Air, exciter.js:80-214 (two 6 kHz Butterworth high-passes, 2x oversampled tanh, additive mix);
//                <PEAKS> also leaves tape warmth's window peaks of its output behind
//   k_win_peak : per-100 ms-window peak of every 4th sample, max over channels (multiband-saturation.js:218-230), when the exciter did not
//   k_warm_env : bypass envelope: soft knee, asymmetric smoothing, one-window lookahead (:232-268)
//   k_warm_punch: Tape Warmth (LR4 3-band split + biased tanh, multiband-saturation.js:280-352) and/or
//                Add Punch (LR4 split + fast/slow followers + smoothed gain, multiband-transient.js:15-263)
// Workgroups: one per (track, channel) and time segment, except k_warm_env (one wavefront per track).
#include "chain.hpp"

namespace wam {

// two more samples of a biquad after the chunk, continuing from its true state
__device__ __forceinline__ void biquad_extra2(const double y[C], const double x[C], double xa, double xb, const Biquad& q, double& ya, double& yb) {
  ya = q.b0 * xa + q.b1 * x[C - 1] + q.b2 * x[C - 2] - q.a1 * y[C - 1] - q.a2 * y[C - 2];
  yb = q.b0 * xb + q.b1 * xa + q.b2 * x[C - 1] - q.a1 * ya - q.a2 * y[C - 1];
}

// PEAKS: the exciter's output is tape warmth's input, whose bypass envelope starts from the per-window peaks of every 4th sample
// (k_win_peak below: a pass over the whole batch at HBM speed).  With PEAKS the kernel takes them from the values it has in
// registers: per chunk two running maxima (the chunk's window and the next one: a chunk crosses at most one boundary), reduced
// over the wavefront - whose 1,024 samples lie in at most two windows, windows being longer than that (checked by the launcher) -
// and merged with atomicMax on the float bit patterns, exactly as k_win_peak merges its partial maxima.
template <bool PEAKS>
__global__ __launch_bounds__(kThreads) void k_exciter(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                      const HeavyPlan* __restrict__ hp, double pre_gain, int64_t seg_tiles, int64_t warm,
                                                      int nch, unsigned* __restrict__ winpeak) {
  __shared__ double lds_scan[kSlotDoubles];
  __shared__ float ctx[1][kCtxFloats];
  const int64_t chan = blockIdx.x;
  const float* x = in + chan * n;
  float* y = out + chan * n;
  LdsSlots slots(lds_scan);
  __shared__ double Ptab[64][4];                 // M^lane of the high-pass (the scans' broadcast steps index it, wam_device.hpp)
  __shared__ double Sst[2][2];                   // tile carries of the two cascaded instances
  if (wave_id() == 0) biquad_lane_power_table(hp->exc_hp, lane_id(), &Ptab[0][0]);
  if (threadIdx.x < 4) Sst[threadIdx.x >> 1][threadIdx.x & 1] = 0;
  lds_barrier();
  const double mix = 0.15 / 100, drive = 0.5, bias = 0.2;     // constants.js:36-42, exciter.js:88
  // segment [T0, T1) of the tiles (blockIdx.y), entered `warm` tiles early: the only state is two 6 kHz high-passes
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const int64_t T0 = (int64_t)blockIdx.y * seg_tiles;
  int64_t T1 = T0 + seg_tiles; if (T1 > ntiles) T1 = ntiles;
  const int64_t Tw = T0 - warm > 0 ? T0 - warm : 0;
  // The next tile's samples are requested before this tile is computed, and nothing else is loaded inside the loop: vmcnt counts
  // loads in order, so waiting for any later load would wait for the prefetch as well.
  float nv[C], nva, nvb;
  auto fetch = [&](int64_t t) {
    const int64_t b = t * kTile + (int64_t)threadIdx.x * C;
    load_chunk(x, n, b, nv);
    nva = b + C < n ? x[b + C] : 0.0f; nvb = b + C + 1 < n ? x[b + C + 1] : 0.0f;
  };
  float pm[2] = { 0.0f, 0.0f };                           // a later segment starts from the samples in memory
  if (Tw > 0 && threadIdx.x == 0) { pm[0] = x[Tw * kTile - 2]; pm[1] = x[Tw * kTile - 1]; }
  // PEAKS: window of this thread's chunk start and the position in it, advanced tile by tile without divisions
  const int ws = PEAKS ? (int)hp->w_ws : 1;
  const int64_t nw = PEAKS ? hp->w_nw : 0;
  int64_t win = PEAKS ? (T0 * kTile + (int64_t)threadIdx.x * C) / ws : 0;
  int wpos = PEAKS ? (int)((T0 * kTile + (int64_t)threadIdx.x * C) % ws) : 0;
  const int tile_q = kTile / ws, tile_r = kTile % ws;
  unsigned* wpk = PEAKS ? winpeak + (chan / nch) * nw : nullptr;
  fetch(Tw);
  for (int64_t t = Tw; t < T1; t++) {
    const HeavyPlan& P = *reload_each_tile(hp);
    const int64_t base = t * kTile + (int64_t)threadIdx.x * C;
    float v[C];
#pragma unroll
    for (int k = 0; k < C; k++) v[k] = nv[k];
    float va = nva, vb = nvb;
    if (t + 1 < T1) fetch(t + 1);
    if (pre_gain != 1.0) {
#pragma unroll
      for (int k = 0; k < C; k++) v[k] = f32r((double)v[k] * pre_gain);
      va = f32r((double)va * pre_gain); vb = f32r((double)vb * pre_gain);
    }
    const float last[2] = { v[C - 2], v[C - 1] };
    float pv[2];
    prev_ctx<2>(last, pv, ctx[0], (int)(t & 1), t == Tw);
    if (t == Tw && Tw > 0 && threadIdx.x == 0) {
      pv[0] = pm[0]; pv[1] = pm[1];
      if (pre_gain != 1.0) { pv[0] = f32r((double)pv[0] * pre_gain); pv[1] = f32r((double)pv[1] * pre_gain); }
    }
    double xin[C], d[C];
#pragma unroll
    for (int k = 0; k < C; k++) { xin[k] = v[k]; d[k] = v[k]; }
    double s0, s1;
    double ea, eb;
    biquad_chunk_lds(d, pv[1], pv[0], P.exc_hp, &Ptab[0][0], Sst[0], slots.next(), s0, s1);
    biquad_extra2(d, xin, va, vb, P.exc_hp, ea, eb);
    double f1[C];
#pragma unroll
    for (int k = 0; k < C; k++) { f1[k] = (double)f32r(d[k]); d[k] = f1[k]; }
    const double f1a = (double)f32r(ea), f1b = (double)f32r(eb);
    double u0, u1;
    double ga, gb;
    biquad_chunk_lds(d, (double)f32r(s0), (double)f32r(s1), P.exc_hp, &Ptab[0][0], Sst[1], slots.next(), u0, u1);
    biquad_extra2(d, f1, f1a, f1b, P.exc_hp, ga, gb);
    // filtered[i-1 .. i+2] around each sample; out-of-range reads AND NaN are 0 (`filtered[k] || 0`, exciter.js:162-184: after a
    // NaN input sample the filter state stays NaN and the reference's wet path falls back to tanh(bias))
    float f[C + 3];                                // (kept as the f32 values they are: half the registers across the shaping loop)
    f[0] = (base - 1 >= 0) ? f32r(u0) : 0.0f;
#pragma unroll
    for (int k = 0; k < C; k++) f[k + 1] = (base + k < n) ? f32r(d[k]) : 0.0f;
    f[C + 1] = (base + C < n) ? f32r(ga) : 0.0f;
    f[C + 2] = (base + C + 1 < n) ? f32r(gb) : 0.0f;
#pragma unroll
    for (int k = 0; k < C + 3; k++) f[k] = (f[k] == f[k]) ? f[k] : 0.0f;
    float o[C];
    // |tanh argument| <= 1 for the whole wavefront's chunks (the Catmull-Rom midpoint overshoots its neighbours by at most 1.25):
    // the Pade form; else the general one.  High-passed programme material is far below the bound.
    float fmx = 0;
#pragma unroll
    for (int k = 0; k < C + 3; k++) fmx = fmaxf(fmx, fabsf(f[k]));
    auto shape = [&](auto tanh_fn) {
#pragma unroll
      for (int k = 0; k < C; k++) {
        const double wetA = tanh_fn((double)f[k + 1] * drive + bias);
        const double sB = catmull_half((double)f[k], (double)f[k + 1], (double)f[k + 2], (double)f[k + 3]);
        const double wetB = tanh_fn(sB * drive + bias);
        const double wet = (wetA + wetB) * 0.5;
        o[k] = f32r((double)v[k] + wet * mix);
      }
    };
    if (__all((double)fmx * (1.25 * drive) + bias <= 1.0)) shape([](double z) { return tanh_small(z); });
    else shape([](double z) { return tanh_fast(z); });
    if (t >= T0) store_chunk(y, n, base, o);
    if constexpr (PEAKS) {
      if (t >= T0) {
        float pa = 0.0f, pb = 0.0f;                  // maxima of |o| over every 4th sample of window `win` / `win + 1` (multiband-saturation.js:218-230)
        if (__any(wpos + C > ws || base + C > n)) {  // a window boundary or the end of the track inside some chunk of the wavefront (one tile in ~5)
#pragma unroll
          for (int k = 0; k < C; k++) {
            int pos = wpos + k;
            const bool nxt = pos >= ws;
            if (nxt) pos -= ws;
            const float a = fabsf(o[k]);
            if ((pos & 3) == 0 && base + k < n) {    // NaN never wins, as in the reference's `sample > peak`
              if (nxt) { if (a > pb) pb = a; } else if (a > pa) pa = a;
            }
          }
        } else {                                     // one window, all sixteen samples inside the track: the four samples k = ph (mod 4)
          const int ph = (-wpos) & 3;
#pragma unroll
          for (int j = 0; j < C / 4; j++) {
            const float a = fabsf(ph == 0 ? o[4 * j] : ph == 1 ? o[4 * j + 1] : ph == 2 ? o[4 * j + 2] : o[4 * j + 3]);
            if (a > pa) pa = a;
          }
        }
        const int64_t wlo = ((int64_t)__builtin_amdgcn_readfirstlane((int)(win >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)win);
        float m0 = win == wlo ? pa : 0.0f, m1 = win == wlo ? pb : pa;      // the wavefront's windows wlo, wlo + 1
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, d)); m1 = fmaxf(m1, __shfl_xor(m1, d)); }
        if (lane_id() == 0) {
          if (m0 > 0 && wlo < nw) atomicMax(&wpk[wlo], __float_as_uint(m0));
          if (m1 > 0 && wlo + 1 < nw) atomicMax(&wpk[wlo + 1], __float_as_uint(m1));
        }
        win += tile_q; wpos += tile_r;
        if (wpos >= ws) { wpos -= ws; win++; }
      }
    }
  }
}

// winpeak: [ntracks][nw] float bit patterns (non-negative floats order like unsigned ints), zero-initialised
__global__ __launch_bounds__(kThreads) void k_win_peak(const float* __restrict__ in, int nch, int64_t n, int64_t ws, int64_t nw,
                                                       unsigned* __restrict__ winpeak, double pre_gain) {
  const int64_t trk = blockIdx.y;
  for (int64_t w = blockIdx.x; w < nw; w += gridDim.x) {
    const int64_t st = w * ws;
    int64_t en = st + ws; if (en > n) en = n;
    float pk = 0;
    for (int64_t i = st + 4 * (int64_t)threadIdx.x; i < en; i += 4 * (int64_t)blockDim.x) {
      for (int c = 0; c < nch; c++) {
        float v = in[(trk * nch + c) * n + i];
        if (pre_gain != 1.0) v = f32r((double)v * pre_gain);
        const float a = fabsf(v);
        if (a > pk) pk = a;                      // NaN never wins, as in the reference's `sample > peak`
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pk = fmaxf(pk, __shfl_xor(pk, d));
    if (lane_id() == 0 && pk > 0) atomicMax(&winpeak[trk * nw + w], __float_as_uint(pk));
  }
}

// One wavefront per track.  The smoothing (multiband-saturation.js:243-253) is a chain of f32 STORES - every step rounds - so it
// is walked in order, by lane 0; what the lanes share is everything around the chain: the window peaks arrive 2,048 at a time
// in coalesced loads, the soft-knee value of each window (:232-241) is computed by all lanes, and the chain itself reads and
// writes LDS four values per access.  (Round 2: one thread per track doing all of it, 36,000 dependent global loads on the
// 60-minute track: 8.1 ms; now the chain's own ~35 cycles per window.)
constexpr int kWarmEnvChunk = 2048;
__global__ __launch_bounds__(64) void k_warm_env(const unsigned* __restrict__ winpeak, int ntracks, int64_t nw, double thr, double kst,
                                                  float* __restrict__ sm) {
#pragma clang fp contract(off)       // smoothed[i - 1] + (envelope[i] - smoothed[i - 1]) * coef: the product is rounded before the sum, as in JavaScript
  __shared__ __align__(16) float ev_s[kWarmEnvChunk];
  __shared__ __align__(16) float sm_s[kWarmEnvChunk];
  const int64_t trk = blockIdx.x;
  const int lane = threadIdx.x;
  const unsigned* wp = winpeak + trk * nw;
  float* s = sm + trk * nw;
  float prev = 0;
  for (int64_t c0 = 0; c0 < nw; c0 += kWarmEnvChunk) {
    const int m = (int)(nw - c0 < kWarmEnvChunk ? nw - c0 : kWarmEnvChunk);
    for (int i = lane; i < m; i += 64) {
      const double peak = (double)__uint_as_float(wp[c0 + i]);
      float ev;
      if (peak >= thr) ev = 1.0f;
      else if (peak >= kst) { const double r = (peak - kst) / (thr - kst); ev = f32r(r * r); }
      else ev = 0.0f;
      ev_s[i] = ev;
    }
    lds_barrier();
    if (lane == 0) {
      int i = 0;
      if (c0 == 0) { prev = ev_s[0]; sm_s[0] = prev; i = 1; }           // smoothed[0] = envelope[0]
      for (; i < m && (i & 3); i++) {
        const float ev = ev_s[i];
        const double coef = ev > prev ? 0.5 : 0.03;
        prev = f32r((double)prev + ((double)ev - (double)prev) * coef);
        sm_s[i] = prev;
      }
      for (; i + 4 <= m; i += 4) {
        const float4 e4 = *reinterpret_cast<const float4*>(&ev_s[i]);
        float4 o;
        { const double c = e4.x > prev ? 0.5 : 0.03; prev = f32r((double)prev + ((double)e4.x - (double)prev) * c); o.x = prev; }
        { const double c = e4.y > prev ? 0.5 : 0.03; prev = f32r((double)prev + ((double)e4.y - (double)prev) * c); o.y = prev; }
        { const double c = e4.z > prev ? 0.5 : 0.03; prev = f32r((double)prev + ((double)e4.z - (double)prev) * c); o.z = prev; }
        { const double c = e4.w > prev ? 0.5 : 0.03; prev = f32r((double)prev + ((double)e4.w - (double)prev) * c); o.w = prev; }
        *reinterpret_cast<float4*>(&sm_s[i]) = o;
      }
      for (; i < m; i++) {
        const float ev = ev_s[i];
        const double coef = ev > prev ? 0.5 : 0.03;
        prev = f32r((double)prev + ((double)ev - (double)prev) * coef);
        sm_s[i] = prev;
      }
    }
    lds_barrier();
    for (int i = lane; i < m; i += 64) s[c0 + i] = sm_s[i];
    lds_barrier();
  }
}
// the look-ahead shift (:255-268): envelope[w] = smoothed[min(w + lookaheadWindows, numWindows - 1)]
__global__ void k_warm_env_shift(const float* __restrict__ sm, int ntracks, int64_t nw, int64_t lw, float* __restrict__ env) {
  const int64_t trk = blockIdx.y;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (int64_t)gridDim.x * blockDim.x) {
    int64_t f = w + (lw > 0 ? lw : 0); if (f > nw - 1) f = nw - 1;
    env[trk * nw + w] = sm[trk * nw + f];
  }
}

// LR4 three-band split of a chunk (multiband-saturation.js:125-148 / multiband-transient.js:94-132).
struct SplitState {                                             // lives in LDS
  double P[4][64][4]; double S[8][2];
  float4 stash[C / 4][kThreads];                                // the band that waits for its turn: 16 VGPRs less across a consume() (conflict-free: lane-major)
};
__device__ __forceinline__ void stash_put(SplitState& st, const float v[C]) {
#pragma unroll
  for (int j = 0; j < C / 4; j++) st.stash[j][threadIdx.x] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}
__device__ __forceinline__ void stash_get(const SplitState& st, float v[C]) {     // same thread, same slots: no barrier needed
#pragma unroll
  for (int j = 0; j < C / 4; j++) { const float4 t = st.stash[j][threadIdx.x]; v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w; }
}
__device__ __forceinline__ void band_put(float4 (*slot)[kThreads], const float v[C]) {
#pragma unroll
  for (int j = 0; j < C / 4; j++) slot[j][threadIdx.x] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}
__device__ __forceinline__ void band_get(float4 (*slot)[kThreads], float v[C]) {
#pragma unroll
  for (int j = 0; j < C / 4; j++) { const float4 t = slot[j][threadIdx.x]; v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w; }
}
__device__ __forceinline__ void split_init(SplitState& st, const Biquad& lp1, const Biquad& hp1, const Biquad& lp2, const Biquad& hp2) {
  if (wave_id() == 0) {
    biquad_lane_power_table(lp1, lane_id(), &st.P[0][0][0]); biquad_lane_power_table(hp1, lane_id(), &st.P[1][0][0]);
    biquad_lane_power_table(lp2, lane_id(), &st.P[2][0][0]); biquad_lane_power_table(hp2, lane_id(), &st.P[3][0][0]);
  }
  if (threadIdx.x < 16) st.S[threadIdx.x >> 1][threadIdx.x & 1] = 0;
  lds_barrier();
}
// Bands are handed to `consume(band_index, float band[C])` one at a time (low, mid, high) so that only one band
// is live in registers while it is processed.
template <class Consume>
__device__ __forceinline__ void split_chunk(const float v[C], float xm1, float xm2, SplitState& st, const Biquad& lp1, const Biquad& hp1,
                                            const Biquad& lp2, const Biquad& hp2, LdsSlots& slots, Consume consume) {
  // The two filters of a split level read the same input, so each cascade stage of the pair runs under one barrier with
  // its two recursions interleaved (biquad_pair_lds): low = lp1(lp1(x)) with midHigh = hp1(hp1(x)), then mid =
  // lp2(lp2(mh)) with high = hp2(hp2(mh)).  The band that waits for its turn is kept as f32 (16 registers).
  double da[C], db[C], a0, a1, b0, b1, c0, c1, e0, e1;
  float band[C];
#pragma unroll
  for (int k = 0; k < C; k++) { da[k] = v[k]; db[k] = v[k]; }
  biquad_pair_lds(da, xm1, xm2, lp1, &st.P[0][0][0], st.S[0], db, xm1, xm2, hp1, &st.P[1][0][0], st.S[2], slots.next(), a0, a1, b0, b1);
#pragma unroll
  for (int k = 0; k < C; k++) { da[k] = (double)f32r(da[k]); db[k] = (double)f32r(db[k]); }
  biquad_pair_lds(da, (double)f32r(a0), (double)f32r(a1), lp1, &st.P[0][0][0], st.S[1],
                  db, (double)f32r(b0), (double)f32r(b1), hp1, &st.P[1][0][0], st.S[3], slots.next(), c0, c1, e0, e1);
  {
    float mh[C];
#pragma unroll
    for (int k = 0; k < C; k++) { band[k] = f32r(da[k]); mh[k] = f32r(db[k]); }
    stash_put(st, mh);                                          // mid + high waits in LDS while the low band is consumed
  }
  const double mh1 = (double)f32r(e0), mh2 = (double)f32r(e1);
  consume(0, band);
  {
    float mh[C];
    stash_get(st, mh);
#pragma unroll
    for (int k = 0; k < C; k++) { da[k] = mh[k]; db[k] = mh[k]; }
  }
  biquad_pair_lds(da, mh1, mh2, lp2, &st.P[2][0][0], st.S[4], db, mh1, mh2, hp2, &st.P[3][0][0], st.S[6], slots.next(), a0, a1, b0, b1);
#pragma unroll
  for (int k = 0; k < C; k++) { da[k] = (double)f32r(da[k]); db[k] = (double)f32r(db[k]); }
  biquad_pair_lds(da, (double)f32r(a0), (double)f32r(a1), lp2, &st.P[2][0][0], st.S[5],
                  db, (double)f32r(b0), (double)f32r(b1), hp2, &st.P[3][0][0], st.S[7], slots.next(), c0, c1, e0, e1);
  {
    float high[C];
#pragma unroll
    for (int k = 0; k < C; k++) { band[k] = f32r(da[k]); high[k] = f32r(db[k]); }
    stash_put(st, high);                                        // and the high band while the mid band is
  }
  consume(1, band);
  stash_get(st, band);
  consume(2, band);
}

// WARM / PUNCH are compile-time so that each instantiation carries only its own filter state: the two stages run as
// two launches (the second in place on the first's output: a tile is fully read before it is written), which costs
// 8 B per sample-channel of extra traffic on a kernel that is bound by f64 issue, not by HBM.
template <bool do_warm, bool do_punch>
__global__ __launch_bounds__(kThreads, 2) void k_warm_punch(const float* __restrict__ in, float* __restrict__ out, int nch, int64_t n,
                                                         const HeavyPlan* __restrict__ hp,
                                                         const float* __restrict__ env, double pre_gain, int64_t seg_tiles, int64_t warm) {
  __shared__ double lds_scan[kSlotDoubles];
  __shared__ float ctx[2][kCtxFloats];
  const int64_t chan = blockIdx.x;
  const int64_t trk = chan / nch;
  const float* x = in + chan * n;
  float* y = out + chan * n;
  const float* ev = env ? env + trk * hp->w_nw : nullptr;
  LdsSlots slots(lds_scan);
  __shared__ SplitState sw_s;                 // one instantiation uses one of the two
  SplitState& sw = sw_s; SplitState& sp = sw_s;
  if (do_warm) split_init(sw, hp->w_lp1, hp->w_hp1, hp->w_lp2, hp->w_hp2);
  if (do_punch) split_init(sp, hp->p_lp1, hp->p_hp1, hp->p_lp2, hp->p_hp2);
  // tile carries of the transient shaper (fast / slow follower and gain smoother per band): the same value in every thread, so they
  // live in LDS like the filters' carries (thread 0 stores, everybody reads at the next tile) instead of 18 VGPRs per thread
  __shared__ double Scar[9];
  if (threadIdx.x < 9) Scar[threadIdx.x] = threadIdx.x < 6 ? 0.0 : 1.0;
  __shared__ double alT[6][C + 2];              // follower_al_table per band: fast [b], slow [3 + b]
  if (do_punch) for (int b = 0; b < 3; b++) { follower_al_table(hp->pf_A[b], hp->pf_R[b], alT[b]); follower_al_table(hp->ps_A[b], hp->ps_R[b], alT[3 + b]); }
  __shared__ float4 bsum[do_punch ? 2 : 1][C / 4][do_punch ? kThreads : 1];   // transient-shaped low and mid bands (own slots per thread: no barrier)
  __shared__ double PsmT[64];                   // the smoother's lane powers: read where they are used, not held across the tile loop
  if (do_punch && threadIdx.x < 64) PsmT[threadIdx.x] = onepole_lane_power(hp->p_smooth, threadIdx.x);
  // bypass-envelope window of this thread's chunk start, advanced tile by tile without divisions
  // segment [T0, T1) of the tiles (blockIdx.y), entered `warm` tiles early (LR4 filters: 2 tiles; the slow transient
  // follower's 250 ms release: 48 tiles)
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const int64_t T0 = (int64_t)blockIdx.y * seg_tiles;
  int64_t T1 = T0 + seg_tiles; if (T1 > ntiles) T1 = ntiles;
  const int64_t Tw = T0 - warm > 0 ? T0 - warm : 0;
  const int64_t ws = hp->w_ws > 0 ? hp->w_ws : 1;
  int64_t win = (Tw * kTile + (int64_t)threadIdx.x * C) / ws, wpos = (Tw * kTile + (int64_t)threadIdx.x * C) % ws;
  const int64_t tile_q = kTile / ws, tile_r = kTile % ws;
  // The next tile's samples (and its two bypass-envelope values) are requested before this tile is computed and nothing else is
  // loaded inside the loop (k_exciter): the loads' latency passes under a tile of arithmetic.
  const int64_t nwl = hp->w_nw - 1;
  float nv[C], ne_here = 0.0f, ne_next = 0.0f;
  auto fetch = [&](int64_t t, int64_t w) {
    load_chunk(x, n, t * kTile + (int64_t)threadIdx.x * C, nv);
    if (do_warm) { ne_here = ev[w < nwl ? w : nwl]; ne_next = ev[w + 1 < nwl ? w + 1 : nwl]; }
  };
  float pm[2] = { 0.0f, 0.0f };                           // a later segment starts from the samples in memory
  if (Tw > 0 && threadIdx.x == 0) { pm[0] = x[Tw * kTile - 2]; pm[1] = x[Tw * kTile - 1]; }
  fetch(Tw, win);
  for (int64_t t = Tw; t < T1; t++) {
    const HeavyPlan& P = *reload_each_tile(hp);
    const int64_t base = t * kTile + (int64_t)threadIdx.x * C;
    const int par = (int)(t & 1);
    float v[C];
#pragma unroll
    for (int k = 0; k < C; k++) v[k] = nv[k];
    const double e_here = (double)ne_here, e_next = (double)ne_next;
    const int64_t to_next = ws - wpos;
    win += tile_q; wpos += tile_r;
    if (wpos >= ws) { wpos -= ws; win++; }
    if (t + 1 < T1) fetch(t + 1, win);
    if (pre_gain != 1.0) {
#pragma unroll
      for (int k = 0; k < C; k++) v[k] = f32r((double)v[k] * pre_gain);
    }
    if constexpr (do_warm) {
      const float last[2] = { v[C - 2], v[C - 1] };
      float pv[2];
      prev_ctx<2>(last, pv, ctx[0], par, t == Tw);
      if (t == Tw && Tw > 0 && threadIdx.x == 0) {
        pv[0] = pm[0]; pv[1] = pm[1];
        if (pre_gain != 1.0) { pv[0] = f32r((double)pv[0] * pre_gain); pv[1] = f32r((double)pv[1] * pre_gain); }
      }
      // envelope[min(floor(i / windowSamples), numWindows - 1)] (multiband-saturation.js:335-337); windowSamples >= C
      // (checked by enqueue_warm_env), so a chunk crosses at most one window boundary
      double acc[C];
      auto saturate = [&](int b, const float band[C]) {
        // saturateBand, multiband-saturation.js:292-308 (gain 0 dB); the three bands are summed left to right (:345)
        const double drive = P.w_drive[b], bias = P.w_bias[b], bo = P.w_bo[b], mk = P.w_mk[b], mix = P.w_mix[b];
        float bmx = 0.0f;
#pragma unroll
        for (int k = 0; k < C; k++) bmx = fmaxf(bmx, fabsf(band[k]));
        auto body = [&](auto tanh_fn) {
#pragma unroll
          for (int k = 0; k < C; k++) {
            if ((k & 3) == 0) __builtin_amdgcn_sched_barrier(0);     // keep a few tanh evaluations in flight, not sixteen
            double e = (k < to_next) ? e_here : e_next;
            if (base + k >= n) e = 0.0;
            const double dry = band[k];
            const double wet = (tanh_fn(drive * (dry + bias)) - bo) * mk;
            const double sa = e * mix;
            const double sat = (double)f32r((dry * (1 - sa) + wet * sa) * 1.0);
            acc[k] = (b == 0) ? sat : acc[k] + sat;
          }
        };
        // |drive (x + bias)| <= 1 for every chunk of the wavefront (drive <= 0.4: bands below ~2.4): the Pade form of tanh
        if (__all(drive * ((double)bmx + fabs(bias)) <= 1.0)) body([](double z) { return tanh_small(z); });
        else body([](double z) { return tanh_fast(z); });
      };
      split_chunk(v, pv[1], pv[0], sw, P.w_lp1, P.w_hp1, P.w_lp2, P.w_hp2, slots, saturate);
#pragma unroll
      for (int k = 0; k < C; k++) v[k] = f32r(acc[k]);
    }
    if constexpr (do_punch) {
      const float last[2] = { v[C - 2], v[C - 1] };
      float pv[2];
      prev_ctx<2>(last, pv, ctx[1], par, t == Tw);
      if (t == Tw && Tw > 0 && threadIdx.x == 0) {
        pv[0] = pm[0]; pv[1] = pm[1];
        if (pre_gain != 1.0) { pv[0] = f32r((double)pv[0] * pre_gain); pv[1] = f32r((double)pv[1] * pre_gain); }
      }
      // the bands' results wait in LDS as the f32 values they are (transient-shaped band, a Float32Array): summed left to right in
      // double at the end (:258-263), they give what a running double sum gives, without 32 registers held across two bands
      auto transient = [&](int b, const float band[C]) {
        // processTransients, multiband-transient.js:194-263
        const double fA = P.pf_A[b], fR = P.pf_R[b], sA = P.ps_A[b], sR = P.ps_R[b], tg = P.p_tg[b] - 1, sg = P.p_sg[b] - 1;
        const double osm = 1 - P.p_smooth.c;
        // rectified input: |band| read through a view (an abs source modifier, no second array of 16 registers).  Samples past the
        // end of the track come after every sample that is stored, so what the followers do with them is immaterial.
        struct AbsView { const float* p; __device__ __forceinline__ float operator[](int k) const { return fabsf(p[k]); } };
        const AbsView a{ band };
        double fe, se;
        double Sfb = Scar[b], Ssb = Scar[3 + b];                  // (read before the solve's first barrier, stored after its last)
        follower_solve2(a, fA, fR, Sfb, sA, sR, Ssb, alT[b], alT[3 + b], slots.next(), fe, se);
        if (threadIdx.x == 0) { Scar[b] = Sfb; Scar[3 + b] = Ssb; }
        double u[C];
#pragma unroll
        for (int k = 0; k < C; k++) {
          fe = follower_step(fe, (double)a[k], fA, fR);
          se = follower_step(se, (double)a[k], sA, sR);
          const double diff = fe - se;
          // strength = min(|diff| / max-guard(slowEnv, 1e-4), 1); gain towards the transient or the sustain target
          const double st = fmin(div_fast(fabs(diff), se > 0.0001 ? se : 0.0001), 1.0);   // den in [1e-4, ~10]
          const double target = 1 + (diff > 0 ? tg : sg) * st;
          u[k] = osm * target;
        }
        double s;
        double Sgb = Scar[6 + b];
        onepole_chunk(u, P.p_smooth, PsmT[lane_id()], Sgb, slots.next(), s);
        if (threadIdx.x == 0) Scar[6 + b] = Sgb;
        float r[C];
#pragma unroll
        for (int k = 0; k < C; k++) r[k] = f32r((double)band[k] * u[k]);
        if (b < 2) { band_put(bsum[b], r); return; }
        float r0[C], r1[C];
        band_get(bsum[0], r0); band_get(bsum[1], r1);
#pragma unroll
        for (int k = 0; k < C; k++) v[k] = f32r(((double)r0[k] + (double)r1[k]) + (double)r[k]);
      };
      split_chunk(v, pv[1], pv[0], sp, P.p_lp1, P.p_hp1, P.p_lp2, P.p_hp2, slots, transient);
    }
    if (t >= T0) store_chunk(y, n, base, v);
  }
}

// ---------------------------------------------------------------------------- launchers
// winpeak (optional, [ntracks][hp_host->w_nw]): also leaves tape warmth's window peaks of the OUTPUT there (zeroed here first);
// exciter_makes_peaks() says whether this sample rate allows it.
bool exciter_makes_peaks(const HeavyPlan& hp_host) { return hp_host.w_ws >= kTile / 4 + C && tuning().exc_peaks != 0; }
int enqueue_exciter(const float* in, float* out, int ntracks, int nch, int64_t n, const HeavyPlan* plan_dev, double pre_gain, double sr, hipStream_t s,
                    unsigned* winpeak, const HeavyPlan* hp_host) {
  if (n <= 0) return WAM_OK;
  const int64_t ntiles = (n + kTile - 1) / kTile, warm = scale_warm(tuning().lti_warm_tiles, sr);
  const int64_t seg = pick_segment_tiles((int64_t)ntracks * nch, ntiles, warm, tuning().fx_seg_tiles);
  const dim3 grid(ntracks * nch, (unsigned)((ntiles + seg - 1) / seg));
  if (winpeak) {
    if (!hp_host || !exciter_makes_peaks(*hp_host)) { set_error("enqueue_exciter: window peaks asked for at a sample rate that does not allow them"); return WAM_ERR_INVALID; }
    WAM_HIP(hipMemsetAsync(winpeak, 0, sizeof(unsigned) * (size_t)hp_host->w_nw * ntracks, s));
    WAM_LAUNCH("k_exciter", k_exciter<true>, grid, dim3(kThreads), s, in, out, n, plan_dev, pre_gain, seg, warm, nch, winpeak);
  } else {
    WAM_LAUNCH("k_exciter", k_exciter<false>, grid, dim3(kThreads), s, in, out, n, plan_dev, pre_gain, seg, warm, nch, (unsigned*)nullptr);
  }
  return WAM_OK;
}

// have_peaks: winpeak was filled by the stage before (enqueue_exciter)
int enqueue_warm_env(const float* in, int ntracks, int nch, int64_t n, const HeavyPlan& hp_host, unsigned* winpeak, float* sm, float* env,
                     double pre_gain, hipStream_t s, bool have_peaks) {
  if (n <= 0) return WAM_OK;
  const int64_t nw = hp_host.w_nw, ws = hp_host.w_ws;
  if (ws < C) { set_error("tape warmth needs a bypass window of at least %d samples (sample rate too low)", C); return WAM_ERR_INVALID; }
  if (!have_peaks) {
    WAM_HIP(hipMemsetAsync(winpeak, 0, sizeof(unsigned) * (size_t)nw * ntracks, s));
    int64_t bx = nw < 512 ? nw : 512;
    WAM_LAUNCH("k_win_peak", k_win_peak, dim3((unsigned)bx, (unsigned)ntracks), dim3(kThreads), s, in, nch, n, ws, nw, winpeak, pre_gain);
  }
  const double thr = std::pow(10, -24 / 20.0), kst = std::pow(10, (-24 - 6) / 20.0);          // multiband-saturation.js:210-211, preset :35
  WAM_LAUNCH("k_warm_env", k_warm_env, dim3((unsigned)ntracks), dim3(64), s, (const unsigned*)winpeak, ntracks, nw, thr, kst, sm);
  int64_t sx = (nw + 255) / 256; if (sx > 256) sx = 256;
  WAM_LAUNCH("k_warm_env", k_warm_env_shift, dim3((unsigned)sx, (unsigned)ntracks), dim3(256), s, (const float*)sm, ntracks, nw, hp_host.w_lw, env);
  return WAM_OK;
}

int enqueue_warm_punch(const float* in, float* out, int ntracks, int nch, int64_t n, const HeavyPlan* plan_dev, int do_warm, int do_punch,
                       const float* env, double pre_gain, double sr, hipStream_t s) {
  if (n <= 0) return WAM_OK;
  if (do_warm && do_punch) { set_error("enqueue_warm_punch: one stage per call (segments read their neighbours' input: no in-place chaining)"); return WAM_ERR_INVALID; }
  if (in == out) { set_error("enqueue_warm_punch: in-place operation is not supported"); return WAM_ERR_INVALID; }
  const int64_t ntiles = (n + kTile - 1) / kTile;
  if (do_warm) {
    const int64_t warm = scale_warm(tuning().lti_warm_tiles, sr), seg = pick_segment_tiles((int64_t)ntracks * nch, ntiles, warm, tuning().fx_seg_tiles);
    WAM_LAUNCH("k_warm", (k_warm_punch<true, false>), dim3(ntracks * nch, (unsigned)((ntiles + seg - 1) / seg)), dim3(kThreads), s, in, out, nch, n, plan_dev, env, pre_gain, seg, warm);
  }
  if (do_punch) {
    // 48 tiles of warm-up: one resident round of workgroups (4 per CU) rather than two
    const int64_t warm = scale_warm(tuning().punch_warm_tiles, sr), seg = pick_segment_tiles((int64_t)ntracks * nch, ntiles, warm, tuning().fx_seg_tiles, 1024);
    WAM_LAUNCH("k_punch", (k_warm_punch<false, true>), dim3(ntracks * nch, (unsigned)((ntiles + seg - 1) / seg)), dim3(kThreads), s, in, out, nch, n, plan_dev, env, pre_gain, seg, warm);
  }
  return WAM_OK;
}

}  // namespace wam
