// chain_tail.hip - stages 8..10 of the export chain:
//   stage 8 gain (normalizer.js:50-72) is folded into the loads of stage 9 (pre-gain G per track)
//   stage 9 mastering soft clipper                  soft-clipper.js:165-294
//       k_sc       : lookahead gain envelope (ramp minimum over the next L samples), release smoothing, gain and the 2x
//                    Catmull-Rom oversampled saturation, one pass
//   stage 10 lookahead true-peak limiter             dsp-worker.js:286-452
//       k_lim_tm   : transient map (10 ms running RMS, 1 ms / 50 ms peak follower, crest factor)
//       k_lim_gain : true-peak driven lookahead ramps (events are serial, each ramp wave-parallel) + release
//       k_lim_apply: gain, oversampled soft knee, and the loudness / true-peak measurement of the result
#include "chain.hpp"
#include <cmath>

namespace wam {

__device__ __forceinline__ int padi(int i) { return i + (i >> 4); }
constexpr int kClipMaxL = 1024;                 // soft-clip look-ahead held in LDS: 0.5 ms up to 2 MHz

// exclusive block sum of one double per thread; total returned in `total`.  lds: 4 doubles.
__device__ __forceinline__ double block_excl_sum(double v, double& total, double* lds) {
  const int lane = lane_id(), wave = wave_id();
  double inc = v;
  inc += dpp_d<0x111, 0xF>(0.0, inc); inc += dpp_d<0x112, 0xF>(0.0, inc); inc += dpp_d<0x114, 0xF>(0.0, inc); inc += dpp_d<0x118, 0xF>(0.0, inc);
  inc += dpp_d<0x142, 0xA>(0.0, inc); inc += dpp_d<0x143, 0xC>(0.0, inc);          // rows, then the row aggregates (wam_device.hpp)
  if (lane == 63) lds[wave] = inc;
  lds_barrier();
  double off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) { if (w < wave) off += lds[w]; tot += lds[w]; }
  total = tot;
  return off + inc - v;
}

// ============================================================================ soft clipper
// One kernel, grid = (segments, tracks): gain envelope, release smoothing and the oversampled saturation of a tile while its
// samples are in registers (the envelope never goes to memory: 8 B per sample-channel read and written, nothing else).
// A sample over the threshold (an "event") pulls the gain of the L samples before it down along a ramp
// (soft-clipper.js:194-219): every event scatters its ramp into an LDS window of the tile's gains with ds_min on the float
// bit patterns (all values are positive, and f32(min(old, new)) == min(old, f32(new))), so the cost follows the number of
// events, not L x samples.  The window is [tile_base, tile_end + 2): the interpolation of a tile's last sample needs the
// gained signal two samples on (:243-268), and a gain is final once every event up to L samples after it has been seen, so
// each tile also looks through the next tile's first L + 2 samples for events (they scatter again, in full, as that tile's
// own: events are ~1e-4 of the samples of a hot master).
// The release pass (:222-231) is the only state that outlives a tile: its memory is exp(-k/(10 ms)), so a segment re-runs
// `warm` >= 1 tiles before its first tile instead of waiting for its predecessor.
template <int NCH>
__global__ __launch_bounds__(kThreads) void k_sc(const float* __restrict__ in, float* __restrict__ out, int64_t n, const ClipPlan* __restrict__ cp,
                                                 const double* __restrict__ gains, int64_t seg_tiles, int64_t warm) {
  constexpr int kMaxL = kClipMaxL, kWin = kTile + 2;
  __shared__ unsigned gwin[kWin + 2];            // gain bits of [tile_base, tile_end + 2)
  __shared__ double rt[kMaxL + 1];               // ramp position m / L  (soft-clipper.js:213-214 with i - start == L)
  __shared__ double lds_scan[kSlotDoubles];
  __shared__ float gx[3][kThreads + 1];          // smoothed gain of every chunk's samples 0, 1, C - 1 (+ the tile's two-sample halo)
  __shared__ float xx[NCH][3][kThreads + 1];     // the same of the gained input
  __shared__ float carry[1 + NCH];               // the previous tile's last gain / inputs
  const ClipPlan& P = *cp;
  const int L = P.L;
  const int64_t trk = blockIdx.y;
  const double G = gains ? gains[trk] : 1.0;
  const double ceil_lin = P.ceil_lin;
  LdsSlots slots(lds_scan);
  double Scur = 1.0;
  const unsigned one = __float_as_uint(1.0f);
  for (int i = threadIdx.x; i < kWin; i += kThreads) gwin[i] = one;
  for (int i = threadIdx.x; i <= L; i += kThreads) rt[i] = (double)i / (double)(L > 1 ? L : 1);
  if (threadIdx.x <= NCH) carry[threadIdx.x] = threadIdx.x == 0 ? 1.0f : 0.0f;
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const int64_t T0 = (int64_t)blockIdx.x * seg_tiles;
  int64_t T1 = T0 + seg_tiles; if (T1 > ntiles) T1 = ntiles;
  const int64_t Tw = T0 - warm > 0 ? T0 - warm : 0;
  lds_barrier();
  auto gained = [&](float x) -> float { return f32r((double)x * G); };               // stage 8's Float32Array (normalizer.js:50-72)
  // soft-clipper.js:198-216 for the sample a = |input[i]| over the threshold: its ramp, clipped to this tile's window
  auto scatter = [&](int64_t i, double a, int64_t tile_base) {
    const double excess = a - P.thr_lin;
    const double normalized = excess / P.range_div;
    const double sat = tanh_fast(normalized * P.drive);
    const double target = P.thr_lin + sat * P.range;
    const double red1 = target / a - 1.0;
    if (i >= L) {
      const int64_t j0 = i - L - tile_base;              // window index of the ramp's first sample (m = 0)
      int m_lo = j0 < 0 ? (int)-j0 : 0, m_hi = (int64_t)kWin - 1 - j0 < L ? (int)(kWin - 1 - j0) : L;
      for (int m = m_lo; m <= m_hi; m++) atomicMin(&gwin[j0 + m], __float_as_uint(f32r(1.0 + red1 * rt[m])));
    } else {                                              // ramp clipped at the start of the track: start = 0 (tile 0 only)
      const double den = (double)(i > 1 ? i : 1);
      for (int64_t j = 0; j <= i; j++) atomicMin(&gwin[j], __float_as_uint(f32r(1.0 + red1 * ((double)j / den))));
    }
  };
  for (int64_t t = Tw; t < T1; t++) {
    const int64_t tile_base = t * kTile;
    const int64_t base = tile_base + (int64_t)threadIdx.x * C;
    // pass 1: which (channel, sample) pairs are over the threshold - one compare per sample, no arithmetic behind it
    float v[NCH][C];
    unsigned ev = 0;
    // the look-ahead samples of the first L + 2 threads and the last thread's halo inputs are requested together with the chunks:
    // loaded where they are used, each would stand between two barriers with the whole workgroup waiting on one wavefront
    float la[NCH], hx[NCH][2];
    const int64_t li = tile_base + kTile + threadIdx.x;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      la[c] = (threadIdx.x < L + 2 && li < n) ? in[(trk * NCH + c) * n + li] : 0.0f;
      if (threadIdx.x == kThreads - 1) {
#pragma unroll
        for (int q = 0; q < 2; q++) hx[c][q] = tile_base + kTile + q < n ? in[(trk * NCH + c) * n + tile_base + kTile + q] : 0.0f;
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      load_chunk(in + (trk * NCH + c) * n, n, base, v[c]);
#pragma unroll
      for (int k = 0; k < C; k++) {
        v[c][k] = gained(v[c][k]);
        if (fabs((double)v[c][k]) > P.thr_lin && base + k < n) ev |= 1u << (c * C + k);
      }
    }
    // pass 2 (rare): one rolled loop over the set bits
    while (ev) {
      const int bit = __ffs((int)ev) - 1;
      ev &= ev - 1;
      const int c = bit / C, k = bit % C;
      scatter(base + k, fabs((double)gained(in[(trk * NCH + c) * n + base + k])), tile_base);
    }
    // the events of the next tile's first L + 2 samples reach back into this window
    for (int q = threadIdx.x; q < L + 2; q += kThreads) {
      const int64_t i = tile_base + kTile + q;
      if (i >= n) break;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const double a = fabs((double)gained(q < kThreads ? la[c] : in[(trk * NCH + c) * n + i]));
        if (a > P.thr_lin) scatter(i, a, tile_base);
      }
    }
    lds_barrier();
    float gs[C], g[C];
#pragma unroll
    for (int k = 0; k < C; k++) { const float gv = __uint_as_float(gwin[threadIdx.x * C + k]); gs[k] = base + k < n ? gv : 1.0f; }
    // release smoothing (soft-clipper.js:222-231) in time order across tiles
    double cur = release_solve(gs, P.rc, P.rcC, Scur, slots.next());
#pragma unroll
    for (int k = 0; k < C; k++) { cur = release_step(cur, (double)gs[k], P.rc); g[k] = f32r(cur); }
    const bool emit = t >= T0;
    if (emit) {
      gx[0][threadIdx.x] = g[0]; gx[1][threadIdx.x] = g[1]; gx[2][threadIdx.x] = g[C - 1];
#pragma unroll
      for (int c = 0; c < NCH; c++) { xx[c][0][threadIdx.x] = v[c][0]; xx[c][1][threadIdx.x] = v[c][1]; xx[c][2][threadIdx.x] = v[c][C - 1]; }
      if (threadIdx.x == kThreads - 1) {                  // the halo: two more steps of the recursion (not carried), two more inputs
        const int64_t e = tile_base + kTile;
        double h = cur;
#pragma unroll
        for (int q = 0; q < 2; q++) {
          h = release_step(h, e + q < n ? (double)__uint_as_float(gwin[kTile + q]) : 1.0, P.rc);
          gx[q][kThreads] = f32r(h);
#pragma unroll
          for (int c = 0; c < NCH; c++) xx[c][q][kThreads] = gained(hx[c][q]);
        }
      }
    }
    lds_barrier();
    if (emit) {
      // gain and 2x Catmull-Rom oversampled saturation (:234-286) of this thread's chunk; s = gained input x envelope at base-1 .. base+C+1
      float gv[C + 3];
#pragma unroll
      for (int k = 0; k < C; k++) gv[k + 1] = g[k];
      gv[0] = threadIdx.x ? gx[2][threadIdx.x - 1] : carry[0];
      gv[C + 1] = gx[0][threadIdx.x + 1]; gv[C + 2] = gx[1][threadIdx.x + 1];
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        float xv[C + 3];
#pragma unroll
        for (int k = 0; k < C; k++) xv[k + 1] = v[c][k];
        xv[0] = threadIdx.x ? xx[c][2][threadIdx.x - 1] : carry[1 + c];
        xv[C + 1] = xx[c][0][threadIdx.x + 1]; xv[C + 2] = xx[c][1][threadIdx.x + 1];
        double s[C + 3];
#pragma unroll
        for (int k = 0; k < C + 3; k++) {
          const int64_t i = base - 1 + k;
          double xi = xv[k], gi = gv[k];
          if (i < 0 || i >= n) { xi = 0; gi = 1; }
          if (!(xi == xi)) xi = 0;                      // input[k] || 0 for NaN (neighbours only; see below)
          if (gi == 0 || !(gi == gi)) gi = 1;           // gainEnvelope[k] || 1
          s[k] = xi * gi;
        }
        float o[C];
#pragma unroll
        for (int k = 0; k < C; k++) {
          // centre sample uses input[i] * gainEnvelope[i] without the || fallbacks (soft-clipper.js:242,249)
          const double sA = (double)xv[k + 1] * (double)gv[k + 1];
          const double sB = catmull_half(s[k], sA, s[k + 2], s[k + 3]);
          double v2[2] = { sA, sB };
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const double a = fabs(v2[q]);
            if (a > ceil_lin) {
              const double ex = (a - ceil_lin) / ceil_lin;
              const double red = 1 - tanh_fast(ex * 2) * 0.1;
              const double m = fmin(a * red, ceil_lin);
              v2[q] = v2[q] > 0 ? m : -m;
            }
          }
          o[k] = f32r((v2[0] + v2[1]) * 0.5);
        }
        store_chunk(out + (trk * NCH + c) * n, n, base, o);
      }
    }
    lds_barrier();                                      // the window and the exchange arrays have been read
    if (threadIdx.x == kThreads - 1) {
      carry[0] = g[C - 1];
#pragma unroll
      for (int c = 0; c < NCH; c++) carry[1 + c] = v[c][C - 1];
    }
    for (int i = threadIdx.x; i < kWin; i += kThreads) gwin[i] = one;
    lds_barrier();
  }
}

// ============================================================================ limiter
// Transient map (dsp-worker.js:306-362).  One workgroup per track.  tm: [ntracks][n] float.
template <int NCH>
__global__ __launch_bounds__(kThreads) void k_lim_tm(const float* __restrict__ in, int64_t n, const LimiterPlan* __restrict__ lp,
                                                     float* __restrict__ tm_out, int64_t seg_tiles, int64_t warm) {
  __shared__ double pre[kTile + kTile / 16 + 8];       // padded inclusive prefix of the tile's mean squares
  extern __shared__ double tail[];                     // the last W prefixes before this tile, relative to this tile's origin (padi(W) + 8
                                                       // doubles of dynamic LDS; W > kTile: a ring, logical index m at (ring + m) mod W)
  __shared__ double lds_scan[kSlotDoubles];
  __shared__ double lds_sum[4];
  const LimiterPlan& P = *lp;
  const int W = P.W;
  const int64_t trk = blockIdx.y;
  float* tmo = tm_out + trk * n;
  LdsSlots slots(lds_scan);
  double Spk = 0;
  int ring = 0;                                        // stays 0 while W <= kTile
  for (int i = threadIdx.x; i < padi(W) + 8; i += kThreads) tail[i] = 0;
  __shared__ double pkT[C + 2];                        // follower_al_table of the peak follower
  follower_al_table(P.pa, P.pr, pkT);
  lds_barrier();
  // segment [T0, T1) of the track's tiles, entered `warm` tiles early: the RMS window is exact once W samples (10 ms; the
  // warm-up is 25 tiles at 48 kHz and scales with the rate) have been seen, and the peak follower forgets as exp(-k / (50 ms))
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const int64_t T0 = (int64_t)blockIdx.x * seg_tiles;
  int64_t T1 = T0 + seg_tiles; if (T1 > ntiles) T1 = ntiles;
  const int64_t Tw = T0 - warm > 0 ? T0 - warm : 0;
  // the next tile's samples are requested before this tile is computed; nothing else is loaded inside the loop (k_exciter)
  float nv[NCH][C];
  auto fetch = [&](int64_t t) {
#pragma unroll
    for (int c = 0; c < NCH; c++) load_chunk(in + (trk * NCH + c) * n, n, t * kTile + (int64_t)threadIdx.x * C, nv[c]);
  };
  fetch(Tw);
  for (int64_t t = Tw; t < T1; t++) {
    const int64_t tile_base = t * kTile;
    const int64_t base = tile_base + (int64_t)threadIdx.x * C;
    double q[C], pk[C];
#pragma unroll
    for (int k = 0; k < C; k++) { q[k] = 0; pk[k] = 0; }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
#pragma unroll
      for (int k = 0; k < C; k++) { const double d = nv[c][k]; q[k] += d * d; pk[k] = fmax(pk[k], fabs(d)); }
    }
    if (t + 1 < T1) fetch(t + 1);
    double run = 0;
#pragma unroll
    for (int k = 0; k < C; k++) { q[k] /= NCH; run += q[k]; q[k] = run; }       // local inclusive prefix
    double total;
    const double off = block_excl_sum(run, total, lds_sum);
#pragma unroll
    for (int k = 0; k < C; k++) pre[padi(threadIdx.x * C + k)] = off + q[k];
    // peak follower (dsp-worker.js:334-351)
    double e = follower_solve(pk, P.pa, P.pr, Spk, pkT, slots.next());      // contains barriers: pre[] is visible after it
    float tmv[C];
#pragma unroll
    for (int k = 0; k < C; k++) {
      e = follower_step(e, pk[k], P.pa, P.pr);
      const int j = threadIdx.x * C + k;
      const int64_t i = base + k;
      const double hi = off + q[k];
      int jr = ring + j; if (jr >= W) jr -= W;                                   // (j < W here whenever the ring has turned)
      const double lo = (j - W >= 0) ? pre[padi(j - W)] : tail[padi(jr)];
      const int64_t ws = i + 1 < W ? i + 1 : W;
      const float rms = f32r(sqrt((hi - lo) / (double)ws));
      const float pke = f32r(e);
      float tmk = 0.0f;
      if (rms > 0.0001f) {
        const double ratio = (double)pke / (double)rms;
        const double crest = 20.0 * 0.30102999566398120 * (double)log2f((float)ratio);     // 20 log10(x) = 20 log10(2) log2(x)
        tmk = f32r(fmin(1.0, fmax(0.0, (crest - 6) / (10 - 6))));
      }
      tmv[k] = tmk;
    }
    if (t >= T0) store_chunk(tmo, n, base, tmv);
    lds_barrier();
    // next tile's tail = the last W prefixes, re-based to the next tile's origin (minus this tile's total): from this tile
    // alone when W <= kTile; else the ring turns by one tile, its newest kTile entries come from this tile and the older
    // ones are re-based in place.  (All reads of tail for this tile lie before the barrier above.)
    if (W <= kTile) {
      for (int m = threadIdx.x; m < W; m += kThreads) tail[padi(m)] = pre[padi(kTile - W + m)] - total;
    } else {
      ring += kTile; if (ring >= W) ring -= W;
      for (int m = threadIdx.x; m < W; m += kThreads) {
        int ph = ring + m; if (ph >= W) ph -= W;
        if (m >= W - kTile) tail[padi(ph)] = pre[padi(kTile - W + m)] - total;
        else tail[padi(ph)] -= total;
      }
    }
    lds_barrier();
  }
}

// Lookahead gain (dsp-worker.js:367-425).  One workgroup per track.  g_out: [ntracks][n] float.
template <int NCH>
__global__ __launch_bounds__(kThreads) void k_lim_gain(const float* __restrict__ in, int64_t n, const LimiterPlan* __restrict__ lp,
                                                       const float* __restrict__ tm, float* __restrict__ g_out, int64_t seg_tiles, int64_t warm) {
  extern __shared__ float gwin_s[];             // gain envelope of [tile_base - L, tile_end): L + kTile floats of dynamic LDS
  __shared__ double lds_scan[kSlotDoubles];
  __shared__ float ctx[NCH][kCtxFloats];
  // (volatile: lanes read what other lanes of the wavefront wrote; in the LDS address space - a generic volatile pointer makes
  // every access a system-coherent FLAT operation)
  volatile float __attribute__((address_space(3)))* gwin = (volatile float __attribute__((address_space(3)))*)gwin_s;
  const LimiterPlan& P = *lp;
  const int L = P.L;
  const int64_t trk = blockIdx.y;
  const float* tmi = tm ? tm + trk * n : nullptr;
  float* go = g_out + trk * n;
  LdsSlots slots(lds_scan);
  double Scur = 1.0;
  for (int i = threadIdx.x; i < L + kTile; i += kThreads) gwin[i] = 1.0f;
  lds_barrier();
  const int lane = lane_id(), wave = wave_id();
  // segment [T0, T1) of the tiles, entered `warm` tiles early (release memory exp(-k / (100 ms)); ramps reach back L)
  const int64_t ntiles = (n + L + kTile - 1) / kTile;
  const int64_t T0 = (int64_t)blockIdx.x * seg_tiles;
  int64_t T1 = T0 + seg_tiles; if (T1 > ntiles) T1 = ntiles;
  const int64_t Tw = T0 - warm > 0 ? T0 - warm : 0;
  for (int64_t t = Tw; t < T1; t++) {
    const int64_t tile_base = t * kTile;
    const int64_t base = tile_base + (int64_t)threadIdx.x * C;
    const int par = (int)(t & 1);
    // per-sample true peak over the channels and the required gain (dsp-worker.js:373-403)
    double tp[C];
#pragma unroll
    for (int k = 0; k < C; k++) tp[k] = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float v[C], xp[3];
      load_chunk(in + (trk * NCH + c) * n, n, base, v);
      const float last[3] = { v[C - 3], v[C - 2], v[C - 1] };
      prev_ctx<3>(last, xp, ctx[c], par, t == Tw);
      if (t == Tw && Tw > 0 && threadIdx.x == 0) {         // a later segment's first tile: the three samples before it come from memory
        const float* xc = in + (trk * NCH + c) * n;
        xp[0] = xc[base - 3]; xp[1] = xc[base - 2]; xp[2] = xc[base - 1];
      }
      double y0 = xp[0], y1 = xp[1], y2 = xp[2];
#pragma unroll
      for (int k = 0; k < C; k++) {
        const double y3 = v[k];
        if (base + k >= 3 && base + k < n) tp[k] = fmax(tp[k], true_peak4(y0, y1, y2, y3));
        y0 = y1; y1 = y2; y2 = y3;
      }
    }
    float tmv[C];
    if (tmi) load_chunk(tmi, n, base, tmv);
    double req[C];
    unsigned evbits = 0;
#pragma unroll
    for (int k = 0; k < C; k++) {
      double ec = P.ceil_lin;
      if (tmi && tmv[k] > 0.5f) ec = P.ceil_lin * (double)exp2f((float)((double)tmv[k] * 0.5 / 20.0 * 3.3219280948873623));   // 10^(tm 0.5/20)
      req[k] = 1.0;
      if (tp[k] > ec && base + k < n) { req[k] = ec / tp[k]; evbits |= 1u << k; }
    }
    // events in time order: waves in turn, lanes in turn, samples in turn; each ramp is applied by the whole wave
    for (int w = 0; w < kWaves; w++) {
      if (w == wave) {
        unsigned long long m = __ballot(evbits != 0);
        while (m) {
          const int src = __ffsll((long long)m) - 1;
          m &= m - 1;
          const unsigned bits = __shfl(evbits, src);
#pragma unroll
          for (int k = 0; k < C; k++) {
            const double rq = __shfl(req[k], src);
            if (bits & (1u << k)) {           // wave-uniform
              const int64_t i = tile_base + (int64_t)(w * 64 + src) * C + k;
              const int64_t tg = i - L > 0 ? i - L : 0;
              const int wl_t = (int)(tg - (tile_base - L));
              const int cnt = (int)(i - tg) + 1;
              const double Gt = (double)gwin[wl_t];
              if (rq < Gt) {
                for (int j = lane; j < cnt; j += 64) {
                  const double prog = (double)j / (double)L;                 // (j - targetIndex) / lookaheadSamples
                  const double sm = Gt + (rq - Gt) * prog;
                  const float old = gwin[wl_t + j];
                  gwin[wl_t + j] = f32r(fmin((double)old, sm));
                }
              }
            }
          }
        }
      }
      lds_barrier();
    }
    // finalised samples [tile_base - L, tile_end - L): release smoothing (dsp-worker.js:416-425)
    const int64_t kb = tile_base - L + (int64_t)threadIdx.x * C;
    float gs[C];
#pragma unroll
    for (int k = 0; k < C; k++) { const float gv = gwin[threadIdx.x * C + k]; gs[k] = (kb + k >= 0 && kb + k < n) ? gv : 1.0f; }
    double cur = release_solve(gs, P.rc, P.rcC, Scur, slots.next());
    float g[C];
#pragma unroll
    for (int k = 0; k < C; k++) { cur = release_step(cur, (double)gs[k], P.rc); g[k] = f32r(cur); }
    if (t >= T0 && kb + C > 0) {
      if (kb >= 0) store_chunk(go, n, kb, g);
      else { for (int k = 0; k < C; k++) if (kb + k >= 0 && kb + k < n) go[kb + k] = g[k]; }
    }
    lds_barrier();
    // slide the window by one tile: [kTile, kTile + L) -> [0, L) (disjoint, L <= kTile), then the rest back to 1.0
    for (int i = threadIdx.x; i < L; i += kThreads) gwin[i] = gwin[kTile + i];
    lds_barrier();
    for (int i = threadIdx.x; i < kTile; i += kThreads) gwin[L + i] = 1.0f;
    lds_barrier();
  }
}

// dsp-worker.js:197-229 applySoftKneeCurve
__device__ __forceinline__ double soft_knee_curve(double s, double ceiling, double knee_start) {
  const double a = fabs(s);
  if (a <= ceiling * 0.9) return s;
  if (a <= knee_start) return s;
  double o;
  if (a <= ceiling) {
    const double t = div_fast(a - knee_start, ceiling - knee_start);      // quotients to <= 1 ulp (div_fast): the result is an f32 store
    const double blend = t * t * (3 - 2 * t);
    o = a + (ceiling - a) * blend * 0.5;
  } else {
    const double nrm = div_fast(a - ceiling, ceiling);
    const double comp = 1 - tanh_fast(nrm * 2) * 0.1;
    o = fmin(ceiling, a * comp);
  }
  return s > 0 ? o : -o;
}

// Apply + soft knee (dsp-worker.js:430-449, :234-281) + final metering.  One workgroup per (track, channel).
__global__ __launch_bounds__(kThreads) void k_lim_apply(const float* __restrict__ in, float* __restrict__ out, int nch, int64_t n,
                                                        const LimiterPlan* __restrict__ lp, const float* __restrict__ genv,
                                                        const MeterPlan* __restrict__ mp, double* __restrict__ hopsums, int64_t hop_stride,
                                                        double* __restrict__ peaks, int64_t seg_tiles, int64_t warm) {
  __shared__ double lds_meter[kMeterLds];
  __shared__ float ctx[2][kCtxFloats];
  const LimiterPlan& P = *lp;
  const int64_t chan = blockIdx.x;
  const int64_t trk = chan / nch;
  const float* x = in + chan * n;
  const float* g = genv + trk * n;
  float* y = out + chan * n;
  // time segment (blockIdx.y); the only carried state is the fused meter's filters and six samples of context
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const Segment sg = make_segment(blockIdx.y, seg_tiles, warm, ntiles, mp ? mp->hop : 0, mp ? mp->nhops : 0);
  __shared__ double mP[2][64 * 4], mS[2][2];        // K-weighting lane powers and tile carries (meter_init_l)
  MeterStateL ms;
  if (mp) meter_init_l(ms, *mp, sg.Tw * kTile, sg, mP[0], mP[1], mS[0], mS[1]);
  lds_barrier();
  for (int64_t t = sg.Tw; t < sg.T1; t++) {
    const int64_t tile_base = t * kTile;
    const int64_t base = tile_base + (int64_t)threadIdx.x * C;
    const int par = (int)(t & 1);
    float v[C], gv[C], s[C];
    load_chunk(x, n, base, v);
    load_chunk(g, n, base, gv);
#pragma unroll
    for (int k = 0; k < C; k++) s[k] = f32r((double)v[k] * (double)gv[k]);
    const float last[3] = { s[C - 3], s[C - 2], s[C - 1] };
    float sp[3];
    prev_ctx<3>(last, sp, ctx[0], par, t == sg.Tw);
    if (t == sg.Tw && sg.Tw > 0 && threadIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < 3; j++) sp[j] = f32r((double)x[base - 3 + j] * (double)g[base - 3 + j]);
    }
    float o[C];
    double y0 = sp[0], y1 = sp[1], y2 = sp[2];
#pragma unroll
    for (int k = 0; k < C; k++) {
      const double y3 = s[k];
      double r = y3;
      if (base + k >= 3) {
        const double m = P.lib_twin ? interp_peak3(y0, y1, y2, y3) : true_peak4(y0, y1, y2, y3);
        if (m > P.ceil_lin) r = y3 * div_fast(P.ceil_lin, m);
      }
      o[k] = f32r(soft_knee_curve(r, P.ceil_lin, P.knee_start));
      y0 = y1; y1 = y2; y2 = y3;
    }
    if (t >= sg.T0) store_chunk(y, n, base, o);
    if (mp) {
      const float lo[3] = { o[C - 3], o[C - 2], o[C - 1] };
      float op[3];
      prev_ctx<3>(lo, op, ctx[1], par, t == sg.Tw);
      meter_chunk_l(o, op, base, tile_base, n, *mp, ms, lds_meter, hopsums + chan * hop_stride);
    }
  }
  if (mp) meter_finish(ms, lds_meter, peaks + chan);
}

// ---------------------------------------------------------------------------- launchers
// A kernel whose static + dynamic LDS exceeds the 64 KB default must be opted in, up to what the device has per workgroup
// (gfx950: 160 KB; asked of the device once, not assumed).  A window that does not fit is the caller's sample rate being too
// high for this device: WAM_ERR_TOO_LARGE with the rate in the message, not a raw HIP launch error.
static int lds_opt_in(const void* kernel, size_t dynamic_bytes, size_t static_bytes, double sr, const char* what) {
  if (dynamic_bytes + static_bytes <= 64 * 1024) return WAM_OK;
  static thread_local int cached_dev = -1;
  static thread_local size_t cached_max = 0;
  int dev = 0; WAM_HIP(hipGetDevice(&dev));
  if (dev != cached_dev) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 64 * 1024; }
    cached_dev = dev; cached_max = (size_t)v;
  }
  if (dynamic_bytes + static_bytes > cached_max) {
    set_error("%s: the window of %zu + %zu bytes at sample rate %.1f does not fit this device's %zu bytes of LDS per workgroup", what, dynamic_bytes, static_bytes, sr, cached_max);
    return WAM_ERR_TOO_LARGE;
  }
  WAM_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynamic_bytes));
  return WAM_OK;
}

int enqueue_soft_clip(const float* in, float* out, int ntracks, int nch, int64_t n, const ClipPlan* plan_dev, const double* gains_dev, double sr,
                      hipStream_t s) {
  if (n <= 0) return WAM_OK;
  if (in == out) { set_error("enqueue_soft_clip: in-place operation is not supported (a tile reads its neighbours' input)"); return WAM_ERR_INVALID; }
  const int64_t warm = scale_warm(tuning().sc_warm_tiles, sr);
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const int64_t seg = pick_segment_tiles(ntracks, ntiles, warm, tuning().env_seg_tiles);
  const unsigned nseg = (unsigned)((ntiles + seg - 1) / seg);
  if (nch == 2) WAM_LAUNCH("k_sc", k_sc<2>, dim3(nseg, ntracks), dim3(kThreads), s, in, out, n, plan_dev, gains_dev, seg, warm);
  else WAM_LAUNCH("k_sc", k_sc<1>, dim3(nseg, ntracks), dim3(kThreads), s, in, out, n, plan_dev, gains_dev, seg, warm);
  return WAM_OK;
}

int enqueue_limiter(const float* in, float* out, float* tm, float* genv, int ntracks, int nch, int64_t n, const LimiterPlan* plan_dev, int L,
                    int preserve, const MeterPlan* meter_dev, double* hopsums, int64_t hop_stride, double* peaks, double sr, hipStream_t s,
                    const Progress* prog) {
  // the worker limiter's own messages (dsp-worker.js:302,364,427,448), at the points where its stages start; the per-channel
  // messages follow the kernel that processes all channels at once
  auto say = [&](double p, const char* status) -> int { return prog ? (*prog)(p, status) : WAM_OK; };
  int prc = say(0.1, "Analyzing transients..."); if (prc) return prc;
  if (n <= 0) {
    if ((prc = say(0.3, "Calculating gain envelope..."))) return prc;
    if ((prc = say(0.6, "Applying limiting..."))) return prc;
    for (int c = 0; c < nch; c++) {
      char msg[64]; snprintf(msg, sizeof(msg), "Processing channel %d/%d...", c + 1, nch);
      if ((prc = say(0.6 + 0.3 * ((double)(c + 1) / nch), msg))) return prc;
    }
    return WAM_OK;
  }
  const int64_t warm = scale_warm(tuning().lim_warm_tiles, sr);
  const int64_t seg = pick_segment_tiles(ntracks, (n + L + kTile - 1) / kTile, warm, tuning().env_seg_tiles);
  if (preserve) {
    const unsigned nseg = (unsigned)(((n + kTile - 1) / kTile + seg - 1) / seg);
    const int W = (int)std::floor(sr * 10 / 1000);                    // = LimiterPlan::W (dsp-worker.js:308-309)
    const size_t tail_bytes = sizeof(double) * (size_t)(W + (W >> 4) + 8);
    int rc = lds_opt_in((const void*)(nch == 2 ? k_lim_tm<2> : k_lim_tm<1>), tail_bytes, sizeof(double) * (kTile + kTile / 16 + 8 + kSlotDoubles + 4) + 64, sr,
                        "limiter transient map (10 ms RMS window)");
    if (rc) return rc;
    if (nch == 2) WAM_LAUNCH_SHMEM("k_lim_tm", k_lim_tm<2>, dim3(nseg, ntracks), dim3(kThreads), tail_bytes, s, in, n, plan_dev, tm, seg, warm);
    else WAM_LAUNCH_SHMEM("k_lim_tm", k_lim_tm<1>, dim3(nseg, ntracks), dim3(kThreads), tail_bytes, s, in, n, plan_dev, tm, seg, warm);
  }
  if ((prc = say(0.3, "Calculating gain envelope..."))) return prc;
  const float* tmi = preserve ? tm : nullptr;
  const unsigned nsegg = (unsigned)(((n + L + kTile - 1) / kTile + seg - 1) / seg);
  const size_t gwin_bytes = sizeof(float) * (size_t)(L + kTile);
  {
    int rc = lds_opt_in((const void*)(nch == 2 ? k_lim_gain<2> : k_lim_gain<1>), gwin_bytes, sizeof(double) * kSlotDoubles + sizeof(float) * 2 * kCtxFloats + 64, sr,
                        "limiter gain envelope (3 ms look-ahead window)");
    if (rc) return rc;
  }
  if (nch == 2) WAM_LAUNCH_SHMEM("k_lim_gain", k_lim_gain<2>, dim3(nsegg, ntracks), dim3(kThreads), gwin_bytes, s, in, n, plan_dev, tmi, genv, seg, warm);
  else WAM_LAUNCH_SHMEM("k_lim_gain", k_lim_gain<1>, dim3(nsegg, ntracks), dim3(kThreads), gwin_bytes, s, in, n, plan_dev, tmi, genv, seg, warm);
  if ((prc = say(0.6, "Applying limiting..."))) return prc;
  {
    const int64_t ntiles = (n + kTile - 1) / kTile, mwarm = meter_warm((int64_t)std::floor(sr * 0.1), sr);
    const int64_t mseg = pick_segment_tiles((int64_t)ntracks * nch, ntiles, mwarm, tuning().meter_seg_tiles);
    WAM_LAUNCH("k_lim_apply", k_lim_apply, dim3(ntracks * nch, (unsigned)((ntiles + mseg - 1) / mseg)), dim3(kThreads), s, in, out, nch, n, plan_dev,
               (const float*)genv, meter_dev, hopsums, hop_stride, peaks, mseg, mwarm);
  }
  for (int c = 0; c < nch; c++) {
    char msg[64]; snprintf(msg, sizeof(msg), "Processing channel %d/%d...", c + 1, nch);
    if ((prc = say(0.6 + 0.3 * ((double)(c + 1) / nch), msg))) return prc;
  }
  return WAM_OK;
}

}  // namespace wam
