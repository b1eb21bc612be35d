// meter.hpp - fused loudness / true-peak metering of a chunk held in registers.
// Used by the stand-alone meter kernel and by every chain kernel that has to
// measure what it just produced (stage 8 needs the loudness of its input, the
// chain reports the loudness of its output).
//
// Reference: web/lib/dsp/lufs.js:17-82 (K-weighting shelf -> high-pass with an f32
// store between and after, 400 ms blocks every 100 ms hop) and
// web/lib/dsp/true-peak.js:46-71.  Per hop (100 ms) sums of squares are
// accumulated in double; the gating kernel adds four hops per block.
#pragma once
#include "wam_device.hpp"

namespace wam {

struct MeterPlan {
  Biquad shelf, hp;
  int64_t hop;        // samples per 100 ms hop
  int64_t nhops;      // complete hops in the track
  int64_t nblocks;    // 400 ms blocks
  int64_t block;      // samples per block: 4 hops + `rem` samples (rem = 0 at the usual rates; 11,025 Hz: 4410 = 4 * 1102 + 2)
  int64_t heads_off;  // rem > 0: offset, inside a channel's hop-sum row, of the "hop heads": z^2 of the first `rem` samples of every hop, 4 slots per hop
  int rem;            // block - 4 * hop, 0..3 (lufs.js:53-54: block = floor(0.4 sr), hop = floor(0.4 sr * 0.25))
  int nch;
  double fallback;
  int too_short;      // duration < 0.4 s -> lufs = fallback (lufs.js:23-26)
};

// A time segment of a (track, channel) walk: tiles [T0, T1), entered at tile Tw <= T0 with cold state (the warm-up
// re-run).  A 100 ms hop belongs to the segment that holds its LAST sample, so the owner also sums the part of the
// hop that lies in its warm-up tiles: every hop sum is accumulated by one workgroup, in tile order - deterministic,
// no atomics.  (Needs warm >= 2 tiles + the settle time of whatever feeds the meter: a hop is 1.17 tiles.)
struct Segment {
  int64_t T0, T1, Tw, own_from, own_to, hop_lo, hop_hi;
};
__device__ __forceinline__ Segment make_segment(int64_t seg_idx, int64_t seg_tiles, int64_t warm, int64_t ntiles, int64_t hop, int64_t nhops) {
  Segment g;
  g.T0 = seg_idx * seg_tiles;
  g.T1 = g.T0 + seg_tiles; if (g.T1 > ntiles) g.T1 = ntiles;
  g.Tw = g.T0 - warm > 0 ? g.T0 - warm : 0;
  g.own_from = g.T0 * kTile;
  g.own_to = g.T1 * kTile;
  g.hop_lo = hop > 0 ? g.own_from / hop : 0;
  g.hop_hi = (g.T1 >= ntiles || hop <= 0) ? nhops : (g.T1 * kTile) / hop;
  if (g.hop_hi > nhops) g.hop_hi = nhops;
  return g;
}

// x[0..C) = this thread's samples (already rounded to f32 by the producing stage),
// xp[0..3) = the three samples before the chunk (xp[2] = x[-1]); base = global
// index of x[0]; n = track length.  lds: 16 doubles (two scan sites) + 4*kMaxK doubles.
constexpr int kMeterLds = 16 + 4 * 4;

// ---- true peak (true-peak.js:53-65): history starts at zeros, evaluated for i >= 3
template <class St>
__device__ __forceinline__ void meter_peak_part(const float x[C], const float xp[3], int64_t base, int64_t n, St& m) {
  double y0 = xp[0], y1 = xp[1], y2 = xp[2];
  double pk = m.peak;
#pragma unroll
  for (int k = 0; k < C; k++) {
    const double y3 = x[k];
    const int64_t i = base + k;
    if (i >= 3 && i < n && i >= m.own_from) pk = fmax(pk, true_peak4(y0, y1, y2, y3));
    y0 = y1; y1 = y2; y2 = y3;
  }
  m.peak = pk;
}

// ---- per-hop sums of squares of the K-weighted chunk v[] (double values that are about to be rounded to f32)
template <class St>
__device__ __forceinline__ void meter_hop_part(const double v[C], int64_t base, int64_t tile_base, int64_t n, const MeterPlan& pl, St& m,
                                               double* lds, double* hopsums) {
  const int64_t lim = pl.nhops * pl.hop;       // samples past the last complete hop never reach a block through a hop sum
  double sA = 0, sB = 0;
  const int64_t to_boundary = pl.hop - m.hop_off;   // samples of this chunk that belong to hop_idx
#pragma unroll
  for (int k = 0; k < C; k++) {
    const double z = (double)f32r(v[k]);
    const double zz = (base + k < lim) ? z * z : 0.0;
    if (k < to_boundary) sA += zz; else sB += zz;
  }
  if (pl.rem > 0) {
    // A block that is `rem` samples longer than four hops ends with the first `rem` samples of a fifth hop (lufs.js:57-65 sums
    // sample ranges): those squares are kept one by one, 4 slots per hop, each written by the one workgroup that owns the sample
    // (its own tiles, not its warm-up), so the gate kernel adds them in a fixed order.
    // (a chunk holds at most one hop start, hop >= C: the head samples are k = k0 .. k0 + rem - 1 with k0 = 0 when the chunk starts
    // inside a head, else the distance to the next hop)
    double* heads = hopsums + pl.heads_off;
    const int off = (int)m.hop_off, hop = (int)pl.hop;
    const int64_t lo = m.own_from > 0 ? m.own_from : 0, hi = m.own_to < n ? m.own_to : n;
#pragma unroll
    for (int k = 0; k < C; k++) {
      int pos = off + k; int64_t h = m.hop_idx;
      if (pos >= hop) { pos -= hop; h++; }
      const int64_t i = base + k;
      if (pos < pl.rem && i >= lo && i < hi) { const double z = (double)f32r(v[k]); heads[4 * h + pos] = z * z; }
    }
  }
  int64_t h_first = tile_base / pl.hop;
  int64_t last = tile_base + kTile - 1; if (last > lim - 1) last = lim - 1;
  int64_t h_last = last >= tile_base ? last / pl.hop : h_first - 1;
  if (h_first < m.hop_lo) h_first = m.hop_lo;
  if (h_last > m.hop_hi - 1) h_last = m.hop_hi - 1;
  double* part = lds + 16;
  for (int64_t h0 = h_first; h0 <= h_last; h0 += 4) {
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t h = h0 + j;
      if (h > h_last) break;
      double c = 0;
      if (h == m.hop_idx) c = sA; else if (h == m.hop_idx + 1) c = sB;
      c = wave_sum_lane63(c);
      if (lane_id() == 63) part[wave_id() * 4 + j] = c;
    }
    lds_barrier();
    if (threadIdx.x < 4) {
      // thread (h & 3) owns hop h in every tile, so the read-modify-write below is same-thread ordered
      const int j = (int)((threadIdx.x - h0) & 3);
      if (h0 + j <= h_last) {
        const double s = ((part[0 * 4 + j] + part[1 * 4 + j]) + part[2 * 4 + j]) + part[3 * 4 + j];
        hopsums[h0 + j] += s;
      }
    }
  }
  // advance chunk position to the next tile
  m.hop_off += kTile;
  while (m.hop_off >= pl.hop) { m.hop_off -= pl.hop; m.hop_idx++; }
}

// The two K-weighting filters' lane powers ([64][4] doubles each) and tile carries ([2] each) live in LDS: see biquad_chunk_lds.
struct MeterStateL {            // carried across tiles, per channel
  const double* P1; const double* P2; double* S1; double* S2;   // lane powers / tile carries of the two filters (LDS)
  double peak;                 // running true peak of this thread
  int64_t hop_idx, hop_off;    // hop index / offset of this thread's chunk start
  int64_t own_from, own_to;    // time segments: the samples [own_from, own_to) whose peaks (and hop heads) this workgroup reports ...
  int64_t hop_lo, hop_hi;      // ... and the hops [hop_lo, hop_hi) it accumulates (each hop has exactly one owner)
};
__device__ __forceinline__ void meter_init_l(MeterStateL& m, const MeterPlan& pl, int64_t first_sample, const Segment& seg,
                                             double* P1, double* P2, double* S1, double* S2) {
  if (wave_id() == 0) { biquad_lane_power_table(pl.shelf, lane_id(), P1); biquad_lane_power_table(pl.hp, lane_id(), P2); }
  if (threadIdx.x < 2) { S1[threadIdx.x] = 0; S2[threadIdx.x] = 0; }
  m.P1 = P1; m.P2 = P2; m.S1 = S1; m.S2 = S2;
  m.peak = 0;
  m.own_from = seg.own_from; m.own_to = seg.own_to; m.hop_lo = seg.hop_lo; m.hop_hi = seg.hop_hi;
  const int64_t b = first_sample + (int64_t)threadIdx.x * C;
  m.hop_idx = b / pl.hop;
  m.hop_off = b - m.hop_idx * pl.hop;
}
__device__ __forceinline__ void meter_chunk_l(const float x[C], const float xp[3], int64_t base, int64_t tile_base, int64_t n,
                                              const MeterPlan& pl, MeterStateL& m, double* lds, double* hopsums) {
  meter_peak_part(x, xp, base, n, m);
  double v[C];
#pragma unroll
  for (int k = 0; k < C; k++) v[k] = (base + k < n) ? (double)x[k] : 0.0;
  double s0, s1;
  biquad_chunk_lds(v, (base - 1 >= 0) ? (double)xp[2] : 0.0, (base - 2 >= 0) ? (double)xp[1] : 0.0, pl.shelf, m.P1, m.S1, lds, s0, s1);
#pragma unroll
  for (int k = 0; k < C; k++) v[k] = (double)f32r(v[k]);
  double t0, t1;
  biquad_chunk_lds(v, (double)f32r(s0), (double)f32r(s1), pl.hp, m.P2, m.S2, lds + 8, t0, t1);
  meter_hop_part(v, base, tile_base, n, pl, m, lds, hopsums);
}

// Two channels of one track in lock-step: the K-weighting stages as fused pairs (one barrier per stage for both channels).
__device__ __forceinline__ void meter_chunk_l2(const float xa[C], const float xpa[3], const float xb[C], const float xpb[3],
                                               int64_t base, int64_t tile_base, int64_t n, const MeterPlan& pl,
                                               MeterStateL& ma, MeterStateL& mb, double* ldsa, double* ldsb, double* hsa, double* hsb) {
  meter_peak_part(xa, xpa, base, n, ma);
  meter_peak_part(xb, xpb, base, n, mb);
  double va[C], vb[C];
#pragma unroll
  for (int k = 0; k < C; k++) { const bool in = base + k < n; va[k] = in ? (double)xa[k] : 0.0; vb[k] = in ? (double)xb[k] : 0.0; }
  double a0, a1, b0, b1, c0, c1, d0, d1;
  const bool h1 = base - 1 >= 0, h2 = base - 2 >= 0;
  biquad_pair_lds(va, h1 ? (double)xpa[2] : 0.0, h2 ? (double)xpa[1] : 0.0, pl.shelf, ma.P1, ma.S1,
                  vb, h1 ? (double)xpb[2] : 0.0, h2 ? (double)xpb[1] : 0.0, pl.shelf, mb.P1, mb.S1, ldsa, a0, a1, b0, b1);
#pragma unroll
  for (int k = 0; k < C; k++) { va[k] = (double)f32r(va[k]); vb[k] = (double)f32r(vb[k]); }
  biquad_pair_lds(va, (double)f32r(a0), (double)f32r(a1), pl.hp, ma.P2, ma.S2,
                  vb, (double)f32r(b0), (double)f32r(b1), pl.hp, mb.P2, mb.S2, ldsb, c0, c1, d0, d1);
  meter_hop_part(va, base, tile_base, n, pl, ma, ldsa, hsa);
  meter_hop_part(vb, base, tile_base, n, pl, mb, ldsb, hsb);
}

// Workgroup-wide final peak, max-ed into peaks[0] (zeroed before the launch; segments of one channel share it).
// Non-negative doubles order like their bit patterns.
template <class St>
__device__ __forceinline__ void meter_finish(St& m, double* lds, double* peak_out) {
  const double w = wave_max(m.peak);
  lds_barrier();
  if (lane_id() == 0) lds[wave_id()] = w;
  lds_barrier();
  if (threadIdx.x == 0) {
    const double pk = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
    atomicMax(reinterpret_cast<unsigned long long*>(peak_out), (unsigned long long)__double_as_longlong(pk));
  }
}

}  // namespace wam
