// chain_front.hip - stages 5..7.5 of the export chain fused into one pass, plus the
// loudness / true-peak measurement of the result that stage 8 needs:
//   final filters (30 Hz biquad HPF, 18 kHz one-pole LPF)   final-filters.js:96-131
//   5-band EQ + Cut Mud (spec-defined biquads)                dsp-worker.js:1222-1267
//   glue compressor, linked side-chain                        dsp-worker.js:1272-1343
//   M/S width + mono bass                                     dsp-worker.js:1122-1217
// One workgroup per track (both channels: the side-chain and the M/S matrix link them).
// Algorithmic traffic: 4 B read + 4 B written per sample-channel.
#include "chain.hpp"
#include <cmath>

namespace wam {

// Filter state lives in LDS (lane powers [64][4] per distinct filter, tile carries [2] per instance): with up to nine
// biquads per channel plus the meter, keeping it in registers cost 400+ VGPRs (one wave per SIMD, spills to AGPRs).
enum { kFHpf = 0, kFEq0 = 1, kFSide = 7, kFShelf = 8, kFKhp = 9, kFNumFilters = 10 };
template <int NCH>
__global__ __launch_bounds__(kThreads, 2) void k_front(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                       const FrontPlan* __restrict__ fp, const MeterPlan* __restrict__ mp,
                                                       double* __restrict__ hopsums, int64_t hop_stride, double* __restrict__ peaks,
                                                       double pre_gain, int64_t seg_tiles, int64_t warm) {
  __shared__ double lds_scan[kSlotDoubles];
  __shared__ double lds_meter[NCH][kMeterLds];
  __shared__ double Pl[kFNumFilters][64][4];
  __shared__ double Sl[NCH * 9 + 1][2];            // per channel: hpf, 6 eq, shelf, khp; + side
  __shared__ float ctx[NCH * 9 + 1][kCtxFloats];   // per channel: hpf, 6 eq, lpf->eq, meter; + side
  __shared__ float bc[NCH];
  const int64_t trk = blockIdx.x;
  const float* x[NCH]; float* y[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) { x[c] = in + (trk * NCH + c) * n; y[c] = out + (trk * NCH + c) * n; }
  const int lane = lane_id();
  LdsSlots slots(lds_scan);

  // time segment (blockIdx.y): tiles [T0, T1) entered `warm` tiles early with cold state; the warm-up re-run rebuilds
  // the filters (30 Hz high-pass: e^-11 per tile) and, when the glue compressor is on, its 250 ms release follower
  const int do_meter = fp->do_meter;
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const Segment sg = make_segment(blockIdx.y, seg_tiles, warm, ntiles, do_meter ? mp->hop : 0, do_meter ? mp->nhops : 0);
  if (wave_id() == 0) {
    biquad_lane_power_table(fp->hpf30, lane, &Pl[kFHpf][0][0]);
#pragma unroll
    for (int b = 0; b < 6; b++) if (b < fp->n_eq) biquad_lane_power_table(fp->eq[b], lane, &Pl[kFEq0 + b][0][0]);
    biquad_lane_power_table(fp->side_hp, lane, &Pl[kFSide][0][0]);
  }
  for (int i = threadIdx.x; i < (NCH * 9 + 1) * 2; i += kThreads) Sl[i >> 1][i & 1] = 0;
  __shared__ double PlpfT[64];                    // lane powers of the one-pole low-pass (read where used: two VGPRs less across the tile)
  if (wave_id() == 1) PlpfT[lane] = onepole_lane_power(fp->lpf18, lane);
  // tile carries of the one-pole low-pass (per channel) and of the glue follower: uniform over the workgroup, so they live in LDS
  // (thread 0 stores after the scan, everybody reads at the next tile) and not in six VGPRs per thread
  __shared__ double Scar[NCH + 1];
  if (threadIdx.x <= NCH) Scar[threadIdx.x] = 0;
  __shared__ double glT[C + 2];                   // follower_al_table of the glue follower
  follower_al_table(fp->glue_A, fp->glue_R, glT);
  MeterStateL ms[NCH];
  if (do_meter) for (int c = 0; c < NCH; c++) meter_init_l(ms[c], *mp, sg.Tw * kTile, sg, &Pl[kFShelf][0][0], &Pl[kFKhp][0][0], Sl[c * 9 + 7], Sl[c * 9 + 8]);
  lds_barrier();

  for (int64_t t = sg.Tw; t < sg.T1; t++) {
    const FrontPlan& P = *reload_each_tile(fp);
    const bool first = t == sg.Tw;
    const int64_t tile_base = t * kTile;
    const int64_t base = tile_base + (int64_t)threadIdx.x * C;
    const int par = (int)(t & 1);
    float w[NCH][C];
    if constexpr (NCH == 2) {
      // Both channels in lock-step: every cascade stage is a fused pair (one barrier for the two channels, their
      // recursions interleaved) - the same filters, separate tile carries and context sites.
      float va[C], vb[C];
      load_chunk(x[0], n, base, va);
      load_chunk(x[1], n, base, vb);
      if (pre_gain != 1.0) {
#pragma unroll
        for (int k = 0; k < C; k++) { va[k] = f32r((double)va[k] * pre_gain); vb[k] = f32r((double)vb[k] * pre_gain); }
      }
      double da[C], db[C];
#pragma unroll
      for (int k = 0; k < C; k++) { da[k] = va[k]; db[k] = vb[k]; }
      double na1 = 0, na2 = 0, nb1 = 0, nb2 = 0;   // the two inputs before the chunk, for the next biquad's FIR part
      bool have_nb = false;
      if (P.do_hpf) {
        const float la[2] = { va[C - 2], va[C - 1] }, lb[2] = { vb[C - 2], vb[C - 1] };
        float pa[2], pb[2];
        prev_ctx2<2>(la, pa, ctx[0], lb, pb, ctx[9], par, first);
        double a0, a1, b0, b1;
        biquad_pair_lds(da, pa[1], pa[0], P.hpf30, &Pl[kFHpf][0][0], Sl[0], db, pb[1], pb[0], P.hpf30, &Pl[kFHpf][0][0], Sl[9], slots.next(), a0, a1, b0, b1);
#pragma unroll
        for (int k = 0; k < C; k++) { da[k] = (double)f32r(da[k]); db[k] = (double)f32r(db[k]); }
        na1 = (double)f32r(a0); na2 = (double)f32r(a1); nb1 = (double)f32r(b0); nb2 = (double)f32r(b1); have_nb = true;
      }
      if (P.do_lpf) {
        // applyOnePoleLP (final-filters.js:68-78): y starts at data[0]
        if (t == 0) {
          if (threadIdx.x == 0) { bc[0] = f32r(da[0]); bc[1] = f32r(db[0]); }
          lds_barrier();
          const float f0 = bc[0], f1 = bc[1];
          if (threadIdx.x == 0) { Scar[0] = (f0 == f0) ? (double)f0 : 0.0; Scar[1] = (f1 == f1) ? (double)f1 : 0.0; }
          lds_barrier();
        }
        double ua[C], ub[C];
#pragma unroll
        for (int k = 0; k < C; k++) { ua[k] = P.lpf_alpha * da[k]; ub[k] = P.lpf_alpha * db[k]; }
        double Sa = Scar[0], Sb = Scar[1];                          // (read before the scan's barrier, stored after it)
        onepole_pair(ua, ub, P.lpf18, PlpfT[lane], Sa, Sb, slots.next());
        if (threadIdx.x == 0) { Scar[0] = Sa; Scar[1] = Sb; }
#pragma unroll
        for (int k = 0; k < C; k++) { da[k] = (double)f32r(ua[k]); db[k] = (double)f32r(ub[k]); }
        have_nb = false;
      }
#pragma unroll
      for (int b = 0; b < 6; b++) {
        if (b >= P.n_eq) break;
        if (!have_nb) {
          const float la[2] = { f32r(da[C - 2]), f32r(da[C - 1]) }, lb[2] = { f32r(db[C - 2]), f32r(db[C - 1]) };
          float pa[2], pb[2];
          prev_ctx2<2>(la, pa, ctx[1 + b], lb, pb, ctx[9 + 1 + b], par, first);
          na1 = pa[1]; na2 = pa[0]; nb1 = pb[1]; nb2 = pb[0];
        }
        double a0, a1, b0, b1;
        biquad_pair_lds(da, na1, na2, P.eq[b], &Pl[kFEq0 + b][0][0], Sl[1 + b], db, nb1, nb2, P.eq[b], &Pl[kFEq0 + b][0][0], Sl[9 + 1 + b], slots.next(), a0, a1, b0, b1);
#pragma unroll
        for (int k = 0; k < C; k++) { da[k] = (double)f32r(da[k]); db[k] = (double)f32r(db[k]); }
        na1 = (double)f32r(a0); na2 = (double)f32r(a1); nb1 = (double)f32r(b0); nb2 = (double)f32r(b1); have_nb = true;
      }
#pragma unroll
      for (int k = 0; k < C; k++) { w[0][k] = f32r(da[k]); w[NCH - 1][k] = f32r(db[k]); }
    } else {
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float v[C];
      load_chunk(x[c], n, base, v);
      if (pre_gain != 1.0) {
#pragma unroll
        for (int k = 0; k < C; k++) v[k] = f32r((double)v[k] * pre_gain);
      }
      double d[C];
#pragma unroll
      for (int k = 0; k < C; k++) d[k] = v[k];
      double nb1 = 0, nb2 = 0;     // the two inputs before the chunk, for the next biquad's FIR part
      bool have_nb = false;
      if (P.do_hpf) {
        const float last[2] = { v[C - 2], v[C - 1] };
        float pv[2];
        prev_ctx<2>(last, pv, ctx[c * 9 + 0], par, first);
        double s0, s1;
        biquad_chunk_lds(d, pv[1], pv[0], P.hpf30, &Pl[kFHpf][0][0], Sl[c * 9 + 0], slots.next(), s0, s1);
#pragma unroll
        for (int k = 0; k < C; k++) d[k] = (double)f32r(d[k]);
        nb1 = (double)f32r(s0); nb2 = (double)f32r(s1); have_nb = true;
      }
      if (P.do_lpf) {
        // applyOnePoleLP (final-filters.js:68-78): y starts at data[0]
        if (t == 0) {
          if (threadIdx.x == 0) bc[c] = f32r(d[0]);
          lds_barrier();
          const float f0 = bc[c];
          if (threadIdx.x == 0) Scar[c] = (f0 == f0) ? (double)f0 : 0.0;
          lds_barrier();
        }
        double u[C];
#pragma unroll
        for (int k = 0; k < C; k++) u[k] = P.lpf_alpha * d[k];
        double s;
        double Sc = Scar[c];
        onepole_chunk(u, P.lpf18, PlpfT[lane], Sc, slots.next(), s);
        if (threadIdx.x == 0) Scar[c] = Sc;
#pragma unroll
        for (int k = 0; k < C; k++) d[k] = (double)f32r(u[k]);
        have_nb = false;
      }
#pragma unroll
      for (int b = 0; b < 6; b++) {
        if (b >= P.n_eq) break;
        if (!have_nb) {
          const float last[2] = { f32r(d[C - 2]), f32r(d[C - 1]) };
          float pv[2];
          prev_ctx<2>(last, pv, ctx[c * 9 + 1 + b], par, first);
          nb1 = pv[1]; nb2 = pv[0];
        }
        double s0, s1;
        biquad_chunk_lds(d, nb1, nb2, P.eq[b], &Pl[kFEq0 + b][0][0], Sl[c * 9 + 1 + b], slots.next(), s0, s1);
#pragma unroll
        for (int k = 0; k < C; k++) d[k] = (double)f32r(d[k]);
        nb1 = (double)f32r(s0); nb2 = (double)f32r(s1); have_nb = true;
      }
#pragma unroll
      for (int k = 0; k < C; k++) w[c][k] = f32r(d[k]);
    }
    }

    if (P.do_glue) {
      // linked side-chain (dsp-worker.js:1287-1293), follower :1304-1312, gain :1318-1324
      float a[C];
#pragma unroll
      for (int k = 0; k < C; k++) {
        float sc;
        if (NCH >= 2) sc = f32r(((double)w[0][k] + (double)w[NCH - 1][k]) * 0.5); else sc = w[0][k];
        a[k] = (base + k < n) ? fabsf(sc) : 0.0f;
      }
      double Senv = Scar[NCH];
      double e = follower_solve(a, P.glue_A, P.glue_R, Senv, glT, slots.next());
      if (threadIdx.x == 0) Scar[NCH] = Senv;
#pragma unroll
      for (int k = 0; k < C; k++) {
        e = follower_step(e, (double)a[k], P.glue_A, P.glue_R);
        float g = 1.0f;
        if (e > P.glue_thr) {
          // 10^(-(20 log10(e/thr)) (1 - 1/ratio) / 20) = (e/thr)^-(1 - 1/ratio); stored to a Float32Array by the reference.
          // Evaluated to ~1e-15: a float32 intrinsic (1 ulp of the store) is enough for this stage alone, but the soft clipper
          // downstream multiplies a one-ulp difference by up to ~550 (tests/probe_hot_margin.py: 4e-5 on hot material).
          g = f32r(pow_fast(div_fast(e, P.glue_thr), -P.glue_k));       // (<= 1 ulp quotient: a third of the IEEE sequence)
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) w[c][k] = f32r((double)w[c][k] * (double)g);
      }
    }

    if (NCH == 2 && P.do_stereo) {
      float mid[C], side[C];
#pragma unroll
      for (int k = 0; k < C; k++) {
        mid[k] = f32r(((double)w[0][k] + (double)w[1][k]) * 0.5);
        side[k] = f32r(((double)w[0][k] - (double)w[1][k]) * 0.5);
      }
      if (P.bass_mono) {
        const float last[2] = { side[C - 2], side[C - 1] };
        float pv[2];
        prev_ctx<2>(last, pv, ctx[NCH * 9], par, first);
        double d[C];
#pragma unroll
        for (int k = 0; k < C; k++) d[k] = side[k];
        double s0, s1;
        biquad_chunk_lds(d, pv[1], pv[0], P.side_hp, &Pl[kFSide][0][0], Sl[NCH * 9], slots.next(), s0, s1);
#pragma unroll
        for (int k = 0; k < C; k++) side[k] = f32r(d[k]);
      }
#pragma unroll
      for (int k = 0; k < C; k++) {
        const float pm = f32r((double)mid[k] * 1.0);
        const float ps = f32r((double)side[k] * P.side_gain);
        w[0][k] = f32r((double)pm + (double)ps);
        w[1][k] = f32r((double)pm - (double)ps);
      }
    }

    if constexpr (NCH == 2) {
      if (t >= sg.T0) { store_chunk(y[0], n, base, w[0]); store_chunk(y[1], n, base, w[1]); }
      if (do_meter) {
        const float la[3] = { w[0][C - 3], w[0][C - 2], w[0][C - 1] }, lb[3] = { w[1][C - 3], w[1][C - 2], w[1][C - 1] };
        float xa[3], xb[3];
        prev_ctx2<3>(la, xa, ctx[8], lb, xb, ctx[9 + 8], par, first);
        meter_chunk_l2(w[0], xa, w[1], xb, base, tile_base, n, *reload_each_tile(mp), ms[0], ms[NCH - 1], lds_meter[0], lds_meter[NCH - 1],
                       hopsums + (trk * NCH) * hop_stride, hopsums + (trk * NCH + 1) * hop_stride);
      }
    } else {
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      if (t >= sg.T0) store_chunk(y[c], n, base, w[c]);
      if (do_meter) {
        const float last[3] = { w[c][C - 3], w[c][C - 2], w[c][C - 1] };
        float xp[3];
        prev_ctx<3>(last, xp, ctx[c * 9 + 8], par, first);
        meter_chunk_l(w[c], xp, base, tile_base, n, *reload_each_tile(mp), ms[c], lds_meter[c], hopsums + (trk * NCH + c) * hop_stride);
      }
    }
    }
  }
  if (do_meter) for (int c = 0; c < NCH; c++) meter_finish(ms[c], lds_meter[c], peaks + trk * NCH + c);
}

int enqueue_front(const float* in, float* out, int ntracks, int nch, int64_t n, const FrontPlan* plan_dev, int do_glue,
                  const MeterPlan* meter_dev, double* hopsums, int64_t hop_stride, double* peaks, double pre_gain, double sr, hipStream_t s) {
  if (n <= 0) return WAM_OK;
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const int64_t mw = meter_warm((int64_t)std::floor(sr * 0.1), sr) + 1;          // +1: the 30 Hz high-pass settles slower than the K-weighting
  const int64_t gw = scale_warm(tuning().glue_warm_tiles, sr);
  const int64_t warm = (do_glue && gw > mw) ? gw : mw;
  // long warm-up (the glue follower's 50 tiles): as few segments as still fill the chip once (2 workgroups per CU resident)
  const int64_t seg = pick_segment_tiles(ntracks, ntiles, warm, tuning().meter_seg_tiles, warm >= 16 ? 512 : 2048);
  const dim3 grid(ntracks, (unsigned)((ntiles + seg - 1) / seg));
  if (nch == 2) WAM_LAUNCH("k_front", k_front<2>, grid, dim3(kThreads), s, in, out, n, plan_dev, meter_dev, hopsums, hop_stride, peaks, pre_gain, seg, warm);
  else WAM_LAUNCH("k_front", k_front<1>, grid, dim3(kThreads), s, in, out, n, plan_dev, meter_dev, hopsums, hop_stride, peaks, pre_gain, seg, warm);
  return WAM_OK;
}

}  // namespace wam
