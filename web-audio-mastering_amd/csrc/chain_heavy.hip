// chain_heavy.hip - the "heavy FX" stages 2..4 of the export chain (dsp-worker.js:1511-1527):
//   k_exciter  : Add Air, exciter.js:80-214 (two 6 kHz Butterworth high-passes, 2x oversampled tanh, additive mix)
//   k_win_peak : per-100 ms-window peak of every 4th sample, max over channels (multiband-saturation.js:218-230)
//   k_warm_env : bypass envelope: soft knee, asymmetric smoothing, one-window lookahead (:232-268)
//   k_warm_punch: Tape Warmth (LR4 3-band split + biased tanh, multiband-saturation.js:280-352) and/or
//                Add Punch (LR4 split + fast/slow followers + smoothed gain, multiband-transient.js:15-263)
// Workgroups: one per (track, channel) except k_warm_env (one thread per track).
#include "chain.hpp"

namespace wam {

// two more samples of a biquad after the chunk, continuing from its true state
__device__ __forceinline__ void biquad_extra2(const double y[C], const double x[C], double xa, double xb, const Biquad& q, double& ya, double& yb) {
  ya = q.b0 * xa + q.b1 * x[C - 1] + q.b2 * x[C - 2] - q.a1 * y[C - 1] - q.a2 * y[C - 2];
  yb = q.b0 * xb + q.b1 * xa + q.b2 * x[C - 1] - q.a1 * ya - q.a2 * y[C - 1];
}

__global__ __launch_bounds__(kThreads) void k_exciter(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                      const HeavyPlan* __restrict__ hp, double pre_gain) {
  __shared__ double lds_scan[kSlotDoubles];
  __shared__ float ctx[1][kCtxFloats];
  const HeavyPlan& P = *hp;
  const int64_t chan = blockIdx.x;
  const float* x = in + chan * n;
  float* y = out + chan * n;
  LdsSlots slots(lds_scan);
  double P1[4];
  biquad_lane_power(P.exc_hp, lane_id(), P1);
  double Sa[2] = {0, 0}, Sb[2] = {0, 0};
  const double mix = 0.15 / 100, drive = 0.5, bias = 0.2;     // constants.js:36-42, exciter.js:88
  const int64_t ntiles = (n + kTile - 1) / kTile;
  for (int64_t t = 0; t < ntiles; t++) {
    const int64_t base = t * kTile + (int64_t)threadIdx.x * C;
    float v[C];
    load_chunk(x, n, base, v);
    float va = base + C < n ? x[base + C] : 0.0f, vb = base + C + 1 < n ? x[base + C + 1] : 0.0f;
    if (pre_gain != 1.0) {
#pragma unroll
      for (int k = 0; k < C; k++) v[k] = f32r((double)v[k] * pre_gain);
      va = f32r((double)va * pre_gain); vb = f32r((double)vb * pre_gain);
    }
    const float last[2] = { v[C - 2], v[C - 1] };
    float pv[2];
    prev_ctx<2>(last, pv, ctx[0], (int)(t & 1), t == 0);
    double xin[C], d[C];
#pragma unroll
    for (int k = 0; k < C; k++) { xin[k] = v[k]; d[k] = v[k]; }
    double s0, s1;
    biquad_chunk(d, pv[1], pv[0], P.exc_hp, P1, Sa[0], Sa[1], slots.next(), s0, s1);
    double ea, eb;
    biquad_extra2(d, xin, va, vb, P.exc_hp, ea, eb);
    double f1[C];
#pragma unroll
    for (int k = 0; k < C; k++) { f1[k] = (double)f32r(d[k]); d[k] = f1[k]; }
    const double f1a = (double)f32r(ea), f1b = (double)f32r(eb);
    double u0, u1;
    biquad_chunk(d, (double)f32r(s0), (double)f32r(s1), P.exc_hp, P1, Sb[0], Sb[1], slots.next(), u0, u1);
    double ga, gb;
    biquad_extra2(d, f1, f1a, f1b, P.exc_hp, ga, gb);
    // filtered[i-1 .. i+2] around each sample; out-of-range reads are 0 (exciter.js:162-184)
    double f[C + 3];
    f[0] = (base - 1 >= 0) ? (double)f32r(u0) : 0.0;
#pragma unroll
    for (int k = 0; k < C; k++) f[k + 1] = (base + k < n) ? (double)f32r(d[k]) : 0.0;
    f[C + 1] = (base + C < n) ? (double)f32r(ga) : 0.0;
    f[C + 2] = (base + C + 1 < n) ? (double)f32r(gb) : 0.0;
    float o[C];
#pragma unroll
    for (int k = 0; k < C; k++) {
      const double wetA = tanh(f[k + 1] * drive + bias);
      const double sB = catmull_half(f[k], f[k + 1], f[k + 2], f[k + 3]);
      const double wetB = tanh(sB * drive + bias);
      const double wet = (wetA + wetB) * 0.5;
      o[k] = f32r((double)v[k] + wet * mix);
    }
    store_chunk(y, n, base, o);
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

__global__ void k_warm_env(const unsigned* __restrict__ winpeak, int ntracks, int64_t nw, double thr, double kst, int64_t lw,
                           float* __restrict__ sm, float* __restrict__ env) {
  const int64_t trk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (trk >= ntracks) return;
  const unsigned* wp = winpeak + trk * nw;
  float* s = sm + trk * nw;
  float* e = env + trk * nw;
  float prev = 0;
  for (int64_t w = 0; w < nw; w++) {
    const double peak = (double)__uint_as_float(wp[w]);
    float ev;
    if (peak >= thr) ev = 1.0f;
    else if (peak >= kst) { const double r = (peak - kst) / (thr - kst); ev = f32r(r * r); }
    else ev = 0.0f;
    if (w == 0) prev = ev;
    else { const double coef = ev > prev ? 0.5 : 0.03; prev = f32r((double)prev + ((double)ev - (double)prev) * coef); }
    s[w] = prev;
  }
  for (int64_t w = 0; w < nw; w++) {
    int64_t f = w + lw; if (f > nw - 1) f = nw - 1;
    e[w] = lw > 0 ? s[f] : s[w];
  }
}

// LR4 three-band split of a chunk (multiband-saturation.js:125-148 / multiband-transient.js:94-132).
struct SplitState { double P[4][4]; double S[8][2]; };
__device__ __forceinline__ void split_init(SplitState& st, const Biquad& lp1, const Biquad& hp1, const Biquad& lp2, const Biquad& hp2) {
  biquad_lane_power(lp1, lane_id(), st.P[0]); biquad_lane_power(hp1, lane_id(), st.P[1]);
  biquad_lane_power(lp2, lane_id(), st.P[2]); biquad_lane_power(hp2, lane_id(), st.P[3]);
#pragma unroll
  for (int i = 0; i < 8; i++) { st.S[i][0] = 0; st.S[i][1] = 0; }
}
__device__ __forceinline__ void split_chunk(const float v[C], float xm1, float xm2, SplitState& st, const Biquad& lp1, const Biquad& hp1,
                                            const Biquad& lp2, const Biquad& hp2, LdsSlots& slots, float lo[C], float mid[C], float hi[C]) {
  double d[C], s0, s1, t0, t1;
  // low = lp1(lp1(x))
#pragma unroll
  for (int k = 0; k < C; k++) d[k] = v[k];
  biquad_chunk(d, xm1, xm2, lp1, st.P[0], st.S[0][0], st.S[0][1], slots.next(), s0, s1);
#pragma unroll
  for (int k = 0; k < C; k++) d[k] = (double)f32r(d[k]);
  biquad_chunk(d, (double)f32r(s0), (double)f32r(s1), lp1, st.P[0], st.S[1][0], st.S[1][1], slots.next(), t0, t1);
#pragma unroll
  for (int k = 0; k < C; k++) lo[k] = f32r(d[k]);
  // midHigh = hp1(hp1(x))
#pragma unroll
  for (int k = 0; k < C; k++) d[k] = v[k];
  biquad_chunk(d, xm1, xm2, hp1, st.P[1], st.S[2][0], st.S[2][1], slots.next(), s0, s1);
#pragma unroll
  for (int k = 0; k < C; k++) d[k] = (double)f32r(d[k]);
  biquad_chunk(d, (double)f32r(s0), (double)f32r(s1), hp1, st.P[1], st.S[3][0], st.S[3][1], slots.next(), t0, t1);
  float mh[C];
#pragma unroll
  for (int k = 0; k < C; k++) mh[k] = f32r(d[k]);
  const double mh1 = (double)f32r(t0), mh2 = (double)f32r(t1);
  // mid = lp2(lp2(mh))
#pragma unroll
  for (int k = 0; k < C; k++) d[k] = mh[k];
  biquad_chunk(d, mh1, mh2, lp2, st.P[2], st.S[4][0], st.S[4][1], slots.next(), s0, s1);
#pragma unroll
  for (int k = 0; k < C; k++) d[k] = (double)f32r(d[k]);
  biquad_chunk(d, (double)f32r(s0), (double)f32r(s1), lp2, st.P[2], st.S[5][0], st.S[5][1], slots.next(), t0, t1);
#pragma unroll
  for (int k = 0; k < C; k++) mid[k] = f32r(d[k]);
  // high = hp2(hp2(mh))
#pragma unroll
  for (int k = 0; k < C; k++) d[k] = mh[k];
  biquad_chunk(d, mh1, mh2, hp2, st.P[3], st.S[6][0], st.S[6][1], slots.next(), s0, s1);
#pragma unroll
  for (int k = 0; k < C; k++) d[k] = (double)f32r(d[k]);
  biquad_chunk(d, (double)f32r(s0), (double)f32r(s1), hp2, st.P[3], st.S[7][0], st.S[7][1], slots.next(), t0, t1);
#pragma unroll
  for (int k = 0; k < C; k++) hi[k] = f32r(d[k]);
}

__global__ __launch_bounds__(kThreads) void k_warm_punch(const float* __restrict__ in, float* __restrict__ out, int nch, int64_t n,
                                                         const HeavyPlan* __restrict__ hp, int do_warm, int do_punch,
                                                         const float* __restrict__ env, double pre_gain) {
  __shared__ double lds_scan[kSlotDoubles];
  __shared__ float ctx[2][kCtxFloats];
  const HeavyPlan& P = *hp;
  const int64_t chan = blockIdx.x;
  const int64_t trk = chan / nch;
  const float* x = in + chan * n;
  float* y = out + chan * n;
  const float* ev = env ? env + trk * P.w_nw : nullptr;
  LdsSlots slots(lds_scan);
  SplitState sw, sp;
  if (do_warm) split_init(sw, P.w_lp1, P.w_hp1, P.w_lp2, P.w_hp2);
  if (do_punch) split_init(sp, P.p_lp1, P.p_hp1, P.p_lp2, P.p_hp2);
  double Sf[3] = {0, 0, 0}, Ss[3] = {0, 0, 0}, Sg[3] = {1, 1, 1};
  const double Psm = onepole_lane_power(P.p_smooth, lane_id());
  const int64_t ntiles = (n + kTile - 1) / kTile;
  for (int64_t t = 0; t < ntiles; t++) {
    const int64_t base = t * kTile + (int64_t)threadIdx.x * C;
    const int par = (int)(t & 1);
    float v[C];
    load_chunk(x, n, base, v);
    if (pre_gain != 1.0) {
#pragma unroll
      for (int k = 0; k < C; k++) v[k] = f32r((double)v[k] * pre_gain);
    }
    if (do_warm) {
      const float last[2] = { v[C - 2], v[C - 1] };
      float pv[2];
      prev_ctx<2>(last, pv, ctx[0], par, t == 0);
      float band[3][C];
      split_chunk(v, pv[1], pv[0], sw, P.w_lp1, P.w_hp1, P.w_lp2, P.w_hp2, slots, band[0], band[1], band[2]);
#pragma unroll
      for (int k = 0; k < C; k++) {
        int64_t wi = (base + k) / P.w_ws; if (wi > P.w_nw - 1) wi = P.w_nw - 1;
        const double e = (base + k < n) ? (double)ev[wi] : 0.0;
        float sat[3];
#pragma unroll
        for (int b = 0; b < 3; b++) {
          // saturateBand, multiband-saturation.js:292-308 (gain 0 dB)
          const double dry = band[b][k];
          const double wet = (tanh(P.w_drive[b] * (dry + P.w_bias[b])) - P.w_bo[b]) * P.w_mk[b];
          const double sa = e * P.w_mix[b];
          sat[b] = f32r((dry * (1 - sa) + wet * sa) * 1.0);
        }
        v[k] = f32r((double)sat[0] + (double)sat[1] + (double)sat[2]);
      }
    }
    if (do_punch) {
      const float last[2] = { v[C - 2], v[C - 1] };
      float pv[2];
      prev_ctx<2>(last, pv, ctx[1], par, t == 0);
      float band[3][C];
      split_chunk(v, pv[1], pv[0], sp, P.p_lp1, P.p_hp1, P.p_lp2, P.p_hp2, slots, band[0], band[1], band[2]);
      float res[3][C];
#pragma unroll
      for (int b = 0; b < 3; b++) {
        // processTransients, multiband-transient.js:194-263
        double a[C];
#pragma unroll
        for (int k = 0; k < C; k++) a[k] = (base + k < n) ? fabs((double)band[b][k]) : 0.0;
        double fe = follower_solve(a, P.pf_A[b], P.pf_R[b], Sf[b], slots.next());
        double se = follower_solve(a, P.ps_A[b], P.ps_R[b], Ss[b], slots.next());
        double u[C];
#pragma unroll
        for (int k = 0; k < C; k++) {
          fe = follower_step(fe, a[k], P.pf_A[b], P.pf_R[b]);
          se = follower_step(se, a[k], P.ps_A[b], P.ps_R[b]);
          const double diff = fe - se;
          double target;
          if (diff > 0) { const double st = se > 0.0001 ? fmin(diff / se, 1.0) : fmin(diff / 0.0001, 1.0); target = 1 + (P.p_tg[b] - 1) * st; }
          else { const double st = se > 0.0001 ? fmin(-diff / se, 1.0) : fmin(-diff / 0.0001, 1.0); target = 1 + (P.p_sg[b] - 1) * st; }
          u[k] = (1 - P.p_smooth.c) * target;
        }
        double s;
        onepole_chunk(u, P.p_smooth, Psm, Sg[b], slots.next(), s);
#pragma unroll
        for (int k = 0; k < C; k++) res[b][k] = f32r((double)band[b][k] * u[k]);
      }
#pragma unroll
      for (int k = 0; k < C; k++) v[k] = f32r((double)res[0][k] + (double)res[1][k] + (double)res[2][k]);
    }
    store_chunk(y, n, base, v);
  }
}

// ---------------------------------------------------------------------------- launchers
int enqueue_exciter(const float* in, float* out, int ntracks, int nch, int64_t n, const HeavyPlan* plan_dev, double pre_gain, hipStream_t s) {
  if (n <= 0) return WAM_OK;
  WAM_LAUNCH("k_exciter", k_exciter, dim3(ntracks * nch), dim3(kThreads), s, in, out, n, plan_dev, pre_gain);
  return WAM_OK;
}

int enqueue_warm_env(const float* in, int ntracks, int nch, int64_t n, const HeavyPlan& hp_host, unsigned* winpeak, float* sm, float* env,
                     double pre_gain, hipStream_t s) {
  if (n <= 0) return WAM_OK;
  const int64_t nw = hp_host.w_nw, ws = hp_host.w_ws;
  WAM_HIP(hipMemsetAsync(winpeak, 0, sizeof(unsigned) * (size_t)nw * ntracks, s));
  int64_t bx = nw < 512 ? nw : 512;
  WAM_LAUNCH("k_win_peak", k_win_peak, dim3((unsigned)bx, (unsigned)ntracks), dim3(kThreads), s, in, nch, n, ws, nw, winpeak, pre_gain);
  const double thr = std::pow(10, -24 / 20.0), kst = std::pow(10, (-24 - 6) / 20.0);          // multiband-saturation.js:210-211, preset :35
  WAM_LAUNCH("k_warm_env", k_warm_env, dim3((unsigned)((ntracks + 63) / 64)), dim3(64), s, (const unsigned*)winpeak, ntracks, nw, thr, kst,
             hp_host.w_lw, sm, env);
  return WAM_OK;
}

int enqueue_warm_punch(const float* in, float* out, int ntracks, int nch, int64_t n, const HeavyPlan* plan_dev, int do_warm, int do_punch,
                       const float* env, double pre_gain, hipStream_t s) {
  if (n <= 0) return WAM_OK;
  WAM_LAUNCH("k_warm_punch", k_warm_punch, dim3(ntracks * nch), dim3(kThreads), s, in, out, nch, n, plan_dev, do_warm, do_punch, env, pre_gain);
  return WAM_OK;
}

}  // namespace wam
