// chain_api.hip - host side of the export chain: plan builders (reference formulas in
// double with the host libm), the single-stage C ABI entry points and the
// RENDER_FULL_CHAIN driver (dsp-worker.js:1468-1667) that strings the kernels together.
#include "chain.hpp"
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

namespace wam {

// ---------------------------------------------------------------------------- plans
void make_front_plan(double sr, int do_hpf, int do_lpf, const double* eq_gains /*5 or NULL*/, int cut_mud,
                     int do_glue, double glue_thr_db, double glue_ratio, double glue_attack, double glue_release,
                     int do_stereo, double width, int bass_mono, double bass_freq, int do_meter, FrontPlan* p) {
  std::memset(p, 0, sizeof(*p));
  p->do_hpf = do_hpf; p->do_lpf = do_lpf; p->do_glue = do_glue; p->do_stereo = do_stereo; p->do_meter = do_meter;
  p->hpf30 = rbj_pass(sr, 30, 0.707, true);                       // final-filters.js:100-102
  {
    const double rc = 1 / (2 * M_PI * 18000);                     // final-filters.js:69-71
    const double dt = 1 / sr;
    const double alpha = dt / (rc + dt);
    p->lpf_alpha = alpha;
    p->lpf18 = make_onepole(1 - alpha);
  }
  static const int types[5] = {0, 1, 1, 1, 2};
  static const double freqs[5] = {80, 250, 1000, 4000, 12000};   // dsp-worker.js:1236-1259
  p->n_eq = 0;
  for (int b = 0; b < 6; b++) p->eq[b] = identity_biquad();
  if (eq_gains) for (int b = 0; b < 5; b++) if (eq_gains[b] != 0 && eq_gains[b] == eq_gains[b]) p->eq[p->n_eq++] = webaudio_biquad(types[b], sr, freqs[b], eq_gains[b], 1.0);
  if (cut_mud) p->eq[p->n_eq++] = webaudio_biquad(1, sr, 250, -3.0, 1.5);                         // dsp-worker.js:1262-1264
  p->glue_thr = std::pow(10, glue_thr_db / 20);                    // dsp-worker.js:1296-1298
  p->glue_k = 1 - 1 / glue_ratio;
  p->glue_A = std::exp(-1 / (glue_attack * sr));
  p->glue_R = std::exp(-1 / (glue_release * sr));
  p->side_hp = (bass_mono && bass_freq > 0) ? stereo_side_hp(sr, bass_freq) : identity_biquad();
  p->bass_mono = (bass_mono && bass_freq > 0) ? 1 : 0;
  p->side_gain = std::pow(10, 0 / 20.0) * width;                   // dsp-worker.js:1187
}

void make_heavy_plan(double sr, int64_t n, HeavyPlan* p) {
  std::memset(p, 0, sizeof(*p));
  p->exc_hp = exciter_hp(sr, 6000);                                        // constants.js:37
  const double Qw = 0.7071067811865476;                                    // multiband-saturation.js:67
  p->w_lp1 = rbj_pass(sr, 200, Qw, false); p->w_hp1 = rbj_pass(sr, 200, Qw, true);
  p->w_lp2 = rbj_pass(sr, 4000, Qw, false); p->w_hp2 = rbj_pass(sr, 4000, Qw, true);
  p->p_lp1 = rbj_pass(sr, 200, 0.707, false); p->p_hp1 = rbj_pass(sr, 200, 0.707, true);   // multiband-transient.js:101-110
  p->p_lp2 = rbj_pass(sr, 4000, 0.707, false); p->p_hp2 = rbj_pass(sr, 4000, 0.707, true);
  static const double drive[3] = {0.2, 0.4, 0.3}, bias[3] = {0.0, 0.1, 0.05}, mix[3] = {0.3, 0.5, 0.4};   // multiband-saturation.js:27-36
  for (int b = 0; b < 3; b++) {
    p->w_drive[b] = drive[b]; p->w_bias[b] = bias[b]; p->w_mix[b] = mix[b];
    p->w_bo[b] = std::tanh(drive[b] * bias[b]);                            // :285
    p->w_mk[b] = 1 / std::tanh(std::fmax(drive[b], 0.1));                  // :290
  }
  p->w_ws = (int64_t)std::floor(sr * 100 / 1000);                          // :206
  p->w_nw = p->w_ws > 0 ? (n + p->w_ws - 1) / p->w_ws : 0;                 // :207
  p->w_lw = p->w_ws > 0 ? (int64_t)std::ceil(sr * 5 / 1000 / (double)p->w_ws) : 0;   // :257
  static const double fa[3] = {0.005, 0.001, 0.003}, fr[3] = {0.050, 0.040, 0.030}, sa[3] = {0.025, 0.020, 0.015},
                      srl[3] = {0.250, 0.200, 0.150}, tg[3] = {2, 6, 2}, sg[3] = {-1, 0, 0};           // multiband-transient.js:25-53
  for (int b = 0; b < 3; b++) {
    p->pf_A[b] = std::exp(-1 / (fa[b] * sr)); p->pf_R[b] = std::exp(-1 / (fr[b] * sr));
    p->ps_A[b] = std::exp(-1 / (sa[b] * sr)); p->ps_R[b] = std::exp(-1 / (srl[b] * sr));
    p->p_tg[b] = std::pow(10, tg[b] / 20); p->p_sg[b] = std::pow(10, sg[b] / 20);
  }
  p->p_smooth = make_onepole(std::exp(-1 / (0.020 * sr)));
}

int make_clip_plan(double sr, double ceiling_db, double lookahead_ms, double release_ms, double drive, ClipPlan* p) {
  std::memset(p, 0, sizeof(*p));
  p->ceil_lin = std::pow(10, ceiling_db / 20);                     // soft-clipper.js:177-180
  p->thr_lin = std::pow(10, (ceiling_db + 3) / 20);
  p->L = (int)std::floor(sr * lookahead_ms / 1000);
  p->rc = std::exp(-1 / (release_ms * sr / 1000));
  p->rcC = std::pow(p->rc, (double)C);
  p->range = p->ceil_lin - p->thr_lin;                             // soft-clipper.js:203-204
  p->range_div = std::fmax(p->range, 0.001);
  p->drive = drive;
  if (p->L < 0 || p->L > 1024) { set_error("soft-clip lookahead of %d samples is outside 0..1024", p->L); return WAM_ERR_INVALID; }   // kClipMaxL, chain_tail.hip
  return WAM_OK;
}

int make_limiter_plan(double sr, double ceil_lin, double lookahead_ms, double release_ms, double knee_db, int preserve, LimiterPlan* p) {
  std::memset(p, 0, sizeof(*p));
  p->ceil_lin = ceil_lin;
  p->knee_db = knee_db;
  p->knee_start = ceil_lin / std::pow(10, knee_db / 20);           // dsp-worker.js:204-205
  p->L = (int)std::floor(sr * lookahead_ms / 1000);                // dsp-worker.js:299-300
  p->rc = std::exp(-1 / (release_ms * sr / 1000));
  p->rcC = std::pow(p->rc, (double)C);
  p->W = (int)std::floor(sr * 10 / 1000);                          // dsp-worker.js:308-309
  p->pa = std::exp(-1 / (0.001 * sr));                             // dsp-worker.js:335-336
  p->pr = std::exp(-1 / (0.050 * sr));
  p->preserve = preserve;
  // The look-ahead window lives in LDS beside one tile (L <= 4096: the slide copies disjoint ranges); the RMS window is a ring
  // of W prefix sums in LDS (W <= 12,288: 160 KB).  3 ms / 10 ms: sample rates up to 1.2 MHz.
  if (p->L < 1 || p->L > kTile || p->W < 1 || p->W > 12288) {
    set_error("limiter lookahead (%d) outside 1..%d or RMS window (%d) outside 1..12288 samples (sample rate %.1f)", p->L, kTile, p->W, sr);
    return WAM_ERR_INVALID;
  }
  return WAM_OK;
}

// ---------------------------------------------------------------------------- staging helpers
struct Staged {            // host planar channels <-> contiguous device [nch][n]
  DevBuf in, out;
  int nch = 0; int64_t n = 0;
  int up(const float* const* ch, int nch_, int64_t n_, hipStream_t s) {
    nch = nch_; n = n_;
    const size_t bytes = sizeof(float) * (size_t)nch * (size_t)(n > 0 ? n : 1);
    int rc = in.alloc(bytes); if (rc) return rc;
    rc = out.alloc(bytes); if (rc) return rc;
    std::vector<HostSeg> segs;
    for (int c = 0; c < nch && n > 0; c++) segs.push_back(HostSeg{ (void*)ch[c], in.as<float>() + (size_t)c * n, sizeof(float) * (size_t)n });
    return staged_copy(segs.data(), (int)segs.size(), true, s);
  }
  int down(const float* src, float* const* o, hipStream_t s) {
    std::vector<HostSeg> segs;
    for (int c = 0; c < nch && n > 0; c++) segs.push_back(HostSeg{ (void*)o[c], (void*)(src + (size_t)c * n), sizeof(float) * (size_t)n });
    return staged_copy(segs.data(), (int)segs.size(), false, s);
  }
};

static int check_io(const float* const* ch, int nch, int64_t n, float* const* out) {
  if (!ch || !out || nch < 1 || nch > 2 || n < 0) { set_error("invalid audio arguments (nch=%d, n=%lld)", nch, (long long)n); return WAM_ERR_INVALID; }
  for (int c = 0; c < nch; c++) if ((!ch[c] || !out[c]) && n > 0) { set_error("channel %d is NULL", c); return WAM_ERR_INVALID; }
  return WAM_OK;
}

template <class T>
static int upload_plan(const T& host, DevBuf& buf, hipStream_t s) {
  int rc = buf.alloc(sizeof(T)); if (rc) return rc;
  WAM_HIP(hipMemcpyAsync(buf.p, &host, sizeof(T), hipMemcpyHostToDevice, s));
  return WAM_OK;
}

static int run_front_only(const float* const* ch, int nch, int64_t n, double sr, const FrontPlan& fp, float* const* out) {
  int rc = check_io(ch, nch, n, out); if (rc) return rc;
  if (!(sr > 0)) { set_error("invalid sample rate"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  Staged st; rc = st.up(ch, nch, n, nullptr); if (rc) return rc;
  DevBuf plan; rc = upload_plan(fp, plan, nullptr); if (rc) return rc;
  rc = enqueue_front(st.in.as<float>(), st.out.as<float>(), 1, nch, n, plan.as<FrontPlan>(), fp.do_glue, nullptr, nullptr, 0, nullptr, 1.0, sr, nullptr);
  if (rc) return rc;
  return st.down(st.out.as<float>(), out, nullptr);
}


// ---------------------------------------------------------------------------- RENDER_FULL_CHAIN driver
static double js_number_or_zero(double v) { return (v == v) ? v : 0.0; }   // Number(x) || 0

// One group of tracks, everything device-resident.  dsp-worker.js:1468-1667.
static int render_group(const float* in, float* out, int T, int nch, int64_t n, double sr, const wam_settings& st, int mode,
                        wam_result* results, hipStream_t stream, const Progress& prog) {
  int rc;
  const bool exp_mode = mode == WAM_MODE_EXPORT;
  const double ig_db = js_number_or_zero(st.input_gain);
  double pending_gain = ig_db != 0 ? std::pow(10, ig_db / 20) : 1.0;       // dsp-worker.js:1493-1496
  const bool do_deharsh = st.deharsh != 0, do_exc = st.add_air != 0, do_warm = st.tape_warmth != 0, do_punch = st.add_punch != 0;
  const double target = st.target_lufs;
  const bool do_norm = exp_mode && st.normalize_loudness && target != 0 && target == target;      // :1602
  const bool do_limit = exp_mode && st.true_peak_limit;
  const double ceiling_db = (st.true_peak_ceiling != 0 && st.true_peak_ceiling == st.true_peak_ceiling) ? st.true_peak_ceiling : -1;   // :1612
  double eq[5] = { js_number_or_zero(st.eq_low), js_number_or_zero(st.eq_low_mid), js_number_or_zero(st.eq_mid),
                   js_number_or_zero(st.eq_high_mid), js_number_or_zero(st.eq_high) };
  bool do_stereo = false; double width = 1.0;
  if (nch == 2) {                                                          // :1582-1588
    const double w = std::isfinite(st.stereo_width) ? st.stereo_width / 100 : 1.0;
    width = std::fmax(0.0, std::fmin(2.0, w));
    do_stereo = st.center_bass || std::fabs(width - 1.0) > 1e-6;
  }

  // ---- plans
  MeterPlan mp; rc = make_meter_plan(nch, n, sr, -14.0, &mp); if (rc) return rc;
  HeavyPlan hp; make_heavy_plan(sr, n, &hp);
  FrontPlan fp;
  make_front_plan(sr, st.clean_low_end != 0, 1, eq, st.cut_mud != 0, st.glue_compression != 0, -18, 3, 0.02, 0.25,
                  do_stereo, width, st.center_bass != 0, 200, (do_norm || !do_limit) ? 1 : 0, &fp);
  ClipPlan cp; LimiterPlan lp;
  if (do_limit) {
    rc = make_clip_plan(sr, ceiling_db, 0.5, 10, 1.5, &cp); if (rc) return rc;
    rc = make_limiter_plan(sr, std::pow(10, ceiling_db / 20), 3, 100, 3, 1, &lp); if (rc) return rc;
  }

  // ---- scratch
  const size_t chan_elems = (size_t)(n > 0 ? n : 1);
  const size_t nw = (size_t)(hp.w_nw > 0 ? hp.w_nw : 1);
  DevBuf B1, tmb, genv, wpk, wsm, wenv, m1b, m2b, plans, gains_b;
  rc = B1.alloc(sizeof(float) * chan_elems * nch * T); if (rc) return rc;
  if (do_limit) { rc = tmb.alloc(sizeof(float) * chan_elems * T); if (rc) return rc; rc = genv.alloc(sizeof(float) * chan_elems * T); if (rc) return rc; }
  if (do_warm) {
    rc = wpk.alloc(sizeof(unsigned) * nw * T); if (rc) return rc;
    rc = wsm.alloc(sizeof(float) * nw * T); if (rc) return rc;
    rc = wenv.alloc(sizeof(float) * nw * T); if (rc) return rc;
  }
  rc = m1b.alloc(meter_scratch_bytes(mp, T, false)); if (rc) return rc;
  rc = m2b.alloc(meter_scratch_bytes(mp, T, false)); if (rc) return rc;
  MeterScratch M1, M2;
  carve_meter_scratch(mp, T, false, m1b.p, &M1);
  carve_meter_scratch(mp, T, false, m2b.p, &M2);
  WAM_HIP(hipMemcpyAsync(M1.plan, &mp, sizeof(mp), hipMemcpyHostToDevice, stream));
  WAM_HIP(hipMemcpyAsync(M2.plan, &mp, sizeof(mp), hipMemcpyHostToDevice, stream));
  const size_t off_h = 0, off_f = align_up(sizeof(HeavyPlan), 256), off_c = off_f + align_up(sizeof(FrontPlan), 256),
               off_l = off_c + align_up(sizeof(ClipPlan), 256), plan_bytes = off_l + align_up(sizeof(LimiterPlan), 256);
  rc = plans.alloc(plan_bytes); if (rc) return rc;
  char* pb = (char*)plans.p;
  WAM_HIP(hipMemcpyAsync(pb + off_h, &hp, sizeof(hp), hipMemcpyHostToDevice, stream));
  WAM_HIP(hipMemcpyAsync(pb + off_f, &fp, sizeof(fp), hipMemcpyHostToDevice, stream));
  if (do_limit) {
    WAM_HIP(hipMemcpyAsync(pb + off_c, &cp, sizeof(cp), hipMemcpyHostToDevice, stream));
    WAM_HIP(hipMemcpyAsync(pb + off_l, &lp, sizeof(lp), hipMemcpyHostToDevice, stream));
  }
  const HeavyPlan* hp_d = (const HeavyPlan*)(pb + off_h);
  const FrontPlan* fp_d = (const FrontPlan*)(pb + off_f);
  const ClipPlan* cp_d = (const ClipPlan*)(pb + off_c);
  const LimiterPlan* lp_d = (const LimiterPlan*)(pb + off_l);
  rc = gains_b.alloc(sizeof(double) * 2 * (size_t)T); if (rc) return rc;
  double* gains_d = gains_b.as<double>();
  double* lufs_in_d = gains_d + T;

  // ---- buffer ping-pong: the last full-buffer write must land in `out`
  int nwrites = (do_deharsh ? 1 : 0) + (do_exc ? 1 : 0) + (do_warm ? 1 : 0) + (do_punch ? 1 : 0);
  if (exp_mode) nwrites += 1 + (do_limit ? 2 : (do_norm ? 1 : 0));
  const float* src = in;
  float* bufs[2] = { B1.as<float>(), out };
  int wi = (nwrites & 1) ? 1 : 0;                     // index of the first destination
  auto next_dst = [&]() { float* d = bufs[wi]; wi ^= 1; return d; };

  rc = prog(0.05, "Creating audio buffer..."); if (rc) return rc;
  if (ig_db != 0) { rc = prog(0.10, "Applying input gain..."); if (rc) return rc; }
  if (do_deharsh) {
    rc = prog(0.15, "Applying hybrid dynamic processor..."); if (rc) return rc;
    float* dst = next_dst();
    rc = enqueue_deharsh(src, dst, T, nch, n, sr, pending_gain, stream); if (rc) return rc;
    src = dst; pending_gain = 1.0;
  }
  bool peaks_made = false;
  if (do_exc) {
    rc = prog(0.3, "Applying exciter..."); if (rc) return rc;
    float* dst = next_dst();
    peaks_made = do_warm && exciter_makes_peaks(hp);      // tape warmth's window peaks come out of the exciter's registers
    rc = enqueue_exciter(src, dst, T, nch, n, hp_d, pending_gain, sr, stream, peaks_made ? wpk.as<unsigned>() : nullptr, &hp); if (rc) return rc;
    src = dst; pending_gain = 1.0;
  }
  if (do_warm) {
    rc = prog(0.45, "Applying multiband saturation..."); if (rc) return rc;
    rc = enqueue_warm_env(src, T, nch, n, hp, wpk.as<unsigned>(), wsm.as<float>(), wenv.as<float>(), pending_gain, stream, peaks_made); if (rc) return rc;
    float* dst = next_dst();
    rc = enqueue_warm_punch(src, dst, T, nch, n, hp_d, 1, 0, wenv.as<float>(), pending_gain, sr, stream); if (rc) return rc;
    src = dst; pending_gain = 1.0;
  }
  if (do_punch) {
    rc = prog(0.55, "Applying multiband transient..."); if (rc) return rc;
    float* dst = next_dst();
    rc = enqueue_warm_punch(src, dst, T, nch, n, hp_d, 0, 1, nullptr, pending_gain, sr, stream); if (rc) return rc;
    src = dst; pending_gain = 1.0;
  }

  const MeterScratch* final_meter = nullptr;
  bool have_gain = false;
  if (!exp_mode) {
    // preview: heavy FX only, then measureLUFS (dsp-worker.js:1530-1551)
    if (nwrites == 0) {
      std::vector<double> g(T, pending_gain);
      WAM_HIP(hipMemcpyAsync(gains_d, g.data(), sizeof(double) * T, hipMemcpyHostToDevice, stream));
      rc = enqueue_gain_meter(src, out, T, nch, n, pending_gain != 1.0 ? gains_d : nullptr, mp, M2, stream); if (rc) return rc;
      WAM_HIP(hipStreamSynchronize(stream));          // g is a host temporary
    } else {
      rc = enqueue_gain_meter(src, nullptr, T, nch, n, nullptr, mp, M2, stream); if (rc) return rc;
    }
    final_meter = &M2;
  } else {
    rc = prog(0.60, "Applying final filters..."); if (rc) return rc;
    rc = prog(0.65, "Applying EQ..."); if (rc) return rc;
    if (st.glue_compression) { rc = prog(0.70, "Applying glue compressor..."); if (rc) return rc; }
    if (do_stereo) { rc = prog(0.72, "Applying stereo processing..."); if (rc) return rc; }
    {
      float* dst = next_dst();
      if (fp.do_meter) WAM_HIP(hipMemsetAsync(M1.hopsums, 0, M1.hopsums_bytes, stream));
      rc = enqueue_front(src, dst, T, nch, n, fp_d, fp.do_glue, M1.plan, M1.hopsums, M1.hop_stride, M1.peaks, pending_gain, sr, stream); if (rc) return rc;
      src = dst; pending_gain = 1.0;
      if (fp.do_meter) { rc = enqueue_gate(mp, T, M1, stream); if (rc) return rc; final_meter = &M1; }
    }
    if (do_norm) {
      rc = prog(0.75, "Analyzing loudness..."); if (rc) return rc;
      rc = enqueue_norm_gain(M1, T, target, gains_d, lufs_in_d, stream); if (rc) return rc;
      have_gain = true;
    }
    if (do_limit) {
      rc = prog(0.85, "Applying soft clipper..."); if (rc) return rc;
      float* dst = next_dst();
      rc = enqueue_soft_clip(src, dst, T, nch, n, cp_d, have_gain ? gains_d : nullptr, sr, stream); if (rc) return rc;
      src = dst;
      rc = prog(0.95, "Applying final limiter..."); if (rc) return rc;
      dst = next_dst();
      WAM_HIP(hipMemsetAsync(M2.hopsums, 0, M2.hopsums_bytes, stream));
      rc = enqueue_limiter(src, dst, tmb.as<float>(), genv.as<float>(), T, nch, n, lp_d, lp.L, 1, M2.plan, M2.hopsums, M2.hop_stride, M2.peaks, sr, stream, &prog); if (rc) return rc;
      src = dst;
      rc = enqueue_gate(mp, T, M2, stream); if (rc) return rc;
      final_meter = &M2;
    } else if (do_norm) {
      float* dst = next_dst();
      rc = enqueue_gain_meter(src, dst, T, nch, n, gains_d, mp, M2, stream); if (rc) return rc;
      src = dst;
      final_meter = &M2;
    }
  }
  // ---- results
  std::vector<GateOut> g(T);
  std::vector<double> gl(2 * (size_t)T, 0.0);
  WAM_HIP(hipMemcpyAsync(g.data(), final_meter->gate, sizeof(GateOut) * T, hipMemcpyDeviceToHost, stream));
  if (have_gain) WAM_HIP(hipMemcpyAsync(gl.data(), gains_d, sizeof(double) * 2 * T, hipMemcpyDeviceToHost, stream));
  WAM_HIP(hipStreamSynchronize(stream));
  for (int t = 0; t < T && results; t++) {
    wam_result& r = results[t];
    r.lufs = r.measured_lufs = finish_lufs(mp, g[t]);
    r.true_peak_db = finish_peak_db(g[t]);
    r.lufs_in = have_gain ? gl[(size_t)T + t] : NAN;
    r.gain_db = (have_gain && std::isfinite(r.lufs_in)) ? target - r.lufs_in : 0.0;
  }
  if (prog.cb && prog.cb(1.0, "Complete", prog.user)) { set_error("Cancelled"); return WAM_ERR_CANCELLED; }
  return WAM_OK;
}

static int render_batch_dev(const float* in, float* out, int T, int nch, int64_t n, double sr, const wam_settings* st, int mode,
                            wam_result* results, hipStream_t stream, wam_progress_cb cb, void* user) {
  if (!in || !out || !st || T < 1 || nch < 1 || nch > 2 || n < 0 || !(sr > 0)) { set_error("invalid render arguments"); return WAM_ERR_INVALID; }
  int rc = ensure_device(); if (rc) return rc;
  const StreamScope scope(stream);
  const Progress prog{cb, user, stream};
  const int group = 256;                              // bounds the scratch: ~2.5 x the group's audio
  for (int t0 = 0; t0 < T; t0 += group) {
    const int g = T - t0 < group ? T - t0 : group;
    const size_t off = (size_t)t0 * nch * (size_t)n;
    rc = render_group(in + off, out + off, g, nch, n, sr, *st, mode, results ? results + t0 : nullptr, stream, prog);
    if (rc) return rc;
  }
  return WAM_OK;
}

}  // namespace wam

using namespace wam;

extern "C" {

int wam_final_filters(const float* const* ch, int nch, int64_t n, double sr, int highpass, int lowpass, float* const* out) {
  FrontPlan fp;
  make_front_plan(sr, highpass != 0, lowpass != 0, nullptr, 0, 0, -18, 3, 0.02, 0.25, 0, 1.0, 0, 200, 0, &fp);
  return run_front_only(ch, nch, n, sr, fp, out);
}

int wam_parametric_eq(const float* const* ch, int nch, int64_t n, double sr, const double gains_db[5], int cut_mud, float* const* out) {
  if (!gains_db) { set_error("gains_db is NULL"); return WAM_ERR_INVALID; }
  FrontPlan fp;
  make_front_plan(sr, 0, 0, gains_db, cut_mud, 0, -18, 3, 0.02, 0.25, 0, 1.0, 0, 200, 0, &fp);
  return run_front_only(ch, nch, n, sr, fp, out);
}

int wam_glue_compressor(const float* const* ch, int nch, int64_t n, double sr, double threshold_db, double ratio, double attack_s,
                        double release_s, float* const* out) {
  if (!(ratio > 0) || !(attack_s > 0) || !(release_s > 0)) { set_error("invalid compressor parameters"); return WAM_ERR_INVALID; }
  FrontPlan fp;
  make_front_plan(sr, 0, 0, nullptr, 0, 1, threshold_db, ratio, attack_s, release_s, 0, 1.0, 0, 200, 0, &fp);
  return run_front_only(ch, nch, n, sr, fp, out);
}

int wam_soft_clip(const float* const* ch, int nch, int64_t n, double sr, double ceiling_db, double lookahead_ms, double release_ms,
                  double drive, float* const* out) {
  int rc = check_io(ch, nch, n, out); if (rc) return rc;
  if (!(sr > 0)) { set_error("invalid sample rate"); return WAM_ERR_INVALID; }
  ClipPlan cp; rc = make_clip_plan(sr, ceiling_db, lookahead_ms, release_ms, drive, &cp); if (rc) return rc;
  rc = ensure_device(); if (rc) return rc;
  Staged st; rc = st.up(ch, nch, n, nullptr); if (rc) return rc;
  DevBuf plan; rc = upload_plan(cp, plan, nullptr); if (rc) return rc;
  rc = enqueue_soft_clip(st.in.as<float>(), st.out.as<float>(), 1, nch, n, plan.as<ClipPlan>(), nullptr, sr, nullptr);
  if (rc) return rc;
  return st.down(st.out.as<float>(), out, nullptr);
}

static int limiter_entry(const float* const* ch, int nch, int64_t n, double sr, double ceiling_linear, double lookahead_ms, double release_ms,
                         double knee_db, int preserve_transients, float* const* out, int lib_twin, wam_progress_cb cb, void* user) {
  int rc = check_io(ch, nch, n, out); if (rc) return rc;
  if (!(sr > 0)) { set_error("invalid sample rate"); return WAM_ERR_INVALID; }
  LimiterPlan lp; rc = make_limiter_plan(sr, ceiling_linear, lookahead_ms, release_ms, knee_db, preserve_transients != 0, &lp); if (rc) return rc;
  lp.lib_twin = lib_twin;
  rc = ensure_device(); if (rc) return rc;
  Staged st; rc = st.up(ch, nch, n, nullptr); if (rc) return rc;
  DevBuf plan, genv, tm; rc = upload_plan(lp, plan, nullptr); if (rc) return rc;
  rc = genv.alloc(sizeof(float) * (size_t)(n > 0 ? n : 1)); if (rc) return rc;
  rc = tm.alloc(sizeof(float) * (size_t)(n > 0 ? n : 1)); if (rc) return rc;
  const Progress prog{cb, user, nullptr};
  rc = enqueue_limiter(st.in.as<float>(), st.out.as<float>(), tm.as<float>(), genv.as<float>(), 1, nch, n, plan.as<LimiterPlan>(), lp.L,
                       preserve_transients != 0, nullptr, nullptr, 0, nullptr, sr, nullptr, &prog);
  if (rc) return rc;
  rc = st.down(st.out.as<float>(), out, nullptr); if (rc) return rc;
  return prog(1.0, "Complete");                                         // the handler's closing message, dsp-worker.js:1673
}
int wam_limiter(const float* const* ch, int nch, int64_t n, double sr, double ceiling_linear, double lookahead_ms, double release_ms,
                double knee_db, int preserve_transients, float* const* out, wam_progress_cb progress, void* user) {
  return limiter_entry(ch, nch, n, sr, ceiling_linear, lookahead_ms, release_ms, knee_db, preserve_transients, out, 0, progress, user);
}
int wam_limiter_lib(const float* const* ch, int nch, int64_t n, double sr, double ceiling_linear, double lookahead_ms, double release_ms,
                    double knee_db, int preserve_transients, float* const* out) {
  return limiter_entry(ch, nch, n, sr, ceiling_linear, lookahead_ms, release_ms, knee_db, preserve_transients, out, 1, nullptr, nullptr);
}

static int run_heavy(const float* const* ch, int nch, int64_t n, double sr, int which, float* const* out) {
  int rc = check_io(ch, nch, n, out); if (rc) return rc;
  if (!(sr > 0)) { set_error("invalid sample rate"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  HeavyPlan hp; make_heavy_plan(sr, n, &hp);
  Staged st; rc = st.up(ch, nch, n, nullptr); if (rc) return rc;
  DevBuf plan; rc = upload_plan(hp, plan, nullptr); if (rc) return rc;
  if (which == 0) {
    rc = enqueue_exciter(st.in.as<float>(), st.out.as<float>(), 1, nch, n, plan.as<HeavyPlan>(), 1.0, sr, nullptr);
  } else if (which == 1) {
    DevBuf wp, sm, env;
    const size_t nw = (size_t)(hp.w_nw > 0 ? hp.w_nw : 1);
    rc = wp.alloc(sizeof(unsigned) * nw); if (rc) return rc;
    rc = sm.alloc(sizeof(float) * nw); if (rc) return rc;
    rc = env.alloc(sizeof(float) * nw); if (rc) return rc;
    rc = enqueue_warm_env(st.in.as<float>(), 1, nch, n, hp, wp.as<unsigned>(), sm.as<float>(), env.as<float>(), 1.0, nullptr);
    if (rc) return rc;
    rc = enqueue_warm_punch(st.in.as<float>(), st.out.as<float>(), 1, nch, n, plan.as<HeavyPlan>(), 1, 0, env.as<float>(), 1.0, sr, nullptr);
    if (rc) return rc;
    return st.down(st.out.as<float>(), out, nullptr);
  } else {
    rc = enqueue_warm_punch(st.in.as<float>(), st.out.as<float>(), 1, nch, n, plan.as<HeavyPlan>(), 0, 1, nullptr, 1.0, sr, nullptr);
  }
  if (rc) return rc;
  return st.down(st.out.as<float>(), out, nullptr);
}

int wam_deharsh_preset(const float* const* ch, int nch, int64_t n, double sr, int preset, float* const* out) {
  int rc = check_io(ch, nch, n, out); if (rc) return rc;
  if (!(sr > 0)) { set_error("invalid sample rate"); return WAM_ERR_INVALID; }
  if (preset != WAM_DEHARSH_GENTLE && preset != WAM_DEHARSH_AGGRESSIVE && preset != WAM_DEHARSH_MASTERING) { set_error("unknown de-harsh preset"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  Staged st; rc = st.up(ch, nch, n, nullptr); if (rc) return rc;
  rc = enqueue_deharsh(st.in.as<float>(), st.out.as<float>(), 1, nch, n, sr, 1.0, nullptr, preset); if (rc) return rc;
  return st.down(st.out.as<float>(), out, nullptr);
}
int wam_deharsh(const float* const* ch, int nch, int64_t n, double sr, float* const* out) { return wam_deharsh_preset(ch, nch, n, sr, WAM_DEHARSH_MASTERING, out); }

int wam_exciter(const float* const* ch, int nch, int64_t n, double sr, float* const* out) { return run_heavy(ch, nch, n, sr, 0, out); }
int wam_tape_warmth(const float* const* ch, int nch, int64_t n, double sr, float* const* out) { return run_heavy(ch, nch, n, sr, 1, out); }
int wam_multiband_transient(const float* const* ch, int nch, int64_t n, double sr, float* const* out) { return run_heavy(ch, nch, n, sr, 2, out); }

int wam_render_chain(const float* const* ch, int nch, int64_t n, double sr, const wam_settings* settings, int mode, float* const* out,
                     wam_result* result, wam_progress_cb progress, void* user) {
  int rc = check_io(ch, nch, n, out); if (rc) return rc;
  if (!settings || !(sr > 0)) { set_error("invalid render arguments"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  Staged st; rc = st.up(ch, nch, n, nullptr); if (rc) return rc;
  rc = render_batch_dev(st.in.as<float>(), st.out.as<float>(), 1, nch, n, sr, settings, mode, result, nullptr, progress, user);
  if (rc) return rc;
  return st.down(st.out.as<float>(), out, nullptr);
}

int wam_batch_render_dev(const float* in_dev, int ntracks, int nch, int64_t n, double sr, const wam_settings* settings, int mode,
                         float* out_dev, wam_result* results, void* stream) {
  return render_batch_dev(in_dev, out_dev, ntracks, nch, n, sr, settings, mode, results, (hipStream_t)stream, nullptr, nullptr);
}

// Host-pointer batch on ONE device.  Tracks go through the device in staging groups of `host_group` tracks (16: eight groups per
// 128 tracks pipeline copies against kernels better than four, and time segmentation fills the chip from 32 channels; measured
// on the MI355X box, 128 tracks, best of three / typical: groups of 8: 382 / 390 ms, 16: 355 / 405 ms, 32: 427 / 435 ms -
// tools/host_path_timing.py with WAM_HOST_SWEEP=1; the run-to-run spread is the host's, the copies are CPU-memcpy bound);
// up to `max_workers` host threads, each with its own stream and device buffers, pull groups from a shared counter, so
// some groups' copies run while other groups' kernels do.  The copies go through the pinned rings of staging.hip (a ring and
// stage_threads copy threads per worker while it copies).  Results are independent of
// the split: tracks never interact.
// Worker streams are kept for the life of the process (per device, handed out again): an idle event of the workspace pool may
// outlive the worker that recorded it, and the runtime looks at the recording stream when such an event is queried later - a
// destroyed stream there surfaced as 'operation not permitted when stream is capturing' on an unrelated launch.
static std::mutex g_wstream_mu;
static std::vector<std::pair<int, hipStream_t>> g_wstreams;
static hipStream_t worker_stream_acquire(int device) {
  {
    std::lock_guard<std::mutex> lk(g_wstream_mu);
    for (size_t i = 0; i < g_wstreams.size(); i++) if (g_wstreams[i].first == device) { hipStream_t s = g_wstreams[i].second; g_wstreams.erase(g_wstreams.begin() + i); return s; }
  }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return s;
}
static void worker_stream_release(int device, hipStream_t s) { std::lock_guard<std::mutex> lk(g_wstream_mu); g_wstreams.emplace_back(device, s); }

static int batch_render_host(int device, const float* const* in, int ntracks, int nch, int64_t n, double sr, const wam_settings* settings, int mode,
                             float* const* out, wam_result* results, int max_workers) {
  const size_t per = (size_t)nch * (size_t)(n > 0 ? n : 1);
  const int group = (int)tuning().host_group;          // tracks per staging group
  const int ngroups = (ntracks + group - 1) / group;
  const int nworkers = ngroups < max_workers ? ngroups : max_workers;
  std::atomic<int> next{0};
  std::vector<int> rcs(nworkers, WAM_OK);
  std::vector<std::string> errs(nworkers);
  auto worker = [&](int w) {
    auto fail = [&](int code) { rcs[w] = code; errs[w] = wam_last_error(); next.store(ngroups); };   // error strings are per thread
    if (bind_device(device) != WAM_OK) return fail(WAM_ERR_NO_DEVICE);
    hipStream_t st = worker_stream_acquire(device);
    if (!st) { set_error("hipStreamCreate failed"); return fail(WAM_ERR_HIP); }
    auto body = [&]() -> int {
      const StreamScope scope(st);
      DevBuf din, dout;
      int r = din.alloc(sizeof(float) * per * group); if (r) return r;
      r = dout.alloc(sizeof(float) * per * group); if (r) return r;
      for (;;) {
        const int gi = next.fetch_add(1);
        if (gi >= ngroups) break;
        const int t0 = gi * group, g = ntracks - t0 < group ? ntracks - t0 : group;
        std::vector<HostSeg> segs;
        for (int t = 0; t < g; t++) for (int c = 0; c < nch && n > 0; c++)
          segs.push_back(HostSeg{ (void*)in[(size_t)(t0 + t) * nch + c], din.as<float>() + ((size_t)t * nch + c) * n, sizeof(float) * (size_t)n });
        r = staged_copy(segs.data(), (int)segs.size(), true, st); if (r) return r;
        r = render_batch_dev(din.as<float>(), dout.as<float>(), g, nch, n, sr, settings, mode, results ? results + t0 : nullptr, st, nullptr, nullptr);
        if (r) return r;
        segs.clear();
        for (int t = 0; t < g; t++) for (int c = 0; c < nch && n > 0; c++)
          segs.push_back(HostSeg{ (void*)out[(size_t)(t0 + t) * nch + c], dout.as<float>() + ((size_t)t * nch + c) * n, sizeof(float) * (size_t)n });
        r = staged_copy(segs.data(), (int)segs.size(), false, st); if (r) return r;
      }
      return WAM_OK;
    };
    const int r = body();
    if (hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError();
    worker_stream_release(device, st);
    if (r) fail(r);
  };
  // (a std::system_error from thread creation must not cross the C ABI: the workers that did start drain the queue)
  std::vector<std::thread> extra;
  try { for (int w = 1; w < nworkers; w++) extra.emplace_back(worker, w); } catch (...) {}
  worker(0);
  for (auto& t : extra) t.join();
  for (int w = 0; w < nworkers; w++) if (rcs[w]) { set_error("%s", errs[w].c_str()); return rcs[w]; }
  return WAM_OK;
}

static int check_batch(const float* const* in, int ntracks, int nch, int64_t n, double sr, const wam_settings* settings, float* const* out) {
  if (!in || !out || !settings || ntracks < 1 || nch < 1 || nch > 2 || n < 0 || !(sr > 0)) { set_error("invalid batch arguments"); return WAM_ERR_INVALID; }
  for (int64_t i = 0; i < (int64_t)ntracks * nch; i++) if ((!in[i] || !out[i]) && n > 0) { set_error("channel pointer %lld is NULL", (long long)i); return WAM_ERR_INVALID; }
  return WAM_OK;
}

int wam_batch_render(const float* const* in, int ntracks, int nch, int64_t n, double sr, const wam_settings* settings, int mode,
                     float* const* out, wam_result* results) {
  int rc = check_batch(in, ntracks, nch, n, sr, settings, out); if (rc) return rc;
  rc = ensure_device(); if (rc) return rc;
  int device = 0; WAM_HIP(hipGetDevice(&device));
  return batch_render_host(device, in, ntracks, nch, n, sr, settings, mode, out, results, 4);
}

// Multi-device batch: entry i of `devices` renders the contiguous track range
// [ceil(i ntracks / ndev), ceil((i + 1) ntracks / ndev)) - track t belongs to entry floor(t ndev / ntracks), the rule of
// wam_dist.shard_range - on HIP device devices[i], from a host thread of its own with its own streams and staging
// buffers.  No collective: tracks are independent, the per-track results land in the caller's array at their track index.
// A device may appear more than once (two workers sharing one GPU).
int wam_batch_render_multi(const float* const* in, int ntracks, int nch, int64_t n, double sr, const wam_settings* settings, int mode,
                           float* const* out, wam_result* results, const int* devices, int ndev) {
  int rc = check_batch(in, ntracks, nch, n, sr, settings, out); if (rc) return rc;
  if (!devices || ndev < 1 || ndev > 64) { set_error("invalid device list (1..64 entries)"); return WAM_ERR_INVALID; }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); set_error("no HIP device available (this backend has no CPU path)"); return WAM_ERR_NO_DEVICE; }
  for (int i = 0; i < ndev; i++) if (devices[i] < 0 || devices[i] >= count) { set_error("devices[%d] = %d out of range (0..%d)", i, devices[i], count - 1); return WAM_ERR_INVALID; }
  int before = -1; (void)hipGetDevice(&before);
  const int per_dev_workers = ndev >= 2 ? 2 : 4;                       // workers per entry (each drives up to stage_threads copy threads)
  std::vector<int> rcs(ndev, WAM_OK);
  std::vector<std::string> errs(ndev);
  auto entry = [&](int i) {
    const int64_t lo = ((int64_t)i * ntracks + ndev - 1) / ndev, hi = ((int64_t)(i + 1) * ntracks + ndev - 1) / ndev;
    if (hi <= lo) return;
    rcs[i] = batch_render_host(devices[i], in + lo * nch, (int)(hi - lo), nch, n, sr, settings, mode, out + lo * nch, results ? results + lo : nullptr, per_dev_workers);
    if (rcs[i]) errs[i] = wam_last_error();
  };
  // (thread creation may throw std::system_error, which must not cross the C ABI: an entry whose thread could not be
  // started runs on the caller's thread after entry 0)
  std::vector<std::thread> threads;
  std::vector<int> inline_entries;
  for (int i = 1; i < ndev; i++) {
    try { threads.emplace_back(entry, i); } catch (...) { inline_entries.push_back(i); }
  }
  entry(0);
  for (int i : inline_entries) entry(i);
  for (auto& t : threads) t.join();
  if (before >= 0) (void)bind_device(before);                           // entry 0 ran on the caller's thread: restore its device
  for (int i = 0; i < ndev; i++) if (rcs[i]) { set_error("device entry %d (device %d): %s", i, devices[i], errs[i].c_str()); return rcs[i]; }
  return WAM_OK;
}

int wam_stereo(const float* const* ch, int64_t n, double sr, double width, int bass_mono, double bass_freq, float* const* out) {
  FrontPlan fp;
  make_front_plan(sr, 0, 0, nullptr, 0, 0, -18, 3, 0.02, 0.25, 1, width, bass_mono, bass_freq, 0, &fp);
  return run_front_only(ch, 2, n, sr, fp, out);
}

}  // extern "C"
