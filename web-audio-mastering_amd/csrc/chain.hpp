// chain.hpp - plan (all host-computed constants) and launcher declarations for the
// export chain, web/workers/dsp-worker.js:1468-1667.
#pragma once
#include "meter_host.hpp"

namespace wam {

struct FrontPlan {             // stages 5, 6, 7, 7.5 (dsp-worker.js:1556-1599)
  int do_hpf, do_lpf, n_eq, do_glue, do_stereo, do_meter;
  Biquad hpf30;                // final-filters.js:96-131
  OnePole lpf18; double lpf_alpha;
  Biquad eq[6];                // dsp-worker.js:1236-1264 (spec-defined)
  double glue_thr, glue_k, glue_A, glue_R;   // dsp-worker.js:1296-1298; k = 1 - 1/ratio
  Biquad side_hp; double side_gain; int bass_mono;   // dsp-worker.js:1160-1187
};

struct ClipPlan {              // stage 9, soft-clipper.js:165-294
  double ceil_lin, thr_lin, range_div, range, drive, rc, rcC;
  int L;
};

struct LimiterPlan {           // stage 10, dsp-worker.js:286-452
  double ceil_lin, knee_db, knee_start, rc, rcC, pa, pr;
  int L, W, preserve;
  int lib_twin;                // 1: the soft-knee pass of web/lib/dsp/limiter.js:81-128 (running maximum starts at 0, not at |y2|)
};

struct HeavyPlan {             // stages 2-4: exciter.js, multiband-saturation.js, multiband-transient.js
  Biquad exc_hp;
  Biquad w_lp1, w_hp1, w_lp2, w_hp2;
  Biquad p_lp1, p_hp1, p_lp2, p_hp2;
  double w_drive[3], w_bias[3], w_mix[3], w_bo[3], w_mk[3];
  int64_t w_ws, w_nw, w_lw;    // bypass-envelope window (samples), count, lookahead windows
  double pf_A[3], pf_R[3], ps_A[3], ps_R[3], p_tg[3], p_sg[3];
  OnePole p_smooth;
};

// Progress reporting of one request (sendProgress, dsp-worker.js:69-71): a stage is reported once the stages before it are
// really done, hence the stream synchronisation.  A non-zero return from the callback cancels the request.
struct Progress {
  wam_progress_cb cb; void* user; hipStream_t stream;
  int operator()(double p, const char* status) const {
    if (!cb) return WAM_OK;
    WAM_HIP(hipStreamSynchronize(stream));
    if (cb(p, status, user)) { set_error("Cancelled"); return WAM_ERR_CANCELLED; }
    return WAM_OK;
  }
};

// ---- launchers (chain_*.hip).  All buffers are device pointers, layout [ntracks][nch][n]. ----
int enqueue_front(const float* in, float* out, int ntracks, int nch, int64_t n, const FrontPlan* plan_dev, int do_glue,
                  const MeterPlan* meter_dev, double* hopsums, int64_t hop_stride, double* peaks, double pre_gain, double sr, hipStream_t s);

int enqueue_soft_clip(const float* in, float* out, int ntracks, int nch, int64_t n, const ClipPlan* plan_dev, const double* gains_dev, double sr,
                      hipStream_t s);
// prog (optional): the worker limiter's own progress messages, dsp-worker.js:302,364,427,448
int enqueue_limiter(const float* in, float* out, float* tm, float* genv, int ntracks, int nch, int64_t n, const LimiterPlan* plan_dev, int L,
                    int preserve, const MeterPlan* meter_dev, double* hopsums, int64_t hop_stride, double* peaks, double sr, hipStream_t s,
                    const Progress* prog = nullptr);

bool exciter_makes_peaks(const HeavyPlan& hp_host);
int enqueue_exciter(const float* in, float* out, int ntracks, int nch, int64_t n, const HeavyPlan* plan_dev, double pre_gain, double sr, hipStream_t s,
                    unsigned* winpeak = nullptr, const HeavyPlan* hp_host = nullptr);
int enqueue_warm_env(const float* in, int ntracks, int nch, int64_t n, const HeavyPlan& hp_host, unsigned* winpeak, float* sm, float* env,
                     double pre_gain, hipStream_t s, bool have_peaks = false);
int enqueue_warm_punch(const float* in, float* out, int ntracks, int nch, int64_t n, const HeavyPlan* plan_dev, int do_warm, int do_punch,
                       const float* env, double pre_gain, double sr, hipStream_t s);
void make_heavy_plan(double sr, int64_t n, HeavyPlan* p);
int enqueue_deharsh(const float* in, float* out, int ntracks, int nch, int64_t n, double sr, double pre_gain, hipStream_t s,
                    int preset = WAM_DEHARSH_MASTERING);

void make_front_plan(double sr, int do_hpf, int do_lpf, const double* eq_gains, int cut_mud,
                     int do_glue, double glue_thr_db, double glue_ratio, double glue_attack, double glue_release,
                     int do_stereo, double width, int bass_mono, double bass_freq, int do_meter, FrontPlan* p);
int make_clip_plan(double sr, double ceiling_db, double lookahead_ms, double release_ms, double drive, ClipPlan* p);
int make_limiter_plan(double sr, double ceil_lin, double lookahead_ms, double release_ms, double knee_db, int preserve, LimiterPlan* p);

}  // namespace wam
