// meter_host.hpp - host-visible declarations of the metering kernels' launchers.
#pragma once
#include "wam_common.hpp"
#include "wam_plan.hpp"
#include "meter.hpp"

namespace wam {

struct GateOut {
  double mean2;      // mean of the blocks kept by both gates
  double peak;       // linear true peak over the track's channels
  int64_t c1, c2;    // blocks kept by the absolute gate / by both gates
  int64_t nblocks;
};

struct MeterScratch {
  MeterPlan* plan;
  double* hopsums; size_t hopsums_bytes; int64_t hop_stride;
  double* peaks;
  double* blocks;
  GateOut* gate;
  uint8_t* mask;
};

int make_meter_plan(int nch, int64_t n, double sr, double fallback, MeterPlan* pl);
size_t meter_scratch_bytes(const MeterPlan& pl, int ntracks, bool want_mask);
void carve_meter_scratch(const MeterPlan& pl, int ntracks, bool want_mask, void* scratch, MeterScratch* s);
// zero hop sums, run k_meter over [ntracks][nch][n] and then k_gate
int enqueue_meter(const float* in_dev, int ntracks, const MeterPlan& pl, int64_t n, const MeterScratch& s, hipStream_t stream);
// the same with the plan handed over as a kernel argument and LUFS / dBTP finished on the device into lufs_dev / dbtp_dev [ntracks]:
// no host <-> device copy, no synchronisation (csrc/noalloc.hip)
int enqueue_meter_to_device(const float* in_dev, int ntracks, const MeterPlan& pl, int64_t n, const MeterScratch& s, double* lufs_dev,
                            double* dbtp_dev, hipStream_t stream);
// out = f32(in * gains[track]) (out may be NULL, gains may be NULL) + metering + gate
int enqueue_gain_meter(const float* in, float* out, int ntracks, int nch, int64_t n, const double* gains, const MeterPlan& pl,
                       const MeterScratch& s, hipStream_t stream);
int enqueue_momentary(const float* in_dev, int nch, int64_t len, double sr, double* sums_dev, hipStream_t stream);
// many windows of one device-resident track [nch][n]: ungated K-weighted sums per (window, channel) / window copies as a batch
int enqueue_momentary_series(const float* in_dev, int nch, int64_t n, const int64_t* starts_dev, const int64_t* lens_dev, int nwin, double sr,
                             double* sums_dev, hipStream_t stream);
int enqueue_gather_windows(const float* in_dev, int nch, int64_t n, const int64_t* starts_dev, int nwin, int64_t len, float* out_dev, hipStream_t stream);
int enqueue_norm_gain(const MeterScratch& s, int ntracks, double target, double* gains, double* lufs_in, hipStream_t stream);
int enqueue_gate(const MeterPlan& pl, int ntracks, const MeterScratch& s, hipStream_t stream);
double finish_lufs(const MeterPlan& pl, const GateOut& g);
double finish_peak_db(const GateOut& g);

}  // namespace wam
