// wam_common.hpp - host-side plumbing shared by the translation units of
// libwam_hip.so: error reporting, device scratch, HIP call checking.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../include/wam.h"

namespace wam {

void set_error(const char* fmt, ...);
int ensure_device();                       // 0 or WAM_ERR_NO_DEVICE; binds the thread to the device chosen by wam_init
int bind_device(int device);               // binds the calling thread to `device`; the process default (wam_init) stays

#define WAM_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      ::wam::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return WAM_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

// Workspace pool (api.hip): device blocks are kept per device after use and handed out again, because a render group
// needs tens of GB of intermediates and hipMalloc / hipFree of that size costs more than the kernels that use it.
// A block returns to the pool when its DevBuf goes out of scope.  On the success path every entry point has synchronised
// its stream by then; on an error path kernels that use the block may still be queued, so pool_free records an event on the
// calling thread's current stream (StreamScope below; the null stream unless an entry point says otherwise) and pool_alloc
// waits for that event before it hands the block to its next owner.  wam_trim_workspace() (include/wam_synth.h) gives the
// memory back to the driver.
void* pool_alloc(size_t bytes, size_t* granted);
void pool_free(void* p, size_t granted);
// Names the stream an entry point enqueues on, for the pool's idle events; nests (restores the previous stream).
struct StreamScope {
  explicit StreamScope(hipStream_t s);
  ~StreamScope();
  StreamScope(const StreamScope&) = delete;
  StreamScope& operator=(const StreamScope&) = delete;
 private:
  hipStream_t prev_;
};

// Host <-> device copies of the host-pointer entry points through a resident pinned ring and a few copy threads (staging.hip).
struct HostSeg { void* host; void* dev; size_t bytes; };
int staged_copy(const HostSeg* segs, int nsegs, bool to_device, hipStream_t stream);   // synchronises `stream` before it returns
void staging_trim();                                                                    // frees the idle pinned rings

// RAII device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0, granted = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) pool_free(p, granted); }
  int alloc(size_t nbytes) {
    if (p) { pool_free(p, granted); p = nullptr; }
    bytes = nbytes;
    p = pool_alloc(nbytes ? nbytes : 16, &granted);
    if (!p) return WAM_ERR_NOMEM;          // pool_alloc set the message
    return WAM_OK;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// per-kernel timing hooks (include/wam_synth.h: wam_profile_enable / wam_profile_read)
int prof_begin(const char* name, hipStream_t s);        // returns a ticket for prof_end (-1 when profiling is off)
void prof_end(int ticket, hipStream_t s);
#define WAM_LAUNCH_SHMEM(name, kern, grid, block, shmem, stream, ...)    \
  do {                                                                   \
    const int prof_ticket_ = ::wam::prof_begin(name, stream);            \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);   \
    ::wam::prof_end(prof_ticket_, stream);                               \
    WAM_HIP(hipGetLastError());                                          \
  } while (0)
#define WAM_LAUNCH(name, kern, grid, block, stream, ...) WAM_LAUNCH_SHMEM(name, kern, grid, block, 0, stream, __VA_ARGS__)

// Time-segmentation knobs (include/wam_synth.h: wam_tune).  Defaults are the production values; tests shrink them so
// that short fixtures cross segment seams.
struct Tuning {
  int64_t dh_seg_frames = 0;      // de-harsh synthesis: frames per workgroup segment, 0 = pick_segment_tiles() over frames
  int64_t dh_exact = 1;           // 1 = de-harsh kernels that reproduce every Float32Array store of the reference (double FFT, sequential
                                  // band sums); 0 = the float32 kernels (1.9x faster, ~1e-7 on the stage's own output, see chain_deharsh.hip)
  int64_t dh_mag_bytes = 0;       // exact de-harsh: scratch for the frames' magnitudes between k_dh_bands_x and k_dh_bandsum (0 = 4 GiB); channels go through in groups that fit
  int64_t dh_warm_frames = 48;    // frames re-run before a segment to rebuild the resonance EMA (0.7^48 = 4e-8) and the overlap-add ring
  int64_t env_seg_tiles = 0;      // soft-clip / limiter envelopes: tiles (4096 samples) per segment, 0 = pick_segment_tiles()
  int64_t sc_warm_tiles = 3;      // soft-clip release: exp(-12288/480) = 8e-12
  int64_t lim_warm_tiles = 25;    // limiter release 100 ms / peak follower 50 ms: exp(-102400/4800) = 5e-10
  int64_t fx_seg_tiles = 0;       // exciter / warmth / punch: tiles per segment, 0 = pick_segment_tiles()
  int64_t lti_warm_tiles = 2;     // kernels whose only state is LTI filters >= 30 Hz: e^-11 per tile at 30 Hz, e^-75 at 200 Hz
  int64_t punch_warm_tiles = 48;  // slow transient follower, 250 ms release: exp(-196608/12000) = 8e-8
  int64_t meter_seg_tiles = 0;    // kernels with a fused meter (k_meter, k_gain_meter, k_lim_apply, k_front): 0 = pick_segment_tiles()
  int64_t meter_warm_tiles = 4;   // a straddling 100 ms hop reaches back 1.17 tiles; K-weighting high-pass e^-20 per tile
  int64_t stft_literal = 0;       // 1 = always use the stage-by-stage STFT kernel (the register-blocked one must equal it bit for bit)
  int64_t glue_warm_tiles = 50;   // glue follower, 250 ms release (+2 for the straddling hop)
  int64_t stage_pinned = 1;       // host-pointer entry points: 1 = copies through the pinned ring (staging.hip), 0 = hipMemcpyAsync on the caller's pageable memory
  int64_t stage_threads = 4;      // copy threads per staged copy (1..8)
  int64_t dh_win_lds = 1;         // 0 = the de-harsh kernels read the Hann table from global memory even when it mirrors (the path a host whose cos() does not mirror takes; test hook)
  int64_t stage_nt = 1;           // staging copies with non-temporal stores (0: plain memcpy)
  int64_t exc_peaks = 1;          // 1 = the exciter also produces tape warmth's window peaks (0: k_win_peak reads the batch again; test hook)
  int64_t host_group = 16;        // wam_batch_render: tracks per staging group (each group is one device batch)
};
Tuning tuning();          // a snapshot (wam_tune is a test / tuning hook: a request that overlaps a wam_tune call may see either value per kernel)

// Warm-up lengths are stated in tiles at 48 kHz; time constants are in seconds, so they grow with the sample rate.
inline int64_t scale_warm(int64_t tiles48k, double sr) {
  const double f = sr > 48000.0 ? sr / 48000.0 : 1.0;
  return (int64_t)std::ceil((double)tiles48k * f - 1e-9);
}
// Fused meter: a straddling 100 ms hop reaches back ceil(hop / 4096) tiles; the rest of the budget (K-weighting settle) scales.
inline int64_t meter_warm(int64_t hop, double sr) {
  const int64_t reach = (hop + 4095) / 4096;
  const int64_t settle = tuning().meter_warm_tiles > 2 ? tuning().meter_warm_tiles - 2 : 1;
  return reach + scale_warm(settle, sr);
}

// Tiles per segment for the kernels that walk a (track, channel) in time order.  With many walkers: enough segments
// for about `want_wgs` workgroups, but never so short that the warm-up re-run costs more than a quarter of the work.
// With few walkers (one track) the chip is idle anyway, so latency wins: segments shrink until the launch has one
// workgroup per resident slot (2 per CU), whatever the warm-up overhead.
inline int64_t pick_segment_tiles(int64_t walkers, int64_t ntiles, int64_t warm, int64_t override_tiles, int64_t want_wgs = 2048,
                                  int64_t slots = 512) {
  if (override_tiles > 0) return override_tiles;
  if (ntiles < 1) return 1;
  if (walkers < 1) walkers = 1;
  const int64_t seg_many = (ntiles * walkers + want_wgs - 1) / want_wgs;
  int64_t seg_fill = (ntiles * walkers + slots - 1) / slots;
  if (seg_fill < 8) seg_fill = 8;
  int64_t floor_seg = 4 * warm < seg_fill ? 4 * warm : seg_fill;
  int64_t seg = seg_many > floor_seg ? seg_many : floor_seg;
  if (seg < 1) seg = 1;
  return seg;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace wam
