// noalloc.hip - wam_batch_meter_dev_noalloc: the device-resident batch meter for callers that may not allocate (code under HIP
// graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels as wam_batch_meter_dev (k_meter, k_gate);
// what that entry point takes from the workspace pool - the plan, per-hop sums, peaks, block energies, gate results - lives in
// a file-scope device array instead, and what it copies to the host is written to the caller's device arrays.
//
// The array exists once per GPU (every device that loads the code object holds its own copy at its own address, until the
// process ends), so its address is looked up per device with hipGetSymbolAddress, with that device current, and kept.  All
// calls on one GPU share that GPU's array: it keeps the last call's contents (whatever must start at zero is cleared on the
// call's stream).  Calls on one GPU are therefore ORDERED on the device: a per-device event, created with the scratch
// look-up (at wam_init: the entry point itself still allocates nothing), is recorded behind a call's last kernel and the
// next call's stream waits for it before its first, so two streams (or two host threads: a per-device mutex keeps one
// call's wait / launches / record together) cannot overwrite each other's plan and hop sums; no host synchronisation.
// Under stream capture the event is left alone (a captured stream may not wait for work outside its graph; inside a graph
// the capture order is the order) - two graphs that hold this call must not be launched to overlap on one GPU.
#include "meter_host.hpp"
#include <mutex>

namespace wam {

constexpr size_t kNoAllocScratchBytes = size_t(64) << 20;      // 64 MiB: 1,024 three-minute stereo tracks need 44.5 MB
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_mu;
static void* g_scratch_ptr[kMaxDevices];                       // per device: the address of that device's copy of `scratch`
static hipEvent_t g_last_call[kMaxDevices];                    // per device: recorded behind the last call's work on the scratch array
static std::mutex g_call_mu[kMaxDevices];                      // per device: one call's wait + launches + record are not interleaved with another's

// Looks the device copy's address up once per device; `device` must be the calling thread's current device.
int noalloc_bind_scratch(int device) {
  if (device < 0 || device >= kMaxDevices) { set_error("device %d outside 0..%d", device, kMaxDevices - 1); return WAM_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_scratch_ptr[device]) return WAM_OK;
  void* p = nullptr;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !p) { (void)hipGetLastError(); set_error("hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e)); return WAM_ERR_HIP; }
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); set_error("hipEventCreate failed on device %d", device); return WAM_ERR_HIP; }
  g_last_call[device] = ev;
  g_scratch_ptr[device] = p;
  return WAM_OK;
}

}  // namespace wam

using namespace wam;

extern "C" int wam_batch_meter_dev_noalloc(const float* in_dev, int ntracks, int nch, int64_t n, double sample_rate, double fallback,
                                           double* lufs_dev, double* dbtp_dev, void* stream) {
  if (!in_dev || ntracks < 1 || nch < 1 || nch > 64 || n < 0 || (!lufs_dev && !dbtp_dev)) { set_error("invalid batch arguments"); return WAM_ERR_INVALID; }
  int rc = ensure_device(); if (rc) return rc;
  int device = -1;
  WAM_HIP(hipGetDevice(&device));
  rc = noalloc_bind_scratch(device); if (rc) return rc;        // a no-op after wam_init(device); the first look-up on a device may load the code object there
  void* const ptr = g_scratch_ptr[device];
  MeterPlan pl;
  rc = make_meter_plan(nch, n, sample_rate, fallback, &pl); if (rc) return rc;
  const size_t need = meter_scratch_bytes(pl, ntracks, false);
  if (need > kNoAllocScratchBytes) {
    set_error("wam_batch_meter_dev_noalloc: %d tracks x %d ch x %lld samples need %zu bytes of scratch, the fixed buffer holds %zu; use wam_batch_meter_dev or split the batch",
              ntracks, nch, (long long)n, need, kNoAllocScratchBytes);
    return WAM_ERR_TOO_LARGE;
  }
  MeterScratch s;
  carve_meter_scratch(pl, ntracks, false, ptr, &s);
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(g_call_mu[device]);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  const bool live = cap == hipStreamCaptureStatusNone;
  if (live) WAM_HIP(hipStreamWaitEvent(st, g_last_call[device], 0));     // (an event never recorded is complete)
  rc = enqueue_meter_to_device(in_dev, ntracks, pl, n, s, lufs_dev, dbtp_dev, st);
  if (live && hipEventRecord(g_last_call[device], st) != hipSuccess) {   // also after a failed enqueue: part of it may be queued
    (void)hipGetLastError();
    if (rc == WAM_OK) { set_error("hipEventRecord failed on device %d", device); rc = WAM_ERR_HIP; }
  }
  return rc;
}
