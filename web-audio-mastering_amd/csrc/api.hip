// api.hip - the C ABI of libwam_hip.so (include/wam.h): argument checking, host
// <-> device staging, and the calls into the kernel launchers.  There is no CPU
// path here: every compute entry point needs a HIP device.
#include "meter_host.hpp"
#include "../../include/wam_synth.h"
#include <atomic>
#include <map>
#include <mutex>
#include <cmath>
#include <cstring>
#include <string>

namespace wam {

static thread_local std::string g_err;
static thread_local int g_device = -1;
static std::atomic<int> g_default_device{-1};   // set by every wam_init; threads that never called wam_init use the latest

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  g_err = buf;
}

static std::mutex g_tune_mu;
static Tuning g_tuning;
Tuning tuning() { std::lock_guard<std::mutex> lk(g_tune_mu); return g_tuning; }
#ifdef WAM_COUNT_ITERS
__device__ unsigned long long g_iter_stats[8];
__device__ unsigned long long g_iter_hist[4];
#endif

// ---- workspace pool -----------------------------------------------------------
// A block comes back to the pool when its DevBuf goes out of scope - on error paths possibly while kernels that use it are
// still queued.  pool_free therefore records an event on the stream the calling entry point works on (StreamScope, thread-
// local; the null stream by default) and the next owner waits for that event before it gets the block.
struct PoolBlock { void* p; size_t bytes; int device; hipEvent_t idle; };
static std::mutex g_pool_mu;
static std::vector<PoolBlock> g_pool;                 // blocks not owned by a DevBuf
static thread_local hipStream_t g_scope_stream = nullptr;
StreamScope::StreamScope(hipStream_t s) : prev_(g_scope_stream) { g_scope_stream = s; }
StreamScope::~StreamScope() { g_scope_stream = prev_; }

// Idle events are recycled per device (a render call releases dozens of blocks: creating and destroying an event for each
// was host overhead on the single-track latency path), and a block's event is waited for AFTER g_pool_mu has been released,
// so one stream that is still busy does not stall every other worker's alloc and free.
static std::vector<std::pair<int, hipEvent_t>> g_spare_events;     // (device, event); guarded by g_pool_mu
static hipEvent_t take_event(int dev) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_spare_events.size(); i++)
      if (g_spare_events[i].first == dev) { const hipEvent_t e = g_spare_events[i].second; g_spare_events.erase(g_spare_events.begin() + i); return e; }
  }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return e;
}
static void give_event(int dev, hipEvent_t e) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_spare_events.size() < 256) { g_spare_events.emplace_back(dev, e); return; }
  (void)hipEventDestroy(e);
}

void* pool_alloc(size_t bytes, size_t* granted) {
  const size_t gran = bytes >= (size_t(1) << 20) ? (size_t(2) << 20) : 256;
  const size_t want = (bytes + gran - 1) / gran * gran;
  int dev = 0; (void)hipGetDevice(&dev);
  PoolBlock hit{nullptr, 0, 0, nullptr};
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int i = 0; i < (int)g_pool.size(); i++) {
      const PoolBlock& b = g_pool[i];
      if (b.device != dev || b.bytes < want || b.bytes > want + want / 4 + 4096) continue;      // best fit, at most 25 % slack
      if (best < 0 || b.bytes < g_pool[best].bytes) best = i;
    }
    if (best >= 0) { hit = g_pool[best]; g_pool.erase(g_pool.begin() + best); }
  }
  if (hit.p) {
    *granted = hit.bytes;
    if (hit.idle) {                                     // the previous owner's last work on this block (usually long done)
      if (hipEventQuery(hit.idle) != hipSuccess) {
        (void)hipGetLastError();
        if (hipEventSynchronize(hit.idle) != hipSuccess) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }   // (never leave a sticky error for the next launch check)
      }
      give_event(dev, hit.idle);
    }
    return hit.p;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {                              // give the idle blocks back and try once more
    (void)hipGetLastError();
    wam_trim_workspace();
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return nullptr; }
  *granted = want;
  return p;
}
void pool_free(void* p, size_t granted) {
  if (!p) return;
  int dev = 0; (void)hipGetDevice(&dev);
  hipEvent_t ev = take_event(dev);
  if (!ev || hipEventRecord(ev, g_scope_stream) != hipSuccess) {
    (void)hipGetLastError();
    if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    (void)hipStreamSynchronize(g_scope_stream);      // no event: make the block idle the slow way
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pool.push_back(PoolBlock{p, granted, dev, ev});
}

// ---- per-kernel timing ------------------------------------------------------
struct ProfEntry { const char* name; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfEntry> g_prof;

// Returns the entry's ticket (-1: profiling off): several host threads may launch at once (wam_batch_render's workers), so
// prof_end names the entry it closes instead of taking the newest one.
int prof_begin(const char* name, hipStream_t s) {
  if (!g_prof_on) return -1;
  ProfEntry e; e.name = name; e.a = e.b = nullptr;
  if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) { if (e.a) (void)hipEventDestroy(e.a); (void)hipGetLastError(); return -1; }
  (void)hipEventRecord(e.a, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(e);
  return (int)g_prof.size() - 1;
}
void prof_end(int ticket, hipStream_t s) {
  if (ticket < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (ticket < (int)g_prof.size()) (void)hipEventRecord(g_prof[ticket].b, s);     // (wam_profile_read clears the log: callers read it between steps)
}

// Binds the calling thread to `device` without touching the process default (worker threads of the batch entry points).
int noalloc_bind_scratch(int device);      // noalloc.hip: per-device address of the fixed scratch array, looked up with the device current
int bind_device(int device) {
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); set_error("hipSetDevice(%d) failed", device); return WAM_ERR_NO_DEVICE; }
  g_device = device;
  (void)noalloc_bind_scratch(device);      // (wam_batch_meter_dev_noalloc reports a failed look-up itself)
  return WAM_OK;
}

int ensure_device() {
  if (g_device >= 0) {
    if (hipSetDevice(g_device) != hipSuccess) { set_error("hipSetDevice(%d) failed", g_device); return WAM_ERR_NO_DEVICE; }
    return WAM_OK;
  }
  const int d = g_default_device.load();
  return wam_init(d >= 0 ? d : 0);
}

// The meters take any channel count, like the reference's loops over numberOfChannels (lufs.js:28-65, true-peak.js:46-71);
// kMaxMeterChannels only bounds the grid arithmetic.
constexpr int kMaxMeterChannels = 64;
static int check_audio(const float* const* ch, int nch, int64_t n, int max_nch = kMaxMeterChannels) {
  if (!ch || nch < 1 || nch > max_nch || n < 0) { set_error("invalid audio arguments (nch=%d, n=%lld; 1..%d channels)", nch, (long long)n, max_nch); return WAM_ERR_INVALID; }
  for (int c = 0; c < nch; c++) if (!ch[c] && n > 0) { set_error("channel %d is NULL", c); return WAM_ERR_INVALID; }
  return WAM_OK;
}

// host planar channels -> contiguous device [nch][n]
static int upload(const float* const* ch, int nch, int64_t n, DevBuf& buf, hipStream_t stream) {
  int rc = buf.alloc(sizeof(float) * (size_t)nch * (size_t)(n > 0 ? n : 1));
  if (rc) return rc;
  std::vector<HostSeg> segs;
  for (int c = 0; c < nch && n > 0; c++) segs.push_back(HostSeg{ (void*)ch[c], buf.as<float>() + (size_t)c * n, sizeof(float) * (size_t)n });
  return staged_copy(segs.data(), (int)segs.size(), true, stream);
}

static int meter_dev(const float* in_dev, int ntracks, int nch, int64_t n, double sr, double fallback,
                     double* lufs, double* dbtp, double* block_ms, uint8_t* gate_mask, int64_t* nblocks_out, hipStream_t stream) {
  const StreamScope scope(stream);
  MeterPlan pl;
  int rc = make_meter_plan(nch, n, sr, fallback, &pl);
  if (rc) return rc;
  const bool want_mask = gate_mask != nullptr;
  DevBuf scratch;
  rc = scratch.alloc(meter_scratch_bytes(pl, ntracks, want_mask));
  if (rc) return rc;
  MeterScratch s;
  carve_meter_scratch(pl, ntracks, want_mask, scratch.p, &s);
  rc = enqueue_meter(in_dev, ntracks, pl, n, s, stream);
  if (rc) return rc;
  std::vector<GateOut> g(ntracks);
  WAM_HIP(hipMemcpyAsync(g.data(), s.gate, sizeof(GateOut) * ntracks, hipMemcpyDeviceToHost, stream));
  if (block_ms && pl.nblocks > 0) WAM_HIP(hipMemcpyAsync(block_ms, s.blocks, sizeof(double) * pl.nblocks, hipMemcpyDeviceToHost, stream));
  if (gate_mask && pl.nblocks > 0) WAM_HIP(hipMemcpyAsync(gate_mask, s.mask, (size_t)pl.nblocks, hipMemcpyDeviceToHost, stream));
  WAM_HIP(hipStreamSynchronize(stream));
  for (int t = 0; t < ntracks; t++) {
    if (lufs) lufs[t] = finish_lufs(pl, g[t]);
    if (dbtp) dbtp[t] = finish_peak_db(g[t]);
  }
  if (nblocks_out) *nblocks_out = pl.too_short ? 0 : pl.nblocks;
  return WAM_OK;
}

}  // namespace wam

using namespace wam;

extern "C" {

const char* wam_last_error(void) { return g_err.c_str(); }
const char* wam_version(void) { return "wam-hip 0.3 (gfx950)"; }
int wam_abi_version(void) { return WAM_ABI_VERSION; }

int wam_debug_counters(uint64_t out[8], int reset) {
  for (int i = 0; i < 8; i++) out[i] = 0;
#ifdef WAM_COUNT_ITERS
  WAM_HIP(hipDeviceSynchronize());
  WAM_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(wam::g_iter_stats), sizeof(uint64_t) * 8));
  uint64_t h[4];                       // the unused half of out[] (sites 3 would be [6], [7]) carries the pair histogram
  WAM_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(wam::g_iter_hist), sizeof(h)));
  out[6] = h[0] << 32 | (h[1] & 0xffffffffull); out[7] = h[2] << 32 | (h[3] & 0xffffffffull);
  if (reset) { uint64_t z[8] = {0}; WAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(wam::g_iter_stats), z, sizeof(z))); WAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(wam::g_iter_hist), z, sizeof(h))); }
#endif
  return WAM_OK;
}
int wam_trim_workspace(void) {
  std::vector<wam::PoolBlock> blocks;
  std::vector<std::pair<int, hipEvent_t>> spare;
  { std::lock_guard<std::mutex> lk(wam::g_pool_mu); blocks.swap(wam::g_pool); spare.swap(wam::g_spare_events); }
  int cur = 0; (void)hipGetDevice(&cur);
  for (const auto& b : blocks) {
    (void)hipSetDevice(b.device);
    if (b.idle) { (void)hipEventSynchronize(b.idle); (void)hipEventDestroy(b.idle); }
    (void)hipFree(b.p);
  }
  for (const auto& e : spare) { (void)hipSetDevice(e.first); (void)hipEventDestroy(e.second); }
  wam::staging_trim();
  (void)hipSetDevice(cur);
  return WAM_OK;
}
int wam_tune(const char* key, int64_t value) {
  if (!key || value < 0) { wam::set_error("wam_tune: bad argument"); return WAM_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(wam::g_tune_mu);
  wam::Tuning& t = wam::g_tuning;
  const std::string k = key;
  if (k == "dh_seg_frames" && (value == 0 || value >= 4)) t.dh_seg_frames = value;
  else if (k == "dh_warm_frames" && value >= 3) t.dh_warm_frames = value;
  else if (k == "dh_exact") t.dh_exact = value;
  else if (k == "dh_mag_bytes" && value >= 0) t.dh_mag_bytes = value;
  else if (k == "env_seg_tiles") t.env_seg_tiles = value;
  else if (k == "sc_warm_tiles" && value >= 1) t.sc_warm_tiles = value;
  else if (k == "lim_warm_tiles" && value >= 1) t.lim_warm_tiles = value;
  else if (k == "fx_seg_tiles") t.fx_seg_tiles = value;
  else if (k == "lti_warm_tiles" && value >= 1) t.lti_warm_tiles = value;
  else if (k == "punch_warm_tiles" && value >= 1) t.punch_warm_tiles = value;
  else if (k == "meter_seg_tiles") t.meter_seg_tiles = value;
  else if (k == "meter_warm_tiles" && value >= 3) t.meter_warm_tiles = value;
  else if (k == "glue_warm_tiles" && value >= 3) t.glue_warm_tiles = value;
  else if (k == "stft_literal") t.stft_literal = value;
  else if (k == "stage_pinned") t.stage_pinned = value;
  else if (k == "stage_threads" && value >= 1 && value <= 8) t.stage_threads = value;
  else if (k == "host_group" && value >= 1 && value <= 256) t.host_group = value;
  else if (k == "exc_peaks") t.exc_peaks = value != 0;
  else if (k == "stage_nt") t.stage_nt = value != 0;
  else if (k == "dh_win_lds") t.dh_win_lds = value != 0;
  else if (k == "reset") t = wam::Tuning();
  else { wam::set_error("wam_tune: unknown key or value out of range: %s = %lld", key, (long long)value); return WAM_ERR_INVALID; }
  return WAM_OK;
}
int wam_profile_enable(int on) { g_prof_on = on != 0; return WAM_OK; }

int wam_profile_read(char* buf, size_t cap) {
  (void)hipDeviceSynchronize();
  std::map<std::string, std::pair<int, double>> agg;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { auto& a = agg[e.name]; a.first++; a.second += ms; }
      (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    g_prof.clear();
  }
  std::string out;
  for (auto& kv : agg) { char line[256]; snprintf(line, sizeof(line), "%s %d %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second); out += line; }
  if (buf && cap > 0) { size_t m = out.size() < cap - 1 ? out.size() : cap - 1; memcpy(buf, out.data(), m); buf[m] = 0; return (int)m; }
  return 0;
}

int wam_device_count(int* count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { n = 0; (void)hipGetLastError(); }
  if (count) *count = n;
  return WAM_OK;
}

int wam_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (this backend has no CPU path)");
    return WAM_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) { set_error("device %d out of range (0..%d)", device, n - 1); return WAM_ERR_INVALID; }
  if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return WAM_ERR_NO_DEVICE; }
  g_device = device;
  g_default_device.store(device);          // threads that never called wam_init (libuv workers) follow the LAST explicit choice
  (void)noalloc_bind_scratch(device);      // set-up for wam_batch_meter_dev_noalloc: once per device, with that device current (it reports a failed look-up itself)
  return WAM_OK;
}

int64_t wam_lufs_num_blocks(int64_t n, double sample_rate) {
  if (!(sample_rate > 0) || n <= 0) return 0;
  const int64_t bs = (int64_t)std::floor(sample_rate * 0.4), hop = (int64_t)std::floor(sample_rate * 0.4 * (1 - 0.75));
  if (hop <= 0 || n < bs) return 0;
  return (n - bs) / hop + 1;
}

int wam_measure_lufs(const float* const* ch, int nch, int64_t n, double sample_rate, double fallback, double* lufs,
                     double* block_ms, uint8_t* gate_mask, int64_t* nblocks_out) {
  int rc = check_audio(ch, nch, n); if (rc) return rc;
  if (!lufs) { set_error("lufs is NULL"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  DevBuf in;
  rc = upload(ch, nch, n, in, nullptr); if (rc) return rc;
  return meter_dev(in.as<float>(), 1, nch, n, sample_rate, fallback, lufs, nullptr, block_ms, gate_mask, nblocks_out, nullptr);
}

int wam_true_peak_db(const float* const* ch, int nch, int64_t n, double* dbtp) {
  int rc = check_audio(ch, nch, n); if (rc) return rc;
  if (!dbtp) { set_error("dbtp is NULL"); return WAM_ERR_INVALID; }
  rc = ensure_device(); if (rc) return rc;
  DevBuf in;
  rc = upload(ch, nch, n, in, nullptr); if (rc) return rc;
  return meter_dev(in.as<float>(), 1, nch, n, 48000.0, -14.0, nullptr, dbtp, nullptr, nullptr, nullptr, nullptr);
}

int wam_meter(const float* const* ch, int nch, int64_t n, double sample_rate, double fallback, double* lufs, double* dbtp) {
  int rc = check_audio(ch, nch, n); if (rc) return rc;
  rc = ensure_device(); if (rc) return rc;
  DevBuf in;
  rc = upload(ch, nch, n, in, nullptr); if (rc) return rc;
  return meter_dev(in.as<float>(), 1, nch, n, sample_rate, fallback, lufs, dbtp, nullptr, nullptr, nullptr, nullptr);
}

int wam_short_term_lufs(const float* const* ch, int nch, int64_t n, double sample_rate, double position, double* lufs) {
  int rc = check_audio(ch, nch, n); if (rc) return rc;
  if (!lufs || !(sample_rate > 0) || !(position >= 0)) { set_error("invalid short-term arguments"); return WAM_ERR_INVALID; }
  const int64_t win = (int64_t)std::floor(sample_rate * 3000 / 1000);              // lufs.js:93-96
  const double sp = std::floor(position * sample_rate);
  if (!(sp < (double)n)) { *lufs = -INFINITY; return WAM_OK; }                     // past the end (also keeps the cast below in range): :98-100
  const int64_t start = (int64_t)sp;
  const int64_t end = n - start > win ? start + win : n;
  if ((double)(end - start) < (double)win * 0.5) { *lufs = -INFINITY; return WAM_OK; }   // :98-100
  std::vector<const float*> view(nch);
  for (int c = 0; c < nch; c++) view[c] = ch[c] + start;
  return wam_measure_lufs(view.data(), nch, end - start, sample_rate, -14.0, lufs, nullptr, nullptr, nullptr);   // :118, default fallback
}

int wam_momentary_lufs(const float* const* ch, int nch, int64_t n, double sample_rate, double position, double* lufs) {
  int rc = check_audio(ch, nch, n); if (rc) return rc;
  if (!lufs || !(sample_rate > 0) || !(position >= 0)) { set_error("invalid momentary arguments"); return WAM_ERR_INVALID; }
  const int64_t win = (int64_t)std::floor(sample_rate * 400 / 1000);               // lufs.js:130-133
  const double sp = std::floor(position * sample_rate);
  if (!(sp < (double)n)) { *lufs = -INFINITY; return WAM_OK; }                     // past the end (also keeps the cast below in range): :135-137
  const int64_t start = (int64_t)sp;
  const int64_t end = n - start > win ? start + win : n;
  if ((double)(end - start) < (double)win * 0.5) { *lufs = -INFINITY; return WAM_OK; }   // :135-137
  rc = ensure_device(); if (rc) return rc;
  const int64_t len = end - start;
  std::vector<const float*> view(nch);
  for (int c = 0; c < nch; c++) view[c] = ch[c] + start;
  DevBuf in, sums;
  rc = upload(view.data(), nch, len, in, nullptr); if (rc) return rc;
  rc = sums.alloc(sizeof(double) * (size_t)nch); if (rc) return rc;
  rc = enqueue_momentary(in.as<float>(), nch, len, sample_rate, sums.as<double>(), nullptr); if (rc) return rc;
  std::vector<double> hs(nch, 0.0);
  WAM_HIP(hipMemcpy(hs.data(), sums.p, sizeof(double) * (size_t)nch, hipMemcpyDeviceToHost));
  double ss = 0;
  for (int c = 0; c < nch; c++) ss += hs[c];
  const double ms = ss / ((double)len * nch);                                       // :167
  *lufs = ms <= 0 ? -INFINITY : -0.691 + 10 * std::log10(ms);                       // :168-170
  return WAM_OK;
}

int wam_lufs_series(const float* const* ch, int nch, int64_t n, double sample_rate, const double* positions, int npos, int momentary,
                    double* lufs) {
  int rc = check_audio(ch, nch, n); if (rc) return rc;
  if (npos < 0 || (npos > 0 && (!positions || !lufs)) || !(sample_rate > 0)) { set_error("invalid series arguments"); return WAM_ERR_INVALID; }
  for (int i = 0; i < npos; i++) if (!(positions[i] >= 0)) { set_error("position %d is negative or NaN", i); return WAM_ERR_INVALID; }
  if (npos == 0) return WAM_OK;
  const int64_t win = (int64_t)std::floor(sample_rate * (momentary ? 400 : 3000) / 1000);       // lufs.js:130-133 / :93-96
  std::vector<int64_t> starts, lens; std::vector<int> where;                                    // windows that are measured at all
  for (int i = 0; i < npos; i++) {
    const double sp = std::floor(positions[i] * sample_rate);
    const int64_t start = sp < 9.0e18 ? (int64_t)sp : INT64_MAX;
    const int64_t len = start < n ? ((n - start < win ? n : start + win) - start) : 0;
    if ((double)len < (double)win * 0.5 || len <= 0) { lufs[i] = -INFINITY; continue; }          // :98-100 / :135-137
    starts.push_back(start); lens.push_back(len); where.push_back(i);
  }
  const int nw = (int)where.size();
  if (nw == 0) return WAM_OK;
  rc = ensure_device(); if (rc) return rc;
  DevBuf in, idx;
  rc = upload(ch, nch, n, in, nullptr); if (rc) return rc;
  rc = idx.alloc(sizeof(int64_t) * 2 * (size_t)nw); if (rc) return rc;
  int64_t* starts_dev = idx.as<int64_t>(); int64_t* lens_dev = starts_dev + nw;
  WAM_HIP(hipMemcpyAsync(starts_dev, starts.data(), sizeof(int64_t) * nw, hipMemcpyHostToDevice, nullptr));
  WAM_HIP(hipMemcpyAsync(lens_dev, lens.data(), sizeof(int64_t) * nw, hipMemcpyHostToDevice, nullptr));
  if (momentary) {
    DevBuf sums;
    rc = sums.alloc(sizeof(double) * (size_t)nw * nch); if (rc) return rc;
    rc = enqueue_momentary_series(in.as<float>(), nch, n, starts_dev, lens_dev, nw, sample_rate, sums.as<double>(), nullptr); if (rc) return rc;
    std::vector<double> hs((size_t)nw * nch);
    WAM_HIP(hipMemcpy(hs.data(), sums.p, sizeof(double) * hs.size(), hipMemcpyDeviceToHost));
    for (int w = 0; w < nw; w++) {
      double ss = 0;
      for (int c = 0; c < nch; c++) ss += hs[(size_t)w * nch + c];
      const double ms = ss / ((double)lens[w] * nch);                                            // :167
      lufs[where[w]] = ms <= 0 ? -INFINITY : -0.691 + 10 * std::log10(ms);                       // :168-170
    }
    return WAM_OK;
  }
  // short-term: full windows as batches of "tracks" through the gated meter; the few clipped by the end one by one.
  // Full windows come first in a stable partition so that their device start indices are contiguous.
  std::vector<int> full, part;
  for (int w = 0; w < nw; w++) (lens[w] == win ? full : part).push_back(w);
  const int kGroup = 256;
  DevBuf batch;
  rc = batch.alloc(sizeof(float) * (size_t)nch * (size_t)win * (size_t)(full.empty() ? 1 : (full.size() < (size_t)kGroup ? full.size() : (size_t)kGroup)));
  if (rc) return rc;
  std::vector<int64_t> gs; std::vector<double> out;
  for (size_t g0 = 0; g0 < full.size(); g0 += kGroup) {
    const int g = (int)(full.size() - g0 < (size_t)kGroup ? full.size() - g0 : (size_t)kGroup);
    gs.resize(g); out.resize(g);
    for (int k = 0; k < g; k++) gs[k] = starts[full[g0 + k]];
    WAM_HIP(hipMemcpy(starts_dev, gs.data(), sizeof(int64_t) * g, hipMemcpyHostToDevice));      // synchronous: gs is reused
    rc = enqueue_gather_windows(in.as<float>(), nch, n, starts_dev, g, win, batch.as<float>(), nullptr); if (rc) return rc;
    rc = meter_dev(batch.as<float>(), g, nch, win, sample_rate, -14.0, out.data(), nullptr, nullptr, nullptr, nullptr, nullptr); if (rc) return rc;   // :118
    for (int k = 0; k < g; k++) lufs[where[full[g0 + k]]] = out[k];
  }
  for (int w : part) {
    WAM_HIP(hipMemcpy(starts_dev, &starts[w], sizeof(int64_t), hipMemcpyHostToDevice));
    rc = enqueue_gather_windows(in.as<float>(), nch, n, starts_dev, 1, lens[w], batch.as<float>(), nullptr); if (rc) return rc;
    rc = meter_dev(batch.as<float>(), 1, nch, lens[w], sample_rate, -14.0, &lufs[where[w]], nullptr, nullptr, nullptr, nullptr, nullptr); if (rc) return rc;
  }
  return WAM_OK;
}

int wam_batch_meter_dev(const float* in_dev, int ntracks, int nch, int64_t n, double sample_rate, double fallback,
                        double* lufs, double* dbtp, void* stream) {
  if (!in_dev || ntracks < 1 || nch < 1 || nch > kMaxMeterChannels || n < 0) { set_error("invalid batch arguments"); return WAM_ERR_INVALID; }
  int rc = ensure_device(); if (rc) return rc;
  return meter_dev(in_dev, ntracks, nch, n, sample_rate, fallback, lufs, dbtp, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

}  // extern "C"
