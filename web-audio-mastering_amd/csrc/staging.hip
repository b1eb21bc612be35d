// staging.hip - host <-> device copies of the host-pointer entry points through a resident pinned ring.
//
// The callers of the C ABI hand over PAGEABLE memory (Float32Array backing stores, numpy arrays).  hipMemcpyAsync from / to such
// memory is staged by the runtime on the calling thread: 28 GB/s up and 17.5 GB/s down for one 3-minute stereo track on the
// MI355X box, against 57 GB/s for the same DMA from pinned memory (tools/probe/host_copy.hip) - half of the drop-in latency of
// wam_render_chain was copies.  Pinning the caller's pages for the call is not available here
// (hipHostRegister of malloc'ed memory is refused on this pool), so the library keeps pinned rings of 32 slots x 4 MiB per device
// and moves the data through them with a few copy threads: a thread owns every nt-th piece and every nt-th slot, fills a slot
// with memcpy and enqueues its DMA at once (H2D), or keeps its slots' DMAs in flight and empties each as its event fires (D2H) -
// 39 GB/s up, 34 GB/s down with four threads.  All DMAs go to the caller's stream, so they are ordered with the kernels there.
#include "wam_common.hpp"
#include <cstring>
#include <cstdint>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <mutex>
#include <thread>

namespace wam {

namespace {
constexpr size_t kPiece = size_t(4) << 20;
constexpr int kSlots = 32;
struct PinRing { char* base; hipEvent_t ev[kSlots]; int device; };
std::mutex g_ring_mu;
std::vector<PinRing*> g_rings;                     // idle rings (grow-only, like the workspace pool; wam_trim_workspace frees them)

PinRing* ring_acquire() {
  int dev = 0; (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_ring_mu);
    for (size_t i = 0; i < g_rings.size(); i++) if (g_rings[i]->device == dev) { PinRing* r = g_rings[i]; g_rings.erase(g_rings.begin() + i); return r; }
  }
  PinRing* r = new PinRing();
  r->device = dev; r->base = nullptr;
  for (auto& e : r->ev) e = nullptr;
  bool ok = hipHostMalloc((void**)&r->base, kPiece * kSlots, hipHostMallocDefault) == hipSuccess;
  for (int s = 0; ok && s < kSlots; s++) ok = hipEventCreateWithFlags(&r->ev[s], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    for (auto& e : r->ev) if (e) (void)hipEventDestroy(e);
    if (r->base) (void)hipHostFree(r->base);
    delete r;
    return nullptr;
  }
  return r;
}
void ring_release(PinRing* r) { std::lock_guard<std::mutex> lk(g_ring_mu); g_rings.push_back(r); }

struct Piece { char* host; char* dev; size_t len; };

// memcpy with non-temporal stores (SSE2: part of every x86-64).  Neither destination is read again by this core - a pinned slot is
// read by the DMA engine, the caller's output array some time later - so the lines need not be fetched for ownership nor kept in
// the cache: half the memory traffic of the store side of a plain memcpy at these sizes (glibc switches to such stores only above
// a threshold that a 4 MiB piece does not reach when many threads share the cache).
inline void copy_streaming(void* dst, const void* src, size_t len) {
#if defined(__x86_64__) && defined(__SSE2__)
  if (!tuning().stage_nt || len < 4096) { std::memcpy(dst, src, len); return; }
  char* d = (char*)dst; const char* s = (const char*)src;
  const size_t head = (16 - ((uintptr_t)d & 15)) & 15;                 // stores must be 16-byte aligned, loads need not be
  if (head) { std::memcpy(d, s, head); d += head; s += head; len -= head; }
  size_t n64 = len / 64;
  for (; n64; n64--, d += 64, s += 64) {
    const __m128i a = _mm_loadu_si128((const __m128i*)s), b = _mm_loadu_si128((const __m128i*)(s + 16));
    const __m128i c = _mm_loadu_si128((const __m128i*)(s + 32)), e = _mm_loadu_si128((const __m128i*)(s + 48));
    _mm_stream_si128((__m128i*)d, a); _mm_stream_si128((__m128i*)(d + 16), b);
    _mm_stream_si128((__m128i*)(d + 32), c); _mm_stream_si128((__m128i*)(d + 48), e);
  }
  _mm_sfence();
  if (len & 63) std::memcpy(d, s, len & 63);
#else
  std::memcpy(dst, src, len);
#endif
}
}  // namespace

void staging_trim() {
  std::vector<PinRing*> rings;
  { std::lock_guard<std::mutex> lk(g_ring_mu); rings.swap(g_rings); }
  for (PinRing* r : rings) {
    (void)hipSetDevice(r->device);
    for (auto& e : r->ev) (void)hipEventDestroy(e);
    (void)hipHostFree(r->base);
    delete r;
  }
}

// Copies every segment between its host and its device address.  to_device: returns when the host data has left the caller's
// buffers and every DMA is complete (the kernels that follow need all of it anyway); from device: returns when the data is in
// the caller's buffers.  Work already enqueued on `stream` is waited for by the first DMA (stream order).
int staged_copy(const HostSeg* segs, int nsegs, bool to_device, hipStream_t stream) {
  size_t total = 0;
  for (int i = 0; i < nsegs; i++) total += segs[i].bytes;
  if (total == 0) return WAM_OK;
  const Tuning tn = tuning();
  PinRing* ring = (tn.stage_pinned && total >= (size_t(1) << 20)) ? ring_acquire() : nullptr;
  if (!ring) {                                       // small, switched off, or no pinned memory to be had: the runtime's own staging
    for (int i = 0; i < nsegs; i++) {
      if (!segs[i].bytes) continue;
      if (to_device) WAM_HIP(hipMemcpyAsync(segs[i].dev, segs[i].host, segs[i].bytes, hipMemcpyHostToDevice, stream));
      else WAM_HIP(hipMemcpyAsync(segs[i].host, segs[i].dev, segs[i].bytes, hipMemcpyDeviceToHost, stream));
    }
    WAM_HIP(hipStreamSynchronize(stream));
    return WAM_OK;
  }
  std::vector<Piece> pieces;
  for (int i = 0; i < nsegs; i++)
    for (size_t off = 0; off < segs[i].bytes; off += kPiece)
      pieces.push_back(Piece{ (char*)segs[i].host + off, (char*)segs[i].dev + off, segs[i].bytes - off < kPiece ? segs[i].bytes - off : kPiece });
  int nt = (int)(tn.stage_threads > 0 ? tn.stage_threads : 4);
  if (nt > 8) nt = 8;
  if ((size_t)nt > pieces.size()) nt = (int)pieces.size();
  const int device = ring->device;
  std::vector<int> rcs(nt, 0);
  // thread k: pieces k, k + nt, ... through the slots s with s % nt == k (disjoint between the threads)
  auto body = [&](int k) {
    if (k > 0 && hipSetDevice(device) != hipSuccess) { rcs[k] = 1; return; }
    std::vector<int> slots;
    for (int s = k; s < kSlots; s += nt) slots.push_back(s);
    const size_t ns = slots.size();
    std::vector<size_t> mine;
    for (size_t p = k; p < pieces.size(); p += nt) mine.push_back(p);
    if (to_device) {
      for (size_t j = 0; j < mine.size(); j++) {
        const Piece& pc = pieces[mine[j]];
        const int s = slots[j % ns];
        if (j >= ns && hipEventSynchronize(ring->ev[s]) != hipSuccess) { rcs[k] = 1; return; }     // the slot's previous DMA
        copy_streaming(ring->base + (size_t)s * kPiece, pc.host, pc.len);
        if (hipMemcpyAsync(pc.dev, ring->base + (size_t)s * kPiece, pc.len, hipMemcpyHostToDevice, stream) != hipSuccess ||
            hipEventRecord(ring->ev[s], stream) != hipSuccess) { rcs[k] = 1; return; }
      }
    } else {
      size_t issued = 0;
      auto issue = [&]() -> bool {
        const Piece& pc = pieces[mine[issued]];
        const int s = slots[issued % ns];
        issued++;
        return hipMemcpyAsync(ring->base + (size_t)s * kPiece, pc.dev, pc.len, hipMemcpyDeviceToHost, stream) == hipSuccess &&
               hipEventRecord(ring->ev[s], stream) == hipSuccess;
      };
      while (issued < mine.size() && issued < ns) if (!issue()) { rcs[k] = 1; return; }
      for (size_t j = 0; j < mine.size(); j++) {
        const Piece& pc = pieces[mine[j]];
        const int s = slots[j % ns];
        if (hipEventSynchronize(ring->ev[s]) != hipSuccess) { rcs[k] = 1; return; }
        copy_streaming(pc.host, ring->base + (size_t)s * kPiece, pc.len);
        if (issued < mine.size() && !issue()) { rcs[k] = 1; return; }                                // the slot is free again
      }
    }
  };
  std::vector<std::thread> helpers;
  std::vector<int> inline_k;
  for (int k = 1; k < nt; k++) {
    try { helpers.emplace_back(body, k); } catch (...) { inline_k.push_back(k); }                    // (no exception crosses the C ABI)
  }
  body(0);
  for (int k : inline_k) body(k);
  for (auto& t : helpers) t.join();
  const hipError_t e = hipStreamSynchronize(stream);      // every slot idle before the ring goes back
  ring_release(ring);
  for (int k = 0; k < nt; k++) if (rcs[k]) { (void)hipGetLastError(); set_error("staged copy failed (%s)", to_device ? "host to device" : "device to host"); return WAM_ERR_HIP; }
  if (e != hipSuccess) { set_error("hipStreamSynchronize failed after a staged copy: %s", hipGetErrorString(e)); return WAM_ERR_HIP; }
  return WAM_OK;
}

}  // namespace wam
