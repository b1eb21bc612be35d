// host_copy.hip - what does it cost to move one 3-minute stereo track (2 x 34.56 MB) between PAGEABLE host memory and the device?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/host_copy tools/probe/host_copy.hip -lpthread && /tmp/host_copy
// Variants: plain hipMemcpyAsync from / to pageable memory (the runtime stages it); a resident
// pinned buffer filled by 1 / 2 / 4 / 8 memcpy threads in chunks, each chunk's DMA enqueued as soon as it is filled.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t n = 8640000, bytes = n * sizeof(float);
  std::vector<float> a(n, 1.0f), b(n, 2.0f), oa(n), ob(n);
  float *d = nullptr, *pin = nullptr;
  CK(hipMalloc(&d, 2 * bytes));
  CK(hipHostMalloc(&pin, 2 * bytes, hipHostMallocDefault));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int reps = 5;
  for (int pass = 0; pass < 2; pass++) {             // pass 0 warms everything up
    double t0 = now();
    for (int r = 0; r < reps; r++) { CK(hipMemcpyAsync(d, a.data(), bytes, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(d + n, b.data(), bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
    double t1 = now();
    for (int r = 0; r < reps; r++) { CK(hipMemcpyAsync(oa.data(), d, bytes, hipMemcpyDeviceToHost, s)); CK(hipMemcpyAsync(ob.data(), d + n, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
    double t2 = now();
    if (pass) printf("pageable hipMemcpyAsync      : H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) / reps * 1e3, 2 * bytes / ((t1 - t0) / reps) / 1e9, (t2 - t1) / reps * 1e3, 2 * bytes / ((t2 - t1) / reps) / 1e9);
    // (hipHostRegister of malloc'ed memory is refused on this pool: 'invalid argument' - no pin-in-place variant)
    t0 = now();
    for (int r = 0; r < reps; r++) { CK(hipMemcpyAsync(d, pin, 2 * bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
    t1 = now();
    for (int r = 0; r < reps; r++) { CK(hipMemcpyAsync(pin, d, 2 * bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
    t2 = now();
    if (pass) printf("pinned DMA alone             : H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) / reps * 1e3, 2 * bytes / ((t1 - t0) / reps) / 1e9, (t2 - t1) / reps * 1e3, 2 * bytes / ((t2 - t1) / reps) / 1e9);
    for (int nt : {1, 2, 4, 8}) {
      const size_t chunk = size_t(4) << 20;            // bytes per piece
      const float* src[2] = { a.data(), b.data() };
      float* dst[2] = { oa.data(), ob.data() };
      const size_t pieces_per = (bytes + chunk - 1) / chunk, pieces = 2 * pieces_per;
      t0 = now();
      for (int r = 0; r < reps; r++) {
        // H2D: thread k fills pieces k, k + nt, ...; each piece's DMA is enqueued by the thread that filled it (own stream per thread)
        std::vector<std::thread> th;
        for (int k = 0; k < nt; k++) th.emplace_back([&, k]() {
          hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
          for (size_t p = k; p < pieces; p += nt) {
            const int c = (int)(p / pieces_per); const size_t off = (p % pieces_per) * chunk, len = off + chunk <= bytes ? chunk : bytes - off;
            memcpy((char*)pin + c * bytes + off, (const char*)src[c] + off, len);
            (void)hipMemcpyAsync((char*)d + c * bytes + off, (char*)pin + c * bytes + off, len, hipMemcpyHostToDevice, st);
          }
          (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st);
        });
        for (auto& t : th) t.join();
      }
      t1 = now();
      for (int r = 0; r < reps; r++) {
        std::vector<std::thread> th;
        for (int k = 0; k < nt; k++) th.emplace_back([&, k]() {
          hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
          hipEvent_t ev[64]; int ne = 0;
          std::vector<size_t> mine;
          for (size_t p = k; p < pieces; p += nt) mine.push_back(p);
          for (size_t p : mine) {
            const int c = (int)(p / pieces_per); const size_t off = (p % pieces_per) * chunk, len = off + chunk <= bytes ? chunk : bytes - off;
            (void)hipMemcpyAsync((char*)pin + c * bytes + off, (char*)d + c * bytes + off, len, hipMemcpyDeviceToHost, st);
            (void)hipEventCreateWithFlags(&ev[ne], hipEventDisableTiming); (void)hipEventRecord(ev[ne++], st);
          }
          int e = 0;
          for (size_t p : mine) {
            const int c = (int)(p / pieces_per); const size_t off = (p % pieces_per) * chunk, len = off + chunk <= bytes ? chunk : bytes - off;
            (void)hipEventSynchronize(ev[e]); (void)hipEventDestroy(ev[e++]);
            memcpy((char*)dst[c] + off, (char*)pin + c * bytes + off, len);
          }
          (void)hipStreamDestroy(st);
        });
        for (auto& t : th) t.join();
      }
      t2 = now();
      if (pass) printf("pinned ring, %d copy thread(s): H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", nt, (t1 - t0) / reps * 1e3, 2 * bytes / ((t1 - t0) / reps) / 1e9, (t2 - t1) / reps * 1e3, 2 * bytes / ((t2 - t1) / reps) / 1e9);
    }
  }
  printf("check %f %f\n", oa[123], ob[456]);
  return 0;
}
