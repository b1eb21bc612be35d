// synth.hip - "synth v1" deterministic synthetic-track generator, HIP statement.
// Bit-identical to oracle/wam_synth.c and tests/golden/synth.mjs (IEEE double
// add/mul with contraction off, u32 integer ops).  Used by bench.py and the
// full-size GPU tests so that multi-GB inputs never cross PCIe.  No counterpart
// in the reference (it is a measurement utility, declared in include/wam_synth.h).
#include "wam_common.hpp"
#include "../../include/wam_synth.h"
#include <cmath>

namespace wam {

__device__ __forceinline__ uint32_t hash32(uint32_t idx, uint32_t ch, uint32_t n) {
  uint32_t h = ((idx + 1u) * 0x9E3779B1u) ^ ((ch + 1u) * 0x85EBCA77u) ^ (n * 0xC2B2AE3Du);
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

__device__ __forceinline__ double psin(uint32_t phase) {
#pragma clang fp contract(off)
  const double x = (double)phase * (1.0 / 4294967296.0);
  const double t = x - 0.5;
  const double a = fabs(t);
  const double p = 8 * t * (1 - 2 * a);
  return p + 0.225 * (p * fabs(p) - p);
}

struct SynthArgs {
  double amp, dc, click_amp, quiet_gain, quiet2_gain, amp_step;
  int64_t quiet_from, quiet_to, quiet2_from, quiet2_to, P;
  uint32_t stepe;
  double sr;
  double W[16];
};

__global__ __launch_bounds__(256) void k_synth(float* __restrict__ out, int nch, int64_t n, int idx0, SynthArgs a) {
#pragma clang fp contract(off)
  const int64_t chan = blockIdx.y;
  const int trk = (int)(chan / nch), ch = (int)(chan % nch);
  const int idx = idx0 + trk;
  const double fc = 110.0 * (ch + 1) * (1 + (idx % 7));
  const uint32_t stepc = (uint32_t)floor(fc * 4294967296.0 / a.sr);
  const double amp = a.amp + a.amp_step * trk;
  float* o = out + chan * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double lp = 0;
    for (int k = 0; k < 16; k++) {
      const int64_t m = i - k;
      if (m < 0) break;
      const double u = (double)hash32((uint32_t)idx, (uint32_t)ch, (uint32_t)m) * (1.0 / 2147483648.0) - 1.0;
      lp = lp + a.W[k] * u;
    }
    const double car = psin((uint32_t)i * stepc);
    const double env = 0.25 + 0.2 * psin((uint32_t)i * a.stepe);
    double click = 0;
    if (a.click_amp != 0) {
      const int64_t k = i % a.P;
      if (k < 192) {
        const double c = 1 - (double)k / 192;
        click = a.click_amp * (c * c) * (((k >> 2) & 1) ? -1 : 1);
      }
    }
    double g = 1.0;
    if (i >= a.quiet_from && i < a.quiet_to) g = a.quiet_gain;
    if (i >= a.quiet2_from && i < a.quiet2_to) g = a.quiet2_gain;
    const double core = env * (0.6 * car + 2.0 * lp) + click;
    const double t1 = amp * g;
    const double t2 = t1 * core;
    o[i] = (float)(a.dc + t2);
  }
}

}  // namespace wam

extern "C" int wam_synth_dev(float* out_dev, int ntracks, int nch, int64_t n, double sample_rate, int idx0,
                             const wam_synth_opts* o, double amp_step, void* stream) {
  using namespace wam;
  if (!out_dev || ntracks < 1 || nch < 1 || n < 0 || !o) { set_error("invalid synth arguments"); return WAM_ERR_INVALID; }
  int rc = ensure_device(); if (rc) return rc;
  if (n == 0) return WAM_OK;
  SynthArgs a;
  a.amp = o->amp; a.dc = o->dc; a.click_amp = o->click_amp; a.quiet_gain = o->quiet_gain; a.quiet2_gain = o->quiet2_gain;
  a.amp_step = amp_step;
  a.quiet_from = o->quiet_from; a.quiet_to = o->quiet_to; a.quiet2_from = o->quiet2_from; a.quiet2_to = o->quiet2_to;
  a.P = (int64_t)std::floor(sample_rate / 2);
  a.stepe = (uint32_t)std::floor(4294967296.0 / (7.3 * sample_rate));
  a.sr = sample_rate;
  a.W[0] = 0.15;
  for (int k = 1; k < 16; k++) a.W[k] = a.W[k - 1] * 0.85;
  int64_t bx = (n + 255) / 256; if (bx > 1024) bx = 1024;
  WAM_LAUNCH("k_synth", k_synth, dim3((unsigned)bx, (unsigned)(ntracks * nch)), dim3(256), (hipStream_t)stream, out_dev, nch, n, idx0, a);
  return WAM_OK;
}

// ---- shader clock under f64 load (bench.py: the peak of its VALU roofline is derived from the clock the chip really runs at) ----
namespace wam {
// Every lane runs eight independent chains of v_fma_f64 (the chain kernels' instruction mix) for `iters` rounds; lane 0 of each
// workgroup reports shader-clock ticks (s_memtime) and constant-rate ticks (s_memrealtime) across its loop.
__global__ __launch_bounds__(256) void k_clock_probe(int64_t iters, unsigned long long* __restrict__ ticks, double* __restrict__ sink) {
  double a0 = 1e-9 * threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 0.999999999, c = 1e-12;
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int64_t i = 0; i < iters; i++) {
    a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
    a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
  const double s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (s == 12345.678) sink[0] = s;                       // keeps the chains alive
}
}  // namespace wam

// Runs the probe for about `millis` ms on every CU (2 workgroups of 4 wavefronts per CU: the chain kernels' occupancy) and returns
// the shader clock in MHz it saw: (shader ticks / constant-rate ticks) x the constant rate the device reports.
extern "C" int wam_probe_shader_clock(double millis, double* mhz) {
  using namespace wam;
  if (!mhz || !(millis > 0)) { set_error("invalid probe arguments"); return WAM_ERR_INVALID; }
  int rc = ensure_device(); if (rc) return rc;
  int dev = 0, cus = 0, wall_khz = 0;
  WAM_HIP(hipGetDevice(&dev));
  WAM_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  WAM_HIP(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
  if (cus <= 0 || wall_khz <= 0) { set_error("device reports no CU count / wall clock rate"); return WAM_ERR_HIP; }
  const int blocks = 2 * cus;
  DevBuf tb, sk;
  rc = tb.alloc(sizeof(unsigned long long) * 2 * blocks); if (rc) return rc;
  rc = sk.alloc(sizeof(double)); if (rc) return rc;
  // a wavefront issues one v_fma_f64 per ~4 cycles per chain slot; 2 wavefronts share a SIMD: 8 FMAs per round ~ 64 cycles
  const int64_t iters = (int64_t)(millis * 1e-3 * 2.0e9 / 64.0) + 1;
  std::vector<unsigned long long> h(2 * (size_t)blocks);
  for (int pass = 0; pass < 2; pass++) {                  // the first pass brings the clock up
    hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(256), 0, nullptr, iters, tb.as<unsigned long long>(), sk.as<double>());
    WAM_HIP(hipGetLastError());
    WAM_HIP(hipMemcpy(h.data(), tb.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  }
  double st = 0, wt = 0;
  for (int b = 0; b < blocks; b++) { st += (double)h[2 * b]; wt += (double)h[2 * b + 1]; }
  if (!(wt > 0)) { set_error("clock probe returned no ticks"); return WAM_ERR_HIP; }
  *mhz = st / wt * (double)wall_khz / 1e3;
  return WAM_OK;
}
