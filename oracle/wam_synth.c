/*
 * TEST INFRASTRUCTURE.  "synth v1" deterministic synthetic-track generator,
 * C statement.  Bit-identical to tests/golden/synth.mjs and to the HIP
 * statement (csrc/synth.hip): IEEE double add/mul and u32 integer ops only.
 * Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include "wam_oracle.h"

static inline uint32_t hash32(uint32_t idx, uint32_t ch, uint32_t n) {
  uint32_t h = ((idx + 1u) * 0x9E3779B1u) ^ ((ch + 1u) * 0x85EBCA77u) ^ (n * 0xC2B2AE3Du);
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

static inline double psin(uint32_t phase) {
  const double x = (double)phase * (1.0 / 4294967296.0);
  const double t = x - 0.5;
  const double a = fabs(t);
  const double p = 8 * t * (1 - 2 * a);
  return p + 0.225 * (p * fabs(p) - p);
}

void ora_synth(int idx, int ch, int64_t n0, int64_t count, double sr, const ora_synth_opts* o, float* out) {
  double W[16];
  W[0] = 0.15;
  for (int k = 1; k < 16; k++) W[k] = W[k - 1] * 0.85;
  const double fc = 110.0 * (ch + 1) * (1 + (idx % 7));
  const uint32_t stepc = (uint32_t)floor(fc * 4294967296.0 / sr);
  const uint32_t stepe = (uint32_t)floor(4294967296.0 / (7.3 * sr));
  const int64_t P = (int64_t)floor(sr / 2);
  for (int64_t j = 0; j < count; j++) {
    const int64_t n = n0 + j;
    double lp = 0;
    for (int k = 0; k < 16; k++) {
      const int64_t m = n - k;
      if (m < 0) break;
      const double u = (double)hash32((uint32_t)idx, (uint32_t)ch, (uint32_t)m) * (1.0 / 2147483648.0) - 1.0;
      lp = lp + W[k] * u;
    }
    const double car = psin((uint32_t)n * stepc);
    const double env = 0.25 + 0.2 * psin((uint32_t)n * stepe);
    double click = 0;
    if (o->click_amp != 0) {
      const int64_t k = n % P;
      if (k < 192) {
        const double c = 1 - (double)k / 192;
        click = o->click_amp * (c * c) * (((k >> 2) & 1) ? -1 : 1);
      }
    }
    double g = 1.0;
    if (n >= o->quiet_from && n < o->quiet_to) g = o->quiet_gain;
    if (n >= o->quiet2_from && n < o->quiet2_to) g = o->quiet2_gain;
    const double core = env * (0.6 * car + 2.0 * lp) + click;
    const double t1 = o->amp * g;
    const double t2 = t1 * core;
    out[j] = (float)(o->dc + t2);
  }
}
