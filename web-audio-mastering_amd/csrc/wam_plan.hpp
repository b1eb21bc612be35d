// wam_plan.hpp - host-side preparation of filter plans (coefficients in IEEE
// double with the host libm, exactly the expressions of the reference) and the
// scan tables the tile engine needs (wam_device.hpp).
#pragma once
#include <cmath>
#include <cstring>
#include "wam_device.hpp"

namespace wam {

inline void mat2_mul_h(const double a[4], const double b[4], double o[4]) {
  double t[4] = { a[0] * b[0] + a[1] * b[2], a[0] * b[1] + a[1] * b[3], a[2] * b[0] + a[3] * b[2], a[2] * b[1] + a[3] * b[3] };
  std::memcpy(o, t, sizeof(t));
}

// Fill the scan tables of a biquad from its five coefficients: the shift of the state coordinates and the powers of the chunk
// transition there, evaluated in quadruple precision on the host (plan_tables.cpp says why).
extern "C" void wam_biquad_scan_tables(double a1, double a2, int chunk, double Mp[7][3], double* tc);
inline void finish_biquad(Biquad& q) { wam_biquad_scan_tables(q.a1, q.a2, C, q.Mp, &q.tc); }
inline Biquad make_biquad(double b0, double b1, double b2, double a1, double a2) {
  Biquad q; q.b0 = b0; q.b1 = b1; q.b2 = b2; q.a1 = a1; q.a2 = a2; finish_biquad(q); return q;
}
inline Biquad identity_biquad() { return make_biquad(1, 0, 0, 0, 0); }

// web/lib/dsp/utils.js:35-50 calcHighShelfCoeffs
inline Biquad k_high_shelf(double sr, double f, double gain_db, double Q) {
  const double A = std::pow(10, gain_db / 40);
  const double w0 = 2 * M_PI * f / sr;
  const double cw = std::cos(w0), sw = std::sin(w0);
  const double alpha = sw / (2 * Q);
  const double b0 = A * ((A + 1) + (A - 1) * cw + 2 * std::sqrt(A) * alpha);
  const double b1 = -2 * A * ((A - 1) + (A + 1) * cw);
  const double b2 = A * ((A + 1) + (A - 1) * cw - 2 * std::sqrt(A) * alpha);
  const double a0 = (A + 1) - (A - 1) * cw + 2 * std::sqrt(A) * alpha;
  const double a1 = 2 * ((A - 1) - (A + 1) * cw);
  const double a2 = (A + 1) - (A - 1) * cw - 2 * std::sqrt(A) * alpha;
  return make_biquad(b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0);
}
// RBJ low/high-pass: utils.js:59-73, final-filters.js:22-43, multiband-transient.js:137-169,
// multiband-saturation.js:63-96
inline Biquad rbj_pass(double sr, double f, double Q, bool highpass) {
  const double w0 = 2 * M_PI * f / sr;
  const double cw = std::cos(w0), sw = std::sin(w0);
  const double alpha = sw / (2 * Q);
  double b0, b1, b2;
  if (highpass) { b0 = (1 + cw) / 2; b1 = -(1 + cw); b2 = (1 + cw) / 2; }
  else          { b0 = (1 - cw) / 2; b1 = 1 - cw;    b2 = (1 - cw) / 2; }
  const double a0 = 1 + alpha, a1 = -2 * cw, a2 = 1 - alpha;
  return make_biquad(b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0);
}
// dsp-worker.js:1161-1170 (side-channel high-pass of the stereo stage; note the operand order)
inline Biquad stereo_side_hp(double sr, double f) {
  const double om = 2 * M_PI * f / sr;
  const double cs = std::cos(om), sn = std::sin(om);
  const double alpha = sn / (2 * 0.7071);
  return make_biquad(((1 + cs) / 2) / (1 + alpha), (-(1 + cs)) / (1 + alpha), ((1 + cs) / 2) / (1 + alpha),
                     (-2 * cs) / (1 + alpha), (1 - alpha) / (1 + alpha));
}
// exciter.js:15-28 (bilinear Butterworth high-pass)
inline Biquad exciter_hp(double sr, double f) {
  const double omega = std::tan(M_PI * f / sr);
  const double om2 = omega * omega;
  const double nn = 1 / (1 + M_SQRT2 * omega + om2);
  return make_biquad(nn, -2 * nn, nn, 2 * (om2 - 1) * nn, (1 - M_SQRT2 * omega + om2) * nn);
}
// Web Audio BiquadFilterNode formulas (a10; the reference stage cannot run: spec-defined)
inline Biquad webaudio_biquad(int type /*0 lowshelf, 1 peaking, 2 highshelf*/, double sr, double f, double gain_db, double Q) {
  const double A = std::pow(10, gain_db / 40);
  const double w0 = 2 * M_PI * f / sr;
  const double cw = std::cos(w0), sw = std::sin(w0);
  double b0, b1, b2, a0, a1, a2;
  if (type == 0) {
    const double aS = sw / 2 * std::sqrt((A + 1 / A) * (1 / 1.0 - 1) + 2);
    const double k = 2 * aS * std::sqrt(A);
    b0 = A * ((A + 1) - (A - 1) * cw + k); b1 = 2 * A * ((A - 1) - (A + 1) * cw); b2 = A * ((A + 1) - (A - 1) * cw - k);
    a0 = (A + 1) + (A - 1) * cw + k; a1 = -2 * ((A - 1) + (A + 1) * cw); a2 = (A + 1) + (A - 1) * cw - k;
  } else if (type == 2) {
    const double aS = sw / 2 * std::sqrt((A + 1 / A) * (1 / 1.0 - 1) + 2);
    const double k = 2 * aS * std::sqrt(A);
    b0 = A * ((A + 1) + (A - 1) * cw + k); b1 = -2 * A * ((A - 1) + (A + 1) * cw); b2 = A * ((A + 1) + (A - 1) * cw - k);
    a0 = (A + 1) - (A - 1) * cw + k; a1 = 2 * ((A - 1) - (A + 1) * cw); a2 = (A + 1) - (A - 1) * cw - k;
  } else {
    const double aQ = sw / (2 * Q);
    b0 = 1 + aQ * A; b1 = -2 * cw; b2 = 1 - aQ * A; a0 = 1 + aQ / A; a1 = -2 * cw; a2 = 1 - aQ / A;
  }
  return make_biquad(b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0);
}

inline OnePole make_onepole(double c) {
  OnePole q; q.c = c;
  double p = 1;
  for (int n = 0; n < C; n++) p *= c;
  q.Mp[0] = p;
  for (int k = 1; k < 7; k++) q.Mp[k] = q.Mp[k - 1] * q.Mp[k - 1];
  return q;
}

}  // namespace wam
