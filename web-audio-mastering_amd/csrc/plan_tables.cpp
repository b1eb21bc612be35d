// plan_tables.cpp - scan tables of a second-order section, computed on the host in quadruple precision (g++: __float128;
// plain arithmetic only, no libquadmath).
//
// The tile engine (wam_device.hpp) rebuilds a biquad's state at every chunk start from zero-state chunk responses and
// powers of the chunk transition M = A^C, A = [[-a1, -a2], [1, 0]].  For the low-frequency filters of the chain (30 Hz
// high-pass, 80 Hz shelf, 200 Hz crossovers) A has a pole pair rho e^{+-i theta} next to z = 1, the companion basis
// (y[n-1], y[n-2]) is nearly degenerate (|A^n| ~ 1 / sin(theta) = 250 at 30 Hz / 48 kHz), and powers formed by repeated
// squaring in double are off by 1e-9: enough to put 2 % of the stage's f32 stores on the other side of a rounding
// boundary than the reference's sequential loop (tools/flip_where.py), which the mastering soft clipper's knee then
// multiplies by ~600 (tests/test_gpu_wholetrack.py).  The scan therefore runs in MODAL coordinates
//     u = T s,  T = [[1, -c], [0, d]],  c = rho cos(theta) = -a1 / 2,  d = rho sin(theta) = sqrt(a2 - c^2),
// where the transition is a rotation-scaling T A^n T^-1 = rho^n [[cos n theta, -sin n theta], [sin n theta, cos n theta]]
// of norm <= 1: products and sums of such matrices lose nothing, and the start states come out to ~3e-13 - the
// accuracy of the sequential double recursion itself.  T is formed from the DOUBLE values of c and d (any invertible T
// gives an exact similarity); the table T A^n T^-1 for those doubles is evaluated here in __float128 and rounded once.
// Sections with real poles (K-weighting high-pass, Q = 0.5: a double pole) keep the companion basis (T = identity), with
// powers that are at least correctly rounded.
#include <cmath>

typedef __float128 q128;

namespace {
struct M2 { q128 a, b, c, d; };
inline M2 mul(const M2& x, const M2& y) { return { x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d }; }
}  // namespace

// Mp[k] = T A^(chunk 2^k) T^-1, k = 0..6, row-major; T3 = { c, d, 1 / d } ({ 0, 1, 1 } when the companion basis is kept).
extern "C" void wam_biquad_scan_tables(double a1, double a2, int chunk, double Mp[7][4], double T3[3]) {
  const double c = -a1 / 2;
  const q128 disc = (q128)a2 - (q128)c * (q128)c;
  double tc = 0, td = 1;
  if (disc > 0) {
    const double d = (double)sqrtl((long double)disc);
    if (d >= 1e-4 && std::isfinite(d)) { tc = c; td = d; }
  }
  T3[0] = tc; T3[1] = td; T3[2] = 1.0 / td;
  const M2 A = { -(q128)a1, -(q128)a2, 1, 0 };
  M2 P = { 1, 0, 0, 1 };
  for (int i = 0; i < chunk; i++) P = mul(A, P);
  const M2 T = { 1, -(q128)tc, 0, (q128)td };
  const M2 Ti = { 1, (q128)tc / (q128)td, 0, (q128)1 / (q128)td };
  for (int k = 0; k < 7; k++) {
    const M2 R = mul(mul(T, P), Ti);
    Mp[k][0] = (double)R.a; Mp[k][1] = (double)R.b; Mp[k][2] = (double)R.c; Mp[k][3] = (double)R.d;
    P = mul(P, P);
  }
}
