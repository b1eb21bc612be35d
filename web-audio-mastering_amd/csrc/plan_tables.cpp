// plan_tables.cpp - scan tables of a second-order section, computed on the host in quadruple precision (g++: __float128;
// plain arithmetic only, no libquadmath).
//
// The tile engine (wam_device.hpp) rebuilds a biquad's state at every chunk start from zero-state chunk responses and
// powers of the chunk transition M = A^C, A = [[-a1, -a2], [1, 0]].  For the low-frequency filters of the chain (30 Hz
// high-pass, 80 Hz shelf, 200 Hz crossovers) A has a pole pair rho e^{+-i theta} next to z = 1, the companion basis
// (y[n-1], y[n-2]) is nearly degenerate (|A^n| ~ 1 / sin(theta) = 250 at 30 Hz / 48 kHz), and powers formed by repeated
// squaring in double are off by 1e-9: enough to put 2 % of the stage's f32 stores on the other side of a rounding
// boundary than the reference's sequential loop (tools/flip_where.py), which the mastering soft clipper's knee then
// multiplies by ~600 (tests/test_gpu_wholetrack.py).  The scans therefore run in SHIFTED coordinates
//     u = T s,  T = [[1, -c], [0, 1]],  c = -a1 / 2  (the real part of the poles),
// where T A T^-1 = c I + N, N = [[0, -delta], [1, 0]], delta = a2 - c^2, N^2 = -delta I.  Every power of the transition is
// then a I + b N = [[a, -delta b], [b, a]] - for a pole pair a rotation-scaling (up to the diagonal scale sqrt(delta)) of norm
// <= 1, for a double pole (K-weighting high-pass, Q = 0.5) the Jordan form, for real poles a hyperbolic rotation - whose products
// and weighted sums lose nothing: the chunk start states come out to ~3e-13, the accuracy of the sequential double recursion
// itself.  c is a double (exactly -a1 / 2), so T is exact; a_n, b_n follow from the recurrence of (c I + N)^n evaluated here in
// __float128 and are rounded once; the table holds (a, b, b' = -delta b) per power.
typedef __float128 q128;

// Mp[k] = (a, b, -delta b) of (c I + N)^(chunk 2^k), k = 0..6; *tc = c.
extern "C" void wam_biquad_scan_tables(double a1, double a2, int chunk, double Mp[7][3], double* tc) {
  const double c = -a1 / 2;
  *tc = c;
  const q128 cq = c, delta = (q128)a2 - cq * cq;
  // (a, b) (c I + N) = (a c - delta b) I + (a + b c) N
  q128 a = 1, b = 0;
  for (int i = 0; i < chunk; i++) { const q128 na = a * cq - delta * b, nb = a + b * cq; a = na; b = nb; }
  for (int k = 0; k < 7; k++) {
    Mp[k][0] = (double)a; Mp[k][1] = (double)b; Mp[k][2] = (double)(-delta * b);
    const q128 na = a * a - delta * b * b, nb = 2 * a * b;          // (a I + b N)^2
    a = na; b = nb;
  }
}
