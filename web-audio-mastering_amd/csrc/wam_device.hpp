// wam_device.hpp - the "tile engine": device-side building blocks shared by every
// kernel of the export chain and the meters (gfx950 only, wave = 64).
//
// Execution model (DESIGN.md "Kernels"): one 256-thread workgroup walks a
// (track, channel-group) along time in tiles of TILE = 256 x C samples, carrying
// all recurrence state from tile to tile in registers / LDS.  Inside a tile every
// thread owns C consecutive samples (64 B of f32 per lane, loaded with 4
// dwordx4), runs the recurrences on its chunk with zero state, the workgroup
// combines the chunk summaries with an associative scan (wave-level shuffles,
// then 4 wave aggregates through LDS) and each thread applies the correction for
// its true start state.  LTI stages (biquads, one-poles) scan 2x2 / 1x1
// state-transition powers; the branchy attack/release followers scan affine maps
// under a decision policy and iterate the policy to its fixed point; the
// release smoothers scan a hysteretic running maximum.
//
// Arithmetic follows the reference's convention: IEEE double for all scalar math,
// one rounding to float where the reference stores into a Float32Array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wam {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int C = 16;                  // samples per thread per tile
constexpr int kTile = kThreads * C;    // 4096 samples

// ---- host-prepared plans ---------------------------------------------------
struct Biquad {            // y[n] = b0 x[n] + b1 x[n-1] + b2 x[n-2] - a1 y[n-1] - a2 y[n-2]
  double b0, b1, b2, a1, a2;
  // The scans run on the state in SHIFTED coordinates u = (y[-1] - tc y[-2], y[-2]), tc = -a1 / 2 = the real part of the
  // poles (plan_tables.cpp).  There the one-sample transition is tc I + N with N = [[0, -delta], [1, 0]], N^2 = -delta I
  // (delta = a2 - tc^2: rho^2 sin^2(theta) for a pole pair rho e^{+-i theta}, 0 for a double pole, negative for real poles), so
  // every power is a I + b N = [[a, -delta b], [b, a]]: a rotation-scaling up to a diagonal scale, whose powers, products and
  // weighted sums lose nothing.  In the companion basis (y[-1], y[-2]) a 30 Hz section has |M^k| ~ 250 with cancellation
  // everywhere, and powers squared up in double put the chunk start states off by 1e-9.  Three numbers per power:
  double tc;
  double Mp[7][3];         // (a, b, -delta b) of the chunk transition (C samples) to the power 2^k, k = 0..6
};
struct OnePole {           // y[n] = c y[n-1] + u[n]
  double c;
  double Mp[7];            // (c^C)^(2^k)
};

// ---- small helpers ---------------------------------------------------------
// Workgroup barrier for data that is exchanged through LDS only.  __syncthreads() fences every address space: the compiler puts
// s_waitcnt vmcnt(0) in front of each one, so a wavefront sits out the latency of the tile's global loads at the first barrier and
// drains its stores of the previous tile there too - memory time added to compute time instead of hidden under it (k_exciter
// reduced to load + store took 9.6 of its 20 ms per 256 tracks).  This one orders LDS only: loads in flight stay in flight
// until their first use, stores retire in the background.  (Nothing in the tile kernels hands data from thread to thread
// through global memory inside a launch; a kernel that does - k_gate's masks - keeps __syncthreads().)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
// The same number as a scalar.  Work that is handed out per WAVEFRONT (frames, segments) must be indexed with this one: derived
// from threadIdx.x the index is a per-lane value to the compiler, and every frame position, loop bound and branch that follows
// from it becomes 64-bit vector arithmetic, exec-mask branches and selects instead of scalar code.
__device__ __forceinline__ int wave_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ float f32r(double v) { return (float)v; }   // the Float32Array store

// Make a plan pointer opaque once per tile: the compiler then reloads plan constants through the
// scalar cache where they are used instead of hoisting hundreds of them out of the tile loop and
// spilling them to VGPR lanes (v_writelane/v_readlane cost VALU issue slots; s_load does not).
// The opaque step also hides where the pointer came from: left generic, every plan constant is then a FLAT vector load (64 lanes
// fetching one address through the texture path, counted on vmcnt AND lgkmcnt, so each forces a full s_waitcnt - 289 of them in
// k_front<2>).  The plans are written by the host before the launch and never by a kernel, so the pointer goes through the
// asm in the constant address space, and the loads behind it are s_load again.
template <class T>
__device__ __forceinline__ const T* reload_each_tile(const T* p) {
  typedef const T __attribute__((address_space(4)))* cptr;
  cptr c = (cptr)p;
  asm volatile("" : "+s"(c));
  return (const T*)c;
}

// num / den for operands in a "safe" range (no overflow / underflow / zero of den, |quotient| moderate): v_rcp_f64 (measured on
// gfx950: 4.6e-8 relative, tools/probe/rcp_probe.hip), ONE Newton step (2.2e-15), one residual correction of the quotient, which
// squares what is left: error <= 1 ulp (the IEEE sequence with div_scale / div_fmas / div_fixup is ~3x the instructions); used
// only where the result is rounded to f32 further down.
__device__ __forceinline__ double div_fast(double num, double den) {
  double y = __builtin_amdgcn_rcp(den);
  y = fma(fma(-den, y, 1.0), y, y);
  const double q = num * y;
  return fma(fma(-den, q, num), y, q);
}

// tanh for |x| <= 1: the [9/8] Pade approximant from the continued fraction x / (1 + x^2 / (3 + x^2 / (5 + ... / 17))), truncation
// error < 2e-17 on that range, evaluated error <= 2.3e-16 - against ~31 issue slots for tanh_fast (range reduction, a degree-12
// exponential, a reciprocal) this is 19.  Callers test the range once per chunk and take tanh_fast otherwise.
__device__ __forceinline__ double tanh_small(double x) {
  const double y = x * x;
  double p = 2.9019636862774117676e-8, q = 1.3058836588248352954e-6;
  p = fma(p, y, 0.000028729440494146376499); q = fma(q, y, 0.00040221216691804927099);
  p = fma(p, y, 0.0039215686274509803922);   q = fma(q, y, 0.027450980392156862745);
  p = fma(p, y, 0.13725490196078431373);     q = fma(q, y, 0.47058823529411764706);
  p = fma(p, y, 1.0);                        q = fma(q, y, 1.0);
  return div_fast(x * p, q);
}

// exp(z) for |z| <= 40, relative error ~1e-15: range reduction by ln 2 (hi / lo), degree-12 Taylor polynomial on |t| <= ln2 / 2.
// tanh_fast and pow_fast are built on it.
__device__ __forceinline__ double exp_fast(double z) {
  const double k = rint(z * 1.4426950408889634);
  double t = fma(-k, 6.93147180369123816490e-01, z);      // ln2 hi / lo
  t = fma(-k, 1.90821492927058770002e-10, t);
  double e = 2.08767569878680989792e-09;                  // 1/12!
  e = fma(e, t, 2.50521083854417187751e-08);
  e = fma(e, t, 2.75573192239858906526e-07);
  e = fma(e, t, 2.75573192239858906526e-06);
  e = fma(e, t, 2.48015873015873015873e-05);
  e = fma(e, t, 1.98412698412698412698e-04);
  e = fma(e, t, 1.38888888888888888889e-03);
  e = fma(e, t, 8.33333333333333333333e-03);
  e = fma(e, t, 4.16666666666666666667e-02);
  e = fma(e, t, 1.66666666666666666667e-01);
  e = fma(e, t, 0.5);
  e = fma(e, t, 1.0);
  e = fma(e, t, 1.0);
  return ldexp(e, (int)k);
}

// tanh in double, |error| < 2e-15 absolute (the callers add the result to a signal that is then
// rounded to f32).  tanh|x| = (1 - t) / (1 + t), t = exp_fast(-2|x|), reciprocal by v_rcp_f64 + two Newton steps.
__device__ __forceinline__ double tanh_fast(double x) {
  const double ax = fabs(x);
  const double z = -2.0 * fmin(ax, 20.0);                 // t < 5e-18 beyond: tanh rounds to 1
  const double t = exp_fast(z);
  const double d = 1.0 + t;
  double y = __builtin_amdgcn_rcp(d);
  y = fma(fma(-d, y, 1.0), y, y);
  y = fma(fma(-d, y, 1.0), y, y);
  const double m = (1.0 - t) * y;
  return copysign(m, x);
}

// sqrt(x) for x = 0 or x in [1e-280, 1e280], error <= 1 ulp: v_rsq_f64 and the Goldschmidt / Newton pair that ocml's sqrt uses,
// without its range scaling and class checks (10 instructions against 16); used where the result is rounded to f32.
__device__ __forceinline__ double sqrt_fast(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  g = fma(fma(-g, g, x), h, g);
  return x > 0 ? g : 0.0;
}

// r^q for r in [2^-100, 2^100], |q ln r| <= 40, relative error ~1e-15: ln r = E ln 2 + 2 atanh((m - 1) / (m + 1)) with the
// mantissa m in [sqrt(1/2), sqrt(2)) (odd series to s^21), then exp_fast.  ~55 instructions against ~250
// for ocml's pow; used where the reference stores the power into a Float32Array and a later stage amplifies any f32-ulp
// difference (the glue gain ahead of the soft clipper).
__device__ __forceinline__ double pow_fast(double r, double q) {
  double m = __builtin_amdgcn_frexp_mant(r);              // [0.5, 1)
  int E = __builtin_amdgcn_frexp_exp(r);
  if (m < 0.70710678118654752) { m *= 2.0; E -= 1; }
  const double s = div_fast(m - 1.0, m + 1.0), s2 = s * s;
  double p = 1.0 / 21.0;
  p = fma(p, s2, 1.0 / 19.0);
  p = fma(p, s2, 1.0 / 17.0);
  p = fma(p, s2, 1.0 / 15.0);
  p = fma(p, s2, 1.0 / 13.0);
  p = fma(p, s2, 1.0 / 11.0);
  p = fma(p, s2, 1.0 / 9.0);
  p = fma(p, s2, 1.0 / 7.0);
  p = fma(p, s2, 1.0 / 5.0);
  p = fma(p, s2, 1.0 / 3.0);
  p = fma(p, s2, 1.0);
  const double ln_r = fma((double)E, 6.93147180559945309417e-01, 2.0 * s * p);
  const double z = q * ln_r;
  return exp_fast(z);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d));
  return v;
}
// DPP moves for the wavefront scans: no LDS round trip (ds_bpermute, which __shfl_up compiles to, takes one per 32 bits) and
// no lane arithmetic; a lane without a source keeps `old`, which the callers set to the identity of their scan.
// row_shr:n = 0x110 + n (within rows of 16 lanes); row_bcast:15 = 0x142 with row mask 0xA (lane 15 of rows 0 / 2 to rows
// 1 / 3); row_bcast:31 = 0x143 with row mask 0xC (lane 31 to rows 2 and 3); wave_shr:1 = 0x138 (lane l - 1).
// One 32-bit DPP move.  update_dpp needs its destination pre-loaded with `old` (a v_mov of the constant before every DPP move);
// when `old` is the constant 0 and no row is masked, bound_ctrl:1 supplies the zero for lanes without a source and the
// pre-load disappears - half of the moves of a scan step on doubles (the low words of 0.0 AND of 1.0 are zero).
template <int CTRL, int RM> __device__ __forceinline__ int dpp_i(int old, int v) {
  if (RM == 0xF && __builtin_constant_p(old) && old == 0) return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
  return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xF, false);
}
template <int CTRL, int RM> __device__ __forceinline__ float dpp_f(float old, float v) {
  return __int_as_float(dpp_i<CTRL, RM>(__float_as_int(old), __float_as_int(v)));
}
template <int CTRL, int RM> __device__ __forceinline__ double dpp_d(double old, double v) {
  return __hiloint2double(dpp_i<CTRL, RM>(__double2hiint(old), __double2hiint(v)), dpp_i<CTRL, RM>(__double2loint(old), __double2loint(v)));
}
// Inclusive scan of affine maps x -> al x + be over the wavefront (lane l: the composition of lanes 0..l, lower lanes
// applied first): Hillis-Steele inside the rows, then the row aggregates by two broadcasts.
template <int CTRL, int RM> __device__ __forceinline__ void affine_step(double& al, double& be) {
  const double ta = dpp_d<CTRL, RM>(1.0, al), tb = dpp_d<CTRL, RM>(0.0, be);
  be = al * tb + be; al = al * ta;
}
__device__ __forceinline__ void wave_scan_affine(double& al, double& be) {
  affine_step<0x111, 0xF>(al, be); affine_step<0x112, 0xF>(al, be); affine_step<0x114, 0xF>(al, be); affine_step<0x118, 0xF>(al, be);
  affine_step<0x142, 0xA>(al, be); affine_step<0x143, 0xC>(al, be);
}

// Sum over the wavefront in a fixed order (deterministic); the total is valid in LANE 63 only.
__device__ __forceinline__ double wave_sum_lane63(double v) {
  v += dpp_d<0x111, 0xF>(0.0, v); v += dpp_d<0x112, 0xF>(0.0, v); v += dpp_d<0x114, 0xF>(0.0, v); v += dpp_d<0x118, 0xF>(0.0, v);
  v += dpp_d<0x142, 0xA>(0.0, v); v += dpp_d<0x143, 0xC>(0.0, v);
  return v;
}

// (a, b, b') of a product of two powers of one transition: (a1 I + b1 N)(a2 I + b2 N), N^2 = -delta I, b' = -delta b
__device__ __forceinline__ void pow3_mul(const double x[3], const double y[3], double o[3]) {
  o[0] = x[0] * y[0] + x[2] * y[1];
  o[1] = x[0] * y[1] + x[1] * y[0];
  o[2] = x[0] * y[2] + x[2] * y[0];
}

// M^lane from the power table (lane < 64)
__device__ __forceinline__ void biquad_lane_power(const Biquad& q, int lane, double P[3]) {
  P[0] = 1; P[1] = 0; P[2] = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (lane & (1 << k)) { double t[3]; pow3_mul(q.Mp[k], P, t); P[0] = t[0]; P[1] = t[1]; P[2] = t[2]; }
  }
}
// The LDS table of a filter's lane powers is component-major, tab[64 c + lane] = (a, b, b')[c] of M^lane: a wavefront's 8-byte
// reads of one component then touch consecutive banks (lane-major made them bank conflicts).  [4][64] doubles are reserved
// per filter by the callers; three rows are used.
__device__ __forceinline__ void biquad_lane_power_table(const Biquad& q, int lane, double* tab) {
  double P[3];
  biquad_lane_power(q, lane, P);
#pragma unroll
  for (int c = 0; c < 3; c++) tab[64 * c + lane] = P[c];
}
__device__ __forceinline__ double onepole_lane_power(const OnePole& q, int lane) {
  double p = 1;
#pragma unroll
  for (int k = 0; k < 6; k++) if (lane & (1 << k)) p *= q.Mp[k];
  return p;
}

// ---- second-order sections ------------------------------------------------------
// One biquad stage over this thread's chunk, in place on x[] (double).  xm1/xm2 = the two inputs before the chunk.
// s0/s1 returns the chunk's start state (y[-1], y[-2]) = the previous thread's last two outputs in double: the next
// cascaded stage takes its xm1/xm2 from them (after the f32 store rounding).  Three phases: feed-forward part in place
// and the recursion from zero state for the chunk's end state only; a scan of the end states; the recursion again from
// the true start state, which reproduces the sequential DF-I evaluation order of utils.js:12-25 given that state.
// The lane powers M^lane ([4][64] doubles per distinct filter, M = the transition over one chunk) and the tile carries
// ([2] doubles per filter instance) live in LDS, not registers: 12 VGPRs per instance that would otherwise be live
// across the whole tile loop, and the table is what the scan's broadcast steps index.  Plds/Slds are written once at
// kernel start (and Slds by thread 0 after each scan); a barrier separates every write from the next read.
struct BiquadMid { double x0, x1, S0, S1; };        // between the two halves: exclusive in-wave state, the tile carry
// First half: feed-forward part in place, zero-state recursion, in-wave scan; lane 63 publishes the wave aggregate in ldsw[0..1].
__device__ __forceinline__ void biquad_pre(double x[C], double xm1, double xm2, const Biquad& q, const double* Plds, const double* Slds,
                                           double* ldsw, BiquadMid& m) {
  const double b0 = q.b0, b1 = q.b1, b2 = q.b2, a1 = q.a1, a2 = q.a2;
  double p1 = 0, p2 = 0, i1 = xm1, i2 = xm2;
#pragma unroll
  for (int n = 0; n < C; n++) {
    const double xi = x[n];
    const double w = b0 * xi + b1 * i1 + b2 * i2;
    i2 = i1; i1 = xi;
    x[n] = w;
    const double y = w - a1 * p1 - a2 * p2;
    p2 = p1; p1 = y;
  }
  double e0 = fma(-q.tc, p2, p1), e1 = p2;           // the chunk's zero-state end state in shifted coordinates (struct Biquad)
  const int lane = lane_id();
  m.S0 = Slds[0]; m.S1 = Slds[1];                    // (modal too) read before the caller's barrier; rewritten after it
  // weighted prefix sum of the end states: inside the rows of 16 lanes by DPP row shifts (uniform powers M^1, M^2, M^4, M^8) ...
#define WAM_SCAN2_ROW(K, CTRL) { const double t0 = dpp_d<CTRL, 0xF>(0.0, e0), t1 = dpp_d<CTRL, 0xF>(0.0, e1); \
    e0 += q.Mp[K][0] * t0 + q.Mp[K][2] * t1; e1 += q.Mp[K][1] * t0 + q.Mp[K][0] * t1; }
  WAM_SCAN2_ROW(0, 0x111) WAM_SCAN2_ROW(1, 0x112) WAM_SCAN2_ROW(2, 0x114) WAM_SCAN2_ROW(3, 0x118)
  {
    // ... then the aggregate of the row(s) below reaches a lane through M^(distance to that aggregate's lane): M^((lane & 15) + 1) for
    // the previous row's lane 15, M^(lane - 31) for lane 31 - both entries of the M^lane table
    const double* Pa = Plds + ((lane & 15) + 1);
    const double ta0 = dpp_d<0x142, 0xA>(0.0, e0), ta1 = dpp_d<0x142, 0xA>(0.0, e1);
    { const double pa = Pa[0], pb = Pa[64], pc = Pa[128]; e0 += pa * ta0 + pc * ta1; e1 += pb * ta0 + pa * ta1; }
    const double* Pb = Plds + (lane >= 32 ? lane - 31 : 0);
    const double tb0 = dpp_d<0x143, 0xC>(0.0, e0), tb1 = dpp_d<0x143, 0xC>(0.0, e1);
    { const double pa = Pb[0], pb = Pb[64], pc = Pb[128]; e0 += pa * tb0 + pc * tb1; e1 += pb * tb0 + pa * tb1; }
  }
  if (lane == 63) { ldsw[0] = e0; ldsw[1] = e1; }
  m.x0 = dpp_d<0x138, 0xF>(0.0, e0); m.x1 = dpp_d<0x138, 0xF>(0.0, e1);
}
// Second half (after a workgroup barrier): the four wave aggregates lds4[2 w .. 2 w + 1] -> this thread's start state, the
// recursion again from it; thread 0 stores the next tile's carry.
__device__ __forceinline__ void biquad_post(double x[C], const Biquad& q, const double* Plds, double* Slds, const double* lds4,
                                            const BiquadMid& m, double& s0, double& s1) {
  const double a1 = q.a1, a2 = q.a2;
  const int lane = lane_id(), wave = wave_id();
  double w0 = m.S0, w1 = m.S1, my0 = m.S0, my1 = m.S1;
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    const double c0 = lds4[2 * w], c1 = lds4[2 * w + 1];
    const double n0 = q.Mp[6][0] * w0 + q.Mp[6][2] * w1 + c0;
    const double n1 = q.Mp[6][1] * w0 + q.Mp[6][0] * w1 + c1;
    w0 = n0; w1 = n1;
    if (w + 1 == wave) { my0 = w0; my1 = w1; }
  }
  if (threadIdx.x == 0) { Slds[0] = w0; Slds[1] = w1; }
  const double* P = Plds + lane;
  const double pa = P[0], pb = P[64], pc = P[128];
  const double u0 = m.x0 + pa * my0 + pc * my1;
  s1 = m.x1 + pb * my0 + pa * my1;                   // back to (y[-1], y[-2]): y[-2] is the second coordinate itself
  s0 = fma(q.tc, s1, u0);
  double p1 = s0, p2 = s1;
#pragma unroll
  for (int n = 0; n < C; n++) {
    const double y = x[n] - a1 * p1 - a2 * p2;
    p2 = p1; p1 = y;
    x[n] = y;
  }
}
__device__ __forceinline__ void biquad_chunk_lds(double x[C], double xm1, double xm2, const Biquad& q, const double* Plds, double* Slds,
                                                 double* lds, double& s0, double& s1) {
  BiquadMid m;
  biquad_pre(x, xm1, xm2, q, Plds, Slds, lds + 2 * wave_id(), m);
  lds_barrier();
  biquad_post(x, q, Plds, Slds, lds, m, s0, s1);
}
// Two independent stages (two filters on one input, or one filter on two channels) under ONE barrier; their dependent
// recursions interleave.  lds: the call site's 16 doubles.
__device__ __forceinline__ void biquad_pair_lds(double xa[C], double xam1, double xam2, const Biquad& qa, const double* Pa, double* Sa,
                                                double xb[C], double xbm1, double xbm2, const Biquad& qb, const double* Pb, double* Sb,
                                                double* lds, double& sa0, double& sa1, double& sb0, double& sb1) {
  BiquadMid ma, mb;
  biquad_pre(xa, xam1, xam2, qa, Pa, Sa, lds + 2 * wave_id(), ma);
  biquad_pre(xb, xbm1, xbm2, qb, Pb, Sb, lds + 8 + 2 * wave_id(), mb);
  lds_barrier();
  biquad_post(xa, qa, Pa, Sa, lds, ma, sa0, sa1);
  biquad_post(xb, qb, Pb, Sb, lds + 8, mb, sb0, sb1);
}

// ---- first-order scan ----------------------------------------------------------
__device__ __forceinline__ void scan1(double& e, const OnePole& q, double Pl, double& S, double* lds) {
  const int lane = lane_id(), wave = wave_id();
  // lane l's chunk is the affine map x -> c^C x + e: the wavefront scan of those maps (rows by DPP, row aggregates by
  // broadcasts) leaves the weighted prefix sum in `be`; its multiplier `al` = c^(C (l + 1)) is what the broadcast steps need
  double al = q.Mp[0];
  wave_scan_affine(al, e);
  if (lane == 63) lds[wave] = e;
  const double x = dpp_d<0x138, 0xF>(0.0, e);
  lds_barrier();
  double w = S, my = S;
#pragma unroll
  for (int k = 0; k < kWaves; k++) {
    w = q.Mp[6] * w + lds[k];
    if (k + 1 == wave) my = w;
  }
  S = w;
  e = x + Pl * my;
}
// y[n] = c y[n-1] + u[n] over the chunk, in place on u[]; returns start state in s.
__device__ __forceinline__ void onepole_chunk(double u[C], const OnePole& q, double Pl, double& S, double* lds, double& s) {
  const double c = q.c;
  double p = 0;
#pragma unroll
  for (int n = 0; n < C; n++) p = c * p + u[n];
  s = p;
  scan1(s, q, Pl, S, lds);
  p = s;
#pragma unroll
  for (int n = 0; n < C; n++) { p = c * p + u[n]; u[n] = p; }
}

// Two independent one-pole sections (one filter, two channels) under one barrier.  lds: 8 doubles.
__device__ __forceinline__ void onepole_pair(double ua[C], double ub[C], const OnePole& q, double Pl, double& Sa, double& Sb, double* lds) {
  const double c = q.c;
  const int lane = lane_id(), wave = wave_id();
  double pa = 0, pb = 0;
#pragma unroll
  for (int n = 0; n < C; n++) { pa = c * pa + ua[n]; pb = c * pb + ub[n]; }
  double ala = q.Mp[0], alb = q.Mp[0];
  wave_scan_affine(ala, pa);
  wave_scan_affine(alb, pb);
  if (lane == 63) { lds[wave] = pa; lds[4 + wave] = pb; }
  const double xa = dpp_d<0x138, 0xF>(0.0, pa), xb = dpp_d<0x138, 0xF>(0.0, pb);
  lds_barrier();
  double wa = Sa, wb = Sb, mya = Sa, myb = Sb;
#pragma unroll
  for (int k = 0; k < kWaves; k++) {
    wa = q.Mp[6] * wa + lds[k]; wb = q.Mp[6] * wb + lds[4 + k];
    if (k + 1 == wave) { mya = wa; myb = wb; }
  }
  Sa = wa; Sb = wb;
  pa = xa + Pl * mya; pb = xb + Pl * myb;
#pragma unroll
  for (int n = 0; n < C; n++) { pa = c * pa + ua[n]; ua[n] = pa; pb = c * pb + ub[n]; ub[n] = pb; }
}

// ---- affine scan (per-thread coefficients), used by the follower policy iteration ----
// Each thread's chunk maps start value e to (al * e + be).  Returns this thread's
// start value given the tile carry S; S advances to the next tile's carry.
template <bool CLAMP1 = false>
__device__ __forceinline__ double scan_affine(double al, double be, double& S, double* lds) {
  const int lane = lane_id(), wave = wave_id();
  wave_scan_affine(al, be);
  if (lane == 63) { lds[2 * wave] = al; lds[2 * wave + 1] = be; }
  const double xa = dpp_d<0x138, 0xF>(1.0, al), xb = dpp_d<0x138, 0xF>(0.0, be);
  lds_barrier();
  double w = S, my = S;
#pragma unroll
  for (int k = 0; k < kWaves; k++) {
    w = lds[2 * k] * w + lds[2 * k + 1];
    if (CLAMP1) w = fmin(w, 1.0);
    if (k + 1 == wave) my = w;
  }
  S = w;
  double r = xa * my + xb;
  // CLAMP1: the recurrence is min(rc cur + (1 - rc), 1): its values never exceed 1, but a composed map can by an ulp,
  // and a start value of 1 + ulp would flip the "g < cur" test of a sample at gain 1.0 back and forth for ever
  if (CLAMP1) r = fmin(r, 1.0);
  return r;
}

// Iteration statistics of the policy iterations (a build with -DWAM_COUNT_ITERS; read with wam_debug_counters):
// [2 site] = solves, [2 site + 1] = iterations; site 0 follower, 1 follower pair, 2 release smoother.
#ifdef WAM_COUNT_ITERS
extern __device__ unsigned long long g_iter_stats[8];
extern __device__ unsigned long long g_iter_hist[4];     // follower pair: solves that took <= 2, 3, 4, > 4 iterations
__device__ __forceinline__ void count_iters(int site, int iters) {
  if (threadIdx.x == 0) { atomicAdd(&g_iter_stats[2 * site], 1ull); atomicAdd(&g_iter_stats[2 * site + 1], (unsigned long long)iters); }
}
#else
__device__ __forceinline__ void count_iters(int, int) {}
#endif

// Workgroup vote: does `p` hold on any thread?  A ballot per wavefront, one flag per wavefront in LDS, ONE barrier
// (HIP's __syncthreads_or makes three, an LDS atomic and ~60 instructions of lane arithmetic).  The flags are double-
// buffered by `parity`, which the caller keeps and starts at 0: two votes on the same buffer then always have another
// barrier between them (inside a solver consecutive votes alternate; between two solver calls lies at least a scan).
__device__ __forceinline__ bool block_any(bool p, int& parity) {
  __shared__ int flags[2][kWaves];
  static_assert(kWaves == 4, "block_any reads the four flags as one int4");
  const bool w = __ballot(p) != 0;
  int* f = flags[parity];
  parity ^= 1;
  if (lane_id() == 0) f[wave_id()] = w;
  lds_barrier();
  const int4 v = *reinterpret_cast<const int4*>(f);
  return (v.x | v.y | v.z | v.w) != 0;
}

// ---- piecewise-affine recurrences: policy iteration ---------------------------------
// A recurrence e[n] = f_n(e[n-1]) whose step is piecewise affine (attack/release
// followers, release smoothers with resets).  Each thread runs its chunk from a
// guessed start value with a functor
//     run(start, al, be, mask)
// that returns the affine map (al, be) its chunk composes to UNDER THE BRANCHES IT
// TOOK and a bitmask of those branches (equal masks must imply equal maps).  The maps are scanned for the start values
// that policy implies; the loop repeats until no thread's mask changes, at which
// point start values, branches and maps are mutually consistent = the sequential
// solution.  Returns this thread's start value; S = tile carry in/out.
// EXACT_GUESS: the caller's first guess is often the solution itself (release smoother on a tile without resets); then the
// first scan's result is compared with it, which ends such a tile after one pass instead of two.
template <bool CLAMP1 = false, bool EXACT_GUESS = false, class Run>
__device__ __forceinline__ double solve_piecewise(Run run, double guess, double& S, double* lds, int max_iter = 260) {
  const double S_in = S;
  double start = guess, prev = guess;
  unsigned pm = 0;
  int it = 0, vp = 0;
  for (; it < max_iter; it++) {
    double al, be;
    unsigned m;
    run(start, al, be, m);
    // Same branches as in the previous pass = same maps = the scan would reproduce the start values this pass ran from:
    // they are the solution and S already holds the carry of that scan.  (The vote is also the barrier that lets lds be
    // rewritten.)
    if (it > 0 && !block_any(m != pm, vp)) break;
    pm = m;
    double Sc = S_in;
    const double ns = scan_affine<CLAMP1>(al, be, Sc, lds);
    S = Sc;
    if (EXACT_GUESS && it == 0 && !block_any(__double_as_longlong(ns) != __double_as_longlong(start), vp)) { start = ns; break; }
    // A comparison that is a tie to the last bit can flip with the ulp its own flip causes: a 2-cycle between two
    // start vectors that are both valid to rounding.  Detected from the 4th pass on (one more vote), then ended.
    if (it >= 3 && !block_any(__double_as_longlong(ns) != __double_as_longlong(prev), vp)) { start = ns; break; }
    prev = start;
    start = ns;
  }
  count_iters(CLAMP1 ? 2 : 0, it + 1);
  return start;
}

// Multiplier table of a follower's chunk maps: tab[p] = A^p R^(C-p), p = 0..C attack steps among the chunk's C (LDS; written by
// threads 0..C of the workgroup, a barrier before the first use is the caller's).
__device__ __forceinline__ void follower_al_table(double A, double R, double* tab) {
  const int p = threadIdx.x;
  if (p <= C) { double v = 1; for (int i = 0; i < C; i++) v *= (i < p ? A : R); tab[p] = v; }
}

// Branchy attack/release follower (dsp-worker.js:1308-1312, :345-349;
// multiband-transient.js:219-230): e' = (a > e) ? A e + (1-A) a : R e + (1-R) a.
// a[] = rectified input of the chunk.  Returns the chunk's true start value; the envelope itself is materialised by the caller
// with the reference's own expression (follower_step).  Tab = follower_al_table(A, R) in LDS.
// The same scheme as the pair solver below, for one follower: two passes in single precision that only have to get the branch
// pattern roughly right (no convergence vote), then double-precision passes in max-form - e' = a + max(A d, R d), d = e - a, no
// compare and no 64-bit select; the pattern bit is the sign of d, the chunk map's multiplier follows from the number of attack
// steps through the table - until the pattern reproduces itself: the answer is the sequential one whatever the first guess was.
// (Round 2's generic policy iteration with per-step selects took 3.4-4.3 double passes of 8 instructions per step here.)
template <class T>
__device__ __forceinline__ double follower_solve(const T& a, double A, double R, double& S, const double* Tab, double* lds) {
  const int lane = lane_id(), wave = wave_id();
  const double S_in = S;
  double s = S;
#ifndef WAM_NO_F32_PRESOLVE
  {
    const float Af = (float)A, Rf = (float)R, Sin = (float)S;
    float st = Sin;
#pragma unroll 1
    for (int pre = 0; pre < 2; pre++) {
      float al = 1.0f, be = 0.0f;
#pragma unroll
      for (int n = 0; n < C; n++) {
        const float an = (float)a[n];
        const float e = al * st + be;
        const float c = (an > e) ? Af : Rf;
        be = c * (be - an) + an;
        al = c * al;
      }
#define WAM_PRE1_STEP(CTRL, RM) { const float ta = dpp_f<CTRL, RM>(1.0f, al), tb = dpp_f<CTRL, RM>(0.0f, be); be = al * tb + be; al = al * ta; }
      WAM_PRE1_STEP(0x111, 0xF) WAM_PRE1_STEP(0x112, 0xF) WAM_PRE1_STEP(0x114, 0xF) WAM_PRE1_STEP(0x118, 0xF)
      WAM_PRE1_STEP(0x142, 0xA) WAM_PRE1_STEP(0x143, 0xC)
#undef WAM_PRE1_STEP
      float* ldsf = (float*)lds + 8 * (pre & 1);                        // two regions of 8 floats in the slot, one per pass
      if (lane == 63) { ldsf[2 * wave] = al; ldsf[2 * wave + 1] = be; }
      const float xa = dpp_f<0x138, 0xF>(1.0f, al), xb = dpp_f<0x138, 0xF>(0.0f, be);
      lds_barrier();
      float w = Sin, my = Sin;
#pragma unroll
      for (int k = 0; k < kWaves; k++) {
        w = ldsf[2 * k] * w + ldsf[2 * k + 1];
        if (k + 1 == wave) my = w;
      }
      st = xa * my + xb;
    }
    lds_barrier();                                              // the loop below rewrites the slot
    s = (double)st;
  }
#endif
  double prev = s;
  unsigned pm = 0;
  const double sg = R >= A ? 1.0 : -1.0, cA = sg * A, cR = sg * R;
  int it = 0, vp = 0;
  for (; it < 260; it++) {
    double e = s;
    unsigned m = 0;
#pragma unroll
    for (int n = 0; n < C; n++) {
      const double an = (double)a[n];
      const double d = e - an;
      m = __builtin_amdgcn_alignbit(m, (unsigned)__double2hiint(d), 31);         // (m << 1) | (d < 0)
      e = fma(sg, fmax(cA * d, cR * d), an);
    }
    // same pattern as in the previous pass = same map = the scan would reproduce the start values: they are the solution and S
    // holds the carry of that scan (the vote is also the barrier that lets lds be rewritten)
    if (it > 0 && !block_any(m != pm, vp)) break;
    pm = m;
    double al = Tab[__builtin_popcount(m)];
    double be = fma(-al, s, e);
    wave_scan_affine(al, be);
    if (lane == 63) { lds[2 * wave] = al; lds[2 * wave + 1] = be; }
    const double xa = dpp_d<0x138, 0xF>(1.0, al), xb = dpp_d<0x138, 0xF>(0.0, be);
    lds_barrier();
    double w = S_in, my = S_in;
#pragma unroll
    for (int k = 0; k < kWaves; k++) {
      w = lds[2 * k] * w + lds[2 * k + 1];
      if (k + 1 == wave) my = w;
    }
    const double ns = xa * my + xb;
    S = w;
    // a tie to the last bit can flip with the ulp its own flip causes (a 2-cycle between two start vectors that are both valid to
    // rounding): detected from the 4th pass on (one more vote), then ended
    if (it >= 3 && !block_any(__double_as_longlong(ns) != __double_as_longlong(prev), vp)) { s = ns; break; }
    prev = s;
    s = ns;
  }
  count_iters(0, it + 1);
  return s;
}
// Two followers over the same input (the fast and slow envelopes of the transient shaper), solved in one iteration
// loop: their runs and scans interleave (instruction-level parallelism for the dependent chains) and they share the
// barriers; the loop ends when both start values reproduce themselves.  lds: 16 doubles.
// a: the rectified input of the chunk, anything indexable that yields f32 values (an array, or a view that rectifies on the fly)
template <class T>
__device__ __forceinline__ void follower_solve2(const T& a, double A1, double R1, double& S1, double A2, double R2, double& S2,
                                                const double* T1, const double* T2, double* lds, double& start1, double& start2) {
  const int lane = lane_id(), wave = wave_id();
  const double S1_in = S1, S2_in = S2;
  double s1 = S1, s2 = S2;
#ifndef WAM_PRESOLVE_PASSES
#define WAM_PRESOLVE_PASSES 2
#endif
#ifndef WAM_NO_F32_PRESOLVE
  // The first passes only have to get the branch pattern roughly right: they run in packed single precision (both
  // followers in one v_pk_* instruction, half the shuffles) and need no convergence vote.  The double-precision loop below
  // starts from their result and still ends only when its own start values reproduce themselves bit for bit, so the
  // answer is unchanged.
  {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f A = { (float)A1, (float)A2 }, R = { (float)R1, (float)R2 }, Sin = { (float)S1, (float)S2 };
    v2f st = Sin;
#pragma unroll 1
    for (int pre = 0; pre < WAM_PRESOLVE_PASSES; pre++) {
      v2f al = { 1.0f, 1.0f }, be = { 0.0f, 0.0f };
#pragma unroll
      for (int n = 0; n < C; n++) {
        const float anf = (float)a[n];
        const v2f an = { anf, anf };
        const v2f e = al * st + be;
        v2f c;
        c.x = (anf > e.x) ? A.x : R.x;
        c.y = (anf > e.y) ? A.y : R.y;
        be = c * (be - an) + an;
        al = c * al;
      }
#define WAM_PRE_STEP(CTRL, RM) { v2f ta, tb; ta.x = dpp_f<CTRL, RM>(1.0f, al.x); ta.y = dpp_f<CTRL, RM>(1.0f, al.y); \
        tb.x = dpp_f<CTRL, RM>(0.0f, be.x); tb.y = dpp_f<CTRL, RM>(0.0f, be.y); be = al * tb + be; al = al * ta; }
      WAM_PRE_STEP(0x111, 0xF) WAM_PRE_STEP(0x112, 0xF) WAM_PRE_STEP(0x114, 0xF) WAM_PRE_STEP(0x118, 0xF)
      WAM_PRE_STEP(0x142, 0xA) WAM_PRE_STEP(0x143, 0xC)
      float* ldsf = (float*)lds + 16 * (pre & 1);                       // the slot's 16 doubles = two regions of 16 floats, one per pass
      if (lane == 63) { ldsf[4 * wave] = al.x; ldsf[4 * wave + 1] = al.y; ldsf[4 * wave + 2] = be.x; ldsf[4 * wave + 3] = be.y; }
      v2f xa, xb;
      xa.x = dpp_f<0x138, 0xF>(1.0f, al.x); xa.y = dpp_f<0x138, 0xF>(1.0f, al.y);
      xb.x = dpp_f<0x138, 0xF>(0.0f, be.x); xb.y = dpp_f<0x138, 0xF>(0.0f, be.y);
      lds_barrier();
      v2f w = Sin, my = Sin;
#pragma unroll
      for (int k = 0; k < kWaves; k++) {
        const v2f wa = { ldsf[4 * k], ldsf[4 * k + 1] }, wb = { ldsf[4 * k + 2], ldsf[4 * k + 3] };
        w = wa * w + wb;
        if (k + 1 == wave) my = w;
      }
      st = xa * my + xb;
    }
    lds_barrier();                                              // the loop below rewrites the slot
    s1 = (double)st.x; s2 = (double)st.y;
  }
#endif
  double p1 = s1, p2 = s2;
  unsigned pm1 = 0, pm2 = 0;                                      // branch patterns of the previous pass, one bit per sample
  // The step e' = (a > e) ? A e + (1 - A) a : R e + (1 - R) a is a + max(A d, R d) with d = e - a when the release is the slower
  // of the two (R >= A; min otherwise): no compare and no 64-bit select in the pass that only has to find the branch pattern
  // and the chunk's end value.  The pattern bit IS the sign of d (one v_alignbit per step), the chunk map's multiplier follows
  // from the number of attack steps alone (T[p] = A^p R^(C-p), a 17-entry LDS table per follower) and its offset from the end
  // value.  The envelope itself is materialised by the caller with the reference's own expression (follower_step).
  // (min(x, y) = -max(-x, -y): the sign rides on the two coefficients and on the final fma, so both orders cost the same)
  const double sg1 = R1 >= A1 ? 1.0 : -1.0, sg2 = R2 >= A2 ? 1.0 : -1.0;
  const double cA1 = sg1 * A1, cR1 = sg1 * R1, cA2 = sg2 * A2, cR2 = sg2 * R2;
  int it = 0, vp = 0;
  for (; it < 260; it++) {
    double e1 = s1, e2 = s2;
    unsigned m1 = 0, m2 = 0;
#pragma unroll
    for (int n = 0; n < C; n++) {
      const double an = (double)a[n];
      const double d1 = e1 - an, d2 = e2 - an;
      m1 = __builtin_amdgcn_alignbit(m1, (unsigned)__double2hiint(d1), 31);      // (m << 1) | (d < 0)
      m2 = __builtin_amdgcn_alignbit(m2, (unsigned)__double2hiint(d2), 31);
      e1 = fma(sg1, fmax(cA1 * d1, cR1 * d1), an);
      e2 = fma(sg2, fmax(cA2 * d2, cR2 * d2), an);
    }
    // Same pattern as in the previous pass = same maps = the scan would reproduce the start values this pass ran from:
    // they are the solution, and S1 / S2 already hold the carries of that scan.  (Also the barrier that lets lds be rewritten.)
    if (it > 0 && !block_any(m1 != pm1 || m2 != pm2, vp)) break;
    pm1 = m1; pm2 = m2;
    double al1 = T1[__builtin_popcount(m1)], al2 = T2[__builtin_popcount(m2)];
    double be1 = fma(-al1, s1, e1), be2 = fma(-al2, s2, e2);
#define WAM_PAIR_STEP(CTRL, RM) { affine_step<CTRL, RM>(al1, be1); affine_step<CTRL, RM>(al2, be2); }
    WAM_PAIR_STEP(0x111, 0xF) WAM_PAIR_STEP(0x112, 0xF) WAM_PAIR_STEP(0x114, 0xF) WAM_PAIR_STEP(0x118, 0xF)
    WAM_PAIR_STEP(0x142, 0xA) WAM_PAIR_STEP(0x143, 0xC)
    if (lane == 63) { lds[4 * wave] = al1; lds[4 * wave + 1] = be1; lds[4 * wave + 2] = al2; lds[4 * wave + 3] = be2; }
    const double xa1 = dpp_d<0x138, 0xF>(1.0, al1), xb1 = dpp_d<0x138, 0xF>(0.0, be1);
    const double xa2 = dpp_d<0x138, 0xF>(1.0, al2), xb2 = dpp_d<0x138, 0xF>(0.0, be2);
    lds_barrier();
    double w1 = S1_in, w2 = S2_in, my1 = S1_in, my2 = S2_in;
#pragma unroll
    for (int k = 0; k < kWaves; k++) {
      w1 = lds[4 * k] * w1 + lds[4 * k + 1];
      w2 = lds[4 * k + 2] * w2 + lds[4 * k + 3];
      if (k + 1 == wave) { my1 = w1; my2 = w2; }
    }
    const double n1 = xa1 * my1 + xb1, n2 = xa2 * my2 + xb2;
    S1 = w1; S2 = w2;
    // A comparison that is a tie to the last bit can flip with the ulp its own flip causes: a 2-cycle between two start
    // vectors that are both valid to rounding.  Detected from the 4th pass on (one vote), then ended.
    if (it >= 3 && !block_any(__double_as_longlong(n1) != __double_as_longlong(p1) || __double_as_longlong(n2) != __double_as_longlong(p2), vp)) {
      s1 = n1; s2 = n2; break;
    }
    p1 = s1; p2 = s2;
    s1 = n1; s2 = n2;
  }
  count_iters(1, it + 1);
#ifdef WAM_COUNT_ITERS
  if (threadIdx.x == 0) { const int h = it + 1 <= 2 ? 0 : (it + 1 == 3 ? 1 : (it + 1 == 4 ? 2 : 3)); atomicAdd(&g_iter_hist[h], 1ull); }
#endif
  start1 = s1; start2 = s2;
}

// one follower step, for the final pass that materialises the envelope
__device__ __forceinline__ double follower_step(double e, double a, double A, double R) {
  return (a > e) ? A * e + (1 - A) * a : R * e + (1 - R) * a;
}

// Release smoother with resets (dsp-worker.js:416-425, soft-clipper.js:222-231):
//   cur' = (g < cur) ? g : min(rc cur + (1 - rc), 1).
// g[] = this chunk's raw gains (f32 values).  rcC = rc^C.  Returns the chunk's start value.
__device__ __forceinline__ double release_solve(const float g[C], double rc, double rcC, double& S, double* lds) {
  const double orc = 1 - rc;
  // (mask: which samples reset.  After the first reset the walk no longer depends on the start value, so equal masks
  // give equal maps.)
  auto run = [&](double cur, double& al, double& be, unsigned& m) {
    m = 0;
#pragma unroll
    for (int n = 0; n < C; n++) {
      const double gn = (double)g[n];
      const bool rs = gn < cur;
      m = (m << 1) | (unsigned)rs;
      if (rs) cur = gn;
      else { cur = rc * cur + orc * 1.0; cur = fmin(cur, 1.0); }
    }
    if (m) { al = 0; be = cur; } else { al = rcC; be = 1 - rcC; }
  };
  // first guess: pure release from the carry (exact when the tile has no reset)
  double guess = S;
  {
    // rc^(C*t) for this thread = rcC^t, by square-and-multiply on the fly
    double p = 1, b = rcC; int t = threadIdx.x;
    while (t) { if (t & 1) p *= b; b *= b; t >>= 1; }
    guess = 1 - p * (1 - S);
  }
  return solve_piecewise<true, true>(run, guess, S, lds);
}
__device__ __forceinline__ double release_step(double cur, double g, double rc) {
  if (g < cur) return g;
  return fmin(rc * cur + (1 - rc) * 1.0, 1.0);
}

// ---- loads / stores ----------------------------------------------------------
// Load this thread's C samples of tile `tile` from p[0..n) (zeros past the end).
__device__ __forceinline__ void load_chunk(const float* __restrict__ p, int64_t n, int64_t base, float v[C]) {
  if (base + C <= n && ((reinterpret_cast<uintptr_t>(p + base) & 15) == 0)) {
    const float4* q = reinterpret_cast<const float4*>(p + base);
#pragma unroll
    for (int k = 0; k < C / 4; k++) { const float4 t = q[k]; v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w; }
  } else {
#pragma unroll
    for (int k = 0; k < C; k++) v[k] = (base + k < n) ? p[base + k] : 0.0f;
  }
}
__device__ __forceinline__ void store_chunk(float* __restrict__ p, int64_t n, int64_t base, const float v[C]) {
  if (base + C <= n && ((reinterpret_cast<uintptr_t>(p + base) & 15) == 0)) {
    float4* q = reinterpret_cast<float4*>(p + base);
#pragma unroll
    for (int k = 0; k < C / 4; k++) q[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < C; k++) if (base + k < n) p[base + k] = v[k];
  }
}

// Previous-thread context: every thread publishes its last K values, thread t
// reads those of t-1; thread 0 reads the previous tile's thread 255 (zeros on
// the first tile).  buf holds 2 x (kThreads + 1) x K floats (double-buffered by tile parity).
template <int K>
__device__ __forceinline__ void prev_context(const float last[K], float prev[K], float* buf, int parity, bool first_tile) {
  float* cur = buf + parity * (kThreads + 1) * K;
  const float* old = buf + (parity ^ 1) * (kThreads + 1) * K;
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; k++) cur[(t + 1) * K + k] = last[k];
  lds_barrier();
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < K; k++) prev[k] = first_tile ? 0.0f : old[kThreads * K + k];
  } else {
#pragma unroll
    for (int k = 0; k < K; k++) prev[k] = cur[t * K + k];
  }
}

// Same exchange for kernels with several call sites per tile: neighbours inside a
// wave come by shuffle, wave edges and the tile carry through a site-private LDS
// record of 2 x kWaves x K floats (double-buffered by tile parity: one barrier per call).
template <int K>
__device__ __forceinline__ void prev_ctx(const float last[K], float prev[K], float* site, int parity, bool first_tile) {
  const int lane = lane_id(), wave = wave_id();
  float* cur = site + parity * kWaves * K;
  const float* old = site + (parity ^ 1) * kWaves * K;
#pragma unroll
  for (int k = 0; k < K; k++) prev[k] = dpp_f<0x138, 0xF>(0.0f, last[k]);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < K; k++) cur[wave * K + k] = last[k];
  }
  lds_barrier();
  if (lane == 0) {
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < K; k++) prev[k] = first_tile ? 0.0f : old[(kWaves - 1) * K + k];
    } else {
#pragma unroll
      for (int k = 0; k < K; k++) prev[k] = cur[(wave - 1) * K + k];
    }
  }
}
// Two sites (two channels) under one barrier.
template <int K>
__device__ __forceinline__ void prev_ctx2(const float lastA[K], float prevA[K], float* siteA, const float lastB[K], float prevB[K], float* siteB,
                                          int parity, bool first_tile) {
  const int lane = lane_id(), wave = wave_id();
  float* curA = siteA + parity * kWaves * K; const float* oldA = siteA + (parity ^ 1) * kWaves * K;
  float* curB = siteB + parity * kWaves * K; const float* oldB = siteB + (parity ^ 1) * kWaves * K;
#pragma unroll
  for (int k = 0; k < K; k++) { prevA[k] = dpp_f<0x138, 0xF>(0.0f, lastA[k]); prevB[k] = dpp_f<0x138, 0xF>(0.0f, lastB[k]); }
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < K; k++) { curA[wave * K + k] = lastA[k]; curB[wave * K + k] = lastB[k]; }
  }
  lds_barrier();
  if (lane == 0) {
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < K; k++) { prevA[k] = first_tile ? 0.0f : oldA[(kWaves - 1) * K + k]; prevB[k] = first_tile ? 0.0f : oldB[(kWaves - 1) * K + k]; }
    } else {
#pragma unroll
      for (int k = 0; k < K; k++) { prevA[k] = curA[(wave - 1) * K + k]; prevB[k] = curB[(wave - 1) * K + k]; }
    }
  }
}
constexpr int kCtxFloats = 2 * kWaves * 3;   // per site, K <= 3

// true-peak.js:13-38: max(|y2|, |p(1/4)|, |p(1/2)|, |p(3/4)|) of the Catmull-Rom segment y1..y2
__device__ __forceinline__ double true_peak4(double y0, double y1, double y2, double y3) {
  const double a0 = -0.5 * y0 + 1.5 * y1 - 1.5 * y2 + 0.5 * y3;
  const double a1 = y0 - 2.5 * y1 + 2 * y2 - 0.5 * y3;
  const double a2 = -0.5 * y0 + 0.5 * y2;
  double pk = fabs(y2);
  pk = fmax(pk, fabs(((a0 * 0.25 + a1) * 0.25 + a2) * 0.25 + y1));
  pk = fmax(pk, fabs(((a0 * 0.5 + a1) * 0.5 + a2) * 0.5 + y1));
  pk = fmax(pk, fabs(((a0 * 0.75 + a1) * 0.75 + a2) * 0.75 + y1));
  return pk;
}
// the three interpolated points only, in interpolateCatmullRom's form (utils.js:143-153; limiter.js:107-116)
__device__ __forceinline__ double interp_peak3(double y0, double y1, double y2, double y3) {
  const double a0 = -0.5 * y0 + 1.5 * y1 - 1.5 * y2 + 0.5 * y3;
  const double a1 = y0 - 2.5 * y1 + 2 * y2 - 0.5 * y3;
  const double a2 = -0.5 * y0 + 0.5 * y2;
  double pk = fabs(a0 * 0.015625 + a1 * 0.0625 + a2 * 0.25 + y1);
  pk = fmax(pk, fabs(a0 * 0.125 + a1 * 0.25 + a2 * 0.5 + y1));
  pk = fmax(pk, fabs(a0 * 0.421875 + a1 * 0.5625 + a2 * 0.75 + y1));
  return pk;
}
__device__ __forceinline__ double catmull_half(double y0, double y1, double y2, double y3) {   // utils.js:143-153 at t = 0.5
  const double a0 = -0.5 * y0 + 1.5 * y1 - 1.5 * y2 + 0.5 * y3;
  const double a1 = y0 - 2.5 * y1 + 2 * y2 - 0.5 * y3;
  const double a2 = -0.5 * y0 + 0.5 * y2;
  return a0 * 0.125 + a1 * 0.25 + a2 * 0.5 + y1;
}

// Two alternating LDS scratch slots for consecutive scans: scan k+2 may overwrite the
// slot of scan k because every thread passed scan k+1's barrier after reading it.
struct LdsSlots {
  double* base; int flip;
  __device__ __forceinline__ explicit LdsSlots(double* b) : base(b), flip(0) {}
  __device__ __forceinline__ double* next() { flip ^= 1; return base + flip * 16; }
};
constexpr int kSlotDoubles = 32;

}  // namespace wam
