"""Host logic of the biquad scans (csrc/plan_tables.cpp): the shift of the state coordinates and the powers of the chunk transition there, checked
against exact rational arithmetic.  The tables are what the start state of every 16-sample chunk is rebuilt from; off by 1e-9
(repeated squaring in double, companion basis) they put 2 % of a 30 Hz high-pass's f32 stores on the other side of a rounding
boundary than the reference's sequential loop, and the mastering soft clipper multiplies such a flip by ~600
(tests/test_gpu_wholetrack.py).  No GPU needed: the function is host code of libwam_hip.so."""
import ctypes as C
import math
import os
from fractions import Fraction as F

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "web-audio-mastering_amd", "libwam_hip.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="libwam_hip.so not built")
CHUNK = 16


def tables(a1, a2):
    lib = C.CDLL(SO)
    mp = ((C.c_double * 3) * 7)()
    tc = C.c_double(0)
    lib.wam_biquad_scan_tables(C.c_double(a1), C.c_double(a2), C.c_int(CHUNK), mp, C.byref(tc))
    return np.array([[v for v in row] for row in mp]), tc.value


def rbj_poles(f, q, sr):          # denominator of the RBJ low / high-pass (utils.js:59-73, final-filters.js:22-43)
    w0 = 2 * math.pi * f / sr
    alpha = math.sin(w0) / (2 * q)
    return -2 * math.cos(w0) / (1 + alpha), (1 - alpha) / (1 + alpha)


def mat_mul(x, y):
    return [[x[0][0] * y[0][0] + x[0][1] * y[1][0], x[0][0] * y[0][1] + x[0][1] * y[1][1]],
            [x[1][0] * y[0][0] + x[1][1] * y[1][0], x[1][0] * y[0][1] + x[1][1] * y[1][1]]]


@pytest.mark.parametrize("f,q,sr", [(30, 0.707, 48000), (30, 0.707, 96000), (80, 0.7071, 48000), (200, 0.707, 44100), (4000, 0.707, 48000),
                                    (38.13547087602444, 0.5, 48000), (100, 0.3, 48000)])
def test_tables_are_correctly_rounded(f, q, sr):
    """Pole pairs, the double pole of the K-weighting high-pass (Q = 0.5) and real poles (Q = 0.3): T A^n T^-1 with
    T = [[1, -c], [0, 1]] in exact rational arithmetic has the form [[a, -delta b], [b, a]], and the table is its correct rounding."""
    a1, a2 = rbj_poles(f, q, sr)
    mp, tc = tables(a1, a2)
    assert tc == -a1 / 2
    delta = F(a2) - F(tc) * F(tc)
    A = [[-F(a1), -F(a2)], [F(1), F(0)]]
    P = [[F(1), F(0)], [F(0), F(1)]]
    for _ in range(CHUNK):
        P = mat_mul(A, P)
    T = [[F(1), -F(tc)], [F(0), F(1)]]
    Ti = [[F(1), F(tc)], [F(0), F(1)]]
    for k in range(7):
        R = mat_mul(mat_mul(T, P), Ti)
        assert R[0][0] == R[1][1] and R[0][1] == -delta * R[1][0]                     # a I + b N, exactly
        want = [float(R[0][0]), float(R[1][0]), float(R[0][1])]                       # float(Fraction) rounds correctly
        assert list(mp[k]) == want, (k, list(mp[k]), want)
        if delta > 0:                                                                 # a pole pair: rho^n (cos, sin / d): nothing to cancel
            assert mp[k][0] ** 2 + float(delta) * mp[k][1] ** 2 <= 1.0 + 1e-12
        P = mat_mul(P, P)


def test_modal_start_states_match_the_sequential_recursion():
    """The scan the kernels run (zero-state chunk responses, combined through the table) against the sequential DF-I loop of
    utils.js:12-25 in double, on the 30 Hz high-pass: start states of full-scale white noise to 1e-10
    (with round 2's companion-basis tables, squared up in double, the start states of the synthetic master were off by up to 3.5e-9)."""
    sr, n = 48000, 4096 * 8
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, n) * 2.5
    w0 = 2 * math.pi * 30 / sr
    cw, alpha = math.cos(w0), math.sin(w0) / (2 * 0.707)
    b0, b1, b2, a1, a2 = ((1 + cw) / 2) / (1 + alpha), -(1 + cw) / (1 + alpha), ((1 + cw) / 2) / (1 + alpha), -2 * cw / (1 + alpha), (1 - alpha) / (1 + alpha)
    y = np.zeros(n)
    x1 = x2 = y1 = y2 = 0.0
    for i in range(n):
        y0 = b0 * x[i] + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2
        y[i] = y0
        x2, x1, y2, y1 = x1, x[i], y1, y0
    mp, tc = tables(a1, a2)
    M = np.array([[mp[0][0], mp[0][2]], [mp[0][1], mp[0][0]]])
    w = b0 * x + b1 * np.concatenate([[0], x[:-1]]) + b2 * np.concatenate([[0, 0], x[:-2]])
    W = w.reshape(-1, CHUNK)
    p1 = np.zeros(len(W)); p2 = np.zeros(len(W))
    for k in range(CHUNK):
        p1, p2 = W[:, k] - a1 * p1 - a2 * p2, p1
    U = np.stack([p1 - tc * p2, p2], 1)
    s = np.zeros(2)
    worst = 0.0
    for c in range(len(W)):
        got = (s[0] + tc * s[1], s[1])
        want = (y[c * CHUNK - 1], y[c * CHUNK - 2]) if c else (0.0, 0.0)
        worst = max(worst, abs(got[0] - want[0]), abs(got[1] - want[1]))
        s = M @ s + U[c]
    assert worst <= 1e-10, worst
