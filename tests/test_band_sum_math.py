"""The identity behind k_dh_bands_x's scan-based band sums (csrc/chain_deharsh.hip, band_energies_scan), checked on the CPU
with numpy: the reference's running sum  s <- f32(f64(s) + a)  (bandEnergies[b] += mag * mag on a Float32Array,
dynamic-processor.js:327-332) equals  s + u * rint(a / u),  u = ulp of s's binade, whenever the sum stays inside that
binade and a / u is not within 2^-20 of a rounding tie - the two cases the kernel hands to the exact serial step."""
import numpy as np


def test_quantised_add_equals_f32_running_sum_step():
    rng = np.random.default_rng(7)
    n = 2_000_000
    s = (rng.random(n, dtype=np.float64) * 10.0 ** rng.uniform(-12, 6, n)).astype(np.float32)
    s = s[s > 0]
    n = len(s)
    # addends: squares of f32 magnitudes, from far below one ulp of s to several times s
    m = (np.sqrt(s.astype(np.float64)) * 10.0 ** rng.uniform(-6, 0.5, n)).astype(np.float32)
    a = m.astype(np.float64) * m.astype(np.float64)          # exact in double (48 bits)
    ref = (s.astype(np.float64) + a).astype(np.float32)      # the reference step: one double add, one f32 store
    mant, ex = np.frexp(s.astype(np.float64))                # s = mant 2^ex, mant in [0.5, 1): binade [2^(ex-1), 2^ex)
    u = np.ldexp(1.0, ex - 24)                               # f32 ulp there
    t = a / u
    r = np.rint(t)
    cand = s.astype(np.float64) + r * u                      # exact: multiples of u below 2^25 u
    _, ex_ref = np.frexp(ref.astype(np.float64))
    same_binade = (ex_ref == ex) & (cand < np.ldexp(1.0, ex))
    away_from_tie = np.abs(np.abs(t - r) - 0.5) > 2.0 ** -20
    ok = same_binade & away_from_tie
    assert ok.mean() > 0.5                                   # the case is the common one
    assert np.array_equal(cand[ok].astype(np.float32), ref[ok])
    assert np.array_equal(cand[ok], ref[ok].astype(np.float64))      # and exactly representable


def test_run_sum_is_order_free_inside_a_binade():
    """Sixteen quantised addends in any order, or summed first and added once, give the bin-by-bin result."""
    rng = np.random.default_rng(8)
    for _ in range(2000):
        s0 = np.float32(rng.uniform(1.0, 1.2) * 2.0 ** rng.integers(-30, 20))
        _, ex = np.frexp(np.float64(s0))
        u = np.ldexp(1.0, int(ex) - 24)
        m = (np.sqrt(np.float64(s0)) * 10.0 ** rng.uniform(-5, -1, 16)).astype(np.float32)
        a = m.astype(np.float64) ** 2
        s = s0
        for x in a:
            s = np.float32(np.float64(s) + x)
        if np.frexp(np.float64(s))[1] != ex or np.any(np.abs(np.abs(a / u - np.rint(a / u)) - 0.5) <= 2.0 ** -20):
            continue
        assert np.float64(s0) + np.sum(np.rint(a / u)) * u == np.float64(s)
