"""GPU: sample-rate conversion (second half).  PARITY UNPINNED: the reference's resampler is Chromium's
OfflineAudioContext (web/ui/renderer.js:172-187), absent here and without fixtures.  What is tested is the entry's own stated
definition (include/wam.h): the length rule of renderer.js:178, agreement with a float64 numpy restatement, identity at equal
rates, and the accuracy of a resampled sine against the analytic one."""
import math

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

wam = pytest.importorskip("wam")


@pytest.mark.parametrize("src,dst", [(48000, 44100), (44100, 48000), (96000, 48000), (48000, 48000)])
def test_resample_matches_definition(src, dst):
    n = 12011
    x = O.synth(5, 2, n, src, {"clickAmp": 0.5})
    for q in (0, 1):
        got = wam.resample(x, src, dst, quality=q)
        assert len(got) == 2 and len(got[0]) == int(math.ceil((n / src) * dst))       # Math.ceil(duration * targetSampleRate)
        for c in range(2):
            want = O.resample_ref(x[c], src, dst, q)
            assert np.max(np.abs(got[c].astype(np.float64) - want)) <= 2e-6, (src, dst, q)
        if src == dst and q == 0:
            assert np.array_equal(got[0], x[0]) and np.array_equal(got[1], x[1])     # integer positions, f = 0


def test_resample_sine_accuracy():
    src, dst, f0, n = 48000, 44100, 1000.0, 48000
    x = (0.5 * np.sin(2 * np.pi * f0 * np.arange(n) / src)).astype(np.float32)
    want = 0.5 * np.sin(2 * np.pi * f0 * np.arange(int(math.ceil(n / src * dst))) / dst)
    lin = wam.resample([x], src, dst, quality=0)[0]
    snc = wam.resample([x], src, dst, quality=1)[0]
    mid = slice(300, len(want) - 300)                                    # away from the zero-padded edges
    assert np.max(np.abs(snc[mid] - want[mid])) <= 2e-5
    assert np.max(np.abs(lin[mid] - want[mid])) <= 0.5 * 0.5 * (math.pi * f0 / src) ** 2 * 1.1    # linear-interpolation bound A (pi f / fs)^2 / 2
    # an alias-prone tone above the new Nyquist (23 kHz -> 22.05 kHz) is attenuated by the sinc kernel, not by the linear one
    hi = (0.5 * np.sin(2 * np.pi * 23000.0 * np.arange(n) / src)).astype(np.float32)
    assert np.max(np.abs(wam.resample([hi], src, dst, quality=1)[0][mid])) <= 1e-4
    assert np.max(np.abs(wam.resample([hi], src, dst, quality=0)[0][mid])) >= 0.1


def test_resample_errors():
    x = [np.zeros(100, np.float32)]
    with pytest.raises(wam.WamError):
        wam.resample(x, 48000, 0)
    with pytest.raises(wam.WamError):
        wam.resample(x, 48000, 44100, quality=7)
