"""Whole tracks, EVERY sample, at the production segmentation (no wam_tune override), against the sequential oracle.

A time segment is entered `warm` tiles early from cold state (DESIGN.md 4 "Segments"); what is left of the cold start at the
seam is a residue of the state (bounded per kernel in DESIGN.md 2), and the mastering soft clipper's knee
(soft-clipper.js:194-231: `max(range, 0.001)` is 0.001, a slope of ~550 around the threshold) multiplies whatever differs ahead
of it.  Windowed or self-consistency checks cannot see an isolated cluster of flipped samples; these tests compare the full
output of ONE track rendered through wam_render_chain with the oracle's render of the same input:

  (a) the 180 s / 48 kHz track at the C2 settings (BASELINE configs[1]);
  (b) the same track hard-driven (+6 dB input, -9 LUFS target, -0.3 dB ceiling): the clipper's worst case;
  (c) 5 min 20 s at 96 kHz stereo with a nearly silent minute (configs[3]'s shape: rate-scaled warm-ups, both gates act);
  (d) 200 s at 44.1 kHz, hard-driven: hops and warmth windows of 4,410 samples (not multiples of a 16-sample chunk), the
      exciter's fused window peaks on their general path.

Bar: 1e-5 max abs on PCM (north_star).  The assertion message carries max |d| and the count of samples above 1e-6.
"""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
wam = pytest.importorskip("wam")

PCM_TOL = 1e-5
C2 = dict(glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1, eqHigh=2)
HOT = dict(C2, inputGain=6, targetLufs=-9, truePeakCeiling=-0.3)

CASES = {
    "c2_180s_48k": (48000, 48000 * 180, C2, {"amp": 1.6, "clickAmp": 0.6}),
    "hot_180s_48k": (48000, 48000 * 180, HOT, {"amp": 1.6, "clickAmp": 0.6}),
    "hot_200s_44k": (44100, 44100 * 200, HOT, {"amp": 1.8, "clickAmp": 0.7}),
    "quiet_320s_96k": (96000, 96000 * 320, C2, {"amp": 1.6, "clickAmp": 0.6, "quietFrom": 96000 * 100, "quietTo": 96000 * 160, "quietGain": 0.003}),
}


@pytest.mark.parametrize("name", list(CASES))
def test_whole_track_every_sample(name):
    sr, n, over, opts = CASES[name]
    x = O.synth(17, 2, n, sr, opts)
    s = dict(wam.DEFAULT_SETTINGS, **over)
    wam.tune("reset")                                   # production segmentation, whatever ran before
    y, res = wam.render_full_chain(x, sr, s, "export")
    yo, ro = O.render_chain(x, sr, s, "export")
    worst, above, where = 0.0, 0, None
    for ch in range(2):
        d = np.abs(y[ch].astype(np.float64) - yo[ch].astype(np.float64))
        k = int(d.argmax())
        if d[k] > worst:
            worst, where = float(d[k]), (ch, k)
        above += int((d > 1e-6).sum())
    msg = f"{name}: max|d| = {worst:.3e} at {where}, {above} of {2 * n} samples above 1e-6"
    print(msg)
    assert worst <= PCM_TOL, msg
    assert abs(res.lufs - ro.lufs) <= 0.01 and abs(res.true_peak_db - ro.true_peak_db) <= 0.01, (res.lufs, ro.lufs, res.true_peak_db, ro.true_peak_db)
    assert abs(res.lufs_in - ro.lufs_in) <= 0.01 and abs(res.gain_db - ro.gain_db) <= 0.01
