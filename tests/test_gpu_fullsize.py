"""BASELINE.json's full sizes on the GPU, through properties that do not need a sequential CPU run of the chain:

* configs[3] - one 60-min stereo 96 kHz track (345.6 M samples per channel): the meter is compared with the CPU oracle
  directly (it finishes in seconds); the chain is rendered under two different time segmentations, which must agree to
  the seam tolerance, and its by-products must be self-consistent (gain = target - measured loudness, output peak
  under the ceiling, finite everywhere).
* configs[2] - 3-min tracks: a track rendered inside a batch equals the same track rendered alone (different
  segmentation again), for every track checked.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

# Two segmentations of the FULL chain.  Round 2 saw isolated clusters differing by up to 1.3e-6 on an hour of audio and put the
# bar at 5e-6; the cause was not the seams but the biquad scans' power tables, whose error depended on where a
# chunk sat in its segment.  With the quadruple-precision tables the two renders of the 60-minute track, and batch against
# single renders of 3-minute tracks, are IDENTICAL (0 of 691,200,000 samples differ; printed below).  The bar allows a handful
# of f32 flips from the warm-up residue (<= 1e-7 of a state), nothing more.
CHAIN_SEAM_TOL = 1e-6
wam = pytest.importorskip("wam")
torch = pytest.importorskip("torch")

SETTINGS = dict(wam.DEFAULT_SETTINGS, glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1, eqHigh=2)


class SynthOpts(C.Structure):
    _fields_ = [("amp", C.c_double), ("dc", C.c_double), ("click_amp", C.c_double),
                ("quiet_from", C.c_int64), ("quiet_to", C.c_int64), ("quiet_gain", C.c_double),
                ("quiet2_from", C.c_int64), ("quiet2_to", C.c_int64), ("quiet2_gain", C.c_double)]


def synth_dev(t, ntracks, idx0, sr, amp=1.6, click=0.6, amp_step=0.0, quiet=None):
    q = quiet or (-1, -1, 0.0)
    o = SynthOpts(amp, 0.0, click, q[0], q[1], q[2], -1, -1, 0.0)
    wam._check(wam.lib().wam_synth_dev(C.c_void_p(t.data_ptr()), C.c_int(ntracks), C.c_int(t.shape[1]), C.c_int64(t.shape[-1]), C.c_double(sr),
                                       C.c_int(idx0), C.byref(o), C.c_double(amp_step), C.c_void_p(0)))


@pytest.fixture(autouse=True)
def _reset_tuning():
    yield
    wam.tune("reset")
    wam.lib().wam_trim_workspace()


def test_longform_60min_96k():
    sr, n = 96000, 96000 * 3600
    x = torch.empty((1, 2, n), dtype=torch.float32, device="cuda")
    synth_dev(x, 1, 17, sr, quiet=(sr * 600, sr * 900, 0.003))          # five nearly silent minutes: both gates act
    # ---- meter at full size against the oracle
    lufs, dbtp = wam.batch_meter_dev(x.data_ptr(), 1, 2, n, sr)
    host = [x[0, 0].cpu().numpy(), x[0, 1].cpu().numpy()]
    assert abs(float(lufs[0]) - O.measure_lufs(host, sr)) <= 1e-6        # bar: 0.01 LU
    assert abs(float(dbtp[0]) - O.true_peak_db(host)) <= 1e-9            # bar: 0.01 dB
    del host
    # ---- chain at full size under two segmentations
    y1 = torch.empty_like(x)
    r1 = wam.batch_render_dev(x.data_ptr(), y1.data_ptr(), 1, 2, n, sr, SETTINGS, "export")[0]
    for k in ("fx_seg_tiles", "meter_seg_tiles", "env_seg_tiles"):
        wam.tune(k, 1500)
    wam.tune("dh_seg_frames", 5000)
    y2 = torch.empty_like(x)
    r2 = wam.batch_render_dev(x.data_ptr(), y2.data_ptr(), 1, 2, n, sr, SETTINGS, "export")[0]
    assert bool(torch.isfinite(y1).all())
    print(f"longform: two segmentations differ by at most {float((y1 - y2).abs().max()):.3e}, {int(((y1 - y2).abs() > 2e-7).sum())} of {2 * n} samples above 2e-7")
    assert float((y1 - y2).abs().max()) <= CHAIN_SEAM_TOL
    assert int(((y1 - y2).abs() > 2e-7).sum()) <= 64                # and they are rare
    assert abs(r1.lufs - r2.lufs) <= 1e-6 and abs(r1.true_peak_db - r2.true_peak_db) <= 1e-5
    assert abs(r1.gain_db - (-14.0 - r1.lufs_in)) <= 1e-9                # normalizer.js:50
    ceil_lin = 10 ** (-1.0 / 20)
    assert float(y1.abs().max()) <= ceil_lin + 1e-7                      # the soft-knee curve clamps every sample to the ceiling (dsp-worker.js:222-226)
    assert 10 ** (r1.true_peak_db / 20) >= float(y1[:, :, 3:].abs().max()) - 1e-7     # the Catmull-Rom measure includes |y2| itself (i >= 3)
    assert r1.true_peak_db < 0.0
    assert -16.5 < r1.lufs < -13.0


def test_batch_track_equals_single_track_3min():
    sr, n, T = 48000, 48000 * 180, 24
    x = torch.empty((T, 2, n), dtype=torch.float32, device="cuda")
    synth_dev(x, T, 100, sr, amp_step=0.02)
    y = torch.empty_like(x)
    res = wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, 2, n, sr, SETTINGS, "export")
    one = torch.empty((1, 2, n), dtype=torch.float32, device="cuda")
    for t in (0, 7, T - 1):
        r = wam.batch_render_dev(x[t:t + 1].data_ptr(), one.data_ptr(), 1, 2, n, sr, SETTINGS, "export")[0]
        print(f"track {t}: batch vs single render differ by at most {float((one[0] - y[t]).abs().max()):.3e}")
        assert float((one[0] - y[t]).abs().max()) <= CHAIN_SEAM_TOL
        assert abs(r.lufs - res[t].lufs) <= 1e-6 and abs(r.lufs_in - res[t].lufs_in) <= 1e-6
    lufs = np.array([r.lufs for r in res])
    assert np.all(np.isfinite(lufs)) and np.all(np.abs(lufs + 14.3) < 1.5)
    # metering of the rendered batch agrees with the numbers the chain reported
    l2, p2 = wam.batch_meter_dev(y.data_ptr(), T, 2, n, sr)
    assert np.max(np.abs(np.array(l2) - lufs)) <= 1e-6
    assert np.max(np.abs(np.array(p2) - np.array([r.true_peak_db for r in res]))) <= 1e-6
