"""Time segmentation: kernels that split a track into segments and rebuild their state by re-running a warm-up
stretch must give the same samples as one sequential walk.  The knobs are shrunk (wam.tune) so that short inputs
cross many segment seams; every case is compared with the sequential CPU oracle, to the PCM bar of 1e-5, and with
the unsegmented GPU result, which it should match far more closely.
"""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
wam = pytest.importorskip("wam")

TOL = 1e-5
SEAM_TOL = 2e-7      # segmented vs unsegmented GPU: warm-up residue + f32 store flips


def maxerr(a, b):
    a = np.stack([np.asarray(x, dtype=np.float64) for x in a])
    b = np.stack([np.asarray(x, dtype=np.float64) for x in b])
    return float(np.max(np.abs(a - b)))


@pytest.fixture(autouse=True)
def _reset_tuning():
    yield
    wam.tune("reset")


@pytest.fixture(scope="module")
def hot():
    sr = 48000
    return sr, O.synth(11, 2, 3 * sr + 777, sr, {"amp": 2.4, "clickAmp": 0.8, "dc": 0.002})


@pytest.mark.parametrize("exact", [1, 0])
def test_deharsh_segments(hot, exact):
    sr, x = hot
    ref = O.deharsh(x, sr)
    wam.tune("dh_exact", exact)                # exact kernels: the warm-up is raised to 96 frames whatever the knob says
    wam.tune("dh_seg_frames", 1 << 20)
    whole = wam.process_hybrid_dynamic(x, sr, "mastering")
    assert maxerr(whole, ref) <= TOL
    for seg, warm in ((16, 48), (7, 40), (64, 48)):
        wam.tune("dh_seg_frames", seg)
        wam.tune("dh_warm_frames", warm)
        y = wam.process_hybrid_dynamic(x, sr, "mastering")
        assert maxerr(y, ref) <= TOL
        assert maxerr(y, whole) <= SEAM_TOL


@pytest.mark.parametrize("preset", ["mastering", "aggressive"])
def test_deharsh_window_from_global_memory_is_bit_identical(hot, preset):
    """The default de-harsh kernels are compiled twice: with the first half of the Hann table in LDS (it mirrors in f32 with this libm's
    cos) and with the table read from global memory (a host whose table does not mirror).  Same numbers either way."""
    sr, x = hot
    want = wam.process_hybrid_dynamic(x, sr, preset)
    wam.tune("dh_win_lds", 0)
    got = wam.process_hybrid_dynamic(x, sr, preset)
    for a, b in zip(got, want):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_soft_clip_segments(hot):
    sr, x = hot
    ref = O.soft_clip(x, sr, -1.0, 0.5, 10.0, 1.5)
    wam.tune("env_seg_tiles", 1 << 20)
    whole = wam.apply_mastering_soft_clip(x, sr, ceiling=-1.0)
    assert maxerr(whole, ref) <= TOL
    for seg, warm in ((1, 3), (2, 4), (5, 3)):
        wam.tune("env_seg_tiles", seg)
        wam.tune("sc_warm_tiles", warm)
        y = wam.apply_mastering_soft_clip(x, sr, ceiling=-1.0)
        assert maxerr(y, ref) <= TOL
        assert maxerr(y, whole) <= SEAM_TOL


@pytest.mark.parametrize("preserve", [True, False])
def test_limiter_segments(hot, preserve):
    sr, x = hot
    ceil = 10 ** (-1 / 20)
    ref = O.limiter(x, sr, ceil, preserve=preserve)
    wam.tune("env_seg_tiles", 1 << 20)
    whole = wam.apply_lookahead_limiter(x, sr, ceil, preserve_transients=preserve)
    assert maxerr(whole, ref) <= TOL
    for seg, warm in ((1, 25), (3, 25), (2, 30)):
        wam.tune("env_seg_tiles", seg)
        wam.tune("lim_warm_tiles", warm)
        y = wam.apply_lookahead_limiter(x, sr, ceil, preserve_transients=preserve)
        assert maxerr(y, ref) <= TOL
        assert maxerr(y, whole) <= SEAM_TOL


def test_chain_segments_long():
    """A 25 s track through the whole chain with small segments everywhere, against the sequential oracle."""
    sr = 48000
    x = O.synth(5, 2, 25 * sr + 123, sr, {"amp": 1.8, "clickAmp": 0.7})
    s = dict(wam.DEFAULT_SETTINGS, glueCompression=True)
    ref, rres = O.render_chain(x, sr, s, "export")
    wam.tune("env_seg_tiles", 16)
    wam.tune("dh_seg_frames", 128)
    wam.tune("fx_seg_tiles", 60)
    y, res = wam.render_full_chain(x, sr, s, "export")
    assert maxerr(y, ref) <= TOL
    assert abs(res.lufs - rres.lufs) <= 0.01


@pytest.mark.parametrize("sr", [48000, 44100, 96000, 11025])
def test_exciter_window_peaks_equal_the_separate_pass(sr):
    """The exciter leaves tape warmth's per-window peaks behind (k_exciter<PEAKS>, every 4th sample of each 100 ms window, maximum
    over the channels); with the hook off, k_win_peak reads the exciter's output again.  Maxima are order-independent: the chain's
    output must be the same to the bit, with one segment and with many seams (window boundaries inside wavefronts, chunks and tiles:
    4,410 and 1,102 are not multiples of 16).  At 11,025 Hz the window (1,102 samples) is shorter than the fused form allows
    (a wavefront's 1,024 samples + a chunk), so both runs take the separate pass."""
    x = O.synth(9, 2, int(7.3 * sr) + 331, sr, {"amp": 0.9, "clickAmp": 0.6})          # quiet enough for the bypass envelope to move
    s = dict(wam.DEFAULT_SETTINGS, addAir=True, tapeWarmth=True)
    for seg in (1 << 20, 5):
        wam.tune("fx_seg_tiles", seg)
        wam.tune("exc_peaks", 0)
        want, wres = wam.render_full_chain(x, sr, s, "export")
        wam.tune("exc_peaks", 1)
        got, gres = wam.render_full_chain(x, sr, s, "export")
        for a, b in zip(got, want):
            assert np.array_equal(np.asarray(a), np.asarray(b)), (sr, seg)
        assert gres.lufs == wres.lufs
    if sr == 48000:
        ref, _ = O.render_chain(x, sr, s, "export")
        assert maxerr(got, ref) <= TOL


def test_fx_segments(hot):
    """exciter / tape warmth / punch: LTI state needs 2 warm-up tiles, the slow transient follower 48."""
    sr, x0 = hot
    x = [np.concatenate([c, c, c, c, c]) for c in x0]          # ~15 s = 176 tiles, so 48 warm-up tiles fit several times
    cases = (("exciter", wam.apply_exciter, O.exciter), ("warmth", wam.apply_tape_warmth, O.warmth), ("punch", wam.apply_multiband_transient, O.punch))
    for name, fn, ofn in cases:
        ref = ofn(x, sr)
        wam.tune("fx_seg_tiles", 1 << 20)
        whole = fn(x, sr)
        assert maxerr(whole, ref) <= TOL, name
        for seg in (8, 50):
            wam.tune("fx_seg_tiles", seg)
            y = fn(x, sr)
            assert maxerr(y, ref) <= TOL, (name, seg)
            assert maxerr(y, whole) <= SEAM_TOL, (name, seg)
        wam.tune("reset")


def test_meter_segments_bit_exact_gating():
    """Segmented metering: every hop sum has one owner workgroup, so block energies agree with the unsegmented walk
    to rounding and the gate masks are identical."""
    sr = 48000
    x = O.synth(2, 2, 30 * sr + 4321, sr, {"amp": 0.9, "quietFrom": 5 * sr, "quietTo": 12 * sr, "quietGain": 0.004})
    ref = O.measure_lufs(x, sr, detail=True)
    wam.tune("meter_seg_tiles", 1 << 20)
    l0, b0, m0 = wam.measure_lufs(x, sr, detail=True)
    p0 = wam.find_true_peak(x)
    for seg in (8, 13, 40):
        wam.tune("meter_seg_tiles", seg)
        l1, b1, m1 = wam.measure_lufs(x, sr, detail=True)
        assert np.array_equal(m1, m0) and np.array_equal(m1, ref[2])
        assert np.max(np.abs(b1 - b0) / np.maximum(b0, 1e-30)) < 1e-9
        assert abs(l1 - l0) < 1e-9 and abs(l1 - ref[0]) < 1e-6
        assert abs(wam.find_true_peak(x) - p0) < 1e-12


def test_front_and_limiter_apply_segments(hot):
    sr, x0 = hot
    x = [np.concatenate([c, c, c, c, c, c, c, c]) for c in x0]          # ~24 s = 282 tiles
    s = dict(wam.DEFAULT_SETTINGS, deharsh=False, addAir=False, tapeWarmth=False, addPunch=False, glueCompression=True,
             eqLow=1, eqMid=-1, eqHigh=2, cutMud=True, stereoWidth=130)
    ref, rres = O.render_chain(x, sr, s, "export")
    wam.tune("meter_seg_tiles", 1 << 20)
    whole, wres = wam.render_full_chain(x, sr, s, "export")
    assert maxerr(whole, ref) <= TOL
    for seg in (60, 110):
        wam.tune("meter_seg_tiles", seg)
        y, res = wam.render_full_chain(x, sr, s, "export")
        assert maxerr(y, ref) <= TOL
        assert maxerr(y, whole) <= 5e-7
        assert abs(res.lufs - wres.lufs) < 1e-6 and abs(res.true_peak_db - wres.true_peak_db) < 1e-6
        assert abs(res.lufs - rres.lufs) <= 0.01
