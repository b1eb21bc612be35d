"""GPU parity of the full export chain (wam_render_chain / wam_batch_render_dev through the C ABI).

Bars (BASELINE.json north_star): rendered PCM within 1e-5 max abs (float32), integrated
LUFS within 0.01 LU, true peak within 0.01 dBTP of the reference.
Checked against (1) the reference's own outputs (goldens produced by driving the real
`self.onmessage` RENDER_FULL_CHAIN handler under Node), (2) the oracle on further settings.
"""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
wam = pytest.importorskip("wam")

PCM_TOL = 1e-5
LUFS_TOL = 0.01
TP_TOL = 0.01

def maxerr(a, b):
    a = np.stack([np.asarray(x, dtype=np.float64) for x in a])
    b = np.stack([np.asarray(x, dtype=np.float64) for x in b])
    assert a.shape == b.shape
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def close(a, b, tol):
    if np.isinf(a) or np.isinf(b):
        assert a == b, (a, b)
    else:
        assert abs(a - b) <= tol, (a, b)


CHAINS = ["chain_nodeharsh_1s", "chain_core_1s", "chain_ceiling0_1s", "chain_preview_1s", "chain_mono_1s",
          "chain_defaults_1s", "chain_c2flat_1s"]


@pytest.mark.parametrize("name", CHAINS)
def test_chain_vs_reference_golden(golden, name):
    c = golden[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    seen = []
    y, res = wam.render_full_chain(x, c["sr"], c["settings"], c["mode"], on_progress=lambda p, s: seen.append((p, s)) or False)
    assert maxerr(y, G.chans(c["out"])) <= PCM_TOL
    close(res.lufs, G.num(c["lufs"]), LUFS_TOL)
    close(res.measured_lufs, G.num(c["measured_lufs"]), LUFS_TOL)
    close(res.true_peak_db, O.true_peak_db(G.chans(c["out"])), TP_TOL)
    # progress protocol: the worker's message stream, the limiter's own messages (dsp-worker.js:302,364,427,448) included
    assert seen == [tuple(p) for p in c["progress"]]


SETTINGS = [
    dict(glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1, eqHigh=2),          # C2: 'aifix' preset (spec-defined EQ)
    dict(cutMud=True, stereoWidth=150, centerBass=False, inputGain=-4.5),
    dict(normalizeLoudness=False),
    dict(truePeakLimit=False, glueCompression=True),
    dict(normalizeLoudness=False, truePeakLimit=False, cleanLowEnd=False, centerBass=False),
    dict(targetLufs=-9, truePeakCeiling=-0.3, inputGain=6),
    dict(addAir=False, tapeWarmth=False, addPunch=False),
]


@pytest.mark.parametrize("over", SETTINGS)
@pytest.mark.parametrize("deharsh", [False, True])
def test_chain_vs_oracle(over, deharsh):
    sr, n = 48000, 60000
    x = O.synth(50, 2, n, sr, {"amp": 1.8, "clickAmp": 0.7, "dc": 0.001})
    s = dict(wam.DEFAULT_SETTINGS, deharsh=deharsh, **over)
    y, res = wam.render_full_chain(x, sr, s, "export")
    yo, ro = O.render_chain(x, sr, s, "export")
    assert maxerr(y, yo) <= PCM_TOL
    close(res.lufs, ro.lufs, LUFS_TOL)
    close(res.true_peak_db, ro.true_peak_db, TP_TOL)
    if np.isnan(ro.lufs_in):
        assert np.isnan(res.lufs_in)
    else:
        close(res.lufs_in, ro.lufs_in, LUFS_TOL)
        close(res.gain_db, ro.gain_db, LUFS_TOL)


def test_chain_edge_inputs():
    sr = 48000
    s = dict(wam.DEFAULT_SETTINGS, deharsh=False)
    # silence: loudness -inf -> stage 8 leaves the buffer alone
    z = [np.zeros(30000, np.float32)] * 2
    y, res = wam.render_full_chain(z, sr, s, "export")     # (the exciter's bias makes the output non-silent, as in the reference)
    yo, ro = O.render_chain(z, sr, s, "export")
    assert maxerr(y, yo) <= PCM_TOL
    close(res.lufs, ro.lufs, LUFS_TOL)
    quiet = dict(s, addAir=False)
    y, res = wam.render_full_chain(z, sr, quiet, "export")
    assert maxerr(y, z) == 0 and res.lufs == float("-inf") and res.true_peak_db == float("-inf")
    # shorter than one LUFS block: loudness falls back to the target -> gain 0 dB
    x = O.synth(51, 2, 9000, sr, {"amp": 1.5})
    y, res = wam.render_full_chain(x, sr, s, "export")
    yo, ro = O.render_chain(x, sr, s, "export")
    assert maxerr(y, yo) <= PCM_TOL
    close(res.lufs, ro.lufs, LUFS_TOL)
    close(res.gain_db, 0.0, 1e-12)


def test_cancel():
    sr = 48000
    x = O.synth(52, 2, 20000, sr, {})
    with pytest.raises(wam.WamError, match="Cancelled"):
        wam.render_full_chain(x, sr, dict(wam.DEFAULT_SETTINGS, deharsh=False), "export", on_progress=lambda p, s: p >= 0.6)


@pytest.mark.parametrize("name", ["chain_180s_nodeharsh", "chain_180s_defaults"])
def test_chain_180s_vs_reference_golden(golden, name):
    c = golden[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    y, res = wam.render_full_chain(x, c["sr"], c["settings"], "export")
    close(res.lufs, G.num(c["lufs"]), LUFS_TOL)
    close(res.true_peak_db, G.num(c["tp_db"]), TP_TOL)
    w = G.f32(c["windows"]).reshape(c["nch"], len(c["win_starts"]), c["win_len"])
    worst = 0.0
    for ch in range(c["nch"]):
        assert abs(float(np.abs(y[ch]).max()) - G.num(c["out_stats"][ch]["maxabs"])) <= PCM_TOL
        for k, s0 in enumerate(c["win_starts"]):
            worst = max(worst, float(np.max(np.abs(y[ch][s0:s0 + c["win_len"]].astype(np.float64) - w[ch, k]))))
    assert worst <= PCM_TOL, worst


def test_batch_render_dev_matches_single():
    import torch
    sr, n, nt = 48000, 50000, 3
    s = dict(wam.DEFAULT_SETTINGS, deharsh=True, glueCompression=True)
    tracks = [O.synth(60 + t, 2, n, sr, {"amp": 1.2 + 0.4 * t, "clickAmp": 0.5}) for t in range(nt)]
    dev = torch.from_numpy(np.stack([np.stack(t) for t in tracks])).cuda()
    out = torch.empty_like(dev)
    res = wam.batch_render_dev(dev.data_ptr(), out.data_ptr(), nt, 2, n, sr, s)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    for t in range(nt):
        y, r = wam.render_full_chain(tracks[t], sr, s, "export")
        assert maxerr(host[t], y) == 0.0          # same kernels, same arithmetic: identical
        assert res[t].lufs == r.lufs and res[t].true_peak_db == r.true_peak_db
        yo, ro = O.render_chain(tracks[t], sr, s, "export")
        assert maxerr(host[t], yo) <= PCM_TOL


def test_host_batch_equals_single():
    """wam_batch_render (host pointers, up to eight staging threads on their own streams, groups of 8): 70 tracks = 9 groups, the
    last one partial; every track equals its single-track render bit for bit, results included."""
    sr, n = 48000, 20000
    s = dict(wam.DEFAULT_SETTINGS, glueCompression=True)
    tracks = [O.synth(100 + t, 2, n, sr, {"amp": 1.2 + 0.01 * t, "clickAmp": 0.5}) for t in range(70)]
    outs, res = wam.batch_render(tracks, sr, s, "export")
    for t in (0, 1, 7, 8, 9, 31, 32, 33, 63, 64, 69):
        y, r = wam.render_full_chain(tracks[t], sr, s, "export")
        assert all(np.array_equal(a, b) for a, b in zip(outs[t], y)), t
        assert res[t].lufs == r.lufs and res[t].true_peak_db == r.true_peak_db
    with pytest.raises(wam.WamError):
        wam.batch_render(tracks[:2], 0.0, s, "export")


def test_staging_variants_move_the_same_bytes():
    """The host-pointer path copies through pinned rings with non-temporal stores by default (csrc/staging.hip); with plain memcpy
    (stage_nt 0) and with the runtime's own staging (stage_pinned 0) the rendered samples are the same to the bit.  1.3 M samples
    per channel: above the 1 MiB threshold of the rings, not a multiple of anything, channels at odd byte offsets of a base array."""
    sr, n = 48000, 1300003
    base = [np.zeros(n + 3, dtype=np.float32) for _ in range(2)]
    x0 = O.synth(3, 2, n, sr, {"amp": 1.3, "clickAmp": 0.5})
    x = []
    for c in range(2):
        v = base[c][1 + c:1 + c + n]                   # 4- and 8-byte offsets from the allocation: unaligned for 16-byte stores
        v[:] = x0[c]
        x.append(v)
    s = dict(wam.DEFAULT_SETTINGS, deharsh=False, addPunch=False)
    try:
        want, wres = wam.render_full_chain(x, sr, s, "export")
        for key, val in (("stage_nt", 0), ("stage_pinned", 0)):
            wam.tune(key, val)
            got, gres = wam.render_full_chain(x, sr, s, "export")
            wam.tune("reset")
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), key
            assert gres.lufs == wres.lufs
    finally:
        wam.tune("reset")


def test_multi_device_batch_equals_single():
    """wam_batch_render_multi: contiguous track ranges per device entry, one host thread + streams per entry,
    results concatenated in track order.  One GPU on the box: devices = [0, 0] are two workers on it; [0, 0, 0] over 23 tracks gives
    ranges of 8, 8, 7.  Every track equals its single-track render bit for bit, results included, and the oracle to the bar."""
    sr, n = 48000, 20000
    s = dict(wam.DEFAULT_SETTINGS, glueCompression=True)
    tracks = [O.synth(200 + t, 2, n, sr, {"amp": 1.2 + 0.02 * t, "clickAmp": 0.5}) for t in range(23)]
    for devices in ([0, 0], [0, 0, 0], [0]):
        outs, res = wam.batch_render(tracks, sr, s, "export", devices=devices)
        for t in (0, 7, 8, 11, 12, 15, 16, 22):
            y, r = wam.render_full_chain(tracks[t], sr, s, "export")
            assert all(np.array_equal(a, b) for a, b in zip(outs[t], y)), (devices, t)
            assert res[t].lufs == r.lufs and res[t].true_peak_db == r.true_peak_db and res[t].gain_db == r.gain_db
    yo, ro = O.render_chain(tracks[12], sr, s, "export")
    assert maxerr(outs[12], yo) <= PCM_TOL
    outs, res = wam.batch_render(tracks[:2], sr, s, "export", devices=[0, 0, 0, 0])        # more entries than tracks: empty ranges
    assert all(np.array_equal(a, b) for a, b in zip(outs[1], wam.render_full_chain(tracks[1], sr, s, "export")[0]))
    with pytest.raises(wam.WamError, match="out of range"):
        wam.batch_render(tracks[:2], sr, s, "export", devices=[0, 99])
    with pytest.raises(wam.WamError):
        wam.batch_render(tracks[:2], sr, s, "export", devices=[])
    assert wam.measure_lufs(tracks[0], sr) == wam.measure_lufs(tracks[0], sr)              # the calling thread still has its device


def test_hot_material_margin():
    """The soft clipper multiplies upstream differences by up to ~550 near its threshold, so a hard-driven master
    is the severest parity case there is.  Every stage ahead of the clipper is faithful to the reference's f32 stores - the
    de-harsh stage included, with its default kernels - and the chain stays within 1e-6 of the oracle.  With the float32
    de-harsh kernels (wam_tune("dh_exact", 0): ~1e-7 on their own output) a small fraction of the samples of such a master
    exceeds the 1e-5 bar; measured and bounded here so that it cannot grow unnoticed (tests/probe_hot_margin.py: 30 s tracks)."""
    sr, n = 48000, 48000 * 10
    x = O.synth(60, 2, n, sr, {"amp": 1.8, "clickAmp": 0.7})
    try:
        for over in (dict(targetLufs=-9, truePeakCeiling=-0.3, inputGain=6), dict(targetLufs=-7, inputGain=3, glueCompression=True)):
            for dh, exact in ((False, 1), (True, 1), (True, 0)):
                wam.tune("dh_exact", exact)
                s = dict(wam.DEFAULT_SETTINGS, deharsh=dh, **over)
                y, _ = wam.render_full_chain(x, sr, s, "export")
                yo, _ = O.render_chain(x, sr, s, "export")
                d = np.concatenate([np.abs(a.astype(np.float64) - b) for a, b in zip(y, yo)])
                if exact:
                    assert d.max() <= 1e-6, (over, dh, d.max())
                else:
                    assert d.max() <= 1e-4 and (d > 1e-5).mean() <= 1e-3, (over, d.max(), (d > 1e-5).mean())
    finally:
        wam.tune("reset")


@pytest.mark.parametrize("n", [1, 2, 3, 5, 16, 17, 255, 257, 2047, 2048, 2049, 2560, 4095, 4097, 4799, 4801, 19199, 19201])
@pytest.mark.parametrize("nch", [1, 2])
def test_ragged_fx_and_chain(n, nch):
    """Tiny and ragged lengths (below one FFT frame, one tile, one LUFS hop / block, around each boundary), mono and stereo: every
    heavy stage and the whole chain against the oracle."""
    sr = 48000
    x = O.synth(70 + n % 7, nch, n, sr, {"amp": 1.9, "clickAmp": 0.6})
    assert maxerr(wam.apply_exciter(x, sr), O.exciter(x, sr)) <= PCM_TOL
    assert maxerr(wam.apply_tape_warmth(x, sr), O.warmth(x, sr)) <= PCM_TOL
    assert maxerr(wam.apply_multiband_transient(x, sr), O.punch(x, sr)) <= PCM_TOL
    assert maxerr(wam.process_hybrid_dynamic(x, sr, "mastering"), O.deharsh(x, sr)) <= PCM_TOL
    s = dict(wam.DEFAULT_SETTINGS, deharsh=True, glueCompression=True, eqLow=1, eqHigh=2, cutMud=True)
    y, res = wam.render_full_chain(x, sr, s, "export")
    yo, ro = O.render_chain(x, sr, s, "export")
    assert maxerr(y, yo) <= PCM_TOL
    close(res.lufs, ro.lufs, LUFS_TOL)
    close(res.true_peak_db, ro.true_peak_db, TP_TOL)


def test_empty_input_everywhere():
    """n = 0: every entry point returns empty outputs (or the documented sentinel) without touching the device's data path."""
    sr = 48000
    z = [np.zeros(0, np.float32)] * 2
    for fn in (wam.apply_exciter, wam.apply_tape_warmth, wam.apply_multiband_transient, wam.apply_final_filters, wam.apply_glue_compressor,
               wam.apply_mastering_soft_clip):
        out = fn(z, sr)
        assert len(out) == 2 and len(out[0]) == 0
    assert len(wam.process_hybrid_dynamic(z, sr, "mastering")[0]) == 0
    y, res = wam.render_full_chain(z, sr, dict(wam.DEFAULT_SETTINGS), "export")
    assert len(y[0]) == 0
    assert wam.measure_lufs(z, sr) == -14.0 and wam.find_true_peak(z) == float("-inf")
    assert len(wam.encode_pcm24(z)) == 0


@pytest.mark.parametrize("sr", [16000, 22050, 32000, 44100, 88200, 96000])
def test_chain_other_sample_rates(sr):
    """Every plan constant that depends on the sample rate (filter coefficients, look-ahead and window lengths, hop and block
    sizes, warm-up lengths, de-harsh band edges): 1.5 s through the whole chain, against the oracle.  Below 32 kHz the stages
    whose corner frequency lies above Nyquist (6 kHz exciter high-pass pair at 16 k is still fine; the 12 kHz shelf is not) are
    switched off: the reference's filters are unstable there and both sides only produce garbage."""
    n = int(sr * 1.5) + 13
    x = O.synth(90, 2, n, sr, {"amp": 1.7, "clickAmp": 0.6})
    s = dict(wam.DEFAULT_SETTINGS, deharsh=True, glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1,
             eqHigh=2 if sr >= 32000 else 0)
    y, res = wam.render_full_chain(x, sr, s, "export")
    yo, ro = O.render_chain(x, sr, s, "export")
    assert np.isfinite(np.stack(yo)).all()
    assert maxerr(y, yo) <= PCM_TOL
    close(res.lufs, ro.lufs, LUFS_TOL)
    close(res.true_peak_db, ro.true_peak_db, TP_TOL)


def test_chain_random_settings():
    """Forty random points of the settings space (every switch, every slider over its UI range, controls.js:66-135), seeded: the
    chain against the oracle, to the plain 1e-5 bar whatever the draw (hard-driven ones with de-harsh included)."""
    rng = np.random.default_rng(2024)
    sr, n = 48000, 40000
    x = O.synth(95, 2, n, sr, {"amp": 1.6, "clickAmp": 0.6, "dc": 0.0005})
    for k in range(40):
        b = lambda: bool(rng.integers(0, 2))
        s = dict(normalizeLoudness=b(), targetLufs=float(rng.choice([-23, -16, -14, -11, -9, -7])), truePeakLimit=b(),
                 truePeakCeiling=float(rng.choice([0, -0.1, -0.3, -1, -2, -3])), cleanLowEnd=b(), glueCompression=b(), deharsh=b(),
                 stereoWidth=float(rng.choice([0, 50, 100, 100, 150, 200])), centerBass=b(), cutMud=b(), addAir=b(), tapeWarmth=b(), autoLevel=False,
                 addPunch=b(), inputGain=float(rng.choice([-12, -4.5, 0, 0, 3, 6, 12])),
                 eqLow=float(rng.integers(-12, 13)) * float(b()), eqLowMid=float(rng.integers(-12, 13)) * float(b()), eqMid=float(rng.integers(-12, 13)) * float(b()),
                 eqHighMid=float(rng.integers(-12, 13)) * float(b()), eqHigh=float(rng.integers(-12, 13)) * float(b()))
        mode = "export" if rng.integers(0, 4) else "preview"
        y, res = wam.render_full_chain(x, sr, s, mode)
        yo, ro = O.render_chain(x, sr, s, mode)
        d = np.concatenate([np.abs(a.astype(np.float64) - c) for a, c in zip(y, yo)])
        assert d.max() <= PCM_TOL, (k, s, d.max())
        close(res.lufs, ro.lufs, LUFS_TOL)
        close(res.true_peak_db, ro.true_peak_db, TP_TOL)
