"""GPU parity on the shapes the reference accepts and round 1 refused, against the reference's own outputs
(tests/golden/gen_shapes.mjs run under Node): meters with 3 / 6 channels, sample rates whose 400 ms block is four hops plus
a remainder (11,025 Hz, 44,056 Hz), limiter / soft clipper / chain at 176.4 and 192 kHz (look-ahead and RMS windows beyond
1,024 samples), and the PROGRESS streams of the NORMALIZE / APPLY_LIMITER messages.  Bars as everywhere: gate masks
identical, LUFS 0.01 LU, true peak 0.01 dB, PCM 1e-5."""
import json
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
wam = pytest.importorskip("wam")

PCM_TOL = 1e-5
with open(os.path.join(G.DATA, "shapes.json")) as f:
    S = json.load(f)["cases"]
METERS = [k for k, c in S.items() if c["kind"] == "meter"]


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


@pytest.fixture(autouse=True)
def _reset_tuning():
    yield
    wam.tune("reset")


@pytest.mark.parametrize("seg", [0, 3])
@pytest.mark.parametrize("name", METERS)
def test_meter_shapes_vs_reference(name, seg):
    c = S[name]
    if seg:
        wam.tune("meter_seg_tiles", seg)                  # short segments: hop sums and hop heads cross many seams
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    lufs, blocks, mask = wam.measure_lufs(x, c["sr"], detail=True)
    want = G.f64(c["block_ms"])
    assert len(blocks) == c["nblocks"] == len(want)
    assert np.max(np.abs(blocks - want) / want) <= 1e-6
    assert np.array_equal(mask, G.u8(c["gate_mask"]))          # block selection identical to the reference's
    close(lufs, G.num(c["lufs"]), 1e-6)
    close(wam.find_true_peak(x), G.num(c["tp_db"]), 1e-9)
    l2, p2 = wam.meter(x, c["sr"])
    assert l2 == lufs and abs(p2 - G.num(c["tp_db"])) <= 1e-9
    for pos, v in c["short_term"]:
        close(wam.measure_short_term_lufs(x, c["sr"], pos), G.num(v), 1e-6)
    for pos, v in c["momentary"]:
        close(wam.measure_momentary_lufs(x, c["sr"], pos), G.num(v), 1e-6)
    st = wam.lufs_series(x, c["sr"], [p for p, _ in c["short_term"]])
    mo = wam.lufs_series(x, c["sr"], [p for p, _ in c["momentary"]], momentary=True)
    for got, (_, v) in list(zip(st, c["short_term"])) + list(zip(mo, c["momentary"])):
        close(float(got), G.num(v), 1e-6)


def test_meter_many_channels_batch_and_dc():
    import torch
    c = S["meter_6ch_44k"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    dev = torch.from_numpy(np.stack([np.stack(x), np.stack(x[::-1])])).cuda()       # two 6-channel tracks, the second with its channels reversed
    lufs, dbtp = wam.batch_meter_dev(dev.data_ptr(), 2, 6, c["n"], c["sr"])
    close(float(lufs[0]), G.num(c["lufs"]), 1e-6)
    close(float(lufs[1]), G.num(c["lufs"]), 1e-6)                                   # the channel mean does not care about the order
    close(float(dbtp[0]), G.num(c["tp_db"]), 1e-9)
    dc = wam.dc_offset(x)
    assert len(dc) == 6 and all(abs(d - O.dc_detect(ch)) <= 1e-12 for d, ch in zip(dc, x))


@pytest.mark.parametrize("sr", [176400, 192000])
def test_high_rate_limiter_softclip_vs_reference(sr):
    tag = f"hi{sr}"
    c = S[tag + "_limiter"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    seen = []
    y = wam.apply_lookahead_limiter(x, sr, c["ceiling_lin"], on_progress=lambda p, s: seen.append([p, s]) or False)
    assert maxerr(y, G.chans(c["out"])) <= PCM_TOL
    assert seen == c["progress"]                                                    # APPLY_LIMITER's PROGRESS stream, 'Complete' included
    assert maxerr(wam.apply_lookahead_limiter(x, sr, c["ceiling_lin"], preserve_transients=False), G.chans(S[tag + "_limiter_np"]["out"])) <= PCM_TOL
    assert maxerr(wam.apply_lookahead_limiter(x, sr, c["ceiling_lin"], twin="lib"), G.chans(S[tag + "_limiter_lib"]["out"])) <= PCM_TOL
    assert maxerr(wam.apply_mastering_soft_clip(x, sr), G.chans(S[tag + "_softclip"]["out"])) <= PCM_TOL


@pytest.mark.parametrize("name", ["hi176400_chain", "hi192000_chain", "chain_11025"])
def test_chain_shapes_vs_reference(name):
    c = S[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    seen = []
    y, res = wam.render_full_chain(x, c["sr"], c["settings"], c["mode"], on_progress=lambda p, s: seen.append([p, s]) or False)
    assert maxerr(y, G.chans(c["out"])) <= PCM_TOL
    close(res.lufs, G.num(c["lufs"]), 0.01)
    assert seen == c["progress"]
    # with de-harsh (unpinned at the fft.js boundary: against the oracle) and under a different segmentation
    s = dict(c["settings"], deharsh=True)
    for k in ("fx_seg_tiles", "meter_seg_tiles", "env_seg_tiles"):
        wam.tune(k, 7)
    y2, r2 = wam.render_full_chain(x, c["sr"], s, c["mode"])
    yo, ro = O.render_chain(x, c["sr"], s, c["mode"])
    assert maxerr(y2, yo) <= PCM_TOL
    close(r2.lufs, ro.lufs, 0.01)
    close(r2.true_peak_db, ro.true_peak_db, 0.01)


def test_very_high_rates_vs_oracle():
    """384 and 768 kHz (Chromium's AudioBuffer ceiling): RMS window 3,840 / 7,680 samples - wider than one tile, the ring of
    prefix sums turns - and look-ahead 1,152 / 2,304 samples."""
    for sr in (384000, 768000):
        n = int(0.12 * sr)
        x = O.synth(49, 2, n, sr, {"amp": 2.2, "clickAmp": 0.9})
        ceil = 10 ** (-1 / 20)
        assert maxerr(wam.apply_lookahead_limiter(x, sr, ceil), O.limiter(x, sr, ceil)) <= PCM_TOL
        assert maxerr(wam.apply_mastering_soft_clip(x, sr), O.soft_clip(x, sr)) <= PCM_TOL


def test_normalize_and_limiter_message_streams():
    for name in ("msg_normalize_limited", "msg_normalize_plain"):
        c = S[name]
        x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
        seen = []
        y, r = wam.normalize(x, c["sr"], c["target"], c["ceiling_db"], on_progress=lambda p, s: seen.append([p, s]) or False)
        assert seen == c["progress"]
        assert maxerr(y, G.chans(c["out"])) <= PCM_TOL
        close(r.current_lufs, G.num(c["currentLUFS"]), 1e-6)
        close(r.final_lufs, G.num(c["finalLUFS"]), 0.01)
        close(r.peak_db, G.num(c["peakDB"]), 0.01)
        close(r.gain_applied_db, G.num(c["gainApplied"]), 1e-6)
        assert bool(r.limiter_applied) == c["limiterApplied"]
    c = S["msg_normalize_silent"]
    seen = []
    z = [np.zeros(c["n"], np.float32)] * 2
    y, r = wam.normalize(z, c["sr"], c["target"], c["ceiling_db"], on_progress=lambda p, s: seen.append([p, s]) or False)
    assert seen == c["progress"] and r.measured == 0 and maxerr(y, z) == 0
    for name in ("msg_limiter", "msg_limiter_mono_defaults"):
        c = S[name]
        x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
        seen = []
        y = wam.apply_lookahead_limiter(x, c["sr"], c["ceiling_lin"], on_progress=lambda p, s: seen.append([p, s]) or False)
        assert seen == c["progress"]
        assert maxerr(y, G.chans(c["out"])) <= PCM_TOL
    with pytest.raises(wam.WamError, match="Cancelled"):
        c = S["msg_limiter"]
        x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
        wam.apply_lookahead_limiter(x, c["sr"], c["ceiling_lin"], on_progress=lambda p, s: p >= 0.3)


@pytest.mark.parametrize("name", [k for k, c in S.items() if c["kind"] == "deharsh"])
def test_deharsh_presets_vs_reference(name):
    """processHybridDynamic 'gentle' / 'aggressive' / no preset (dynamic-processor.js:538-566, :595-607) against the reference's
    output (made with the substitute DFT: unpinned at the fft.js boundary) and, longer and across segment boundaries, the oracle."""
    c = S[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    y = wam.process_hybrid_dynamic(x, c["sr"], c["preset"]) if c["preset"] else wam.process_hybrid_dynamic(x, c["sr"])
    assert maxerr(y, G.chans(c["out"])) <= PCM_TOL


@pytest.mark.parametrize("preset", ["gentle", "aggressive", "no-such-preset"])
def test_deharsh_presets_segments_vs_oracle(preset):
    sr, n = 48000, 48000 * 5 + 777
    x = O.synth(61, 2, n, sr, {"amp": 1.8, "clickAmp": 0.7})
    yo = O.deharsh(x, sr, preset)
    assert maxerr(O.deharsh(x, sr, "gentle"), O.deharsh(x, sr, "aggressive")) > 1e-2       # the presets do differ on this signal
    try:
        for seg in (0, 16, 50):
            wam.tune("dh_seg_frames", seg)
            y = wam.process_hybrid_dynamic(x, sr, preset)
            assert maxerr(y, yo) <= PCM_TOL, seg
    finally:
        wam.tune("reset")
    # fully wet: nothing after the last frame's end ((F - 1) hop + N) survives, the input does not leak through
    F = (n - 2048) // 512
    end = (F - 1) * 512 + 2048
    assert end < n and all(np.all(ch[end:] == 0) for ch in y)
