"""GPU parity: HIP meters (through the C ABI) against the oracle and the reference goldens.

Bars (BASELINE.json north_star): gating block selection bit-exact, |dLUFS| <= 0.01 LU,
|dTP| <= 0.01 dB.  We hold much tighter: 1e-9 LU / dB, block energies 1e-12 relative.
"""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu

wam = pytest.importorskip("wam")

LUFS_TOL = 1e-9
TP_TOL = 1e-9


def _close(a, b, tol):
    if np.isinf(a) or np.isinf(b):
        assert a == b
    else:
        assert abs(a - b) <= tol, (a, b)


def gate_margin(blocks, mask):
    """Relative distance of the nearest block energy to either gate threshold (for the report)."""
    kept1 = blocks[blocks > 1e-7]
    m = np.min(np.abs(blocks - 1e-7) / 1e-7) if len(blocks) else np.inf
    if len(kept1):
        thr = kept1.sum() / len(kept1) * 0.1
        m = min(m, np.min(np.abs(blocks - thr) / thr))
    return m


METER = ["meter_10s_48k", "meter_gated_48k", "meter_mono_44k", "meter_96k", "meter_short", "meter_exact_block", "meter_silence"]


@pytest.mark.parametrize("name", METER)
def test_meter_vs_golden_and_oracle(golden, name):
    c = golden[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    lufs, blocks, mask = wam.measure_lufs(x, c["sr"], detail=True)
    tp = wam.find_true_peak(x)
    _close(lufs, G.num(c["lufs"]), LUFS_TOL)
    _close(tp, G.num(c["tp_db"]), TP_TOL)
    _close(wam.measure_lufs(x, c["sr"], fallback=-20.0), G.num(c["lufs_fallback_m20"]), LUFS_TOL)
    ol, ob, om = O.measure_lufs(x, c["sr"], detail=True)
    assert len(blocks) == len(ob)
    if len(ob):
        assert np.array_equal(mask, om), "gate selection differs from the oracle"
        # block energies: different (parallel) summation order and scan association than the sequential
        # reference; quiet blocks after loud ones carry the loud part's rounding noise
        rel = float(np.max(np.abs(blocks - ob) / np.maximum(ob, 1e-30)))
        absr = float(np.max(np.abs(blocks - ob)) / max(float(ob.max()), 1e-30))
        assert rel < 1e-6 and absr < 1e-8, (rel, absr)
        assert gate_margin(ob, om) > 1e-9
    if "gate_mask" in c:
        assert np.array_equal(mask, G.u8(c["gate_mask"])), "gate selection differs from the reference"
    l2, t2 = wam.meter(x, c["sr"])
    _close(l2, lufs, 0)
    _close(t2, tp, 0)


def test_meter_180s(golden):
    c = golden["meter_180s_48k"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    lufs, blocks, mask = wam.measure_lufs(x, c["sr"], detail=True)
    _close(lufs, G.num(c["lufs"]), LUFS_TOL)
    _close(wam.find_true_peak(x), G.num(c["tp_db"]), TP_TOL)
    assert len(blocks) == 1797
    assert int((mask >= 1).sum()) == c["ngate_abs"] and int((mask == 2).sum()) == c["ngate_rel"]
    ol, ob, om = O.measure_lufs(x, c["sr"], detail=True)
    assert np.array_equal(mask, om)


def test_true_peak_special(golden):
    for k, v in golden["tp_special"]["vectors"].items():
        _close(wam.find_true_peak([G.f32(v["input"])]), G.num(v["tp_db"]), TP_TOL)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 15, 16, 17, 4095, 4096, 4097, 19200, 24001])
def test_ragged_lengths(n):
    sr = 48000
    x = O.synth(21, 2, n, sr, {"amp": 1.2}) if n else [np.zeros(0, np.float32)] * 2
    _close(wam.find_true_peak(x), O.true_peak_db(x) if n else float("-inf"), TP_TOL)
    _close(wam.measure_lufs(x, sr), O.measure_lufs(x, sr) if n else -14.0, LUFS_TOL)


def test_batch_meter_dev():
    torch = pytest.importorskip("torch")
    sr, n, nt = 48000, 60000, 5
    tracks = [O.synth(30 + t, 2, n, sr, {"amp": 0.5 + 0.3 * t}) for t in range(nt)]
    host = np.stack([np.stack(t) for t in tracks])          # [nt][2][n]
    dev = torch.from_numpy(host).cuda()
    l, t = wam.batch_meter_dev(dev.data_ptr(), nt, 2, n, sr)
    for k in range(nt):
        _close(l[k], O.measure_lufs(tracks[k], sr), LUFS_TOL)
        _close(t[k], O.true_peak_db(tracks[k]), TP_TOL)


def test_short_term_momentary(golden):
    """a4: measureShortTermLUFS / measureMomentaryLUFS against the reference goldens and the oracle (incl. truncated windows)."""
    c = golden["meter_10s_48k"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    for pos, v in c["short_term"]:
        got, want = wam.measure_short_term_lufs(x, c["sr"], pos), G.num(v)
        assert (got == want) if not np.isfinite(want) else abs(got - want) <= 1e-6
    for pos, v in c["momentary"]:
        got, want = wam.measure_momentary_lufs(x, c["sr"], pos), G.num(v)
        assert (got == want) if not np.isfinite(want) else abs(got - want) <= 1e-9
    for pos in (0.0, 1.234, 7.9, 8.4, 9.7, 9.85, 12.0):
        for fn, ofn in ((wam.measure_short_term_lufs, O.short_term_lufs), (wam.measure_momentary_lufs, O.momentary_lufs)):
            got, want = fn(x, c["sr"], pos), ofn(x, c["sr"], pos)
            assert (got == want) if not np.isfinite(want) else abs(got - want) <= 1e-6


def _series_cases():
    import json, os
    with open(os.path.join(G.DATA, "series.json")) as f:
        return json.load(f)["cases"]


def test_lufs_series(golden):
    """8f rank 4: the short-term / momentary series in one call equals the reference's per-position calls (tests/golden/gen_series.mjs)
    and the single-position entry points."""
    for name, c in _series_cases().items():
        if "positions" not in c:
            continue
        x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
        st = wam.lufs_series(x, c["sr"], c["positions"], momentary=False)
        mo = wam.lufs_series(x, c["sr"], c["positions"], momentary=True)
        for k, pos in enumerate(c["positions"]):
            _close(st[k], G.num(c["short_term"][k]), 1e-6)
            _close(mo[k], G.num(c["momentary"][k]), 1e-9)
        for k in (0, 3, len(c["positions"]) - 4):
            _close(st[k], wam.measure_short_term_lufs(x, c["sr"], c["positions"][k]), 1e-12)
            _close(mo[k], wam.measure_momentary_lufs(x, c["sr"], c["positions"][k]), 1e-12)
    # more positions than one gather group (256), unsorted, duplicates
    c = _series_cases()["series_12s_48k"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    rng = np.random.default_rng(5)
    pos = np.concatenate([rng.uniform(0, 12.5, 300), [0.0, 0.0, 9.0]])
    st = wam.lufs_series(x, c["sr"], pos)
    for k in (0, 17, 299, 300, 301, 302):
        _close(st[k], O.short_term_lufs(x, c["sr"], float(pos[k])), 1e-6)
    assert len(wam.lufs_series(x, c["sr"], [])) == 0
    with pytest.raises(wam.WamError):
        wam.lufs_series(x, c["sr"], [1.0, -0.5])


def test_extract_peaks(golden):
    """8f rank 4: waveform peaks identical to the reference's extractPeaks and to the oracle at full size."""
    for name, c in _series_cases().items():
        if "num_peaks" not in c:
            continue
        x = G.f32(c["input"]) if c["input"] else O.synth(c["idx"], c["nch"], c["n_synth"], c["sr"], c["opts"])[0][:c["n"]]
        assert np.array_equal(wam.extract_peaks(x, c["num_peaks"]), G.f32(c["peaks"])), name
    x = O.synth(3, 1, 8640000, 48000, {"clickAmp": 0.7})[0]
    assert np.array_equal(wam.extract_peaks(x, 800), O.extract_peaks(x, 800))
    assert np.array_equal(wam.extract_peaks(x[:777], 1000), np.zeros(1000, np.float32))
    assert len(wam.extract_peaks(x[:100], 0)) == 0
