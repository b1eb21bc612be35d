"""GPU parity of the callers / data formats either side of the chain: DC offset, NORMALIZE
message, 24-bit PCM quantiser + WAV header, spectrogram STFT."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
wam = pytest.importorskip("wam")


def maxerr(a, b):
    a = np.stack([np.asarray(x, dtype=np.float64) for x in a])
    b = np.stack([np.asarray(x, dtype=np.float64) for x in b])
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def test_dc_offset(golden):
    c = golden["stages_hot"]
    x = O.synth(c["idx"], 2, c["n"], c["sr"], c["opts"])
    dc = wam.dc_offset(x)
    want = [G.num(v) for v in c["stages"]["dc"]["detect"]]
    assert np.allclose(dc, want, rtol=0, atol=1e-15)
    dc2, y = wam.dc_offset(x, remove=True)
    assert maxerr(y, G.chans(c["stages"]["dc"]["out"])) <= 6e-8          # bit-exact unless the mean differs in its last bit
    z = [np.full(1000, 1e-12, np.float32)]
    dcz, yz = wam.dc_offset(z, remove=True)                                # |dc| <= 1e-10: left alone (dc-offset.js:70)
    assert np.array_equal(yz[0], z[0])


def test_pcm24(golden):
    S = golden["stages_hot"]["stages"]
    lim = G.chans(S["limiter"]["out"])
    assert np.array_equal(wam.encode_pcm24(lim), G.u8(S["pcm24"]["out"]))          # identical 24-bit words
    e = golden["pcm24_edges"]
    assert np.array_equal(wam.encode_pcm24(G.chans(e["in"])), G.u8(e["out"]))
    assert bytes(wam.wav_header(2, golden["stages_hot"]["n"], 48000, 24)).hex() == S["pcm24"]["header_hex"]
    assert bytes(wam.wav_header(2, golden["stages_hot"]["n"], 48000, 16)).hex() == S["pcm16_none"]["header_hex"]
    for n in [1, 2, 3, 5, 1023]:
        for nch in [1, 2]:
            x = O.synth(70, nch, n, 48000, {"amp": 3.0})
            assert np.array_equal(wam.encode_pcm24(x), O.pcm24(x))


@pytest.mark.parametrize("name", ["spectrogram_2048", "spectrogram_4096"])
def test_stft(golden, name):
    c = golden[name]
    x = O.synth(c["idx"], 2, c["n"], c["sr"], c["opts"])
    db, cols = wam.stft_db(x, c["sr"], c["nfft"], c["hop"], columns=True)
    want = G.f32(c["spectrum_db"]).reshape(db.shape)
    # same radix-2 butterflies with the same f32 stores: only log10's last bit can differ
    assert float(np.max(np.abs(db.astype(np.float64) - want))) <= 2e-5
    assert float(np.max(np.abs(cols.astype(np.float64) - G.f32(c["columns"]).reshape(cols.shape)))) <= 2e-5
    assert float(np.mean(db == want)) > 0.99


def test_stft_batch_vs_oracle():
    sr = 48000
    x = O.synth(71, 2, 4096 + 1024 * 40 + 123, sr, {"amp": 1.0, "clickAmp": 0.4})
    db = wam.stft_db(x, sr, 4096, 1024)
    want = O.stft_db(x, sr, 4096, 1024)
    assert db.shape == want.shape == (41, 2048)
    assert float(np.max(np.abs(db.astype(np.float64) - want))) <= 2e-5
    mono = wam.stft_db([x[0]], sr, 2048, 512)
    assert float(np.max(np.abs(mono.astype(np.float64) - O.stft_db([x[0]], sr, 2048, 512)))) <= 2e-5


@pytest.mark.parametrize("amp,target,ceiling", [(1.6, -14.0, -1.0), (0.2, -14.0, -1.0), (1.0, -20.0, -1.0), (0.0, -14.0, -1.0)])
def test_normalize_message(amp, target, ceiling):
    sr = 48000
    x = O.synth(72, 2, 60000, sr, {"amp": amp, "clickAmp": 0.5 * (amp > 0)})
    y, r = wam.normalize(x, sr, target, ceiling)
    yo, ro = O.normalize(x, sr, target, ceiling)
    assert r.limiter_applied == ro.limiter_applied and r.measured == ro.measured
    assert maxerr(y, yo) <= 1e-5
    for a, b in [(r.current_lufs, ro.current_lufs), (r.final_lufs, ro.final_lufs), (r.peak_db, ro.peak_db), (r.gain_applied_db, ro.gain_applied_db)]:
        if np.isinf(b):
            assert a == b
        else:
            assert abs(a - b) <= 0.01


def test_batch_stft_dev_matches_single():
    """configs[4] entry point: device-resident batch == per-track host call (same kernel, track dimension added)."""
    import ctypes as C
    import torch
    sr, n, T = 48000, 20000, 3
    tracks = [O.synth(60 + t, 2, n, sr, {"amp": 0.8}) for t in range(T)]
    x = torch.tensor(np.stack([np.stack(t) for t in tracks]), device="cuda")
    nfr = (n - 4096) // 1024 + 1
    out = torch.empty((T, nfr, 2048), dtype=torch.float32, device="cuda")
    cols = torch.empty((T, nfr, 256), dtype=torch.float32, device="cuda")
    wam._check(wam.lib().wam_batch_stft_db_dev(C.c_void_p(x.data_ptr()), C.c_int(T), C.c_int(2), C.c_int64(n), C.c_double(sr), C.c_int(4096), C.c_int(1024),
                                               C.c_void_p(out.data_ptr()), C.c_void_p(cols.data_ptr()), C.c_void_p(0)))
    for t in range(T):
        ref, rcols = O.stft_db(tracks[t], sr, 4096, 1024, columns=True)
        assert np.max(np.abs(out[t].cpu().numpy() - ref)) <= 2e-5          # dB; same bar as test_stft
        assert np.max(np.abs(cols[t].cpu().numpy() - rcols)) <= 2e-5


@pytest.mark.parametrize("nfft,hop", [(4096, 1024), (2048, 512)])
def test_stft_register_blocked_is_bit_identical(nfft, hop):
    """k_stft_fast regroups the reference's radix-2 stages but performs the same butterflies with the same f32 stores."""
    sr = 48000
    x = O.synth(73, 2, nfft + hop * 25 + 77, sr, {"amp": 1.2, "clickAmp": 0.5})
    try:
        fast, fcols = wam.stft_db(x, sr, nfft, hop, columns=True)
        wam.tune("stft_literal", 1)
        lit, lcols = wam.stft_db(x, sr, nfft, hop, columns=True)
    finally:
        wam.tune("reset")
    assert np.array_equal(fast, lit) and np.array_equal(fcols, lcols)
    mono_f = wam.stft_db([x[1]], sr, nfft, hop)
    wam.tune("stft_literal", 1)
    try:
        assert np.array_equal(mono_f, wam.stft_db([x[1]], sr, nfft, hop))
    finally:
        wam.tune("reset")


def test_pcm16_modes_vs_reference_and_oracle():
    """16-bit encoder branches through the C ABI: identical bytes to the reference (seeded Math.random goldens) and, on a
    longer input that spans many noise-shaping runs, to the sequential oracle."""
    import json, os
    MODE = {"none": 0, "tpdf": 1, "noise-shaped": 2}
    with open(os.path.join(G.DATA, "pcm16.json")) as f:
        cases = json.load(f)["cases"]
    for name, c in cases.items():
        x = G.chans(c["input"])
        d = None if c["mode"] == "none" else G.f32(c["dither"])[: c["nch"] * c["n"]]
        got = wam.encode_pcm16(x, MODE[c["mode"]], d)
        assert np.array_equal(got, G.u8(c["out"])), name
    rng = np.random.default_rng(5)
    n = 70001
    x = O.synth(74, 2, n, 48000, {"amp": 1.9, "clickAmp": 0.8})
    x[0][1000:1100] = np.nan                                             # NaN input: zeros out, and poisons the noise shaper's channel from there on
    d = ((rng.random(2 * n) + rng.random(2 * n)) - 1).astype(np.float32)
    for mode in (0, 1, 2):
        assert np.array_equal(wam.encode_pcm16(x, mode, d), O.pcm16(x, mode, d)), mode
    # internal generator: TPDF on (-1, 1): mean 0, variance 1/6, and it changes with the seed
    z = [np.zeros(200000, dtype=np.float32)]
    a = wam.encode_pcm16(z, 1, None, seed=1).view(np.int16).astype(np.float64)
    b = wam.encode_pcm16(z, 1, None, seed=2).view(np.int16).astype(np.float64)
    assert set(np.unique(a)) <= {-1.0, 0.0, 1.0} and not np.array_equal(a, b)
    assert abs(a.mean()) < 0.01 and abs(np.mean(a == 0) - 0.75) < 0.01   # P(|d| < 0.5) = 0.75 for a triangular density
