"""(Test infrastructure, run by hand on a GPU box.)  How far is the de-harsh stage from bit-faithful?  Share of output samples that
equal the oracle's exactly, and the error distribution of the rest - default (float32) kernels and the exact mode."""
import sys, time
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, wam
import oracle_lib as O
sr, n = 48000, 48000 * 10
x = O.synth(60, 2, n, sr, {"amp": 1.8, "clickAmp": 0.7})
yo = O.deharsh(x, sr)
for exact in (0, 1):
    wam.tune("dh_exact", exact)
    y = wam.process_hybrid_dynamic(x, sr, "mastering")
    t0 = time.perf_counter(); y = wam.process_hybrid_dynamic(x, sr, "mastering"); dt = time.perf_counter() - t0
    for c in range(2):
        d = np.abs(y[c].astype(np.float64) - yo[c])
        ulp = np.spacing(np.abs(yo[c]).astype(np.float32)).astype(np.float64)
        print(f"exact={exact} ch{c}: identical {np.mean(d == 0) * 100:.4f} %, within 1 ulp {np.mean(d <= ulp) * 100:.4f} %, max {d.max():.2e} ({(d / ulp).max():.1f} ulp)  [{dt * 1e3:.1f} ms host call]")
for over in (dict(targetLufs=-9, truePeakCeiling=-0.3, inputGain=6), dict(targetLufs=-7, inputGain=3, glueCompression=True)):
    s = dict(wam.DEFAULT_SETTINGS, deharsh=True, **over)
    yo, _ = O.render_chain(x, sr, s, "export")
    for exact in (0, 1):
        wam.tune("dh_exact", exact)
        y, _ = wam.render_full_chain(x, sr, s, "export")
        d = np.concatenate([np.abs(a.astype(np.float64) - b) for a, b in zip(y, yo)])
        print(f"chain {over} exact={exact}: max {d.max():.2e}, > 1e-5: {(d > 1e-5).mean():.2e}, > 1e-6: {(d > 1e-6).mean():.2e}")
wam.tune("dh_exact", 0)
wam.tune("reset")
