import sys
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, wam
import oracle_lib as O
sr, n = 48000, 48000 * 30
x = O.synth(61, 2, n, sr, {"amp": 1.8, "clickAmp": 0.7})
for tag, over in (("hot, no glue", dict(targetLufs=-7, inputGain=3)), ("hot, glue", dict(targetLufs=-7, inputGain=3, glueCompression=True)),
                  ("hot, glue, no fx", dict(targetLufs=-7, inputGain=3, glueCompression=True, addAir=False, tapeWarmth=False, addPunch=False)),
                  ("hot, no glue, no fx", dict(targetLufs=-7, inputGain=3, addAir=False, tapeWarmth=False, addPunch=False)),
                  ("hot, only punch", dict(targetLufs=-7, inputGain=3, addAir=False, tapeWarmth=False)),
                  ("hot, only warmth", dict(targetLufs=-7, inputGain=3, addAir=False, addPunch=False)),
                  ("hot, only air", dict(targetLufs=-7, inputGain=3, tapeWarmth=False, addPunch=False))):
    s = dict(wam.DEFAULT_SETTINGS, deharsh=False, **over)
    y, _ = wam.render_full_chain(x, sr, s, "export")
    yo, _ = O.render_chain(x, sr, s, "export")
    d = np.concatenate([np.abs(a.astype(np.float64) - b) for a, b in zip(y, yo)])
    print(f"{tag}: max {d.max():.2e}, > 1e-6: {(d > 1e-6).sum()}, > 1e-5: {(d > 1e-5).sum()}", flush=True)
g = wam.apply_glue_compressor(x, sr) if hasattr(wam, "apply_glue_compressor") else None
