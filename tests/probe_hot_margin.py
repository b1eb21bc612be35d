"""(Test infrastructure, run by hand: python tests/probe_hot_margin.py on a GPU box.)  Worst-case probe: long hot tracks through the full chain with de-harsh on, GPU vs CPU oracle (the soft clipper amplifies upstream differences)."""
import sys, time
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, wam
import oracle_lib as O
sr, n = 48000, 48000 * 30
for seed, over in ((60, dict(targetLufs=-9, truePeakCeiling=-0.3, inputGain=6)), (61, dict(targetLufs=-7, inputGain=3, glueCompression=True)),
                   (62, dict(glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1, eqHigh=2))):
    x = O.synth(seed, 2, n, sr, {"amp": 1.8, "clickAmp": 0.7})
    for dh in (False, True):
        s = dict(wam.DEFAULT_SETTINGS, deharsh=dh, **over)
        y, _ = wam.render_full_chain(x, sr, s, "export")
        yo, _ = O.render_chain(x, sr, s, "export")
        d = np.concatenate([np.abs(a.astype(np.float64) - b) for a, b in zip(y, yo)])
        print(f"seed {seed} deharsh {int(dh)}: max {d.max():.2e}, samples > 1e-6: {(d > 1e-6).sum()}, > 1e-5: {(d > 1e-5).sum()} of {d.size}", flush=True)
