import sys, os, ctypes as C
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, "tests")
import torch, numpy as np, wam
from test_gpu_fullsize import synth_dev
sr = 96000; n = sr * 600
x = torch.empty((1, 2, n), dtype=torch.float32, device="cuda")
synth_dev(x, 1, 17, sr, quiet=(sr * 100, sr * 150, 0.003))
base = dict(normalizeLoudness=False, truePeakLimit=False, cleanLowEnd=False, glueCompression=False, deharsh=False, stereoWidth=100, centerBass=False,
            cutMud=False, addAir=False, tapeWarmth=False, addPunch=False, inputGain=0, targetLufs=-14, truePeakCeiling=-1,
            eqLow=0, eqLowMid=0, eqMid=0, eqHighMid=0, eqHigh=0, autoLevel=False)
cases = {"all": None, "all-dh": dict(deharsh=False), "all-punch": dict(addPunch=False), "all-warm": dict(tapeWarmth=False), "all-glue": dict(glueCompression=False),
         "all-limit": dict(truePeakLimit=False), "all-norm": dict(normalizeLoudness=False), "all-air": dict(addAir=False)}
full = dict(wam.DEFAULT_SETTINGS, glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1, eqHigh=2)
for name, over in cases.items():
    s = full if over is None else dict(full, **over)
    wam.tune("reset")
    y1 = torch.empty_like(x); r1 = wam.batch_render_dev(x.data_ptr(), y1.data_ptr(), 1, 2, n, sr, s, "export")[0]
    for k in ("fx_seg_tiles", "meter_seg_tiles", "env_seg_tiles"): wam.tune(k, 1500)
    wam.tune("dh_seg_frames", 5000)
    y2 = torch.empty_like(x); r2 = wam.batch_render_dev(x.data_ptr(), y2.data_ptr(), 1, 2, n, sr, s, "export")[0]
    d = (y1 - y2).abs(); m = float(d.max()); idx = int(d.view(-1).argmax()) % n
    print(f"{name:8s} maxdiff {m:.3e} at sample {idx} (tile {idx // 4096}, t={idx / sr:.2f}s) lufs {r1.lufs:.6f}/{r2.lufs:.6f} gain {r1.gain_db:.9f}/{r2.gain_db:.9f} count>2e-7: {int((d > 2e-7).sum())}", flush=True)
    if name == "all":
        bad = torch.nonzero(d[0, 0] > 2e-7).view(-1).cpu().numpy()
        if len(bad):
            br = np.split(bad, np.nonzero(np.diff(bad) > 20000)[0] + 1)
            for c in br[:12]: print("   cluster", c[0], c[-1], len(c), "tiles", c[0] // 4096, c[-1] // 4096, "max", float(d[0, 0, c[0]:c[-1] + 1].max()))
