"""k_front's time per 256-track group with stages switched off one at a time (settings variants), to see which phase carries the
stalls.  python tools/front_phases.py"""
import sys
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, ".")
import torch, wam
import bench
T, sr, n = 256, 48000, 48000 * 180
wam.init(0)
x = torch.empty((T, 2, n), dtype=torch.float32, device="cuda")
bench.synth_dev(x, T, 0, {"amp": 1.6, "clickAmp": 0.6})
y = torch.empty_like(x)
C2 = dict(wam.DEFAULT_SETTINGS, glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1, eqHigh=2)
variants = [("all on", {}), ("no glue", {"glueCompression": False}), ("no EQ", {"eqLow": 0, "eqLowMid": 0, "eqMid": 0, "eqHighMid": 0, "eqHigh": 0}),
            ("no normalise (no fused meter)", {"normalizeLoudness": False}), ("no stereo stage", {"centerBass": False}), ("no final filters", {"cleanLowEnd": False})]
for name, over in variants:
    s = dict(C2, **over)
    wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, 2, n, sr, s, "export")
    wam.lib().wam_profile_enable(1); bench.profile_read()
    for _ in range(2):
        wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, 2, n, sr, s, "export")
    torch.cuda.synchronize()
    p = bench.profile_read(); wam.lib().wam_profile_enable(0)
    c, ms = p["k_front"]
    print(f"{name:32s} k_front {ms / c:7.2f} ms", flush=True)
