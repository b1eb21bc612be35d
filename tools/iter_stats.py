"""Average policy-iteration counts per solve (needs tools/libwam_hip_count.so, a -DWAM_COUNT_ITERS build)."""
import sys, os, ctypes as C
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, ".")
import torch, wam
wam.LIB_PATH = os.path.abspath("tools/libwam_hip_count.so")
import bench
sr, n, T = 48000, 48000 * 180, 16
x = torch.empty((T, 2, n), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
bench.synth_dev(x, T, 0, bench.SYNTH, amp_step=0.02)
base = dict(normalizeLoudness=False, truePeakLimit=False, cleanLowEnd=False, glueCompression=False, deharsh=False, stereoWidth=100, centerBass=False,
            cutMud=False, addAir=False, tapeWarmth=False, addPunch=False, inputGain=0, targetLufs=-14, truePeakCeiling=-1,
            eqLow=0, eqLowMid=0, eqMid=0, eqHighMid=0, eqHigh=0, autoLevel=False)
def stats(tag):
    out = (C.c_uint64 * 8)()
    wam._check(wam.lib().wam_debug_counters(out, 1))
    names = ("follower", "follower pair", "release")
    if out[6] or out[7]:
        print("   pair histogram (<=2, 3, 4, >4 iterations):", out[6] >> 32, out[6] & 0xffffffff, out[7] >> 32, out[7] & 0xffffffff)
    print(tag, "; ".join(f"{names[s]}: {out[2*s+1] / out[2*s]:.2f} iters x {out[2*s]}" for s in range(3) if out[2 * s]))
stats("warmup")
for tag, over in (("deharsh", dict(deharsh=True)), ("punch", dict(addPunch=True)), ("glue", dict(glueCompression=True)),
                  ("limit (soft clip + limiter)", dict(truePeakLimit=True, inputGain=4)), ("full", None)):
    s = bench.C2_SETTINGS if over is None else dict(base, **over)
    wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, 2, n, sr, s, "export")
    stats(tag)
