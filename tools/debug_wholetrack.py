"""Localise a whole-track deviation from the oracle (tests/test_gpu_wholetrack.py): is it a seam residue (goes away when the
track is one segment) or arithmetic (stays), and which stage feeds it.  Runs on the GPU box: python tools/debug_wholetrack.py [case]"""
import sys
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
import wam
from test_gpu_wholetrack import CASES

name = sys.argv[1] if len(sys.argv) > 1 else "hot_180s_48k"
sr, n, over, opts = CASES[name]
x = O.synth(17, 2, n, sr, opts)
full = dict(wam.DEFAULT_SETTINGS, **over)


def report(tag, y, yo):
    d = np.stack([np.abs(a.astype(np.float64) - b.astype(np.float64)) for a, b in zip(y, yo)])
    m = d.max(axis=0)
    k = int(m.argmax())
    bad = np.nonzero(m > 1e-6)[0]
    print(f"{tag:28s} max|d| {m[k]:.3e} at {k} (tile {k // 4096}, t={k / sr:.3f}s)  >1e-6: {len(bad)}", flush=True)
    if len(bad):
        cl = np.split(bad, np.nonzero(np.diff(bad) > 20000)[0] + 1)
        for c in cl[:10]:
            print(f"      cluster {c[0]}..{c[-1]} ({len(c)} samples, tiles {c[0] // 4096}..{c[-1] // 4096}) max {m[c[0]:c[-1] + 1].max():.3e}")


def one_segment():
    for k in ("fx_seg_tiles", "meter_seg_tiles", "env_seg_tiles"):
        wam.tune(k, 1 << 20)
    wam.tune("dh_seg_frames", 1 << 24)


variants = {"all": {}, "-deharsh": dict(deharsh=False), "-punch": dict(addPunch=False), "-warmth": dict(tapeWarmth=False), "-air": dict(addAir=False),
            "-glue": dict(glueCompression=False), "-limiter": dict(truePeakLimit=False), "-fx": dict(deharsh=False, addPunch=False, tapeWarmth=False, addAir=False),
            "-bass": dict(centerBass=False), "-hpf": dict(cleanLowEnd=False), "-eq": dict(eqLow=0, eqLowMid=0, eqMid=0, eqHighMid=0, eqHigh=0)}
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(variants)
for v in only:
    s = dict(full, **variants[v])
    yo, ro = O.render_chain(x, sr, s, "export")
    wam.tune("reset")
    y, r = wam.render_full_chain(x, sr, s, "export")
    report(v + " default segs", y, yo)
    one_segment()
    y1, r1 = wam.render_full_chain(x, sr, s, "export")
    report(v + " one segment", y1, yo)
    print(f"      lufs_in gpu {r.lufs_in:.9f} one-seg {r1.lufs_in:.9f} oracle {ro.lufs_in:.9f}; gain {r.gain_db:.9f} {r1.gain_db:.9f} {ro.gain_db:.9f}")
wam.tune("reset")
