"""Where in the tile do the LTI stages' f32 flips against the oracle sit?  python tools/flip_where.py"""
import sys
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
import wam

sr, n = 48000, 48000 * 60
x = O.synth(17, 2, n, sr, {"amp": 3.2, "clickAmp": 0.6})


def where(tag, g, o):
    g = np.stack(g); o = np.stack(o)
    ne = (g != o)
    print(f"{tag:18s} differing {int(ne.sum()):8d} ({ne.mean():.2e}) max|d| {np.abs(g.astype(np.float64) - o).max():.3e}")
    idx = np.nonzero(ne[0])[0]
    if len(idx):
        print("    by sample-in-chunk:", np.bincount(idx % 16, minlength=16).tolist())
        th = (idx % 4096) // 16
        print("    by wave:", np.bincount(th // 64, minlength=4).tolist(), " lanes 0..7 of a wave:", np.bincount(th % 64, minlength=64)[:8].tolist(),
              " lanes 56..63:", np.bincount(th % 64, minlength=64)[56:].tolist())
        print("    first differing samples:", idx[:12].tolist())


where("hpf only", wam.apply_final_filters(x, sr, True, False), O.final_filters(x, sr, True, False))
where("lpf only", wam.apply_final_filters(x, sr, False, True), O.final_filters(x, sr, False, True))
for k, gains in enumerate(([1, 0, 0, 0, 0], [0, -2, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, -1, 0], [0, 0, 0, 0, 2])):
    where(f"eq band {k}", wam.apply_parametric_eq(x, sr, gains, False), O.parametric_eq(x, sr, gains, False))
where("stereo", wam.apply_stereo(x, sr, 1.0, True, 200.0), O.stereo(x, sr, 1.0, True, 200.0))
where("warmth", wam.apply_tape_warmth(x, sr), O.warmth(x, sr))
where("punch", wam.apply_multiband_transient(x, sr), O.punch(x, sr))
where("exciter", wam.apply_exciter(x, sr), O.exciter(x, sr))
