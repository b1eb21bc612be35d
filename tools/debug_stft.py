"""wave-per-frame STFT kernel against the literal loop nest: where do they differ?  python tools/debug_stft.py"""
import sys
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
import wam
sr = 48000
def run(x, variant):
    wam.tune("stft_literal", variant)
    try:
        return wam.stft_db(x, sr, 4096, 1024)
    finally:
        wam.tune("reset")
x = O.synth(73, 2, 4096 + 1024 * 5 + 77, sr, {"amp": 1.2, "clickAmp": 0.5})
for name, xx in (("synth stereo", x), ("synth mono", [x[0]])):
    a, b, c = run(xx, 0), run(xx, 1), run(xx, 2)
    d = np.abs(a.astype(np.float64) - b)
    print(name, "wave vs literal: max", d.max(), "differing", int((a != b).sum()), "of", a.size, "| blocked vs literal differing", int((c != b).sum()))
    if (a != b).any():
        fr, bn = np.nonzero(a != b)
        print("   frames", np.unique(fr)[:10], "bins (first 20)", bn[:20], "bin mod 64 hist", np.bincount(bn % 64, minlength=64)[:16], "bin//64 hist", np.bincount(bn // 64, minlength=32))
        print("   sample values wave", a[fr[0], bn[0]], "literal", b[fr[0], bn[0]])
for pos in (0, 1, 64, 65, 2048, 4095):
    imp = np.zeros(4096, np.float32); imp[pos] = 1.0
    a, b = run([imp], 0), run([imp], 1)
    print("impulse at", pos, "max diff", np.abs(a.astype(np.float64) - b).max(), "differing", int((a != b).sum()))
