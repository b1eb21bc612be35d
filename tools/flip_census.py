"""Per-stage census of f32 rounding flips over a WHOLE track: every stage is given the oracle's own input for that stage, on
the GPU and on the oracle, and the outputs are compared sample by sample (identical / one ulp / more).  Shows which stages
leak differences into the mastering soft clipper, whose knee multiplies them by ~600 (tests/test_gpu_wholetrack.py).
python tools/flip_census.py [case]"""
import sys
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
import wam
from test_gpu_wholetrack import CASES

name = sys.argv[1] if len(sys.argv) > 1 else "hot_180s_48k"
sr, n, over, opts = CASES[name]
x = O.synth(17, 2, n, sr, opts)
s = dict(wam.DEFAULT_SETTINGS, **over)


def census(tag, g, o):
    g = np.stack(g); o = np.stack(o)
    d = np.abs(g.astype(np.float64) - o.astype(np.float64))
    ne = int((g != o).sum())
    ulp = np.spacing(np.maximum(np.abs(o), np.float32(1e-30)).astype(np.float32)).astype(np.float64)
    many = int((d > 1.5 * ulp).sum())
    print(f"{tag:14s} differing samples {ne:8d} of {g.size} ({ne / g.size:.2e}), more than one ulp: {many:6d}, max|d| {d.max():.3e}", flush=True)


gain = 10 ** (s["inputGain"] / 20)
cur = [(c.astype(np.float64) * gain).astype(np.float32) for c in x]          # dsp-worker.js:1493-1503
for tag, fg, fo in (("deharsh", lambda c: wam.process_hybrid_dynamic(c, sr, "mastering"), lambda c: O.deharsh(c, sr)),
                    ("exciter", lambda c: wam.apply_exciter(c, sr), lambda c: O.exciter(c, sr)),
                    ("warmth", lambda c: wam.apply_tape_warmth(c, sr), lambda c: O.warmth(c, sr)),
                    ("punch", lambda c: wam.apply_multiband_transient(c, sr), lambda c: O.punch(c, sr)),
                    ("final filters", lambda c: wam.apply_final_filters(c, sr, True, True), lambda c: O.final_filters(c, sr, True, True)),
                    ("eq", lambda c: wam.apply_parametric_eq(c, sr, [1, -2, 1, -1, 2], False), lambda c: O.parametric_eq(c, sr, [1, -2, 1, -1, 2], False)),
                    ("glue", lambda c: wam.apply_glue_compressor(c, sr), lambda c: O.glue(c, sr)),
                    ("stereo", lambda c: wam.apply_stereo(c, sr, 1.0, True, 200.0), lambda c: O.stereo(c, sr, 1.0, True, 200.0))):
    g, o = fg(cur), fo(cur)
    census(tag, g, o)
    cur = o
lufs = O.measure_lufs(cur, sr)
gdb = s["targetLufs"] - lufs
print(f"normalise: oracle lufs {lufs:.9f} gpu {wam.measure_lufs(cur, sr):.9f}")
g, o = wam.apply_gain(cur, gdb), [(c.astype(np.float64) * 10 ** (gdb / 20)).astype(np.float32) for c in cur]
census("gain", g, o)
cur = o
ceil_db = s["truePeakCeiling"]
g, o = wam.apply_mastering_soft_clip(cur, sr, ceil_db), O.soft_clip(cur, sr, ceil_db)
census("soft clip", g, o)
thr = 10 ** ((ceil_db + 3) / 20)
a = np.abs(np.stack(cur).astype(np.float64))
print(f"soft-clip threshold {thr:.6f}: samples above {int((a > thr).sum())}, within 1e-3 above {int(((a > thr) & (a < thr + 1e-3)).sum())}, within 1e-4 above {int(((a > thr) & (a < thr + 1e-4)).sum())}")
cur = o
g, o = wam.apply_lookahead_limiter(cur, sr, 10 ** (ceil_db / 20)), O.limiter(cur, sr, 10 ** (ceil_db / 20))
census("limiter", g, o)
