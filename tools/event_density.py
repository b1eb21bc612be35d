"""How often do the soft clipper and limiter engage on the bench input?  (fraction of samples over their thresholds)"""
import sys
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, ".")
import torch, wam, bench
sr, n, T = 48000, 48000 * 180, 8
x = torch.empty((T, 2, n), dtype=torch.float32, device="cuda")
bench.synth_dev(x, T, 0, bench.SYNTH, amp_step=0.3 / 1024 * 128)
print("input peak", float(x.abs().max()), "rms", float(x.pow(2).mean().sqrt()))
s = dict(bench.C2_SETTINGS, truePeakLimit=False, normalizeLoudness=False)
y = torch.empty_like(x)
res = wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, 2, n, sr, s, "export")      # everything before stage 8
s2 = dict(bench.C2_SETTINGS, truePeakLimit=False)
res2 = wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, 2, n, sr, s2, "export")    # + normalisation gain
thr = 10 ** ((-1 + 3) / 20); ceil = 10 ** (-1 / 20)
print("after stage 8: peak", float(y.abs().max()), "gain_db", [round(r.gain_db, 2) for r in res2][:4])
print("fraction |x| > soft-clip threshold %.3f:" % thr, float((y.abs() > thr).float().mean()))
print("fraction |x| > ceiling %.3f:" % ceil, float((y.abs() > ceil).float().mean()))
