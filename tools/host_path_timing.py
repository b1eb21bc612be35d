"""PCIe-inclusive timing of the host-pointer entry points on one 3-min stereo 48 kHz track."""
import sys, time
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, ".")
import torch, numpy as np, wam
import bench
sr, n = 48000, 48000 * 180
xd = torch.empty((1, 2, n), dtype=torch.float32, device="cuda")
bench.synth_dev(xd, 1, 3, {"amp": 1.6, "clickAmp": 0.6})          # the library's own generator (csrc/synth.hip), copied to pageable host memory
x = [xd[0, 0].cpu().numpy(), xd[0, 1].cpu().numpy()]
s = dict(wam.DEFAULT_SETTINGS, glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1, eqHigh=2)
wam.init(0)
import os
pinned = int(os.environ.get("WAM_STAGE_PINNED", "1"))
wam.tune("stage_pinned", pinned)          # 0 = hipMemcpyAsync on the caller's pageable memory (round 2), 1 = the pinned ring of staging.hip
print(f"stage_pinned = {pinned}")
for name, fn in (("render_full_chain", lambda: wam.render_full_chain(x, sr, s, "export")), ("meter", lambda: wam.meter(x, sr)),
                 ("encode_pcm24", lambda: wam.encode_pcm24(x))):
    fn()
    t0 = time.perf_counter(); k = 5
    for _ in range(k): fn()
    dt = (time.perf_counter() - t0) / k
    print(f"{name}: {dt * 1e3:.1f} ms per 3-min track incl. H2D/D2H from pageable host memory = {180 / dt:.0f} x realtime")

# the host-pointer batch (two staging threads, copies of one group overlap the other group's kernels)
T = 128
xb = torch.empty((T, 2, n), dtype=torch.float32, device="cuda")
bench.synth_dev(xb, T, 0, {"amp": 1.6, "clickAmp": 0.6})
host = xb.cpu().numpy()
del xb
tracks = [[host[t, 0], host[t, 1]] for t in range(T)]
wam.batch_render(tracks[:4], sr, s, "export")
t0 = time.perf_counter()
wam.batch_render(tracks, sr, s, "export")
dt = time.perf_counter() - t0
print(f"batch_render: {T} tracks in {dt * 1e3:.0f} ms incl. H2D/D2H from pageable host memory = {T * 180 / dt:.0f} x realtime")
outs = [np.ones(n, dtype=np.float32) for _ in range(2 * T)]           # output arrays a caller reuses: written once, so their pages exist (np.zeros would map them lazily)
t0 = time.perf_counter()
wam.batch_render(tracks, sr, s, "export", outs=outs)
dt = time.perf_counter() - t0
print(f"batch_render into reused output arrays: {T} tracks in {dt * 1e3:.0f} ms = {T * 180 / dt:.0f} x realtime")

if os.environ.get("WAM_HOST_NT"):
    for nt in (0, 1, 0, 1, 0, 1):
        wam.tune("stage_nt", nt)
        t0 = time.perf_counter()
        wam.batch_render(tracks, sr, s, "export")
        dt = time.perf_counter() - t0
        print(f"stage_nt {nt}: {T} tracks in {dt * 1e3:.0f} ms = {T * 180 / dt:.0f} x realtime", flush=True)
    for nt in (0, 1, 0, 1):
        wam.tune("stage_nt", nt)
        t0 = time.perf_counter(); k = 5
        for _ in range(k): wam.render_full_chain(x, sr, s, "export")
        dt = (time.perf_counter() - t0) / k
        print(f"stage_nt {nt}: render_full_chain {dt * 1e3:.1f} ms", flush=True)
if os.environ.get("WAM_HOST_SWEEP"):
    for hg in (8, 16, 32):
        for th in (4, 8):
            wam.tune("host_group", hg); wam.tune("stage_threads", th)
            wam.batch_render(tracks[:hg], sr, s, "export")
            res = []
            for reuse in (False, True, False, True, False, True):
                t0 = time.perf_counter()
                wam.batch_render(tracks, sr, s, "export", outs=outs if reuse else None)
                res.append((time.perf_counter() - t0) * 1e3)
            print(f"host_group {hg:3d} stage_threads {th}: fresh outputs {min(res[0::2]):.0f} ms (min of 3; all {[round(r) for r in res[0::2]]}), reused {min(res[1::2]):.0f} ms ({[round(r) for r in res[1::2]]})"
                  f" = {T * 180 / min(res[0::2]) * 1e3:.0f} / {T * 180 / min(res[1::2]) * 1e3:.0f} x realtime", flush=True)
