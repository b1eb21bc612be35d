"""Does running two render groups at once (two host threads, two streams) beat running them back to back?"""
import sys, time, threading
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, ".")
import torch, wam, bench
sr, n, T = 48000, 48000 * 180, 1024
x = torch.empty((T, 2, n), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
bench.synth_dev(x, T, 0, bench.SYNTH, amp_step=0.3 / T)
s = dict(bench.C2_SETTINGS)
wam.init(0)
esz = 2 * n * 4
def run(lo, hi, stream):
    wam.init(0)
    wam.batch_render_dev(x.data_ptr() + lo * esz, y.data_ptr() + lo * esz, hi - lo, 2, n, sr, s, "export", stream=stream)
run(0, 256, 0)
torch.cuda.synchronize(); t0 = time.perf_counter(); run(0, T, 0); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
ref = y[::97].clone()
st = [torch.cuda.Stream(), torch.cuda.Stream()]
for parts in (2, 4):
    y.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(k * T // parts, (k + 1) * T // parts, st[k % 2].cuda_stream)) for k in range(parts)]
    # two at a time
    for k in range(0, parts, 2):
        for t in th[k:k + 2]: t.start()
        for t in th[k:k + 2]: t.join()
    torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    print(f"one call: {t1 * 1e3:.0f} ms; {parts} parts, two at a time on two streams: {t2 * 1e3:.0f} ms; identical: {bool(torch.equal(ref, y[::97]))}", flush=True)
