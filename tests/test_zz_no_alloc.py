"""wam_batch_meter_dev_noalloc (include/wam.h): the batch meter for callers that may not allocate - fixed per-GPU scratch
array, results into the caller's device arrays - against the existing entry point wam_batch_meter_dev on the same inputs,
on two calls in a row (the second smaller: the shared scratch keeps the first call's contents) and then a larger one again."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
wam = pytest.importorskip("wam")
torch = pytest.importorskip("torch")


class SynthOpts(C.Structure):
    _fields_ = [("amp", C.c_double), ("dc", C.c_double), ("click_amp", C.c_double),
                ("quiet_from", C.c_int64), ("quiet_to", C.c_int64), ("quiet_gain", C.c_double),
                ("quiet2_from", C.c_int64), ("quiet2_to", C.c_int64), ("quiet2_gain", C.c_double)]


def synth(T, nch, n, sr, idx0, quiet=None):
    x = torch.empty((T, nch, n), dtype=torch.float32, device="cuda")
    q = quiet or (-1, -1, 0.0)
    o = SynthOpts(1.1, 0.0, 0.5, q[0], q[1], q[2], -1, -1, 0.0)
    wam._check(wam.lib().wam_synth_dev(C.c_void_p(x.data_ptr()), C.c_int(T), C.c_int(nch), C.c_int64(n), C.c_double(sr), C.c_int(idx0), C.byref(o),
                                       C.c_double(0.01), C.c_void_p(0)))
    return x


def noalloc(x, sr, fallback=-14.0):
    T, nch, n = x.shape
    lufs = torch.full((T,), float("nan"), dtype=torch.float64, device="cuda")
    dbtp = torch.full((T,), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    wam._check(wam.lib().wam_batch_meter_dev_noalloc(C.c_void_p(x.data_ptr()), C.c_int(T), C.c_int(nch), C.c_int64(n), C.c_double(sr), C.c_double(fallback),
                                                     C.c_void_p(lufs.data_ptr()), C.c_void_p(dbtp.data_ptr()), C.c_void_p(0)))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0              # the device's free memory did not move: nothing was allocated
    return lufs.cpu().numpy(), dbtp.cpu().numpy()


def check(x, sr):
    T, nch, n = x.shape
    want_l, want_p = wam.batch_meter_dev(x.data_ptr(), T, nch, n, sr)
    got_l, got_p = noalloc(x, sr)
    for a, b in list(zip(got_l, want_l)) + list(zip(got_p, want_p)):
        if np.isinf(b):
            assert a == b
        else:
            assert abs(a - b) <= 1e-9, (a, b)                  # same kernels; only log10 runs on the device instead of the host (bar: 0.01)


def test_noalloc_meter_matches_batch_meter_twice():
    wam.init(0)
    sr = 48000
    check(synth(40, 2, 150000, sr, 500, quiet=(40000, 90000, 0.002)), sr)       # gates act
    check(synth(7, 1, 30011, sr, 600), sr)                                         # smaller, mono, ragged: stale scratch beyond it
    check(synth(3, 2, 9000, sr, 610), sr)                                          # shorter than one block: the fallback value
    check(synth(64, 2, 100000, 44100, 620), 44100)                                 # and larger again, another rate
    z = torch.zeros((2, 2, 50000), dtype=torch.float32, device="cuda")
    l, p = noalloc(z, sr)
    assert np.all(np.isneginf(l)) and np.all(np.isneginf(p))                       # silence


def test_noalloc_calls_on_two_streams_do_not_race():
    """Two calls back to back on two non-blocking streams, no host synchronisation between them: both carve the same per-GPU
    scratch array, so the second must wait on the device for the first (the per-device event of noalloc.hip).  Without that
    order the second call's plan and cleared hop sums overwrite the first's while its k_meter / k_gate are running."""
    wam.init(0)
    sr = 48000
    a = synth(96, 2, 48000 * 20, sr, 700, quiet=(200000, 500000, 0.002))           # ~10 ms of kernels: still running when the next call is enqueued
    b = synth(5, 2, 70001, sr, 800)
    want = [wam.batch_meter_dev(x.data_ptr(), x.shape[0], x.shape[1], x.shape[2], sr) for x in (a, b)]
    for rep in range(3):
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        torch.cuda.synchronize()
        for x, st in ((a, s1), (b, s2), (a, s2), (b, s1)):
            T, nch, n = x.shape
            lufs = torch.full((T,), float("nan"), dtype=torch.float64, device="cuda")
            dbtp = torch.full((T,), float("nan"), dtype=torch.float64, device="cuda")
            outs.append((lufs, dbtp))
        torch.cuda.synchronize()
        for (x, st), (lufs, dbtp) in zip(((a, s1), (b, s2), (a, s2), (b, s1)), outs):
            T, nch, n = x.shape
            wam._check(wam.lib().wam_batch_meter_dev_noalloc(C.c_void_p(x.data_ptr()), C.c_int(T), C.c_int(nch), C.c_int64(n), C.c_double(sr), C.c_double(-14.0),
                                                             C.c_void_p(lufs.data_ptr()), C.c_void_p(dbtp.data_ptr()), C.c_void_p(st.cuda_stream)))
        torch.cuda.synchronize()
        for k, (lufs, dbtp) in enumerate(outs):
            wl, wp = want[k & 1]
            assert np.max(np.abs(lufs.cpu().numpy() - wl)) <= 1e-9, (rep, k)
            assert np.max(np.abs(dbtp.cpu().numpy() - wp)) <= 1e-9, (rep, k)


def test_noalloc_meter_rejects_what_does_not_fit():
    wam.init(0)
    x = torch.zeros((1, 1, 16), dtype=torch.float32, device="cuda")
    lufs = torch.zeros(1, dtype=torch.float64, device="cuda")
    rc = wam.lib().wam_batch_meter_dev_noalloc(C.c_void_p(x.data_ptr()), C.c_int(200000), C.c_int(2), C.c_int64(48000 * 180), C.c_double(48000.0), C.c_double(-14.0),
                                               C.c_void_p(lufs.data_ptr()), C.c_void_p(0), C.c_void_p(0))
    assert rc == -6 and b"fixed buffer holds" in wam.lib().wam_last_error()       # WAM_ERR_TOO_LARGE, before anything is launched
