"""Batches larger than one render group (csrc/chain_api.hip render_batch_dev: groups of 256 tracks, each with its own
offsets into in / out / results): 601 short tracks - a count that is not a multiple of 256 - through the device-resident
batch entry points; the tracks either side of every group boundary must equal their single-track renders bit for bit
(same kernels, same segmentation at this length), agree with the oracle to the parity bar, and carry their own results."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
wam = pytest.importorskip("wam")
torch = pytest.importorskip("torch")

PCM_TOL = 1e-5
EDGE_TRACKS = (0, 1, 254, 255, 256, 257, 511, 512, 513, 599, 600)
T, SR, IDX0, AMP0, STEP, CLICK = 601, 48000, 300, 1.2, 0.001, 0.5


class SynthOpts(C.Structure):
    _fields_ = [("amp", C.c_double), ("dc", C.c_double), ("click_amp", C.c_double),
                ("quiet_from", C.c_int64), ("quiet_to", C.c_int64), ("quiet_gain", C.c_double),
                ("quiet2_from", C.c_int64), ("quiet2_to", C.c_int64), ("quiet2_gain", C.c_double)]


def synth_batch(nch, n):
    x = torch.empty((T, nch, n), dtype=torch.float32, device="cuda")
    o = SynthOpts(AMP0, 0.0, CLICK, -1, -1, 0.0, -1, -1, 0.0)
    wam._check(wam.lib().wam_synth_dev(C.c_void_p(x.data_ptr()), C.c_int(T), C.c_int(nch), C.c_int64(n), C.c_double(SR), C.c_int(IDX0), C.byref(o),
                                       C.c_double(STEP), C.c_void_p(0)))
    return x


def host_track(x, t):
    """Track t on the host; also checks that the device generator and the oracle's agree (same seed = global index, same level)."""
    h = [x[t, c].cpu().numpy() for c in range(x.shape[1])]
    ref = O.synth(IDX0 + t, x.shape[1], x.shape[2], SR, {"amp": float(np.float64(AMP0) + np.float64(STEP) * t), "clickAmp": CLICK})
    assert all(np.array_equal(a, b) for a, b in zip(h, ref)), t
    return h


@pytest.mark.parametrize("nch,mode", [(2, "export"), (1, "export"), (2, "preview")])
def test_render_batch_crosses_group_boundaries(nch, mode):
    n = 20000
    s = dict(wam.DEFAULT_SETTINGS, glueCompression=True, eqLow=1, eqHigh=2)
    x = synth_batch(nch, n)
    y = torch.full_like(x, float("nan"))
    res = wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, nch, n, SR, s, mode)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())                                   # every track of every group was written
    for t in EDGE_TRACKS:
        h = host_track(x, t)
        ys, r = wam.render_full_chain(h, SR, s, mode)
        got = y[t].cpu().numpy()
        assert all(np.array_equal(got[c], ys[c]) for c in range(nch)), t
        for k in ("lufs", "measured_lufs", "true_peak_db", "gain_db"):
            assert getattr(res[t], k) == getattr(r, k), (t, k)
        assert res[t].lufs_in == r.lufs_in or (np.isnan(res[t].lufs_in) and np.isnan(r.lufs_in))
        yo, ro = O.render_chain(h, SR, s, mode)
        assert float(np.max(np.abs(got.astype(np.float64) - np.stack(yo)))) <= PCM_TOL, t
        assert abs(res[t].lufs - ro.lufs) <= 0.01 and abs(res[t].true_peak_db - ro.true_peak_db) <= 0.01
    # all 601 result rows belong to their own track: the output metered on its own gives the numbers the chain reported
    if mode == "export":
        lufs, dbtp = wam.batch_meter_dev(y.data_ptr(), T, nch, n, SR)
        assert np.max(np.abs(np.array(lufs) - np.array([r.lufs for r in res]))) <= 1e-9
        assert np.max(np.abs(np.array(dbtp) - np.array([r.true_peak_db for r in res]))) <= 1e-9


def test_meter_batch_crosses_group_boundaries():
    n = 30000
    x = synth_batch(2, n)
    lufs, dbtp = wam.batch_meter_dev(x.data_ptr(), T, 2, n, SR)
    assert len(set(np.round(np.array(lufs), 9))) > T // 2                    # distinct tracks, distinct numbers
    for t in EDGE_TRACKS:
        h = host_track(x, t)
        l1, p1 = wam.meter(h, SR)
        assert lufs[t] == l1 and dbtp[t] == p1, t
        assert abs(lufs[t] - O.measure_lufs(h, SR)) <= 1e-6 and abs(dbtp[t] - O.true_peak_db(h)) <= 1e-9


def test_stft_batch_crosses_group_boundaries():
    nfft, hop = 4096, 1024
    n = nfft + 5 * hop + 17
    x = synth_batch(2, n)
    nfr = (n - nfft) // hop + 1
    spec = torch.full((T, nfr, nfft // 2), float("nan"), dtype=torch.float32, device="cuda")
    cols = torch.full((T, nfr, 256), float("nan"), dtype=torch.float32, device="cuda")
    wam._check(wam.lib().wam_batch_stft_db_dev(C.c_void_p(x.data_ptr()), C.c_int(T), C.c_int(2), C.c_int64(n), C.c_double(SR), C.c_int(nfft), C.c_int(hop),
                                               C.c_void_p(spec.data_ptr()), C.c_void_p(cols.data_ptr()), C.c_void_p(0)))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(spec).all()) and bool(torch.isfinite(cols).all())
    for t in EDGE_TRACKS:
        h = host_track(x, t)
        s1, c1 = wam.stft_db(h, SR, nfft, hop, columns=True)
        assert np.array_equal(spec[t].cpu().numpy(), s1) and np.array_equal(cols[t].cpu().numpy(), c1), t
        so = O.stft_db(h, SR, nfft, hop)
        assert float(np.max(np.abs(s1.astype(np.float64) - so))) <= 2e-5                          # dB; the bar of tests/test_gpu_extras.py::test_stft
