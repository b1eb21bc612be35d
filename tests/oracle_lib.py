"""ctypes binding of oracle/libwam_oracle.so (TEST INFRASTRUCTURE: the checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libwam_oracle.so")

FP = C.POINTER(C.c_float)
FPP = C.POINTER(FP)


class Settings(C.Structure):
    """Mirror of wam_settings (include/wam.h)."""
    _fields_ = [
        ("normalize_loudness", C.c_int32), ("target_lufs", C.c_double),
        ("true_peak_limit", C.c_int32), ("true_peak_ceiling", C.c_double),
        ("clean_low_end", C.c_int32), ("glue_compression", C.c_int32), ("deharsh", C.c_int32),
        ("stereo_width", C.c_double), ("center_bass", C.c_int32), ("cut_mud", C.c_int32),
        ("add_air", C.c_int32), ("tape_warmth", C.c_int32), ("auto_level", C.c_int32),
        ("add_punch", C.c_int32), ("input_gain", C.c_double),
        ("eq_low", C.c_double), ("eq_low_mid", C.c_double), ("eq_mid", C.c_double),
        ("eq_high_mid", C.c_double), ("eq_high", C.c_double),
    ]

    @classmethod
    def from_js(cls, s):
        """Build from the reference's flat settings object (web/ui/controls.js:66-89)."""
        def num(v, d=0.0):
            try:
                f = float(v)
            except (TypeError, ValueError):
                return d
            return f
        return cls(
            int(bool(s.get("normalizeLoudness"))), num(s.get("targetLufs")),
            int(bool(s.get("truePeakLimit"))), num(s.get("truePeakCeiling")),
            int(bool(s.get("cleanLowEnd"))), int(bool(s.get("glueCompression"))), int(bool(s.get("deharsh"))),
            num(s.get("stereoWidth"), float("nan")), int(bool(s.get("centerBass"))), int(bool(s.get("cutMud"))),
            int(bool(s.get("addAir"))), int(bool(s.get("tapeWarmth"))), int(bool(s.get("autoLevel"))),
            int(bool(s.get("addPunch"))), num(s.get("inputGain")),
            num(s.get("eqLow")), num(s.get("eqLowMid")), num(s.get("eqMid")), num(s.get("eqHighMid")), num(s.get("eqHigh")))


class Result(C.Structure):
    _fields_ = [("lufs", C.c_double), ("measured_lufs", C.c_double), ("true_peak_db", C.c_double),
                ("lufs_in", C.c_double), ("gain_db", C.c_double)]


class NormalizeResult(C.Structure):
    _fields_ = [("current_lufs", C.c_double), ("final_lufs", C.c_double), ("peak_db", C.c_double),
                ("gain_applied_db", C.c_double), ("limiter_applied", C.c_int32), ("measured", C.c_int32)]


class SynthOpts(C.Structure):
    _fields_ = [("amp", C.c_double), ("dc", C.c_double), ("click_amp", C.c_double),
                ("quiet_from", C.c_int64), ("quiet_to", C.c_int64), ("quiet_gain", C.c_double),
                ("quiet2_from", C.c_int64), ("quiet2_to", C.c_int64), ("quiet2_gain", C.c_double)]

    @classmethod
    def from_js(cls, o):
        o = o or {}
        return cls(o.get("amp", 1.0), o.get("dc", 0.0), o.get("clickAmp", 0.0),
                   o.get("quietFrom", -1), o.get("quietTo", -1), o.get("quietGain", 0.0),
                   o.get("quiet2From", -1), o.get("quiet2To", -1), o.get("quiet2Gain", 0.0))


class Biquad(C.Structure):
    _fields_ = [("b0", C.c_double), ("b1", C.c_double), ("b2", C.c_double), ("a1", C.c_double), ("a2", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.ora_measure_lufs.restype = C.c_double
        _lib.ora_true_peak_db.restype = C.c_double
        _lib.ora_true_peak_sample.restype = C.c_double
        _lib.ora_short_term_lufs.restype = C.c_double
        _lib.ora_momentary_lufs.restype = C.c_double
        _lib.ora_dc_detect.restype = C.c_double
        _lib.ora_dc_remove.restype = C.c_double
    return _lib


def chan_ptrs(chans):
    """list of contiguous float32 arrays -> float** (keeps refs alive via the returned tuple)."""
    arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in chans]
    pp = (FP * len(arrs))(*[a.ctypes.data_as(FP) for a in arrs])
    return pp, arrs


def synth(idx, nch, n, sr, opts=None, n0=0):
    o = SynthOpts.from_js(opts)
    out = []
    for ch in range(nch):
        a = np.empty(n, dtype=np.float32)
        lib().ora_synth(C.c_int(idx), C.c_int(ch), C.c_int64(n0), C.c_int64(n), C.c_double(sr), C.byref(o), a.ctypes.data_as(FP))
        out.append(a)
    return out


def _run_stage(fn, chans, args):
    n = len(chans[0])
    pin, keep = chan_ptrs(chans)
    outs = [np.empty(n, dtype=np.float32) for _ in chans]
    pout, keep2 = chan_ptrs(outs)
    fn(pin, *args(n), pout)
    return keep2


def measure_lufs(chans, sr, fallback=-14.0, detail=False):
    n = len(chans[0])
    pin, keep = chan_ptrs(chans)
    cap = max(1, n // max(1, int(math.floor(sr * 0.4 * 0.25))) + 2)
    blocks = np.zeros(cap, dtype=np.float64)
    mask = np.zeros(cap, dtype=np.uint8)
    nb = C.c_int64(0)
    v = lib().ora_measure_lufs(pin, C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_double(fallback),
                               blocks.ctypes.data_as(C.POINTER(C.c_double)), mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(nb))
    if detail:
        return v, blocks[:nb.value].copy(), mask[:nb.value].copy()
    return v


def true_peak_db(chans):
    pin, keep = chan_ptrs(chans)
    return lib().ora_true_peak_db(pin, C.c_int(len(chans)), C.c_int64(len(chans[0])))


def true_peak_sample(y):
    a = (C.c_double * 4)(*y)
    return lib().ora_true_peak_sample(a)


def short_term_lufs(chans, sr, pos):
    pin, keep = chan_ptrs(chans)
    return lib().ora_short_term_lufs(pin, C.c_int(len(chans)), C.c_int64(len(chans[0])), C.c_double(sr), C.c_double(pos))


def momentary_lufs(chans, sr, pos):
    pin, keep = chan_ptrs(chans)
    return lib().ora_momentary_lufs(pin, C.c_int(len(chans)), C.c_int64(len(chans[0])), C.c_double(sr), C.c_double(pos))


def extract_peaks(x, num_peaks=800):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(num_peaks, dtype=np.float32)
    lib().ora_extract_peaks(x.ctypes.data_as(FP), C.c_int64(len(x)), C.c_int(num_peaks), out.ctypes.data_as(FP))
    return out


def dc_detect(x):
    a = np.ascontiguousarray(x, dtype=np.float32)
    return lib().ora_dc_detect(a.ctypes.data_as(FP), C.c_int64(len(a)))


def dc_remove(x):
    a = np.array(x, dtype=np.float32, copy=True)
    dc = lib().ora_dc_remove(a.ctypes.data_as(FP), C.c_int64(len(a)))
    return dc, a


def final_filters(chans, sr, highpass=True, lowpass=True):
    return _run_stage(lib().ora_final_filters, chans,
                      lambda n: (C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_int(int(highpass)), C.c_int(int(lowpass))))


def parametric_eq(chans, sr, gains, cut_mud=False):
    g = (C.c_double * 5)(*gains)
    return _run_stage(lib().ora_parametric_eq, chans,
                      lambda n: (C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), g, C.c_int(int(cut_mud))))


def glue(chans, sr, thr=-18.0, ratio=3.0, attack=0.02, release=0.25):
    return _run_stage(lib().ora_glue, chans,
                      lambda n: (C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_double(thr), C.c_double(ratio),
                                 C.c_double(attack), C.c_double(release)))


def stereo(chans, sr, width, bass_mono, bass_freq=200.0):
    return _run_stage(lib().ora_stereo, chans,
                      lambda n: (C.c_int64(n), C.c_double(sr), C.c_double(width), C.c_int(int(bass_mono)), C.c_double(bass_freq)))


def normalize_skip_limiter(chans, sr, target):
    n = len(chans[0])
    pin, keep = chan_ptrs(chans)
    outs = [np.empty(n, dtype=np.float32) for _ in chans]
    pout, keep2 = chan_ptrs(outs)
    li, ti = C.c_double(0), C.c_double(0)
    applied = lib().ora_normalize_skip_limiter(pin, C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_double(target), pout,
                                               C.byref(li), C.byref(ti))
    return keep2, applied, li.value, ti.value


def normalize(chans, sr, target=-14.0, ceiling_db=-1.0):
    n = len(chans[0])
    pin, keep = chan_ptrs(chans)
    outs = [np.empty(n, dtype=np.float32) for _ in chans]
    pout, keep2 = chan_ptrs(outs)
    res = NormalizeResult()
    lib().ora_normalize(pin, C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_double(target), C.c_double(ceiling_db), pout, C.byref(res))
    return keep2, res


def soft_clip(chans, sr, ceiling_db=-1.0, lookahead_ms=0.5, release_ms=10.0, drive=1.5):
    return _run_stage(lib().ora_soft_clip, chans,
                      lambda n: (C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_double(ceiling_db), C.c_double(lookahead_ms),
                                 C.c_double(release_ms), C.c_double(drive)))


def limiter(chans, sr, ceiling_lin, lookahead_ms=3.0, release_ms=100.0, knee_db=3.0, preserve=True):
    return _run_stage(lib().ora_limiter, chans,
                      lambda n: (C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_double(ceiling_lin), C.c_double(lookahead_ms),
                                 C.c_double(release_ms), C.c_double(knee_db), C.c_int(int(preserve))))


def limiter_lib(chans, sr, ceiling_lin, lookahead_ms=3.0, release_ms=100.0, knee_db=3.0, preserve=True):
    return _run_stage(lib().ora_limiter_lib, chans,
                      lambda n: (C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_double(ceiling_lin), C.c_double(lookahead_ms),
                                 C.c_double(release_ms), C.c_double(knee_db), C.c_int(int(preserve))))


def soft_knee_oversampled(x, ceiling, knee_db=3.0):
    a = np.ascontiguousarray(x, dtype=np.float32)
    o = np.empty_like(a)
    lib().ora_soft_knee_oversampled(a.ctypes.data_as(FP), C.c_int64(len(a)), C.c_double(ceiling), C.c_double(knee_db), o.ctypes.data_as(FP))
    return o


def _simple(fn):
    def run(chans, sr):
        return _run_stage(fn, chans, lambda n: (C.c_int(len(chans)), C.c_int64(n), C.c_double(sr)))
    return run


def exciter(chans, sr):
    return _simple(lib().ora_exciter)(chans, sr)


DEHARSH_PRESETS = {"gentle": 0, "aggressive": 1, "mastering": 2}     # any other name is 'gentle' (dynamic-processor.js:604-606)


def deharsh(chans, sr, preset="mastering"):
    return _run_stage(lib().ora_deharsh_preset, chans,
                      lambda n: (C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_int(DEHARSH_PRESETS.get(preset, 0))))


def warmth(chans, sr):
    return _simple(lib().ora_warmth)(chans, sr)


def punch(chans, sr):
    return _simple(lib().ora_punch)(chans, sr)


def pcm24(chans):
    n = len(chans[0])
    pin, keep = chan_ptrs(chans)
    out = np.empty(3 * n * len(chans), dtype=np.uint8)
    lib().ora_pcm24(pin, C.c_int(len(chans)), C.c_int64(n), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def wav_header(nch, n, sr, bits):
    h = np.zeros(44, dtype=np.uint8)
    lib().ora_wav_header(C.c_int(nch), C.c_int64(n), C.c_int(sr), C.c_int(bits), h.ctypes.data_as(C.POINTER(C.c_uint8)))
    return h


def pcm16(chans, mode, dither=None):
    n = len(chans[0])
    pin, keep = chan_ptrs(chans)
    out = np.empty(2 * len(chans) * n, dtype=np.uint8)
    d = np.ascontiguousarray(dither, dtype=np.float32) if dither is not None else None
    lib().ora_pcm16(pin, C.c_int(len(chans)), C.c_int64(n), C.c_int(mode), d.ctypes.data_as(FP) if d is not None else None,
                    out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def stft_db(chans, sr, nfft, hop, columns=False):
    n = len(chans[0])
    pin, keep = chan_ptrs(chans)
    nfr = (n - nfft) // hop + 1 if n >= nfft else 0
    out = np.empty((nfr, nfft // 2), dtype=np.float32)
    cols = np.empty((nfr, 256), dtype=np.float32) if columns else None
    lib().ora_stft_db(pin, C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.c_int(nfft), C.c_int(hop), out.ctypes.data_as(FP),
                      cols.ctypes.data_as(FP) if columns else None)
    return (out, cols) if columns else out


def render_chain(chans, sr, settings, mode="export"):
    n = len(chans[0])
    pin, keep = chan_ptrs(chans)
    outs = [np.empty(n, dtype=np.float32) for _ in chans]
    pout, keep2 = chan_ptrs(outs)
    s = settings if isinstance(settings, Settings) else Settings.from_js(settings)
    res = Result()
    lib().ora_render_chain(pin, C.c_int(len(chans)), C.c_int64(n), C.c_double(sr), C.byref(s), C.c_int(1 if mode == "export" else 0), pout, C.byref(res))
    return keep2, res


def resample_ref(x, src_rate, dst_rate, quality):
    """numpy float64 restatement of wam_resample's OWN definition (include/wam.h; the reference's resampler is Chromium's and
    unpinned): position p = i * src / dst, zero outside the buffer; quality 0 linear, 1 Kaiser-windowed sinc (cutoff 0.98 x lower Nyquist, 64 zeros, beta 8.6)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    n_out = int(math.ceil((n / src_rate) * dst_rate))
    p = np.arange(n_out, dtype=np.float64) * (src_rate / dst_rate)
    xp = np.concatenate([x, [0.0, 0.0]])
    if quality == 0:
        k = np.floor(p).astype(np.int64)
        f = p - k
        k = np.minimum(k, n)
        return ((1.0 - f) * xp[k] + f * xp[np.minimum(k + 1, n + 1)]).astype(np.float32)
    fc = min(1.0, dst_rate / src_rate) * 0.98
    H = 64 / fc
    out = np.zeros(n_out)
    i0b = np.i0(8.6)
    for d in range(-int(math.ceil(H)) - 1, int(math.ceil(H)) + 2):
        k = np.floor(p).astype(np.int64) + d
        t = p - k
        u = t / H
        ok = (k >= 0) & (k <= n - 1) & (t >= -H) & (t <= H)          # k in [ceil(p - H), floor(p + H)]
        w = np.where(u * u < 1.0, np.i0(8.6 * np.sqrt(np.maximum(0.0, 1.0 - u * u))) / i0b, 0.0)
        out += np.where(ok, xp[np.clip(k, 0, n + 1)] * (fc * np.sinc(fc * t) * w), 0.0)
    return out.astype(np.float32)
