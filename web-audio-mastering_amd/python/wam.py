"""ctypes binding of libwam_hip.so (include/wam.h) + a Python mirror of the reference's
JavaScript surface (lib/dsp/index.js function names, DSPWorkerInterface methods).

Plumbing only: numpy arrays in, numpy arrays out, every call goes through the C ABI.
There is no fallback: if the HIP library is missing or no device is present the
calls raise WamError.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WAM_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libwam_hip.so")   # (override: timing experiments with variant builds)

FP = C.POINTER(C.c_float)
ABI_VERSION = 3            # WAM_ABI_VERSION of include/wam.h this file mirrors


class WamError(RuntimeError):
    pass


class Settings(C.Structure):
    """wam_settings == the reference's flat settings object (web/ui/controls.js:66-89)."""
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
        def num(v, d=0.0):
            try:
                return float(v)
            except (TypeError, ValueError):
                return d
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


PROGRESS_CB = C.CFUNCTYPE(C.c_int, C.c_double, C.c_char_p, C.c_void_p)

# default UI settings of the reference (web/index.html:178-295)
DEFAULT_SETTINGS = dict(
    normalizeLoudness=True, targetLufs=-14, truePeakLimit=True, truePeakCeiling=-1, cleanLowEnd=True,
    glueCompression=False, deharsh=True, stereoWidth=100, centerBass=True, cutMud=False, addAir=True,
    tapeWarmth=True, autoLevel=False, addPunch=True, inputGain=0,
    eqLow=0, eqLowMid=0, eqMid=0, eqHighMid=0, eqHigh=0)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WamError(f"{LIB_PATH} not built (run `make -C web-audio-mastering_amd/csrc`); there is no CPU path")
        _lib = C.CDLL(LIB_PATH)
        if not hasattr(_lib, "wam_abi_version") or _lib.wam_abi_version() != ABI_VERSION:
            got = _lib.wam_abi_version() if hasattr(_lib, "wam_abi_version") else "none"
            _lib = None
            raise WamError(f"{LIB_PATH} exports ABI version {got}, this binding was written against {ABI_VERSION} (include/wam.h)")
        _lib.wam_last_error.restype = C.c_char_p
        _lib.wam_version.restype = C.c_char_p
        _lib.wam_lufs_num_blocks.restype = C.c_int64
        _lib.wam_lufs_num_blocks.argtypes = [C.c_int64, C.c_double]
        if hasattr(_lib, "wam_stft_num_frames"):
            _lib.wam_stft_num_frames.restype = C.c_int64
    return _lib


def _check(rc):
    if rc != 0:
        raise WamError(f"wam error {rc}: {lib().wam_last_error().decode()}")


def init(device=0):
    _check(lib().wam_init(C.c_int(device)))


def tune(key, value=0):
    """Time-segmentation knobs (include/wam_synth.h wam_tune); tune("reset") restores the defaults."""
    _check(lib().wam_tune(key.encode(), C.c_int64(value)))


def device_count():
    n = C.c_int(0)
    lib().wam_device_count(C.byref(n))
    return n.value


def _ptrs(chans):
    arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in chans]
    pp = (FP * len(arrs))(*[a.ctypes.data_as(FP) for a in arrs])
    return pp, arrs


def _outs(chans):
    outs = [np.empty(len(chans[0]), dtype=np.float32) for _ in chans]
    pp = (FP * len(outs))(*[a.ctypes.data_as(FP) for a in outs])
    return pp, outs


# ---- metering (lib/dsp/index.js names) ---------------------------------------
def measure_lufs(chans, sr, fallback=-14.0, detail=False):
    """measureLUFS(audioBuffer, fallbackLufs) - web/lib/dsp/lufs.js:17"""
    pin, keep = _ptrs(chans)
    n = len(keep[0])
    out = C.c_double(0)
    if detail:
        cap = max(1, lib().wam_lufs_num_blocks(n, float(sr)))
        blocks = np.zeros(cap, dtype=np.float64)
        mask = np.zeros(cap, dtype=np.uint8)
        nb = C.c_int64(0)
        _check(lib().wam_measure_lufs(pin, C.c_int(len(keep)), C.c_int64(n), C.c_double(sr), C.c_double(fallback), C.byref(out),
                                      blocks.ctypes.data_as(C.POINTER(C.c_double)), mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(nb)))
        return out.value, blocks[:nb.value], mask[:nb.value]
    _check(lib().wam_measure_lufs(pin, C.c_int(len(keep)), C.c_int64(n), C.c_double(sr), C.c_double(fallback), C.byref(out), None, None, None))
    return out.value


def measure_short_term_lufs(chans, sr, position=0.0):
    """measureShortTermLUFS(audioBuffer, position) - web/lib/dsp/lufs.js:91"""
    pin, keep = _ptrs(chans)
    out = C.c_double(0)
    _check(lib().wam_short_term_lufs(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), C.c_double(sr), C.c_double(position), C.byref(out)))
    return out.value


def measure_momentary_lufs(chans, sr, position=0.0):
    """measureMomentaryLUFS(audioBuffer, position) - web/lib/dsp/lufs.js:128"""
    pin, keep = _ptrs(chans)
    out = C.c_double(0)
    _check(lib().wam_momentary_lufs(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), C.c_double(sr), C.c_double(position), C.byref(out)))
    return out.value


def lufs_series(chans, sr, positions, momentary=False):
    """measureShortTermLUFS / measureMomentaryLUFS (web/lib/dsp/lufs.js:91,128) at many positions, one upload"""
    pin, keep = _ptrs(chans)
    pos = np.ascontiguousarray(positions, dtype=np.float64)
    out = np.zeros(len(pos), dtype=np.float64)
    DP = C.POINTER(C.c_double)
    _check(lib().wam_lufs_series(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), C.c_double(sr), pos.ctypes.data_as(DP), C.c_int(len(pos)),
                                 C.c_int(1 if momentary else 0), out.ctypes.data_as(DP)))
    return out


def resample(chans, src_rate, dst_rate, quality=1):
    """resampleAudioBuffer - web/ui/renderer.js:172 (self-defined, parity unpinned: include/wam.h)"""
    pin, keep = _ptrs(chans)
    n = len(keep[0])
    lib().wam_resample_length.restype = C.c_int64
    n_out = lib().wam_resample_length(C.c_int64(n), C.c_double(src_rate), C.c_double(dst_rate))
    outs = [np.empty(n_out, dtype=np.float32) for _ in keep]
    pout = (FP * len(outs))(*[a.ctypes.data_as(FP) for a in outs])
    _check(lib().wam_resample(pin, C.c_int(len(keep)), C.c_int64(n), C.c_double(src_rate), C.c_double(dst_rate), C.c_int(quality), pout, C.c_int64(n_out)))
    return outs


def extract_peaks(ch0, num_peaks=800):
    """extractPeaks(audioBuffer, numPeaks) - web/ui/waveform.js:164 (channel 0 only)"""
    x = np.ascontiguousarray(ch0, dtype=np.float32)
    out = np.zeros(max(num_peaks, 0), dtype=np.float32)
    _check(lib().wam_extract_peaks(x.ctypes.data_as(FP), C.c_int64(len(x)), C.c_int(num_peaks), out.ctypes.data_as(FP)))
    return out


def find_true_peak(chans):
    """findTruePeak(audioBuffer) - web/lib/dsp/true-peak.js:46"""
    pin, keep = _ptrs(chans)
    out = C.c_double(0)
    _check(lib().wam_true_peak_db(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), C.byref(out)))
    return out.value


def meter(chans, sr, fallback=-14.0):
    pin, keep = _ptrs(chans)
    l, t = C.c_double(0), C.c_double(0)
    _check(lib().wam_meter(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), C.c_double(sr), C.c_double(fallback), C.byref(l), C.byref(t)))
    return l.value, t.value


class LoadAnalysis(C.Structure):
    _fields_ = [("dc", C.c_double * 2), ("dc_percent", C.c_double), ("original_lufs", C.c_double), ("original_true_peak_db", C.c_double),
                ("gain_db", C.c_double), ("dc_significant", C.c_int32), ("dc_removed", C.c_int32), ("measured", C.c_int32),
                ("limiter_applied", C.c_int32)]


def analyze_on_load(chans, sr, target_lufs=-14.0, ceiling_db=-1.0):
    """web/app.js:823-872 in one call: (cleaned, normalized, LoadAnalysis)"""
    pin, keep = _ptrs(chans)
    pc, cleaned = _outs(keep)
    pn, normalized = _outs(keep)
    res = LoadAnalysis()
    _check(lib().wam_analyze_on_load(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), C.c_double(sr), C.c_double(target_lufs), C.c_double(ceiling_db),
                                     pc, pn, C.byref(res)))
    return cleaned, normalized, res


def dc_offset(chans, remove=False):
    """detectDCOffsetBuffer / removeDCOffset - web/lib/dsp/dc-offset.js:38,67"""
    pin, keep = _ptrs(chans)
    dc = (C.c_double * len(keep))()
    if remove:
        pout, outs = _outs(keep)
        _check(lib().wam_dc_offset(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), dc, C.c_int(1), pout))
        return list(dc), outs
    _check(lib().wam_dc_offset(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), dc, C.c_int(0), None))
    return list(dc)


# ---- rendering -----------------------------------------------------------------
def render_full_chain(chans, sr, settings, mode="export", on_progress=None):
    """DSPWorkerInterface.renderFullChain - web/workers/worker-interface.js:259"""
    pin, keep = _ptrs(chans)
    pout, outs = _outs(keep)
    s = settings if isinstance(settings, Settings) else Settings.from_js(settings)
    res = Result()
    cb = _progress_cb(on_progress)
    _check(lib().wam_render_chain(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), C.c_double(sr), C.byref(s),
                                  C.c_int(1 if mode == "export" else 0), pout, C.byref(res), cb, None))
    return outs, res


def _stage(fn, chans, *args):
    pin, keep = _ptrs(chans)
    pout, outs = _outs(keep)
    _check(fn(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), *args, pout))
    return outs


def apply_final_filters(chans, sr, highpass=True, lowpass=True):
    return _stage(lib().wam_final_filters, chans, C.c_double(sr), C.c_int(int(highpass)), C.c_int(int(lowpass)))


def apply_mastering_soft_clip(chans, sr, ceiling=-1.0, lookahead_ms=0.5, release_ms=10.0, drive=1.5):
    return _stage(lib().wam_soft_clip, chans, C.c_double(sr), C.c_double(ceiling), C.c_double(lookahead_ms), C.c_double(release_ms), C.c_double(drive))


def apply_glue_compressor(chans, sr, threshold=-18.0, ratio=3.0, attack=0.02, release=0.25):
    return _stage(lib().wam_glue_compressor, chans, C.c_double(sr), C.c_double(threshold), C.c_double(ratio), C.c_double(attack), C.c_double(release))


def apply_stereo(chans, sr, width=1.0, bass_mono=True, bass_freq=200.0):
    pin, keep = _ptrs(chans)
    pout, outs = _outs(keep)
    _check(lib().wam_stereo(pin, C.c_int64(len(keep[0])), C.c_double(sr), C.c_double(width), C.c_int(int(bass_mono)), C.c_double(bass_freq), pout))
    return outs


def apply_parametric_eq(chans, sr, gains, cut_mud=False):
    g = (C.c_double * 5)(*gains)
    return _stage(lib().wam_parametric_eq, chans, C.c_double(sr), g, C.c_int(int(cut_mud)))


def _progress_cb(on_progress):
    return PROGRESS_CB(lambda p, st, u: int(bool(on_progress(p, st.decode())))) if on_progress else C.cast(None, PROGRESS_CB)


def apply_lookahead_limiter(chans, sr, ceiling_linear, lookahead_ms=3.0, release_ms=100.0, knee_db=3.0, preserve_transients=True, twin="worker",
                            on_progress=None):
    """twin="worker": applyLookaheadLimiterToChannels (dsp-worker.js:286; message APPLY_LIMITER with its PROGRESS stream);
    twin="lib": applyLookaheadLimiter (web/lib/dsp/limiter.js:155)"""
    if twin == "lib":
        return _stage(lib().wam_limiter_lib, chans, C.c_double(sr), C.c_double(ceiling_linear), C.c_double(lookahead_ms), C.c_double(release_ms),
                      C.c_double(knee_db), C.c_int(int(preserve_transients)))
    pin, keep = _ptrs(chans)
    pout, outs = _outs(keep)
    cb = _progress_cb(on_progress)
    _check(lib().wam_limiter(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), C.c_double(sr), C.c_double(ceiling_linear), C.c_double(lookahead_ms),
                             C.c_double(release_ms), C.c_double(knee_db), C.c_int(int(preserve_transients)), pout, cb, None))
    return outs


def apply_exciter(chans, sr):
    return _stage(lib().wam_exciter, chans, C.c_double(sr))


DEHARSH_PRESETS = {"gentle": 0, "aggressive": 1, "mastering": 2}


def process_hybrid_dynamic(chans, sr, preset="gentle"):
    """processHybridDynamic(buffer, preset = 'gentle'), dynamic-processor.js:595-609 (an unknown name is 'gentle', :604-606);
    the export chain's stage is preset='mastering'."""
    return _stage(lib().wam_deharsh_preset, chans, C.c_double(sr), C.c_int(DEHARSH_PRESETS.get(preset, 0)))


def apply_tape_warmth(chans, sr):
    return _stage(lib().wam_tape_warmth, chans, C.c_double(sr))


def apply_multiband_transient(chans, sr):
    return _stage(lib().wam_multiband_transient, chans, C.c_double(sr))


def apply_gain(chans, gain_db):
    """applyGain(audioBuffer, gainDB) - web/lib/dsp/normalizer.js:103"""
    return _stage(lib().wam_apply_gain, chans, C.c_double(gain_db))


def normalize(chans, sr, target_lufs=-14.0, ceiling_db=-1.0, on_progress=None):
    """DSPWorkerInterface.normalize - worker-interface.js:197"""
    pin, keep = _ptrs(chans)
    pout, outs = _outs(keep)
    res = NormalizeResult()
    cb = _progress_cb(on_progress)
    _check(lib().wam_normalize(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), C.c_double(sr), C.c_double(target_lufs), C.c_double(ceiling_db), pout, C.byref(res),
                               cb, None))
    return outs, res


def encode_pcm24(chans):
    """encodeWAV 24-bit sample loop - web/ui/encoder.js:116-123"""
    pin, keep = _ptrs(chans)
    out = np.empty(3 * len(keep) * len(keep[0]), dtype=np.uint8)
    _check(lib().wam_encode_pcm24(pin, C.c_int(len(keep)), C.c_int64(len(keep[0])), out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def encode_pcm16(chans, dither_mode=1, dither=None, seed=0):
    """encodeWAV 16-bit sample loop - web/ui/encoder.js:97-115.  dither_mode 0 none, 1 tpdf, 2 noise-shaped."""
    pin, keep = _ptrs(chans)
    n = len(keep[0])
    out = np.empty(2 * len(keep) * n, dtype=np.uint8)
    d = np.ascontiguousarray(dither, dtype=np.float32) if dither is not None else None
    if d is not None and d.size < len(keep) * n:
        raise WamError("dither needs nch * n values")
    _check(lib().wam_encode_pcm16(pin, C.c_int(len(keep)), C.c_int64(n), C.c_int(dither_mode), d.ctypes.data_as(FP) if d is not None else None,
                                  C.c_uint32(seed), out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def wav_header(nch, n, sr, bits):
    h = np.zeros(44, dtype=np.uint8)
    _check(lib().wam_wav_header(C.c_int(nch), C.c_int64(n), C.c_int(sr), C.c_int(bits), h.ctypes.data_as(C.POINTER(C.c_uint8))))
    return h


def stft_db(chans, sr, nfft=4096, hop=1024, columns=False):
    pin, keep = _ptrs(chans)
    n = len(keep[0])
    nfr = (n - nfft) // hop + 1 if n >= nfft else 0
    out = np.empty((nfr, nfft // 2), dtype=np.float32)
    cols = np.empty((nfr, 256), dtype=np.float32) if columns else None
    _check(lib().wam_stft_db(pin, C.c_int(len(keep)), C.c_int64(n), C.c_double(sr), C.c_int(nfft), C.c_int(hop), out.ctypes.data_as(FP),
                             cols.ctypes.data_as(FP) if columns else None))
    return (out, cols) if columns else out


def batch_render(tracks, sr, settings, mode="export", devices=None, outs=None):
    """wam_batch_render / wam_batch_render_multi: host-pointer batch; tracks = list of tracks, each a list of nch float32 arrays of
    one common length; devices = list of HIP device indices (contiguous track ranges, one per entry) or None for the current device.
    Returns (outputs in the same shape, results)."""
    nch, n = len(tracks[0]), len(tracks[0][0])
    keep = [np.ascontiguousarray(c, dtype=np.float32) for t in tracks for c in t]
    if outs is None:                                   # (fresh arrays: their pages are first touched by the copy-out, inside the call)
        outs = [np.empty(n, dtype=np.float32) for _ in keep]
    else:                                              # the caller's own output arrays, flat [track * nch + channel], reused between calls
        assert len(outs) == len(keep) and all(o.dtype == np.float32 and o.flags.c_contiguous and len(o) == n for o in outs)
    pin = (FP * len(keep))(*[a.ctypes.data_as(FP) for a in keep])
    pout = (FP * len(outs))(*[a.ctypes.data_as(FP) for a in outs])
    s = settings if isinstance(settings, Settings) else Settings.from_js(settings)
    res = (Result * len(tracks))()
    if devices is None:
        _check(lib().wam_batch_render(pin, C.c_int(len(tracks)), C.c_int(nch), C.c_int64(n), C.c_double(sr), C.byref(s),
                                      C.c_int(1 if mode == "export" else 0), pout, res))
    else:
        dv = (C.c_int * len(devices))(*devices)
        _check(lib().wam_batch_render_multi(pin, C.c_int(len(tracks)), C.c_int(nch), C.c_int64(n), C.c_double(sr), C.byref(s),
                                            C.c_int(1 if mode == "export" else 0), pout, res, dv, C.c_int(len(devices))))
    return [outs[i * nch:(i + 1) * nch] for i in range(len(tracks))], res


# ---- device-resident batch (pointers are raw device addresses, e.g. torch.Tensor.data_ptr()) ----
def batch_meter_dev(in_ptr, ntracks, nch, n, sr, fallback=-14.0, stream=0):
    l = (C.c_double * ntracks)()
    t = (C.c_double * ntracks)()
    _check(lib().wam_batch_meter_dev(C.c_void_p(in_ptr), C.c_int(ntracks), C.c_int(nch), C.c_int64(n), C.c_double(sr), C.c_double(fallback), l, t, C.c_void_p(stream)))
    return np.array(l), np.array(t)


def batch_render_dev(in_ptr, out_ptr, ntracks, nch, n, sr, settings, mode="export", stream=0):
    s = settings if isinstance(settings, Settings) else Settings.from_js(settings)
    res = (Result * ntracks)()
    _check(lib().wam_batch_render_dev(C.c_void_p(in_ptr), C.c_int(ntracks), C.c_int(nch), C.c_int64(n), C.c_double(sr), C.byref(s),
                                      C.c_int(1 if mode == "export" else 0), C.c_void_p(out_ptr), res, C.c_void_p(stream)))
    return res
