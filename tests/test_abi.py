"""CPU-side checks of the product library: it loads, exports every symbol include/wam.h
declares, and refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "web-audio-mastering_amd", "libwam_hip.so")


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "wam.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wam_[a-z0-9_]+)\s*\(", src)) - {"wam_progress_cb"})


@pytest.mark.skipif(not os.path.exists(SO), reason="libwam_hip.so not built")
def test_exports_every_declared_symbol():
    lib = C.CDLL(SO)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


@pytest.mark.skipif(not os.path.exists(SO), reason="libwam_hip.so not built")
def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    lib = C.CDLL(SO)
    lib.wam_last_error.restype = C.c_char_p
    x = np.zeros(48000, np.float32)
    FP = C.POINTER(C.c_float)
    pp = (FP * 1)(x.ctypes.data_as(FP))
    out = C.c_double(0)
    rc = lib.wam_true_peak_db(pp, C.c_int(1), C.c_int64(len(x)), C.byref(out))
    assert rc == -2, rc     # WAM_ERR_NO_DEVICE
    assert b"no HIP device" in lib.wam_last_error()
