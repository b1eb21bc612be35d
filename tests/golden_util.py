"""Loader for tests/golden/data (outputs of the reference's JS, see tests/golden/gen_golden.mjs)."""
import json
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")


def num(v):
    if isinstance(v, str):
        return {"inf": float("inf"), "-inf": float("-inf"), "nan": float("nan")}[v]
    return float(v)


def load():
    with open(os.path.join(DATA, "golden.json")) as f:
        return json.load(f)["cases"]


def f32(name):
    return np.fromfile(os.path.join(DATA, name), dtype=np.float32)


def f64(name):
    return np.fromfile(os.path.join(DATA, name), dtype=np.float64)


def u8(name):
    return np.fromfile(os.path.join(DATA, name), dtype=np.uint8)


def chans(names):
    return [f32(n) for n in names]
