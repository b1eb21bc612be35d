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


def gate_margin(blocks):
    """Relative distance of the closest block energy to either gate threshold of lufs.js:67-78 (1e-7 absolute; 0.1 x the mean
    of the blocks above it).  The GPU's k_gate sums the blocks as a tree, not in block order, which moves the relative threshold
    by ~1e-16: the gate masks of a fixture are only pinned if no block sits that close to a threshold."""
    import numpy as np
    b = np.asarray(blocks, dtype=np.float64)
    if b.size == 0:
        return float("inf")
    m = float(np.min(np.abs(b - 1e-7) / 1e-7))
    kept = b[b > 1e-7]
    if kept.size:
        thr = 0.1 * float(np.mean(kept))
        m = min(m, float(np.min(np.abs(b - thr) / thr)))
    return m
