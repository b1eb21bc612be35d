"""Pins the CPU oracle on the shapes round 1 refused (tests/golden/gen_shapes.mjs: the reference's own JavaScript under
Node): meters with 3 and 6 channels, sample rates whose 400 ms block is not four hops (11,025 Hz, 44,056 Hz), the
limiter / soft clipper / chain at 176.4 and 192 kHz, and the NORMALIZE message's numbers.  BIT-EXACT like
tests/test_oracle_golden.py."""
import json
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from test_oracle_golden import close, same


def shapes():
    with open(os.path.join(G.DATA, "shapes.json")) as f:
        return json.load(f)["cases"]


S = shapes()
METERS = [k for k, c in S.items() if c["kind"] == "meter"]


@pytest.mark.parametrize("name", METERS)
def test_meter_shapes(name):
    c = S[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    lufs, blocks, mask = O.measure_lufs(x, c["sr"], detail=True)
    close(lufs, G.num(c["lufs"]))
    close(lufs, G.num(c["lufs_worker"]))
    close(O.true_peak_db(x), G.num(c["tp_db"]))
    close(O.true_peak_db(x), G.num(c["tp_db_worker"]))
    assert len(blocks) == c["nblocks"]
    assert np.array_equal(blocks, G.f64(c["block_ms"]))
    assert np.array_equal(mask, G.u8(c["gate_mask"]))
    assert G.gate_margin(blocks) > 1e-12                 # (tests/golden_util.py: the GPU gate sums in tree order)
    for pos, v in c["short_term"]:
        close(O.short_term_lufs(x, c["sr"], pos), G.num(v))
    for pos, v in c["momentary"]:
        close(O.momentary_lufs(x, c["sr"], pos), G.num(v))


@pytest.mark.parametrize("sr", [176400, 192000])
def test_high_rate_stages(sr):
    tag = f"hi{sr}"
    c = S[tag + "_limiter"]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    same(O.limiter(x, sr, c["ceiling_lin"]), G.chans(c["out"]))
    same(O.limiter(x, sr, c["ceiling_lin"], preserve=False), G.chans(S[tag + "_limiter_np"]["out"]))
    same(O.limiter_lib(x, sr, c["ceiling_lin"]), G.chans(S[tag + "_limiter_lib"]["out"]))
    same(O.soft_clip(x, sr), G.chans(S[tag + "_softclip"]["out"]))


@pytest.mark.parametrize("name", ["hi176400_chain", "hi192000_chain", "chain_11025"])
def test_chain_shapes(name):
    c = S[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    o, r = O.render_chain(x, c["sr"], c["settings"], c["mode"])
    same(o, G.chans(c["out"]))
    close(r.lufs, G.num(c["lufs"]))


def test_normalize_and_limiter_messages():
    for name in ("msg_normalize_limited", "msg_normalize_plain"):
        c = S[name]
        x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
        o, r = O.normalize(x, c["sr"], c["target"], c["ceiling_db"])
        same(o, G.chans(c["out"]))
        close(r.current_lufs, G.num(c["currentLUFS"]))
        close(r.final_lufs, G.num(c["finalLUFS"]))
        close(r.peak_db, G.num(c["peakDB"]))
        close(r.gain_applied_db, G.num(c["gainApplied"]))
        assert bool(r.limiter_applied) == c["limiterApplied"]
    c = S["msg_normalize_silent"]
    z = [np.zeros(c["n"], np.float32)] * 2
    o, r = O.normalize(z, c["sr"], c["target"], c["ceiling_db"])
    assert r.measured == 0 and not c["has_gain"] and r.current_lufs == G.num(c["currentLUFS"]) == float("-inf")
    for name in ("msg_limiter", "msg_limiter_mono_defaults"):
        c = S[name]
        x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
        same(O.limiter(x, c["sr"], c["ceiling_lin"]), G.chans(c["out"]))


@pytest.mark.parametrize("name", [k for k, c in S.items() if c["kind"] == "deharsh"])
def test_deharsh_presets(name):
    """processHybridDynamic 'gentle' / 'aggressive' / no preset given (dynamic-processor.js:538-566, :595-607).  Substitute DFT on
    both sides as in test_oracle_golden.py: unpinned at the fft.js boundary, bit-exact everywhere else."""
    c = S[name]
    x = O.synth(c["idx"], c["nch"], c["n"], c["sr"], c["opts"])
    same(O.deharsh(x, c["sr"], c["preset"] or "gentle"), G.chans(c["out"]))
