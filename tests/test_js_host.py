"""The JavaScript host layer (web-audio-mastering_amd/js/index.mjs over napi/wam.node), run under Node.

Mirrors the reference's caller contract: web/workers/worker-interface.js (promise API, FIFO, error strings)
and web/lib/dsp/index.js (sync names).  The checks themselves live in tests/js/host_test.mjs.
"""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "js", "host_test.mjs")
ADDON = os.path.join(ROOT, "web-audio-mastering_amd", "napi", "wam.node")

needs_node = pytest.mark.skipif(shutil.which("node") is None, reason="node not on PATH")


def _run(mode):
    assert os.path.exists(ADDON), "napi/wam.node not built (run __graft_entry__.build())"
    p = subprocess.run(["node", SCRIPT, mode], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["ok"] and r["checks"] > 0
    return r


@needs_node
def test_js_surface_and_errors_cpu():
    _run("cpu")


@needs_node
@pytest.mark.gpu
def test_js_host_matches_goldens_gpu():
    r = _run("gpu")
    assert r["checks"] > 20
