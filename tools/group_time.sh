#!/bin/bash
# On the GPU box: one 256-track render group through bench.py, per-kernel milliseconds per launch.  Usage: bash tools/group_time.sh <tag> [bench args]
TAG=${1:-g}; shift
mkdir -p gpurun_out/r3
timeout -k 10 240 python3 bench.py --tracks 256 --no-cpu-baseline --steps 3 "$@" > gpurun_out/r3/$TAG.log 2>&1 || { tail -5 gpurun_out/r3/$TAG.log; exit 1; }
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r3/$TAG.log").read().strip().splitlines()[-1])
print("$TAG", round(d["ms_per_step"], 2), "ms/step;", {k: round(v["total_ms"] / v["launches"], 2) for k, v in d["kernels"].items()})
PY
