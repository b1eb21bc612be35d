#!/bin/bash
# On the GPU box: one-group chain bench of the product library under several values of one wam_tune knob.
# usage: bash tools/bench_tune.sh <knob> "<values>" "<kernels>"
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; VALS=$2; KS=$3
cd $R
for v in $VALS; do
  out=$(timeout -k 10 120 python3 bench.py --tracks 256 --steps 2 --warmup 1 --no-cpu-baseline --tune $K=$v 2>/dev/null | tail -1)
  rc=$?
  if [ $rc -ne 0 ]; then echo "$K=$v FAILED rc=$rc"; exit $rc; fi
  python3 - "$K=$v" "$KS" <<PY
import json, sys
d = json.loads('''$out''')
ks = sys.argv[2].split()
print(f"{sys.argv[1]:24s} step {d['ms_per_step']:8.2f} " + " ".join(f"{k} {d['kernels'][k]['total_ms'] / d['steps']:7.2f}" for k in ks if k in d['kernels']))
PY
done
