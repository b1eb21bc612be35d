#!/bin/bash
# On the GPU box: one-group chain bench for every var/*.so (and the product library), kernel ms per step of the kernels named.
# usage: bash tools/bench_variants.sh "k_punch k_warm" [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
KS=$1; shift
cd $R
for so in "" var/*.so; do
  name=${so:-product}
  out=$(WAM_LIB_PATH=${so:+$R/$so} timeout -k 10 120 python3 bench.py --tracks 256 --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1)
  rc=$?
  if [ $rc -ne 0 ]; then echo "$name FAILED rc=$rc"; exit $rc; fi
  python3 - "$name" "$KS" <<PY
import json, sys
d = json.loads('''$out''')
ks = sys.argv[2].split()
print(f"{sys.argv[1]:24s} step {d['ms_per_step']:8.2f} " + " ".join(f"{k} {d['kernels'][k]['total_ms'] / d['steps']:7.2f}" for k in ks if k in d['kernels']))
PY
done
