#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash profiles/refresh.sh <tag>'): the default bench line, the same command under
# rocprofv3 --kernel-trace --stats, and the PMC passes (128 tracks = one launch per kernel and step).  Everything lands
# in gpurun_out/<tag>/; copy what is to be judged into profiles/ afterwards (profiles/ingest.py).
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 200 python3 bench.py --workload meter > $OUT/bench_meter.json 2>> $OUT/bench.err
timeout -k 10 200 python3 bench.py --workload stft > $OUT/bench_stft.json 2>> $OUT/bench.err
timeout -k 10 200 python3 bench.py --workload single --steps 20 --warmup 3 > $OUT/bench_single.json 2>> $OUT/bench.err
timeout -k 10 200 python3 bench.py --workload longform --steps 5 > $OUT/bench_longform.json 2>> $OUT/bench.err
timeout -k 10 200 python3 bench.py --tracks 128 --no-cpu-baseline > $OUT/bench_T128.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
cd $R
bash profiles/collect_pmc.sh $TAG --tracks-per-gpu 128
bash profiles/collect_pmc.sh $TAG/meter --workload meter --tracks-per-gpu 128
bash profiles/collect_pmc.sh $TAG/stft --workload stft --tracks-per-gpu 128
echo done
