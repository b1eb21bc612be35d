#!/bin/bash
# Runs on the GPU box (WAM_COMMIT=$(git rev-parse --short HEAD); gpurun -- "WAM_COMMIT=$WAM_COMMIT bash profiles/refresh.sh <tag> <round prefix>"):
# the PMC passes first (128 tracks = one launch per kernel and step), their summaries into profiles/ of the box's copy, then the
# bench lines - which therefore price their kernels with counters of this very build - and the default bench command under
# rocprofv3 --kernel-trace --stats.  Everything lands in gpurun_out/<tag>/; `python profiles/ingest.py <tag> <prefix>` afterwards
# copies what is to be judged into profiles/ (and rebuilds the same PMC summaries from the raw passes).
set -e
TAG=${1:-r01}
PRE=${2:-$TAG}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
bash profiles/collect_pmc.sh $TAG --tracks-per-gpu 128
bash profiles/collect_pmc.sh $TAG/meter --workload meter --tracks-per-gpu 128
bash profiles/collect_pmc.sh $TAG/stft --workload stft --tracks-per-gpu 128
cd $R
python3 profiles/ingest.py $TAG $PRE --pmc-only
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 200 python3 bench.py --workload meter > $OUT/bench_meter.json 2>> $OUT/bench.err
timeout -k 10 200 python3 bench.py --workload stft > $OUT/bench_stft.json 2>> $OUT/bench.err
timeout -k 10 200 python3 bench.py --workload single --steps 20 --warmup 3 > $OUT/bench_single.json 2>> $OUT/bench.err
timeout -k 10 200 python3 bench.py --workload longform --steps 5 > $OUT/bench_longform.json 2>> $OUT/bench.err
timeout -k 10 200 python3 bench.py --tracks 128 --no-cpu-baseline > $OUT/bench_T128.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
echo done
