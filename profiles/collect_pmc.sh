#!/bin/bash
# Collects the PMC passes for bench.py (one pass per run, as MI355X_MICROARCH.md prescribes). Usage on the GPU box:
#   bash profiles/collect_pmc.sh <tag> [bench args...]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/sq -o sq --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/sq.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $OUT/sq2 -o sq2 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/sq2.log 2>&1 || echo "sq2 pass failed (optional)"
(cd $R && python3 -c 'import bench; print(bench.source_digest())') > $OUT/source_digest.txt
find $OUT -name '*.csv' | head -20
