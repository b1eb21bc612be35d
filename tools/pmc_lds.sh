#!/bin/bash
# On the GPU box: one PMC pass with the LDS counters over the chain at 128 tracks (one launch per kernel).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3/pmc_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL -d $OUT -o lds --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --tracks-per-gpu 128 "$@" > $OUT/lds.log 2>&1
find $OUT -name '*counter_collection.csv' | head
