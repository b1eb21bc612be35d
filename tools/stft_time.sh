#!/bin/bash
# On the GPU box: the STFT workload through bench.py for each variant library given (var/libwam_<name>.so), ms per 1,024 tracks.
mkdir -p gpurun_out/r3
for v in "$@"; do
  if [ "$v" = "base" ]; then unset WAM_LIB_PATH; else export WAM_LIB_PATH=var/libwam_$v.so; fi
  timeout -k 10 200 python3 bench.py --workload stft --no-cpu-baseline --steps 3 > gpurun_out/r3/stft_$v.log 2>&1 || { tail -3 gpurun_out/r3/stft_$v.log; exit 1; }
  python3 -c "
import json
d = json.loads(open('gpurun_out/r3/stft_$v.log').read().strip().splitlines()[-1])
print('$v', round(d['ms_per_step'], 2), 'ms')"
done
