#!/bin/bash
# rocprofv3 kernel stats of one bench_configs.py configuration: tools/prof_config.sh c3|c4 [args...]  -> gpurun_out/prof_<cfg>/
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
cfg=$1; shift
mkdir -p gpurun_out/prof_$cfg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$cfg -o $cfg -- python3 tools/bench_configs.py $cfg "$@" > gpurun_out/prof_$cfg/run.log 2>&1
grep "^{" gpurun_out/prof_$cfg/run.log | cut -c1-200
head -8 gpurun_out/prof_$cfg/${cfg}_kernel_stats.csv | cut -c1-200
