#!/bin/bash
# On the GPU box: the numbers DESIGN.md section 6 / README quote, one log.  usage: tools/collect_round_numbers.sh <tag>
TAG=${1:-r02}; O=gpurun_out/numbers_$TAG.log; : > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; }
run python tools/bench_configs.py c3
run python tools/bench_configs.py c4
run python tools/bench_configs.py c1
run python tools/example_rates.py
run python tools/time_to_solution.py 4096
run python tools/setup_time.py 4096
run env QUICK=1 python tools/microbench_iter2.py 4096 0,0 f64
run env QUICK=1 python tools/microbench_iter2.py 4096 0,0 f32 abs
run python tools/microbench_iter2.py 4096 0 f32
run python tools/microbench_iter_mc.py 4096 3
run python bench.py --no-pair --no-cpu-baseline
run env PROST_ITER2_NO_RING=1 python bench.py --no-cpu-baseline
grep -v "^$" $O | cut -c1-400
