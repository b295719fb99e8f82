#!/bin/bash
# round 5, first GPU call: the suite on the modified tree, this box's baseline lines for C2 / C3 / C4 (before the round's kernel
# work), the instruction-mix PMC passes of the shipping kernels.   usage: bash tools/collect_r05_a.sh
R=$PWD; O=$R/gpurun_out/r05a; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1 > $O/bench_driver_invocation.json
python3 bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python3 bench.py --config c3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3.json
python3 bench.py --config c4 2>$O/bench_c4.err | tail -1 > $O/bench_c4.json
python3 bench.py --stepsize boyd --residual-iter 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_boyd_r1_4096.json
python3 bench.py --stepsize boyd --residual-iter 1 --size 1024 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_boyd_r1_1024.json
python3 bench.py --stepsize boyd --residual-iter 1 --size 256 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_boyd_r1_256.json
python3 bench.py --stepsize boyd --residual-iter 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_boyd_r10_4096.json
for f in bench_driver_invocation bench bench_c3 bench_c4 bench_boyd_r1_4096; do python3 -c "
import json,sys
d=json.load(open('$O/$f.json')); r=d.get('roofline',{}); c=d.get('cpu_baseline',{})
print('$f', d['value'], r.get('kernel'), r.get('avg_launch_ms'), r.get('frac'), 'cpu', c.get('value'), c.get('cores'), c.get('threads_probed'), c.get('single_thread_value'), (c.get('reference_build') or {}).get('value'), c.get('port_over_reference_one_thread'))
"; done
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_c4 -- python3 $R/bench.py --config c4 --no-cpu-baseline > $O/bench_c4_under_rocprof.json 2>/dev/null )
bash tools/collect_r05_instmix.sh $O/instmix > $O/instmix.log 2>&1
cat $O/instmix/r05_pmc_instmix.txt | head -150
