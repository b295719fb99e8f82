#!/bin/bash
# On the GPU box: everything profiles/r03_* and DESIGN.md section 6 quote.  usage: bash tools/collect_r03.sh   (outputs under gpurun_out/r03/)
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
# ---- C2 (headline): bench lines, rocprofv3 kernel stats, HBM traffic / issue counters (separate --pmc passes)
bash tools/profile_bench.sh r03 > $O/profile_bench.log 2>&1
# ---- C3 / C4: the bench lines the driver can reproduce, and kernel stats of the same command
python3 bench.py --config c3 2>/dev/null | tail -1 > $O/bench_c3.json
python3 bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4.json
python3 bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3_short.json
python3 bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_short.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c3 -- python3 $R/bench.py --config c3 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c4 -- python3 $R/bench.py --config c4 --no-cpu-baseline > $O/bench_c4_under_rocprof.json 2>/dev/null
cd $R
bash tools/run_pmc_iter3d_x2.sh r03 > $O/pmc_c3.txt 2>&1
# ---- two ranks on this GPU (host transport), started by bench.py itself
PROST_BENCH_TRANSPORT=host python3 bench.py --gpus 2 --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_two_ranks_host_transport.json
# ---- the other numbers
bash tools/collect_round_numbers.sh r03 > /dev/null 2>&1; cp gpurun_out/numbers_r03.log $O/numbers.log
python3 tools/inpaint_rate.py 2048 3 >> $O/numbers.log 2>&1
python3 tools/inpaint_rate.py 4096 1 >> $O/numbers.log 2>&1
X2_ONLY=1 X2_COLS=0 python3 tools/microbench_iter3d.py 2048 2048 64 0 >> $O/numbers.log 2>&1
X2_ONLY=1 X2_COLS=0 python3 tools/microbench_iter3d.py 2048 2048 64 0 f64 >> $O/numbers.log 2>&1
ls $O | head -50
