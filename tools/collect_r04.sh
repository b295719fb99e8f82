#!/bin/bash
# On the GPU box: everything profiles/r04_* and DESIGN.md section 6 quote for round 4.  usage: bash tools/collect_r04.sh   (outputs under gpurun_out/r04c/)
R=$PWD; O=$R/gpurun_out/r04c; mkdir -p $O
# ---- C2 (headline): the driver's invocation, the default run, fp64
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_invocation.json
python3 bench.py 2>/dev/null | tail -1 > $O/bench.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_invocation_2.json
python3 bench.py --dtype f64 2>/dev/null | tail -1 > $O/bench_f64.json
python3 bench.py --dtype f64 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f64_driver_invocation.json
# ---- the reference's DEFAULT backend options (pdhg.m:4-14: boyd, residual_iter = 1): rule on the device / on the host
for dev in 1 0; do
  for size in 4096 1024 256; do
    PROST_BENCH_DEVICE_RULES=$dev python3 bench.py --stepsize boyd --residual-iter 1 --size $size --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_boyd_r1_${size}_dev$dev.json
  done
  PROST_BENCH_DEVICE_RULES=$dev python3 bench.py --stepsize boyd --residual-iter 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_boyd_r10_4096_dev$dev.json
done
# ---- C3 / C4
python3 bench.py --config c3 2>/dev/null | tail -1 > $O/bench_c3.json
python3 bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4.json
# ---- rocprofv3 kernel stats of the same commands
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_short -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_short_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_f64 -- python3 $R/bench.py --dtype f64 --no-cpu-baseline > $O/bench_f64_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_boyd_r1 -- python3 $R/bench.py --stepsize boyd --residual-iter 1 --no-cpu-baseline > $O/bench_boyd_r1_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_sparse -- python3 $R/tools/sparse_rof_rate.py 2048 2000 1 > $O/sparse_rof_under_rocprof.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_sparse_generic -- python3 $R/tools/sparse_rof_rate.py 2048 600 0 > $O/sparse_rof_generic_under_rocprof.txt 2>/dev/null
# ---- HBM traffic (separate --pmc passes): fp32 and fp64 pair kernels, the position-dependent instance
for dt in f32 f64; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o pmc_fetch_$dt -- python3 $R/bench.py --dtype $dt --steps 60 --warmup 10 --prelude-iters 0 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o pmc_write_$dt -- python3 $R/bench.py --dtype $dt --steps 60 --warmup 10 --prelude-iters 0 --no-cpu-baseline > /dev/null 2>&1
done
# C3: kernel stats, traffic and the instruction mix of the 3-D pair kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_c3 -- python3 $R/bench.py --config c3 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O -o pmc_c3_$c -- python3 $R/bench.py --config c3 --steps 40 --warmup 10 --prelude-iters 0 --no-cpu-baseline > /dev/null 2>&1
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o pmc_fetch_sparse -- python3 $R/tools/sparse_rof_rate.py 4096 60 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o pmc_write_sparse -- python3 $R/tools/sparse_rof_rate.py 4096 60 1 > /dev/null 2>&1
cd $R
python3 - <<PY > $O/pmc_summary.txt
# mean counter value per launch, FULL-SIZE launches only (smaller grids = the code-object warm-up on a tiny problem)
import csv, collections, glob
for tag in ("pmc_fetch_f32", "pmc_write_f32", "pmc_fetch_f64", "pmc_write_f64", "pmc_fetch_sparse", "pmc_write_sparse", "pmc_c3_FETCH_SIZE", "pmc_c3_WRITE_SIZE", "pmc_c3_SQ_INSTS_VALU", "pmc_c3_SQ_INSTS_SALU", "pmc_c3_SQ_INSTS_BRANCH"):
    files = glob.glob("$O/**/%s_counter_collection.csv" % tag, recursive=True)
    rows = [r for f in files[:1] for r in csv.DictReader(open(f)) if "fused" in r["Kernel_Name"]]
    big = collections.defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        big[k] = max(big[k], int(r["Grid_Size"]))
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if int(r["Grid_Size"]) != big[k]: continue
        agg[(k, r["Counter_Name"], big[k])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"], big[k])].add(r["Dispatch_Id"])
    for (k, c, g), v in sorted(agg.items()):
        print(tag, k[-84:], "grid %d work-items" % g, c, "mean per launch = %.6g over %d launches" % (v / len(n[(k, c, g)]), len(n[(k, c, g)])))
PY
cat $O/pmc_summary.txt
for f in $(find $O -name "*_kernel_stats.csv"); do echo $f; python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:7]: print('  ', r['Name'][:110].replace('void prost_hip::',''), r['Calls'], r['AverageNs'], r['Percentage'])"; done
for f in $O/bench*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('$f'.split('/')[-1], 'value', round(d['value']), 'iterate_only', round(d['iterate_only_it_per_s']), 'kernel', r.get('kernel'), 'avg_ms', r.get('avg_launch_ms'), 'timed', r.get('launches_timed'), 'frac', r.get('frac'), 'phys', r.get('frac_hbm_traffic'), 'cols', r.get('chunk_cols'))
except Exception as e: print('$f', 'ERR', e)"; done
cat $O/sparse_rof_under_rocprof.txt $O/sparse_rof_generic_under_rocprof.txt
# ---- the examples as written, the generic path under the default options, what a stamp costs
{ python3 tools/rof_primal_rate.py 700 464 3 3000; python3 tools/rof_primal_rate.py 700 464 1 3000; python3 tools/rof_primal_rate.py 2048 2048 3 1000; python3 tools/rof_primal_rate.py 2048 2048 1 1000; } 2>/dev/null | tee $O/rof_primal_example_rates.txt
{ python3 tools/generic_rule_rate.py 256 256 3000; python3 tools/generic_rule_rate.py 700 464 3000; python3 tools/generic_rule_rate.py 1024 1024 2000; python3 tools/generic_rule_rate.py 2048 2048 600; } 2>/dev/null | tee $O/generic_device_rules.txt
{ python3 tools/sparse_rof_rate.py 2048 2000 1; python3 tools/sparse_rof_rate.py 2048 2000 2; python3 tools/sparse_rof_rate.py 4096 600 1; python3 tools/sparse_rof_rate.py 2048 2000 1 boyd 1; } 2>/dev/null | tee $O/sparse_rof_rates.txt
tools/bin/stamp_probe | tee $O/stamp_probe.txt
