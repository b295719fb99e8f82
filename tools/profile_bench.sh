#!/bin/bash
# On the GPU box: bench lines (driver invocation + default) + rocprofv3 kernel stats + HBM traffic / issue counters (separate --pmc passes).
# usage: tools/profile_bench.sh <tag>      (outputs under gpurun_out/prof_<tag>/, copy the summaries to profiles/)
R=$PWD; TAG=${1:-r02}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_invocation.json
python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_default.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_invocation_2.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_short -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_short_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o pmc_fetch -- python3 $R/bench.py --steps 60 --warmup 10 --prelude-iters 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o pmc_write -- python3 $R/bench.py --steps 60 --warmup 10 --prelude-iters 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o pmc_valu -- python3 $R/bench.py --steps 60 --warmup 10 --prelude-iters 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY > $O/pmc_summary.txt
# mean counter value per launch, FULL-SIZE launches only (bench.py first runs the same kernels on a 64 x 64 problem to load the
# code objects: those launches have a smaller grid and are left out)
import csv, collections, glob
for tag in ("pmc_fetch", "pmc_write", "pmc_valu"):
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
        print(tag, k[-72:], "grid %d work-items" % g, c, "mean per launch = %.6g over %d launches" % (v / len(n[(k, c, g)]), len(n[(k, c, g)])))
PY
cat $O/pmc_summary.txt
for f in $(find $O -name "stats_kernel_stats.csv" -o -name "stats_short_kernel_stats.csv"); do echo $f; python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:6]: print('  ', r['Name'][:100].replace('void prost_hip::',''), r['Calls'], r['AverageNs'], r['Percentage'])"; done
for f in bench_driver_invocation bench_driver_invocation_2 bench_default; do python3 -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']
print('$f', 'value', round(d['value']), 'iterate_only', round(d['iterate_only_it_per_s']), 'ms/step', d['ms_per_step'], 'kernel', r['kernel'], 'avg_ms', r['avg_launch_ms'], 'timed', r['launches_timed'], 'frac', round(r['frac'],3), 'phys', r['frac_hbm_traffic'], 'cols', r['chunk_cols'])
print('   all', {k:(round(v['avg_launch_ms'],4), v['launches_timed'], v['chunk_cols']) for k,v in r['all_kernels'].items()}, d.get('cpu_baseline'))"; done
