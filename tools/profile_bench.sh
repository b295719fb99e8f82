#!/bin/bash
# On the GPU box: bench line + rocprofv3 kernel stats + HBM traffic counters (separate --pmc passes).
R=$PWD; mkdir -p $R/gpurun_out/prof
python bench.py 2>/dev/null | tail -1 > $R/gpurun_out/bench_r01.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof -o pmc_fetch -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof -o pmc_write -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, collections
for tag in ("pmc_fetch", "pmc_write"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open("$R/gpurun_out/prof/%s_counter_collection.csv" % tag)):
        k = r["Kernel_Name"].split("(")[0]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for (k, c), v in sorted(agg.items()):
        if "fused" in k: print(tag, k[-60:], c, "mean per launch = %.6g KiB over %d launches" % (v / n[(k, c)], n[(k, c)]))
PY
head -6 $R/gpurun_out/prof/r01_kernel_stats.csv | cut -c1-200
python3 -c "import json; d=json.load(open('$R/gpurun_out/bench_r01.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])"
