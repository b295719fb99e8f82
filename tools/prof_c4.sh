#!/bin/bash
# kernel-level profile of the C4 ADMM configuration: tools/prof_c4.sh [device|host]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mode=${1:-device}
mkdir -p gpurun_out/prof_c4_$mode
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4_$mode -o c4 -- python3 tools/bench_configs.py c4 1024 $mode > gpurun_out/prof_c4_$mode/run.log 2>&1
tail -2 gpurun_out/prof_c4_$mode/run.log
f=$(find gpurun_out/prof_c4_$mode -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print("%-90s calls %6s avg_us %9.2f  pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
