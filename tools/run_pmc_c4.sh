#!/bin/bash
# HBM traffic counters of the kernels of the four-launch CG round at the C4 size (separate --pmc passes): tools/run_pmc_c4.sh
R=$PWD
cd /tmp; export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_c4_$grp
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_c4_$grp -o p -- python3 $R/bench.py --config c4 --steps 30 --warmup 5 --prelude-iters 0 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); big = collections.defaultdict(int)
rows = []
for f in sorted(glob.glob("/tmp/pmc_c4_*/**/*counter_collection.csv", recursive=True)):
    rows += [r for r in csv.DictReader(open(f)) if "cg_step" in r["Kernel_Name"] or "op_stage_kernel" in r["Kernel_Name"]]
for r in rows:
    k = r["Kernel_Name"].split("(")[0]; big[k] = max(big[k], int(r["Grid_Size"]))
for r in rows:                                   # full-size launches only (bench.py first runs the kernels on a 32 x 32 problem)
    k = r["Kernel_Name"].split("(")[0]
    if int(r["Grid_Size"]) != big[k]: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, d in sorted(agg.items()):
    print(k.replace("void prost_hip::", "")[:90])
    for c, v in sorted(d.items()): print("   %-12s %.6g KiB per launch over %d launches" % (c, v / max(1, len(cnt[(k, c)])), len(cnt[(k, c)])))
PY
