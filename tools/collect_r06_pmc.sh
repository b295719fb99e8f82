#!/bin/bash
# On the GPU box: HBM traffic counters (separate --pmc passes, FETCH_SIZE / WRITE_SIZE) of the kernels round 6 added or re-measured:
# the K-iteration kernel of the tolerance class at the headline size, the CG-round kernels of C4 at 2048^2 (beyond the Infinity Cache)
# and of the warp-matrix config c4w.  usage: bash tools/collect_r06_pmc.sh [c4w]  -> gpurun_out/r06/pmc_traffic_raw.txt (c4w: only the c4w runs ->
# pmc_traffic_c4w_raw.txt: the two-launch rounds with gathered rows, round 6)
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {   # tag, bench args...
  local tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${tag}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$c -o p -- python3 $R/bench.py "$@" --no-cpu-baseline --no-kernel-timing --prelude-iters 0 > /dev/null 2>&1
  done
  python3 - $tag "$*" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); big = collections.defaultdict(int)
rows = []
for f in sorted(glob.glob("/tmp/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True)):
    rows += [r for r in csv.DictReader(open(f)) if any(t in r["Kernel_Name"] for t in ("fused_iter", "cg_step", "cg_pixel", "op_stage_kernel"))]
for r in rows:
    k = r["Kernel_Name"].split("(")[0]; big[k] = max(big[k], int(r["Grid_Size"]))
for r in rows:                                   # full-size launches only (bench.py first runs the kernels on a tiny problem)
    k = r["Kernel_Name"].split("(")[0]
    if int(r["Grid_Size"]) != big[k]: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("== %s: bench.py %s" % (tag, sys.argv[2]))
for k, d in sorted(agg.items()):
    print(k.replace("void prost_hip::", "")[:100], " grid", big[k], "work-items")
    for c, v in sorted(d.items()): print("   %-12s %.6g KiB per launch over %d launches" % (c, v / max(1, len(cnt[(k, c)])), len(cnt[(k, c)])))
PY
  rm -rf /tmp/pmc_${tag}_FETCH_SIZE /tmp/pmc_${tag}_WRITE_SIZE
}
if [ "$1" = "c4w" ]; then
{
run c4w_1024 --config c4w --steps 20 --warmup 5
run c4w_2048 --config c4w --size 2048 --steps 12 --warmup 3
} > $O/pmc_traffic_c4w_raw.txt 2>&1
cat $O/pmc_traffic_c4w_raw.txt
exit 0
fi
{
run c2_fmad --arithmetic fmad --steps 60 --warmup 10
run c2_exact --no-fmad --steps 60 --warmup 10
run c3_fmad --config c3 --arithmetic fmad --steps 20 --warmup 4
run c4_2048 --config c4 --size 2048 --steps 12 --warmup 3
run c4w_1024 --config c4w --steps 20 --warmup 5
run c4w_2048 --config c4w --size 2048 --steps 12 --warmup 3
} > $O/pmc_traffic_raw.txt 2>&1
cat $O/pmc_traffic_raw.txt
