#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) + VALU busy of the 3-D kernels: tools/run_pmc_iter3d.sh [cols]
R=$PWD; cd /tmp; export TMPDIR=/tmp
cols=${1:-18}
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc3d -o $tag -- python3 $R/tools/microbench_iter3d.py 2048 2048 64 $cols > /dev/null 2>&1
done
python3 - <<PY
import csv, collections
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open("$R/gpurun_out/pmc3d/%s_counter_collection.csv" % tag)):
        k = r["Kernel_Name"].split("(")[0]
        if "fused" not in k: continue
        agg[(k[-70:], r["Counter_Name"])] += float(r["Counter_Value"]); n[(k[-70:], r["Counter_Name"])] += 1
    for (k, c), v in sorted(agg.items()):
        print("%-72s %-22s mean per launch = %.6g over %d launches" % (k, c, v / n[(k, c)], n[(k, c)]))
PY
