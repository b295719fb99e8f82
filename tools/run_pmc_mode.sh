#!/bin/bash
# usage: tools/run_pmc_mode.sh <tag> <pmc_iter.py mode> <cols>   -- issue / wait counters of one kernel instance (separate --pmc passes)
R=$PWD; TAG=$1; MODE=$2; COLS=$3
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcm_${TAG}_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcm_${TAG}_$i -o p -- python3 $R/tools/pmc_iter.py 4096 $MODE $COLS 12 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in sorted(glob.glob("/tmp/pmcm_${TAG}_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "fused_iter2d_x2" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-48:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, d in agg.items():
    print("$TAG", k)
    for c, v in sorted(d.items()): print("   %-28s %.5g per launch" % (c, v / max(1, len(cnt[(k, c)]))))
PY
