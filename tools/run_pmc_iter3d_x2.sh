#!/bin/bash
# issue / wait / traffic counters of the 3-D double-iteration kernel (separate --pmc passes): tools/run_pmc_iter3d_x2.sh [tag]
R=$PWD; TAG=${1:-x2}
cd /tmp; export TMPDIR=/tmp
export X2_ONLY=1 X2_COLS=0
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcx2_${TAG}_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcx2_${TAG}_$i -o p -- python3 $R/tools/microbench_iter3d.py 2048 2048 64 0 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in sorted(glob.glob("/tmp/pmcx2_${TAG}_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "fused_iter3d_x2" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, d in agg.items():
    print("$TAG", k)
    for c, v in sorted(d.items()): print("   %-28s %.5g per launch" % (c, v / max(1, len(cnt[(k, c)]))))
PY
