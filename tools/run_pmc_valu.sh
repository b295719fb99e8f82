#!/bin/bash
# usage: tools/run_pmc_valu.sh <tag> <python args...>  -- VALU instruction mix of every kernel
R=$PWD; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcv_$TAG -o p$i -- python3 "$@" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob("$R/gpurun_out/pmcv_$TAG/*counter_collection.csv")):
    seen=set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-70:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen.add((k, r["Dispatch_Id"]))
    for k,_ in seen: cnt[(k,f)] += 1
for k, d in agg.items():
    n = max(v for (kk,f),v in cnt.items() if kk==k)
    print(k, "dispatches/pass", n)
    for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, v / n))
PY
