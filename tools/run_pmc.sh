#!/bin/bash
# usage: tools/run_pmc.sh <tag> <python args...>   (on the GPU box; one --pmc pass per counter group)
R=$PWD; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE TCC_MISS" "WRITE_SIZE TCC_HIT TCC_REQ" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_INT64" \
           "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o p$i -- python3 "$@" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob("$R/gpurun_out/pmc_$TAG/*counter_collection.csv")):
    seen=set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r["Dispatch_Id"]))
    for k,_ in seen: cnt[(k,f)] += 1
for k, d in agg.items():
    n = max(v for (kk,f),v in cnt.items() if kk==k)
    print(k, "dispatches/pass", n)
    for c, v in sorted(d.items()): print("   %-34s %.4g per dispatch" % (c, v / n))
PY
