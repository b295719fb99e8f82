#!/bin/bash
# On the GPU box: instruction mix and stall breakdown (SQ counters) of the kernels that SHIP in round 5 -- the fp32 pair kernel
# (plain + residual instance), its fp64 twin, the single residual launch of the reference's default options, the 3-D pair kernel.
# One rocprofv3 --pmc pass per counter group (8 SQ slots per pass), the program directly behind `--`.
# usage: bash tools/collect_r05_instmix.sh [outdir]      -> <outdir>/instmix_<cfg>.txt  (copied to profiles/r05_pmc_instmix.txt)
R=$PWD; O=${1:-$R/gpurun_out/r05_instmix}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
avail() { for c in "$@"; do grep -qw "$c" $O/counters_list.txt && echo -n "$c "; done; }
G1=$(avail SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU)
G2=$(avail SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM)
G3=$(avail SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64)
G4=$(avail SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM)
G5=$(avail SQ_WAIT_INST_LDS SQ_INSTS_WAVE32 SQ_THREAD_CYCLES_VALU SQ_INSTS_SENDMSG SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_EXP_GDS SQ_ACTIVE_INST_FLAT)
echo "groups: [$G1] [$G2] [$G3] [$G4] [$G5]" > $O/groups.txt
run_cfg() {   # tag, bench args...
  local tag=$1; shift
  local i=0
  for grp in "$G1" "$G2" "$G3" "$G4" "$G5"; do
    i=$((i+1))
    [ -z "$grp" ] && continue
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -o p$i -- python3 $R/bench.py "$@" --no-cpu-baseline --prelude-iters 0 > $O/${tag}_p$i.log 2>&1
  done
  python3 - "$O/$tag" > $O/instmix_$tag.txt <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(lambda: collections.defaultdict(set))
big = collections.defaultdict(int)
rows = []
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fused" not in k and "cg_" not in k and "op_stage" not in k and "c4_" not in k:
            continue
        rows.append(r); big[k] = max(big[k], int(r["Grid_Size"]))
for r in rows:
    k = r["Kernel_Name"]
    if int(r["Grid_Size"]) != big[k]:
        continue                      # smaller grids: the code-object warm-up on a tiny problem
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg):
    print(k.split("(")[0], " grid", big[k], "work-items")
    for c in sorted(agg[k]):
        n = len(disp[k][c])
        print("   %-28s %14.6g per launch (%d launches)" % (c, agg[k][c] / n, n))
PY
}
run_cfg c2_f32 --steps 60 --warmup 10
run_cfg c2_f64 --dtype f64 --steps 60 --warmup 10
run_cfg c2_boyd_r1 --stepsize boyd --residual-iter 1 --steps 60 --warmup 10
run_cfg c3_f32 --config c3 --steps 20 --warmup 4
cat $O/groups.txt $O/instmix_*.txt > $O/r05_pmc_instmix.txt
# drop the raw per-dispatch csv files (tens of MB): the summaries are what profiles/ keeps
find $O -name "*.csv" -size +2M -delete
