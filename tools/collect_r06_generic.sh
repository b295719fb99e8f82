#!/bin/bash
# On the GPU box: the DEFAULT generic PDHG iteration (separate operator products, residual sums in the prox launches, rule on the device) at the
# deblurring shape, 2048^2 fp32, boyd / residual_iter 1: kernel durations (rocprofv3 --kernel-trace --stats) and, per launch of every kernel,
# FETCH_SIZE / WRITE_SIZE / SQ_WAIT_* / VALU counters (separate --pmc passes).  usage: bash tools/collect_r06_generic.sh [mask]  -> gpurun_out/r06/pmc_generic.txt
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O; MASK=${1:-4}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/gstats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gstats -o s -- python3 $R/tools/generic_rule_rate.py 2048 2048 300 1 100 $MASK > $O/generic_rate_under_rocprof.txt 2>&1
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
G2="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE"
i=0
for grp in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1))
  rm -rf /tmp/gpmc$i
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/gpmc$i -o p -- python3 $R/tools/generic_rule_rate.py 2048 2048 12 1 6 $MASK > /dev/null 2>&1
done
python3 - > $O/pmc_generic.txt <<'PY'
import collections, csv, glob
stats = {}
for f in glob.glob("/tmp/gstats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(lambda: collections.defaultdict(set)); big = collections.defaultdict(int)
rows = []
for f in sorted(glob.glob("/tmp/gpmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        rows.append((k, r)); big[k] = max(big[k], int(r["Grid_Size"]))
for k, r in rows:
    if int(r["Grid_Size"]) != big[k]: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
print("default generic PDHG iteration, deblurring shape 2048^2 fp32, boyd / residual_iter 1 (tools/collect_r06_generic.sh); per launch of each kernel")
for k, (calls, us, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    if pct < 0.5: continue
    print("%s\n   calls %d  avg %.1f us  %.1f %% of the GPU time  grid %d work-items" % (k.replace("void prost_hip::", "")[:150], calls, us, pct, big.get(k, 0)))
    d = agg.get(k, {})
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fe = d["FETCH_SIZE"] / len(disp[k]["FETCH_SIZE"]); wr = d["WRITE_SIZE"] / len(disp[k]["WRITE_SIZE"])
        mb = (2 * fe + wr) * 1024 / 1e6
        print("   traffic: FETCH_SIZE %.0f KiB (x2) + WRITE_SIZE %.0f KiB = %.1f MB per launch -> %.0f GB/s = %.3f of 8 TB/s" % (fe, wr, mb, mb / 1e3 / (us * 1e-6), mb / 1e3 / (us * 1e-6) / 8000))
    for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
              "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"):
        if c in d:
            v = d[c] / len(disp[k][c])
            share = c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") and d.get("SQ_WAVE_CYCLES")
            extra = "  (%.0f %% of the wave cycles)" % (100 * v / (d["SQ_WAVE_CYCLES"] / len(disp[k]["SQ_WAVE_CYCLES"]))) if share else ""
            print("   %-22s %12.6g%s" % (c, v, extra))
PY
cat $O/generic_rate_under_rocprof.txt | tail -2
cat $O/pmc_generic.txt
