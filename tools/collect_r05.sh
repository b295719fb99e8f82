#!/bin/bash
# On the GPU box: everything profiles/r05_* and DESIGN.md section 6 quote for round 5.  usage: bash tools/collect_r05.sh   (outputs under gpurun_out/r05c/)
R=$PWD; O=$R/gpurun_out/r05c; mkdir -p $O
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|Socket|NUMA|Thread|Core|^CPU\(s\)" ) > $O/host_info.txt 2>&1
# ---- C2 (headline): the driver's invocation, the default run, fp64
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_invocation.json
python3 bench.py 2>/dev/null | tail -1 > $O/bench.json
python3 bench.py --dtype f64 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f64.json
# ---- the reference's DEFAULT backend options (pdhg.m:4-14: boyd, residual_iter = 1)
for size in 4096 1024 256; do
  python3 bench.py --stepsize boyd --residual-iter 1 --size $size --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_boyd_r1_${size}.json
done
# ---- C3 / C4 (+ C3 with the default options: rule on the device / on the host)
python3 bench.py --config c3 2>/dev/null | tail -1 > $O/bench_c3.json
for dev in 1 0; do PROST_BENCH_DEVICE_RULES=$dev python3 bench.py --config c3 --stepsize boyd --residual-iter 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3_boyd_r1_dev$dev.json; done
python3 bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4.json
python3 bench.py --config c4 --dtype f64 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_f64.json
# ---- rocprofv3 kernel stats of the same commands
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_c3 -- python3 $R/bench.py --config c3 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_c4 -- python3 $R/bench.py --config c4 --no-cpu-baseline > $O/bench_c4_under_rocprof.json 2>/dev/null
# ---- HBM traffic of the C4 round kernels (separate --pmc passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O -o pmc_c4_$c -- python3 $R/bench.py --config c4 --steps 60 --warmup 10 --prelude-iters 0 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python3 - <<PY > $O/pmc_summary.txt
import csv, collections, glob
for tag in ("pmc_c4_FETCH_SIZE", "pmc_c4_WRITE_SIZE"):
    files = glob.glob("$O/**/%s_counter_collection.csv" % tag, recursive=True)
    rows = [r for f in files[:1] for r in csv.DictReader(open(f)) if "cg_pixel" in r["Kernel_Name"] or "op_stage" in r["Kernel_Name"]]
    big = collections.defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        big[k] = max(big[k], int(r["Grid_Size"]))
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if int(r["Grid_Size"]) != big[k]: continue
        agg[(k, r["Counter_Name"], big[k])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"], big[k])].add(r["Dispatch_Id"])
    for (k, c, g), v in sorted(agg.items()):
        print(tag, k[-84:], "grid %d work-items" % g, c, "mean per launch = %.6g over %d launches" % (v / len(n[(k, c, g)]), len(n[(k, c, g)])))
PY
cat $O/pmc_summary.txt
for f in $(find $O -name "*_kernel_stats.csv"); do echo $f; python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:9]: print('  ', r['Name'][:110].replace('void prost_hip::',''), r['Calls'], r['AverageNs'], r['Percentage'])"; done
for f in $O/bench*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline',{}); c=d.get('cpu_baseline') or {}
    print('$f'.split('/')[-1], 'value', round(d['value'],1), 'path', d['config'].get('path'), 'kernel', r.get('kernel'), 'avg_ms', r.get('avg_launch_ms'), 'timed', r.get('launches_timed'), 'frac', r.get('frac'), 'phys', r.get('frac_hbm_traffic'), 'frac_iteration', r.get('frac_iteration'), 'cpu', c.get('value'), c.get('cores'), c.get('threads_probed'), c.get('port_over_reference_one_thread'))
except Exception as e: print('$f', 'ERR', e)"; done
{ python3 tools/generic_rule_rate.py 256 256 3000; python3 tools/generic_rule_rate.py 700 464 3000; python3 tools/generic_rule_rate.py 1024 1024 2000; python3 tools/generic_rule_rate.py 2048 2048 600; } 2>/dev/null | tee $O/generic_op_fusion_rates.txt
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*counter_collection.csv" -size +1M -delete
# ---- where the generic path spends its time at 2048^2: kernel stats, operator inside the prox kernels (mask 1) / separate products (mask 4)
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_generic2048 -- python3 $R/tools/generic_rule_rate.py 2048 2048 300 1 100 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_generic2048_separate -- python3 $R/tools/generic_rule_rate.py 2048 2048 300 1 100 4 > /dev/null 2>&1
# ---- instruction mix of the prox launches that apply the operator (short runs: every dispatch is serialised under --pmc)
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
G2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
G3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT"
i=0
for grp in "$G1" "$G2" "$G3" "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/gpmc$i -o p$i -- python3 $R/tools/generic_rule_rate.py 2048 2048 12 1 6 1 > /dev/null 2>&1
done
cd $R
python3 - "$O" > $O/pmc_generic_op_kernels.txt <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(lambda: collections.defaultdict(set)); big = collections.defaultdict(int)
rows = []
for f in sorted(glob.glob(sys.argv[1] + "/gpmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "prox_elem_vec" not in k and "fold_sums" not in k: continue
        rows.append((k, r)); big[k] = max(big[k], int(r["Grid_Size"]))
for k, r in rows:
    if int(r["Grid_Size"]) != big[k]: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg):
    print(k, " grid", big[k], "work-items")
    w = agg[k].get("SQ_WAVES", 0) / max(1, len(disp[k].get("SQ_WAVES", [1])))
    for c in sorted(agg[k]):
        n = len(disp[k][c]); v = agg[k][c] / n
        print("   %-26s %14.6g per launch %10.1f per wavefront (%d launches)" % (c, v, v / max(w, 1), n))
PY
cat $O/pmc_generic_op_kernels.txt
find $O -name "*counter_collection.csv" -size +200k -delete
python3 -c "
import csv
for f in ('stats_generic2048', 'stats_generic2048_separate'):
  print(f)
  for r in list(csv.DictReader(open('$O/' + f + '_kernel_stats.csv')))[:12]: print('  ', r['Name'][:150].replace('void prost_hip::',''), r['Calls'], r['AverageNs'], r['Percentage'])"
find $O -name "*kernel_trace.csv" -size +1M -delete
du -sh $R/gpurun_out
