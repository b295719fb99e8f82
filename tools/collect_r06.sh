#!/bin/bash
# On the GPU box: the bench lines and rocprofv3 kernel summaries profiles/r06_* and DESIGN.md section 6 quote.  usage: bash tools/collect_r06.sh
R=$PWD; O=$R/gpurun_out/r06c; mkdir -p $O
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|Socket|NUMA|Thread|Core|^CPU\(s\)" ) > $O/host_info.txt 2>&1
# ---- C2 (headline): the driver's invocation twice (cpu_baseline reproducibility), the default run, fp64
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_invocation.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_invocation_again.json
python3 bench.py 2>/dev/null | tail -1 > $O/bench.json
python3 bench.py --dtype f64 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f64.json
python3 bench.py --stepsize boyd --residual-iter 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_boyd_r1_4096.json
python3 bench.py --residual-iter 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_residual_iter_100.json
# ---- C3, C4 (1024^2 and 2048^2), c4w (1024^2 and 2048^2)
python3 bench.py --config c3 2>/dev/null | tail -1 > $O/bench_c3.json
python3 bench.py --config c4 2>/dev/null | tail -1 > $O/bench_c4.json
python3 bench.py --config c4 --size 2048 --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $O/bench_c4_2048.json
python3 bench.py --config c4w 2>/dev/null | tail -1 > $O/bench_c4w.json
python3 bench.py --config c4w --size 2048 --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $O/bench_c4w_2048.json
# ---- rocprofv3 kernel stats of the same commands
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_driver -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_c3 -- python3 $R/bench.py --config c3 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats_c4w -- python3 $R/bench.py --config c4w --no-cpu-baseline > $O/bench_c4w_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kernels -- python3 $R/tools/bench_kernels.py > $O/kernels_raw.jsonl 2> $O/kernels.err
cd $R
python3 tools/annotate_kernels.py $O/kernels_raw.jsonl profiles/r06_kernels.jsonl > $O/kernels.jsonl
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for f in $(find $O -name "*_kernel_stats.csv"); do echo $f; python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:8]: print('  ', r['Name'][:120].replace('void prost_hip::',''), r['Calls'], r['AverageNs'], r['Percentage'])"; done
python3 - $O <<'PY'
import glob, json, sys
r3 = lambda v: None if v is None else round(v, 3)
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline", {}); c = d.get("cpu_baseline") or {}; rf = d.get("roofline_fmad") or {}
        print(f.split("/")[-1], "value", round(d["value"], 1), "fmad", round(d.get("value_fmad") or 0, 1), d["config"].get("path"), r.get("kernel"),
              "ms", r3(r.get("avg_launch_ms")), "frac", r3(r.get("frac")), "traffic frac", r3(r.get("frac_hbm_traffic")),
              "| fmad kernel", rf.get("kernel"), r3(rf.get("avg_launch_ms")), r3(rf.get("frac")), r3(rf.get("frac_hbm_traffic")),
              "| cpu", r3(c.get("value")), r3(c.get("value_min")), r3(c.get("value_max")), r3(c.get("port_over_reference_one_thread")))
    except Exception as e:
        print(f, "ERR", e)
PY
