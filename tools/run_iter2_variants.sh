# A/B of the PROST_ITER2_OPT experiment variants (library built with -DPROST_ITER2_VARIANTS)
for o in 0 0 $@; do echo "== OPT $o"; PROST_ITER2_OPT=$o QUICK=1 python tools/microbench_iter2.py 4096 24,24 f32 | grep double; done
