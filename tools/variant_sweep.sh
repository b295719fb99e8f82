#!/bin/bash
# experiment: iteration-kernel variants (bit0 nt stores, bit1 nt loads, bit2 no register prefetch)
for rep in 1 2; do for v in 0 1 2 3 4 5 6 7; do
  PROST_HIP_ITER_VARIANT=$v python tools/microbench_fused.py 4096 all iter 2>&1 | grep -E "float32 cols=12 " | sed "s/^/var=$v /" | cut -c1-90
done; done
