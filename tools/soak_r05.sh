#!/bin/bash
# On the GPU box: the randomised differential run on the round-5 build (product-only switches drawn per case: operator inside the prox
# launches, device-resident rule, two-launch CG rounds) -> gpurun_out/r05_fuzz_final.log (copied to profiles/r05_fuzz_final.log)
O=$PWD/gpurun_out; mkdir -p $O
{
  date
  python3 tools/fuzz_parity.py --mode generic --cases 200000 --seed 61 --budget-s ${1:-420}
  python3 tools/fuzz_parity.py --mode generic --cases 200000 --seed 62 --budget-s ${1:-420}
  python3 tools/fuzz_parity.py --mode fused --cases 200000 --seed 63 --budget-s ${2:-240}
  python3 tools/fuzz_parity.py --mode large --cases 2000 --seed 64 --budget-s ${3:-120}
  python3 tools/fuzz_parity.py --mode sharded --cases 2000 --seed 65 --budget-s ${3:-120}
} > $O/r05_fuzz_final.log 2>&1
tail -c 3000 $O/r05_fuzz_final.log | cut -c1-1500
