#!/bin/bash
# GPU: latency configurations against the RPR_GEMM_DEEP threshold (4-stage 128x64 kernel up to that many tiles)
for d in 128 256 400 800; do
  for cfg in "--batch 1 --beams 1000" "--batch 1" "--batch 8" "--batch 64"; do
    v=$(RPR_GEMM_DEEP=$d python bench.py $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2))")
    echo "deep=$d $cfg -> $v ms"
  done
done
