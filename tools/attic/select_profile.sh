#!/bin/bash
# GPU: per-step durations of select_kernel at a given beam/batch. Usage: bash tools/select_profile.sh <tag> <beams> <batch>
tag=${1:-selprof}; out=$PWD/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/prof -o t -- python bench.py --beams ${2:-1000} --batch ${3:-1} --steps 1 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --docs 1000000 > $out/bench.json 2> $out/bench.log
f=$(find $out/prof -name "*_results.db" | head -1)
python tools/select_steps.py "$f" | tail -4
python tools/rocpd_summary.py "$f" $out/kernel_stats.csv
head -12 $out/kernel_stats.csv | cut -c1-160
rm -rf $out/prof
