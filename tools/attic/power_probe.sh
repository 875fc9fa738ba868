#!/bin/bash
# GPU: board power and clocks sampled while the headline bench runs (is the GEMM-dominated step at the power cap?)
O=$GRAFT_REPO_ROOT/gpurun_out/power; mkdir -p $O
rocm-smi --showmaxpower --showpower --showclocks > $O/idle.txt 2>&1
python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "" > $O/bench.json 2> $O/bench.log &
BP=$!
for i in $(seq 1 400); do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use" | tr '\n' ' ' >> $O/samples.txt
  echo >> $O/samples.txt
  kill -0 $BP 2>/dev/null || break
  sleep 0.15
done
wait $BP
cat $O/idle.txt | grep -iE "power|sclk|mclk" | head -8
echo ---
python - <<PY
import re
rows = []
for line in open("$O/samples.txt"):
    pw = re.search(r"Power \(W\): ([0-9.]+)", line); sc = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", line)
    use = re.search(r"GPU use \(%\): (\d+)", line)
    if pw and sc and use: rows.append((float(pw.group(1)), int(sc.group(1)), int(use.group(1))))
busy = [r for r in rows if r[2] >= 90]
print(len(rows), "samples,", len(busy), "with GPU use >= 90 %")
if busy:
    print("power W: min %.0f mean %.0f max %.0f" % (min(b[0] for b in busy), sum(b[0] for b in busy) / len(busy), max(b[0] for b in busy)))
    print("sclk MHz: min %d mean %.0f max %d" % (min(b[1] for b in busy), sum(b[1] for b in busy) / len(busy), max(b[1] for b in busy)))
    print("last 12 busy samples:", busy[-12:])
PY
head -c 300 $O/bench.json
