cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r02x}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for Q in 2048; do
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 --batch $Q > $OUT/bench_q$Q.json 2> $OUT/bench_q$Q.log
python - <<PY
import json
d = json.loads(open("$OUT/bench_q$Q.json").read().strip().splitlines()[-1])
print("Q=$Q", round(d["value"], 1), "q/s", round(d["ms_per_step"],1), "ms", d["kernel_breakdown_ms"], "frac", round(d["roofline"]["frac"],4), "attn", round(d["roofline_hbm"]["achieved"]), "GB/s")
PY
done


