cd $GRAFT_REPO_ROOT
run() { env $3 python bench.py --batch $1 --beams $2 --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3 Q=$1 B=$2', round(d['value'],1), 'q/s', round(d['ms_per_step'],2), 'ms')"; }
for e in X=1 RPR_GEMM_SKINNY=256 RPR_GEMM_SKINNY=192 RPR_GEMM_SKINNY=448; do
  for q in 1 16 24 32 40; do run $q 10 $e; done
done
