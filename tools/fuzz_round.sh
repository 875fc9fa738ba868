#!/bin/bash
# via gpurun: end-of-round randomised sweeps (search: 4 seeds x N cases; fine-tune step: 24 cases)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-fuzz}; mkdir -p $O
N=${2:-300}
cd $GRAFT_REPO_ROOT
for seed in 501 502 503 504; do
  timeout 1500 python tools/fuzz_parity.py $N $seed > $O/fuzz_parity_$seed.log 2>&1; echo "seed $seed rc=$?: $(tail -1 $O/fuzz_parity_$seed.log)"
done
timeout 1200 python tools/fuzz_train.py 24 55 > $O/fuzz_train.log 2>&1; echo "train rc=$?: $(tail -1 $O/fuzz_train.log)"
