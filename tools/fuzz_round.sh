#!/bin/bash
# via gpurun: end-of-round randomised sweeps (search: 4 seeds x N cases; fine-tune step: 24 cases). Usage: fuzz_round.sh TAG [N] [first seed]
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-fuzz}; mkdir -p $O
N=${2:-300}
cd $GRAFT_REPO_ROOT
S0=${3:-501}
for seed in $S0 $((S0 + 1)) $((S0 + 2)) $((S0 + 3)); do
  timeout 1500 python tools/fuzz_parity.py $N $seed > $O/fuzz_parity_$seed.log 2>&1; echo "seed $seed rc=$?: $(tail -1 $O/fuzz_parity_$seed.log)"
done
timeout 1200 python tools/fuzz_train.py 24 55 > $O/fuzz_train.log 2>&1; echo "train rc=$?: $(tail -1 $O/fuzz_train.log)"
