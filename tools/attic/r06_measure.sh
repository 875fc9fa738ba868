#!/bin/bash
# via gpurun: the round's measurement set — bench + rocprofv3 kernel stats + PMC passes of the headline (tools/profile_round.sh),
# PMC traffic of the small-batch searches, the CLI end to end. Usage: tools/r06_measure.sh
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06 > gpurun_out/prof_r06.log 2>&1; tail -5 gpurun_out/prof_r06.log
bash tools/small_batch_pmc.sh r06_sbpmc > gpurun_out/r06_sbpmc.log 2>&1; tail -25 gpurun_out/r06_sbpmc.log
mkdir -p gpurun_out/r06_cli
python tools/cli_end_to_end.py --out gpurun_out/r06_cli/cli_end_to_end.json > gpurun_out/r06_cli/log.txt 2>&1; tail -45 gpurun_out/r06_cli/log.txt
