#!/bin/bash
# round 6, run 27: the whole GPU suite with the shipped defaults, smoke(), and the round profile (bench + kernel stats + PMC traffic)
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
( time python -m pytest tests -q -m gpu --durations=15 ) > gpurun_out/r06/suite_final_e.log 2>&1
tail -5 gpurun_out/r06/suite_final_e.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke_final_e.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r06/smoke_final_e.log
bash tools/profile_round.sh r06_e > gpurun_out/r06/profile_e.log 2>&1
tail -3 gpurun_out/r06/profile_e.log | cut -c1-1200
