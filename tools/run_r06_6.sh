set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests -x -q -m gpu --durations=25 2>&1 | tail -45 > gpurun_out/r06/suite_a.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke3.log 2>&1
bash tools/profile_round.sh r06_a > gpurun_out/r06_a_profile.log 2>&1
