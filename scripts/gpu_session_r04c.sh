set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04c_pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 -o b1 -- python $GRAFT_REPO_ROOT/scripts/tracklet_b1_profile.py > $GRAFT_REPO_ROOT/gpurun_out/r04c_b1_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_b1 -name "*kernel_stats*" | head
cp $(find /tmp/prof_b1 -name "*kernel_stats.csv" | head -1) gpurun_out/r04c_b1_kernel_stats.csv
cat gpurun_out/r04c_pytest.log; tail -5 gpurun_out/r04c_b1_prof.log
