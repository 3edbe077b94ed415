set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
python tests/diagnostics/dbg_oracle_case.py > gpurun_out/r06b/oracle_case.txt 2>&1; tail -5 gpurun_out/r06b/oracle_case.txt
timeout 600 python -m pytest tests/test_hip_round6.py -x -q -s 2>&1 | grep -vE "^(Saving|Loading|Producing|Evaluating|test accuracy|avg softmax|$)" | tail -40 > gpurun_out/r06b/pytest_round6.txt; tail -25 gpurun_out/r06b/pytest_round6.txt
