set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$1', 'ms/step %.3f (min %.3f med %.3f)' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_median']), {n: round(v['avg_ms'],3) for n,v in k.items()})"; }
{
for rep in 1 2 3; do
  for v in "RBNN_X3_L1_DEFER=0" "RBNN_X3_L1_DEFER=1"; do
    env $v python bench.py --workload fc2 --steps 10 --warmup 2 --cpu-seconds 0 --no-other-mode 2>/dev/null | line "fc2 rep$rep $v"
  done
done
for v in "RBNN_X3_L1_DEFER=0" "RBNN_X3_L1_DEFER=1"; do
  env $v python bench.py --workload fc2_1024 --steps 4 --warmup 1 --cpu-seconds 0 --no-other-mode 2>/dev/null | line "fc2_1024 $v"
done
} 2>&1 | tee gpurun_out/r06e/fc2_defer_ab.txt
timeout 900 python -m pytest tests/test_hip_triple.py -x -q -k fc2 2>&1 | tail -3
