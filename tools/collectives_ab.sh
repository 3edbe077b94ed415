#!/bin/bash
# usage (GPU box): tools/collectives_ab.sh <tag>   -> gpurun_out/<tag>/collectives_ab.txt
# Why does the 1-rank C2 step with the collectives forced on cost more than the plain step, kernels included?  Same box,
# alternating: (a) plain step; (b) the sharded launch sequence with the collectives replaced by no-ops (RBNN_FAKE_COLLECTIVES=1): what the
# SEQUENCE costs — separate reduce / loss / dZ / slab-sum kernels instead of the fused tail, G materialised; (c) the same with the 1-rank RCCL
# all-reduces; (d) plain step with the fused tail switched off (the sequence of round 3).
TAG=${1:-r04f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$1', 'ms/step %.3f' % d['ms_per_step'], {n: round(v['avg_ms'],3) for n,v in k.items()}, 'draw %.3f' % d['svi']['draw_ms'])"; }
run_plain() { python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-other-mode 2>/dev/null | line "$1"; }
run_tr() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$2 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-seconds 0 --no-other-mode 2>/dev/null | line "$1"; }
{
for rep in 1 2; do
  run_plain "rep$rep (a) plain, fused tail          "
  RBNN_FUSED_TAIL=0 run_plain "rep$rep (d) plain, separate tail kernels"
  RBNN_FORCE_COLLECTIVES=1 RBNN_FAKE_COLLECTIVES=1 run_tr "rep$rep (b) sharded sequence, no-op comm " $rep
  RBNN_FORCE_COLLECTIVES=1 run_tr "rep$rep (c) sharded sequence, 1-rank RCCL" $((rep+2))
done
} 2>&1 | tee $OUT/collectives_ab.txt
