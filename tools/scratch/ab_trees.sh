#!/bin/bash
# Same-box A/B of two source trees: put an older checkout (git archive <rev> | tar -x -C abtree; build it) next to the working tree, then
#   gpurun -- 'bash tools/scratch/ab_trees.sh [workload]'
# times the workload from both, alternating, twice (boxes differ by several per cent on the power-limited kernels: only same-box numbers compare).
wl=${1:-c2}
for rep in 1 2; do
  for t in abtree .; do
    (cd $t && echo "== $t" && python bench.py --workload $wl --steps 10 --warmup 2 --cpu-seconds 0 --no-other-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})")
  done
done
