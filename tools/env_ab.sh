#!/bin/bash
# usage: tools/env_ab.sh <workload> "<ENV=val ...>" "<ENV=val ...>" ...   times a bench workload (auto mode) under each environment, twice, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
wl=$1; shift
cd $R
for rep in 1 2; do
  for e in "$@"; do
    echo "== $e"
    env $e python bench.py --workload $wl --steps 5 --warmup 1 --cpu-seconds 0 --no-other-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})"
  done
done
