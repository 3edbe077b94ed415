#!/bin/bash
# one rocprofv3 --kernel-trace --stats pass of a bench workload + the summary: tools/prof_quick.sh <tag> <workload> [bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; wl=$2; shift 2
OUT=$R/gpurun_out/$tag/$wl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $R/bench.py --workload $wl --cpu-seconds 0 --steps 5 --warmup 2 "$@" > $OUT/trace.log 2>&1
rc=$?
cd $R && python tools/summarize_profile.py $OUT $wl > $OUT/summary.txt 2>&1
rm -rf $OUT/trace
head -16 $OUT/summary.txt | cut -c1-130
grep '^{"metric' $OUT/trace.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})"
exit $rc
