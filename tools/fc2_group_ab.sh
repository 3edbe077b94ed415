#!/bin/bash
# usage (on the GPU box): tools/fc2_group_ab.sh <tag>     -> gpurun_out/<tag>/fc2_group_ab.txt
# A/B of the fc2 sample grouping (AttackEngine._fc2_groups): bench --workload fc2 per (forward group, backward group), same box, plus
# the WRITE_SIZE / FETCH_SIZE counters of the forward kernels for three forward group sizes.
set -u
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$1', 'ms/step %.3f' % d['ms_per_step'], {n: round(v['avg_ms'],3) for n,v in k.items()}, {n: v['launches_per_pass'] for n,v in k.items()})"; }
{
echo "# fc2-512, S=100, N=10000, FGSM, stored posterior, triple mode; group = samples per reused hidden image (0 = all 100 at once)"
for rep in 1 2; do
for g in "0 0" "3 0" "6 0" "13 0" "16 0" "25 0" "50 0" "0 5" "0 10" "0 20" "0 50" "6 10" "16 20"; do
  set -- $g
  RBNN_FC2_GROUP_FWD=$1 RBNN_FC2_GROUP_BWD=$2 python bench.py --workload fc2 --steps 10 --warmup 2 --cpu-seconds 0 --no-other-mode --posterior stored 2>/dev/null | line "rep$rep fwd_group=$1 bwd_group=$2"
done
done
} 2>&1 | tee $OUT/fc2_group_ab.txt
cd /tmp
for g in 0 6 16; do
  for c in WRITE_SIZE FETCH_SIZE; do
    RBNN_FC2_GROUP_FWD=$g RBNN_FC2_GROUP_BWD=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_g${g}_$c -o run -- python3 $R/bench.py --workload fc2 --steps 2 --warmup 1 --cpu-seconds 0 --no-other-mode --posterior stored > $OUT/pmc_g${g}_$c.log 2>&1
    python3 - <<PY | tee -a $OUT/fc2_group_ab.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_g${g}_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fc_forward_x3" in r["Kernel_Name"]:
            acc["layer2" if "true" in r["Kernel_Name"].split("fc_forward_x3_kernel")[1][:60].split(",")[-1] else "layer1"].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("pmc fwd_group=$g $c", k, "launches", len(v), "KB per launch %.4g" % (sum(v) / len(v)), "KB per pass %.4g" % (sum(v) / len(v) * (1 if $g == 0 else -(-100 // $g))))
PY
    rm -rf $OUT/pmc_g${g}_$c
  done
done
