cd /tmp && export TMPDIR=/tmp
for wl in "c5 --points 512 --iters 3 --steps 1 --warmup 1" "conv --steps 3 --warmup 1"; do
  set -- $wl
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/q
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q/trace -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --cpu-seconds 0 --no-other-mode > /dev/null 2>&1
  python3 - <<'PY'
import csv,os
f=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/q/trace/run_kernel_stats.csv'
for r in list(csv.DictReader(open(f)))[:9]:
    print(r['Name'][:60].ljust(60), r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
  echo ----
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/q
