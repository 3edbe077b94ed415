#!/bin/bash
# Run on the GPU box (via gpurun): usage: tools/profile_round.sh <tag> <stage>       -> gpurun_out/<tag>/...   (copy what should be judged into profiles/<tag>/)
#   stage tests : the GPU test tier (-s: statistics printed), smoke()
#   stage bench : every bench workload (JSON lines) + the 1-rank forced-collectives run
#   stage prof  : rocprofv3 --kernel-trace --stats and the separate --pmc passes of the workloads in WLS (default "c2 c5 conv"; QUICK=1: c2 only;
#                 also: c3 c4 fc2 fc2_1024 conv1024 — five bench runs per workload, so three or four workloads per 20-minute call)
# (a gpurun call is limited to 20 minutes: one stage per call)
set -u
TAG=${1:-r03}
STAGE=${2:-bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
if [ "$STAGE" = "tests" ]; then
  # (a heartbeat: the filtered log below appears only when pytest ends, and gpurun takes seven silent minutes for a hang)
  (while sleep 60; do echo "[profile_round] tests running"; done) & HB=$!
  (timeout 1100 python -m pytest tests -m gpu -q -s 2>&1 | grep -vE "^(Saving|Loading|Producing|Evaluating|test accuracy|avg softmax|vanishing|increasing|null|image_idx| === |min = |$)" | tail -600) > $OUT/pytest_gpu.log
  kill $HB 2>/dev/null
  tail -2 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
fi
if [ "$STAGE" = "bench" ]; then
  python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
  python bench.py --workload c3 --steps 2 --warmup 1 --cpu-seconds 10 2>/dev/null | tail -1 > $OUT/bench_c3.json
  python bench.py --workload c4 --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_c4.json
  python bench.py --workload c5 --points 1024 --iters 10 --steps 1 --warmup 1 --cpu-seconds 10 2>/dev/null | tail -1 > $OUT/bench_c5_n1024_t10.json
  python bench.py --workload conv --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_conv.json
  python bench.py --workload fc2 --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_fc2.json
  python bench.py --workload fc2_1024 --steps 3 --warmup 1 --cpu-seconds 10 2>/dev/null | tail -1 > $OUT/bench_fc2_1024.json
  python bench.py --workload conv1024 --steps 3 --warmup 1 --cpu-seconds 10 2>/dev/null | tail -1 > $OUT/bench_conv1024.json
  python bench.py --workload c1 --steps 200 --warmup 20 --cpu-seconds 5 2>/dev/null | tail -1 > $OUT/bench_c1.json
  python bench.py --workload eval --steps 10 --warmup 2 --cpu-seconds 10 2>/dev/null | tail -1 > $OUT/bench_eval.json
  RBNN_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-seconds 0 --no-other-mode 2>/dev/null | tail -1 > $OUT/bench_c2_torchrun_1rank_forced_collectives.json
  for f in $OUT/bench*.json; do python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$(basename $f)', d['precision_mode'], '%.4g' % d['value'], '%.4g ms' % d['ms_per_step'])"; done
fi
if [ "$STAGE" = "prof" ]; then
cd /tmp
prof() {   # prof <subdir> <workload> <points> <samples per GPU> <extra bench args...>   (points / samples: what the command below runs at)
  local sub=$1 wl=$2 pts=$3 smp=$4; shift 4
  mkdir -p $OUT/$sub
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$sub/trace -o run -- python3 $R/bench.py --workload $wl --cpu-seconds 0 "$@" > $OUT/$sub/trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$sub/pmc_fetch -o run -- python3 $R/bench.py --workload $wl --cpu-seconds 0 "$@" > $OUT/$sub/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$sub/pmc_write -o run -- python3 $R/bench.py --workload $wl --cpu-seconds 0 "$@" > $OUT/$sub/pmc_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/$sub/pmc_sq -o run -- python3 $R/bench.py --workload $wl --cpu-seconds 0 "$@" > $OUT/$sub/pmc_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/$sub/pmc_lds -o run -- python3 $R/bench.py --workload $wl --cpu-seconds 0 "$@" > $OUT/$sub/pmc_lds.log 2>&1
  python3 $R/tools/summarize_profile.py $OUT/$sub $wl $pts $smp > $OUT/$sub/summary.txt 2>&1
  cp $OUT/$sub/trace/run_kernel_stats.csv $OUT/$sub/kernel_stats.csv 2>/dev/null
  rm -rf $OUT/$sub/trace/run_kernel_trace.csv $OUT/$sub/pmc_*/run_kernel_trace.csv      # bulky; the summary keeps the per-kernel numbers
  head -30 $OUT/$sub/summary.txt
}
WLS=${WLS:-"c2 c5 conv"}
[ -n "${QUICK:-}" ] && WLS="c2"
for wl in $WLS; do
  case $wl in
    c2)       prof c2 c2 10000 100 --steps 5 --warmup 2 ;;
    c3)       prof c3 c3 10000 500 --iters 2 --steps 1 --warmup 1 --no-other-mode ;;
    c4)       prof c4 c4 10000 250 --steps 2 --warmup 1 --no-other-mode ;;
    fc2)      prof fc2 fc2 10000 100 --steps 3 --warmup 1 --no-other-mode ;;
    fc2_1024) prof fc2_1024 fc2_1024 10000 100 --steps 3 --warmup 1 --no-other-mode ;;
    c5)       prof c5 c5 512 62 --points 512 --iters 3 --steps 1 --warmup 1 --no-other-mode ;;
    conv)     prof conv conv 2048 16 --steps 3 --warmup 1 ;;
    conv1024) prof conv1024 conv1024 1024 16 --steps 3 --warmup 1 --no-other-mode ;;
    eval)     prof eval eval 10000 100 --steps 3 --warmup 1 --no-other-mode ;;
    *) echo "unknown workload $wl" ;;
  esac
done
python3 - <<PY
import json, glob, os
out = {}
for f in sorted(glob.glob("$OUT/*/pmc_traffic.json")):
    d = json.load(open(f))
    src = d.pop("source", None)
    out.update(d)
    out.setdefault("source", []).append(src)
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
print(sorted(k for k in out if k != "source"))
PY
fi
