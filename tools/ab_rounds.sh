#!/bin/bash
# Same-box A/B of a bench workload between an older tree (abtree/<tag>: `git archive <commit> | tar -x -C abtree/<tag>` + build, done on the build host;
# abtree/ is git-ignored but travels with gpurun) and the current tree, alternating runs on ONE box (boxes differ by several per cent on the
# power-limited kernels).  usage: tools/ab_rounds.sh <tag> <rounds> <bench args...>   -> gpurun_out/ab/<tag>_<workload>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; rounds=$2; shift 2
mkdir -p $R/gpurun_out/ab
out=$R/gpurun_out/ab/${tag}_$(echo "$@" | tr -c 'a-zA-Z0-9' '_' | cut -c1-60).txt
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%-8s" % sys.argv[1], "%.5g" % d["value"], "%.4g ms/step" % d["ms_per_step"], {k: round(v["avg_ms"], 3) for k, v in d["roofline"]["kernels"].items()})'
for i in $(seq $rounds); do
  (cd $R/abtree/$tag && python bench.py "$@" --cpu-seconds 0 --no-other-mode 2>/dev/null | tail -1 | python3 -c "$fmt" $tag) | tee -a $out
  (cd $R && python bench.py "$@" --cpu-seconds 0 --no-other-mode 2>/dev/null | tail -1 | python3 -c "$fmt" now) | tee -a $out
done
