#!/bin/bash
# usage: tools/run_variants_triple.sh "<flags1>" "<flags2>" ...   each a set of -D flags for tools/ablate_triple.hip (GPU box)
mkdir -p gpurun_out/abl
i=0
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRBNN_ALLOW_ABLATION -DRBNN_FAST_BUILD $f -o /tmp/tvar_$i tools/ablate_triple.hip 2>/dev/null &
  i=$((i+1))
  if [ $((i % 8)) -eq 0 ]; then wait; fi
done
wait
i=0
for f in "$@"; do echo "== $f"; timeout -k 5 120 /tmp/tvar_$i; i=$((i+1)); done 2>&1 | tee gpurun_out/abl/variants_triple.log
