#!/bin/bash
# usage: tools/run_variants.sh "<flags1>" "<flags2>" ...   each a set of -D flags for tools/ablate_split.hip (GPU box)
mkdir -p gpurun_out/abl
i=0
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRBNN_ALLOW_ABLATION -DRBNN_FAST_BUILD $f -o /tmp/svar_$i tools/ablate_split.hip 2>/dev/null &
  i=$((i+1))
done
wait
i=0
for f in "$@"; do echo "== $f"; /tmp/svar_$i; i=$((i+1)); done 2>&1 | tee gpurun_out/abl/variants_split.log
