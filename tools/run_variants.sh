#!/bin/bash
# usage: tools/run_variants.sh "<flags1>" "<flags2>" ...   each a set of -D flags for tools/ablate.hip (GPU box)
mkdir -p gpurun_out/abl
i=0
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRBNN_ALLOW_ABLATION -DRBNN_FAST_BUILD $f -o /tmp/var_$i tools/ablate.hip 2>/dev/null &
  i=$((i+1))
done
wait
i=0
for f in "$@"; do echo "== $f"; /tmp/var_$i; i=$((i+1)); done 2>&1 | tee gpurun_out/abl/variants.log
