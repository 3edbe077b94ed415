#!/bin/bash
# usage (on the GPU box): [LEVELS='1 2'] [EXTRA='-D...'] tools/dense_stamps.sh [bench args...]   — builds the library with -DRBNN_DENSE_STAMPS=1, then =2, prints the per-segment
# cycles of conv_bwd_dense_x3_kernel for the bench workload given (default: c5 at 512 points), and rebuilds the library as the sources say
export RBNN_ALLOW_ABLATION=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/abl
ARGS=${@:-"--workload c5 --points 512 --iters 3 --steps 1 --warmup 1 --cpu-seconds 0 --no-other-mode"}
for lvl in ${LEVELS:-1 2}; do
  echo "== RBNN_DENSE_STAMPS=$lvl $EXTRA"
  (cd $R/robustbnns_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRBNN_ALLOW_ABLATION -DRBNN_DENSE_STAMPS=$lvl $EXTRA -c rbnn_conv_x3.hip -o rbnn_conv_x3.o 2>/dev/null && \
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o librbnn_hip.so rbnn_kernels.o rbnn_conv.o rbnn_conv_x3.o rbnn_split.o rbnn_triple.o rbnn_svi.o rbnn_lowdim.o) && \
  (cd $R && python tools/dense_stamps.py $ARGS 2>/dev/null | grep -E "^wave|^   " )
done 2>&1 | tee -a $R/gpurun_out/abl/dense_stamps.log
cd $R && unset RBNN_ALLOW_ABLATION && python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1 && echo "[dense_stamps] library rebuilt without diagnostic flags"
