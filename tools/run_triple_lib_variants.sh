#!/bin/bash
# usage: tools/run_triple_lib_variants.sh <workload> "<flags1>" "<flags2>" ...   rebuilds librbnn_hip.so with extra -D flags for rbnn_triple.hip ON
# the GPU box and times a bench workload (auto mode only) for each flag set; the tree's library is rebuilt without flags at the end
mkdir -p gpurun_out/abl
# flag sets that contain an ablation switch (RBNN_*_ABL_*) must also carry -DRBNN_ALLOW_ABLATION (csrc/rbnn_common.hpp); the runs below are allowed to
# load such a library (RBNN_ALLOW_ABLATION=1 here only), everything else refuses it (rbnn_build_flags(), robustbnns_amd/_hip.py)
export RBNN_ALLOW_ABLATION=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
wl=$1; shift
cd $R/robustbnns_amd/csrc
for f in "$@"; do
  echo "== $f"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $f -c rbnn_triple.hip -o rbnn_triple.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o librbnn_hip.so rbnn_kernels.o rbnn_conv.o rbnn_conv_x3.o rbnn_split.o rbnn_triple.o rbnn_svi.o rbnn_lowdim.o && \
  (cd $R && python bench.py --workload $wl --steps 5 --warmup 1 --cpu-seconds 0 --no-other-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})")
done 2>&1 | tee $R/gpurun_out/abl/triple_lib_variants.log
cd $R && unset RBNN_ALLOW_ABLATION && python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1 && echo "[run_triple_lib_variants] library rebuilt without variant flags"
