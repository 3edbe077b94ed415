R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/robustbnns_amd/csrc
run() {
  echo "== conv1_pool flags: $*"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include "$@" -c rbnn_conv.hip -o rbnn_conv.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o librbnn_hip.so rbnn_kernels.o rbnn_conv.o rbnn_conv_x3.o rbnn_split.o rbnn_triple.o rbnn_svi.o rbnn_lowdim.o && \
  (cd $R && for i in 1 2 3 4; do timeout 200 python tools/scratch/dbg_conv_concurrent.py 2>&1 | grep "both at once. [0-9]* of"; done)
}
run -DDIAG_SRC0_BCAST
run -DDIAG_NOTHING
