R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/robustbnns_amd/csrc
for f in "-DRBNN_X3FWD_PAIR13=0" "-DRBNN_X3FWD_PAIR13=1"; do
  echo "== $f"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $f -c rbnn_conv_x3.hip -o rbnn_conv_x3.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o librbnn_hip.so rbnn_kernels.o rbnn_conv.o rbnn_conv_x3.o rbnn_split.o rbnn_triple.o rbnn_svi.o rbnn_lowdim.o && \
  (cd $R && for i in 1 2 3; do timeout 200 python -m pytest tests/test_hip_sharded_2rank.py -m gpu -q -x -k "conv" 2>&1 | grep -E "AssertionError: rank 0|passed|failed" | head -3; done)
done
