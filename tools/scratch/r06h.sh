# conv1_pool_x3_kernel: fragment pipeline on / off, same box (rocprofv3 kernel stats of the c5 pass at N = 512, S = 62 and of the conv workload)
export RBNN_ALLOW_ABLATION=1
R=$GRAFT_REPO_ROOT
cd $R/robustbnns_amd/csrc
for f in "-DRBNN_C1X3_PIPE=0" "-DRBNN_C1X3_PIPE=1" "-DRBNN_C1X3_PIPE=0" "-DRBNN_C1X3_PIPE=1"; do
  echo "== $f"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $f -c rbnn_conv_x3.hip -o rbnn_conv_x3.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o librbnn_hip.so rbnn_kernels.o rbnn_conv.o rbnn_conv_x3.o rbnn_split.o rbnn_triple.o rbnn_svi.o rbnn_lowdim.o && \
  (cd $R && bash tools/prof_quick.sh r06h c5 --points 512 --iters 3 --steps 1 --warmup 1 --no-other-mode 2>&1 | grep -E "conv1_pool|^[0-9.e+]+ " | cut -c1-150; \
   bash tools/prof_quick.sh r06h conv --steps 3 --warmup 1 --no-other-mode 2>&1 | grep -E "conv1_pool" | cut -c1-150)
done 2>&1 | tee $R/gpurun_out/r06h_conv1_pipe_ab.txt
cd $R && python -m pytest tests/test_hip_round6.py -x -q 2>&1 | tail -2
