export RBNN_ALLOW_ABLATION=1
R=$GRAFT_REPO_ROOT
cd $R/robustbnns_amd/csrc
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$1', 'ms/step %.3f (min %.3f med %.3f)' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_median']), {n: round(v['avg_ms'],3) for n,v in k.items()})"; }
for f in "-DRBNN_X3_L2_CFG=4,4,2,4" "-DRBNN_X3_L2_CFG=2,8,4,2" "-DRBNN_X3_L2_CFG=4,4,2,4" "-DRBNN_X3_L2_CFG=2,8,4,2"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$f" -c rbnn_triple.hip -o rbnn_triple.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o librbnn_hip.so rbnn_kernels.o rbnn_conv.o rbnn_conv_x3.o rbnn_split.o rbnn_triple.o rbnn_svi.o rbnn_lowdim.o && \
  (cd $R && python bench.py --workload fc2 --steps 10 --warmup 2 --cpu-seconds 0 --no-other-mode 2>/dev/null | line "fc2 $f"; \
   python bench.py --workload fc2_1024 --steps 4 --warmup 1 --cpu-seconds 0 --no-other-mode 2>/dev/null | line "fc2_1024 $f")
done 2>&1 | tee $R/gpurun_out/r06n_l2_cfg_ab.txt
cd $R && timeout 600 python -m pytest tests/test_hip_triple.py -x -q -k fc2 2>&1 | tail -2
