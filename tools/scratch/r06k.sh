export RBNN_ALLOW_ABLATION=1
R=$GRAFT_REPO_ROOT
cd $R/robustbnns_amd/csrc
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$1', 'ms/step %.3f (min %.3f med %.3f)' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_median']), {n: round(v['avg_ms'],3) for n,v in k.items()})"; }
for f in "-DRBNN_X3_FWD_PRIO=0" "-DRBNN_X3_FWD_PRIO=1" "-DRBNN_X3_FWD_PRIO=2" "-DRBNN_X3_FWD_PRIO=0" "-DRBNN_X3_FWD_PRIO=1" "-DRBNN_X3_FWD_PRIO=2"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $f -c rbnn_triple.hip -o rbnn_triple.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o librbnn_hip.so rbnn_kernels.o rbnn_conv.o rbnn_conv_x3.o rbnn_split.o rbnn_triple.o rbnn_svi.o rbnn_lowdim.o && \
  (cd $R && python bench.py --workload c2 --steps 30 --warmup 3 --cpu-seconds 0 --no-other-mode 2>/dev/null | line "c2 $f")
done 2>&1 | tee $R/gpurun_out/r06k_fwd_prio_ab.txt
cd $R && python tools/kernel_resources.py --hazards | tail -1
cd $R && python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1; timeout 600 python -m pytest tests/test_hip_triple.py tests/test_hip_parity.py -x -q 2>&1 | tail -2
cd $R && bash tools/prof_quick.sh r06k c5 --points 512 --iters 3 --steps 1 --warmup 1 --no-other-mode 2>&1 | grep -E "conv1_pool|conv2_pool_x3|conv_bwd_dense" | cut -c1-130
